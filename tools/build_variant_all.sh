#!/bin/bash
# Build an experimental copy of the WHOLE library with extra -D flags (constants of common.h reach every source):
#   tools/build_variant_all.sh NAME -DRUART_C8_SA_LO=11 -DRUART_C8_SA_HI=0
# -> build/libruart_hip_NAME.so (same ABI; use it with RUART_HIP_LIB=build/libruart_hip_NAME.so).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/$name
objs=""
for f in gemm gemm_corr gemm_tn bert_kernels bert_train_kernels bert_train_attn bert_forward sdnet_attention sdnet_lstm sdnet_gemm sdnet_optim sdnet_scorer phoc; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I ruart_amd/csrc -Wno-unused-result -Wno-pass-failed "$@" \
    -c ruart_amd/csrc/$f.hip -o build/$name/$f.o 2>/dev/null &
  objs="$objs build/$name/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libruart_hip_$name.so $objs
echo built build/libruart_hip_$name.so
