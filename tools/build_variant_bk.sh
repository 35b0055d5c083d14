#!/bin/bash
# Build an experimental copy of the library with extra -D flags for bert_kernels.hip only:
#   tools/build_variant_bk.sh NAME -DRUART_ABL_ATTN_NOSTORE   -> build/libruart_hip_NAME.so (use with RUART_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I ruart_amd/csrc -Wno-unused-result -Wno-pass-failed "$@" \
  -c ruart_amd/csrc/bert_kernels.hip -o build/bert_kernels_$name.o 2>/dev/null
objs=""
for f in gemm gemm_corr gemm_tn bert_train_kernels bert_train_attn bert_forward sdnet_attention sdnet_lstm sdnet_gemm sdnet_optim sdnet_scorer phoc; do objs="$objs ruart_amd/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libruart_hip_$name.so build/bert_kernels_$name.o $objs
echo built build/libruart_hip_$name.so
