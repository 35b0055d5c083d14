#!/bin/bash
# Build an experimental copy of the library with extra -D flags for ONE source:  tools/build_variant_one.sh NAME sdnet_attention -DWLN_BLOCKS=64
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I ruart_amd/csrc -Wno-unused-result -Wno-pass-failed "$@" \
  -c ruart_amd/csrc/$src.hip -o build/${src}_$name.o 2>/dev/null
objs=""
for f in gemm gemm_corr gemm_tn bert_kernels bert_train_kernels bert_train_attn bert_forward sdnet_attention sdnet_lstm sdnet_gemm sdnet_optim sdnet_scorer phoc; do
  if [ $f = $src ]; then objs="$objs build/${src}_$name.o"; else objs="$objs ruart_amd/csrc/$f.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libruart_hip_$name.so $objs
echo built build/libruart_hip_$name.so
