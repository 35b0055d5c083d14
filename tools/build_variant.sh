#!/bin/bash
# Build an experimental copy of the library with extra -D flags for the encoder GEMM sources (gemm.hip, gemm_corr.hip, gemm_tn.hip):
#   tools/build_variant.sh NAME -DRUART_P8_ABLATE=3
# -> build/libruart_hip_NAME.so (same ABI; use it with RUART_HIP_LIB=build/libruart_hip_NAME.so).  Needs the normal build first.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
for f in gemm gemm_corr gemm_tn; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I ruart_amd/csrc -Wno-unused-result -Wno-pass-failed "$@" \
    -c ruart_amd/csrc/$f.hip -o build/${f}_$name.o 2>/dev/null
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libruart_hip_$name.so build/gemm_$name.o build/gemm_corr_$name.o build/gemm_tn_$name.o \
  ruart_amd/csrc/bert_kernels.o ruart_amd/csrc/bert_forward.o ruart_amd/csrc/sdnet_attention.o ruart_amd/csrc/sdnet_lstm.o ruart_amd/csrc/sdnet_gemm.o ruart_amd/csrc/sdnet_optim.o ruart_amd/csrc/sdnet_scorer.o ruart_amd/csrc/phoc.o ruart_amd/csrc/bert_train_kernels.o ruart_amd/csrc/bert_train_attn.o
echo built build/libruart_hip_$name.so
