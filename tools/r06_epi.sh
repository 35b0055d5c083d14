#!/bin/bash
# round 6: epilogue trims of the fp16c GEMM (scaled fp8 conversion, v_fma_mix, GELU as max - |x| r, 32-bit row offsets) against the tree before
# them (build/libruart_hip_base.so): kernel tests, then the four projections and the folded encoder pass, interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_16c or fold or split or encoder or layernorm or attention or embed" > $O/pytest_epi.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_epi.log
for i in 1 2 3; do
  timeout -k 10 120 python3 tools/gemm_corr_bench.py --rows 42752 > $O/epi_new_$i.log 2>&1; grep -h "ff1\|ao \|layer" $O/epi_new_$i.log | sed 's/^/new  /'
  RUART_HIP_LIB=build/libruart_hip_base.so timeout -k 10 120 python3 tools/gemm_corr_bench.py --rows 42752 > $O/epi_base_$i.log 2>&1; grep -h "ff1\|ao \|layer" $O/epi_base_$i.log | sed 's/^/base /'
done
for i in 1 2; do
  timeout -k 10 120 python3 tools/encoder_kernel_times.py > $O/enc_new_$i.log 2>&1; head -6 $O/enc_new_$i.log | cut -c1-110 | sed 's/^/new  /'
  RUART_HIP_LIB=build/libruart_hip_base.so timeout -k 10 120 python3 tools/encoder_kernel_times.py > $O/enc_base_$i.log 2>&1; head -6 $O/enc_base_$i.log | cut -c1-110 | sed 's/^/base /'
done
