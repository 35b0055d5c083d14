#!/bin/bash
# round 6 timing diagnostic: both e4m3 companions of four columns in ONE 8-byte store (-DRUART_ABL_SPLIT8: wrong operand layout, right
# instruction counts) against the two 4-byte stores of the product - what the epilogue's store INSTRUCTION count is worth
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
for i in 1 2 3 4; do
  for v in new split8; do
    RUART_HIP_LIB=build/libruart_hip_$v.so timeout -k 10 120 python3 tools/gemm_corr_bench.py --rows 42752 --iters 40 > $O/s8_${v}_$i.log 2>&1
    echo "$v $(grep -h 'ff1' $O/s8_${v}_$i.log | sed 's/.*f16+fp8 *\([0-9.]*\) us.*/ff1 \1/') $(grep -h 'ff2' $O/s8_${v}_$i.log | sed 's/.*f16+fp8 *\([0-9.]*\) us.*/ff2 \1/') $(grep -h 'qkv' $O/s8_${v}_$i.log | sed 's/.*f16+fp8 *\([0-9.]*\) us.*/qkv \1/')"
  done
done
