#!/bin/bash
# round 6: socket power and shader clock (rocm-smi, sampled from outside) while the fp16c QKV product runs back to back - the product, and the
# diagnostic builds with one matrix instruction in both runs of the K loop (build/libruart_hip_st_allf8.so / st_allf16.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/power.log; : > $L
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" | head -8 | tee -a $L
for v in stamps st_allf8 st_allf16; do
  echo "== $v" | tee -a $L
  RUART_HIP_LIB=build/libruart_hip_$v.so SECONDS_=9 timeout -k 10 120 python3 tools/r06_gemm_loop.py 2>/dev/null >> $L &
  pid=$!
  sleep 5
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -i "package power\|sclk" | tr '\n' ' ' | tee -a $L; echo | tee -a $L; sleep 1; done
  wait $pid || exit 1
  tail -1 $L
done
