#!/bin/bash
# round 6: socket power and clock beside the plain f16 QKV product - the product and its loop ablations (ab1 no prefetch issue in the loop,
# ab2 no fragment reads after the first K-tile, ab3 both: wrong numbers, right timing); K = 768 and K = 3072 (longer loops, same epilogue)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/power16.log; : > $L
for k in 768 3072; do
for v in "" ab1 ab2 ab3; do
  lib=ruart_amd/libruart_hip.so; [ -n "$v" ] && lib=build/libruart_hip_$v.so
  K_=$k RUART_HIP_LIB=$lib SECONDS_=7 timeout -k 10 120 python3 tools/r06_gemm16_loop.py 2>/dev/null >> $L &
  pid=$!
  sleep 4
  for i in 1 2; do rocm-smi --showpower --showclocks 2>&1 | grep -i "package power\|sclk" | tr '\n' ' ' | sed 's/GPU\[0\]\t*: //g' >> $L; echo >> $L; sleep 1; done
  wait $pid || exit 1
done
done
cat $L
