#!/bin/bash
# round 6: pure MFMA streams at the power cap, per instruction type (build/r06_mfma_power from tools/r06_mfma_power.hip) with rocm-smi sampled beside them
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/mfma_power.log; : > $L
for v in 0 1 2 3 4; do
  timeout -k 10 60 build/r06_mfma_power 6 $v >> $L 2>&1 &
  pid=$!
  sleep 3
  for i in 1 2; do rocm-smi --showpower --showclocks 2>&1 | grep -i "package power\|sclk" | tr '\n' ' ' | sed 's/GPU\[0\]\t*: //g' >> $L; echo >> $L; sleep 1; done
  wait $pid || exit 1
done
cat $L
