#!/bin/bash
# round 6: is the host's enqueue time on the step's critical path under the round-5 schedule?  A busy-wait of N us at the start of the trunk forward
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline --steps 40"
for i in 1 2; do
  for d in 0 500 1000 2000; do
    RUART_DIAGNOSTICS=1 RUART_ABL_HOST_DELAY_US=$d timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('host delay $d us: ms/step %.2f median %.2f host enqueue %.2f' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['host_enqueue_median']))"
  done
done
