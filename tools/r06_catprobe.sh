#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/cat_probe.log; : > $L
for i in 1 2 3; do
  for c in 0 1; do
    CATPROBE=$c timeout -k 10 200 python3 tools/r06_cat_probe.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline --steps 40 2>$O/cat_probe.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cat probe $c: ms/step %.2f median %.2f' % (d['ms_per_step'], d['step_ms']['median']))" | tee -a $L || { tail -5 $O/cat_probe.err; exit 1; }
    tail -1 $O/cat_probe.err | tee -a $L
  done
done
