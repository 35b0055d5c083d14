#!/bin/bash
# round 4, session 2: where do the occasional slow bench runs come from?  twelve identical runs, per-step times of each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  RUART_BENCH_STEP_TIMES=1 python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline > $O/spread_$i.json 2> $O/spread_$i.err
  python3 -c "
import json,sys
d=json.loads(open('$O/spread_$i.json').read().strip().splitlines()[-1]); print('run $i', d['ms_per_step'])"
  grep "per-step ms" $O/spread_$i.err | cut -c1-220
done
