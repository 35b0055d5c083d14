#!/bin/bash
# round 6: encoder prefix mask x trunk suffix mask
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFGS=${CFGS:-224:0 224:-160 216:-160 232:-160 240:-160 232:-176 240:-192 248:-192}
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline --steps 40"
for i in $(seq 1 ${ROUNDS:-2}); do
  for cfg in $CFGS; do
    e=${cfg%%:*}; t=${cfg#*:}
    RUART_PREFETCH_CUS=$e RUART_TRUNK_CUS=$t timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('encoder $e trunk $t: ms/step %.2f median %.2f' % (d['ms_per_step'], d['step_ms']['median']))"
  done
done
