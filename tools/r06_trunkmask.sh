#!/bin/bash
# round 6: the trunk's three streams confined to the LAST n CUs (the 32 outside the encoder's 224-CU prefix mask + n - 32 inside it)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline --steps 40"
for i in $(seq 1 ${ROUNDS:-2}); do
  for c in ${CUS:-0 -64 -96 -128 -160}; do
    RUART_TRUNK_CUS=$c timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('trunk CUs $c: ms/step %.2f median %.2f' % (d['ms_per_step'], d['step_ms']['median']))"
  done
done
