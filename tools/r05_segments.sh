#!/bin/bash
# round 5: the trunk's single-stream pieces as captured graphs (RUART_GRAPH_SEGMENTS=1) against the eager trunk, by trunk stream priority
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for cfg in "seg_low:RUART_GRAPH_SEGMENTS=1" "seg_prio0:RUART_GRAPH_SEGMENTS=1 RUART_TRUNK_PRIORITY=0" "seg_high:RUART_GRAPH_SEGMENTS=1 RUART_TRUNK_PRIORITY=-1" "eager_prio0:RUART_TRUNK_PRIORITY=0" "eager_low:RUART_X=0" "seg_prio0_1s:RUART_GRAPH_SEGMENTS=1 RUART_TRUNK_PRIORITY=0 RUART_STREAMS=0"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline --warmup 8 --steps 40 > $O/$n.json 2> $O/$n.err || tail -5 $O/$n.err
  python3 -c "
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('%-14s %.2f ms median %.2f host enqueue %.2f' % ('$n', d['ms_per_step'], d['step_ms']['median'], d['step_ms']['host_enqueue_median']))"
done
