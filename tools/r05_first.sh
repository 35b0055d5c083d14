#!/bin/bash
# round 5, call 1: the new bench line (step_ms, timeline_ms) at trunk priority LOW (default) / NORMAL, interleaved; then the graph probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-bert512"
for i in 1 2; do
  for v in 1 0; do
    RUART_TRUNK_PRIORITY=$v RUART_BENCH_STEP_TIMES=1 $B > $O/first_p${v}_$i.json 2> $O/first_p${v}_$i.err || tail -3 $O/first_p${v}_$i.err
    tail -c 1500 $O/first_p${v}_$i.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prio $v run $i:', d['ms_per_step'], d['step_ms'], d['timeline_ms'], d['roofline']['avg_launch_us'], d['roofline']['alone']['avg_launch_us'])" 2>/dev/null || python3 -c "
import json
d=json.loads(open('$O/first_p${v}_$i.json').read().strip().splitlines()[-1])
print('prio $v run $i:', d['ms_per_step'], d['step_ms'], d['timeline_ms'], d['roofline']['avg_launch_us'], d['roofline']['alone']['avg_launch_us'])"
  done
done
timeout -k 10 300 python3 tools/r05_graph_probe.py > $O/graph_probe_1s.log 2>&1; tail -12 $O/graph_probe_1s.log
timeout -k 10 300 python3 tools/r05_graph_probe.py --streams > $O/graph_probe_3s.log 2>&1; tail -12 $O/graph_probe_3s.log
