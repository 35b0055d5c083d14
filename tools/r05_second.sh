#!/bin/bash
# round 5, call 2: trunk graphs through torch today (graph_timing), the step with them, and a long run's step distribution
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', d['ms_per_step'], {k:v for k,v in d['step_ms'].items() if k!='what'}, {k:v for k,v in (d['timeline_ms'] or {}).items() if k!='what'}, d['roofline'] and d['roofline']['avg_launch_us'])"; }
timeout -k 10 400 python3 tools/graph_timing.py > $O/graph_timing.log 2>&1; grep "graph=" $O/graph_timing.log
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
$B --steps 300 --no-roofline > $O/long300.json 2> $O/long300.err; python3 -c "
import json
d=json.loads(open('$O/long300.json').read().strip().splitlines()[-1]); print('300 steps', d['ms_per_step'], d['step_ms'])"
RUART_BENCH_STEP_TIMES=1 $B --steps 300 --no-roofline 2>&1 >/dev/null | grep "per-step" > $O/long300_steps.log
$B --graph-trunk 1 > $O/graphtrunk1.json 2> $O/graphtrunk1.err && show $O/graphtrunk1.json "graph-trunk 1:"
$B > $O/graphtrunk0.json 2> $O/graphtrunk0.err && show $O/graphtrunk0.json "eager:"
RUART_STREAMS=0 $B --graph-trunk 1 > $O/graphtrunk1_1s.json 2> $O/graphtrunk1_1s.err && show $O/graphtrunk1_1s.json "graph-trunk 1, one stream:"
