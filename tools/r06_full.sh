#!/bin/bash
# round 6: the whole GPU suite, smoke(), and the default bench line (everything on)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?"; tail -4 $O/pytest_gpu.log
timeout -k 10 200 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout -k 10 500 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['step_ms'], d['parity'], d['roofline']['frac'], d['roofline'].get('alone',{}).get('frac'), d['bert512'].get('frac_of_peak') if d.get('bert512') else None)
PY
