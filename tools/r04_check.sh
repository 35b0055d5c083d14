#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/check_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/check_tests.log
python3 tools/x3_step_shapes.py > $O/x3_shapes_new2.log 2>&1; tail -1 $O/x3_shapes_new2.log
B="python3 bench.py --no-cpu-baseline --no-bert512"
for i in 1 2; do
  RUART_HIP_LIB=build/libruart_hip_oldx3.so $B > $O/ck_step_old_$i.json 2>/dev/null; $B > $O/ck_step_new_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/ck_step_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], 'timed avg us', r['avg_launch_us'], 'alone', r['alone']['avg_launch_us'], 'parity', d['parity']['max_abs_err_vs_reference'])
PY
