#!/bin/bash
# round 4: trunk GEMM (ruart_gemm_x3) with branch-free loads + two-step prefetch against round 3's kernel: parity tests, per-shape times, step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 -m pytest tests/test_gpu_kernels.py -x -q -k "x3 or gemm or linear or mm" > $O/x3_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/x3_tests.log
RUART_HIP_LIB=build/libruart_hip_oldx3.so python3 tools/x3_step_shapes.py > $O/x3_shapes_old.log 2>&1; tail -1 $O/x3_shapes_old.log
python3 tools/x3_step_shapes.py > $O/x3_shapes_new.log 2>&1; tail -1 $O/x3_shapes_new.log
B="python3 bench.py --no-cpu-baseline --no-bert512"
for i in 1 2; do
  RUART_HIP_LIB=build/libruart_hip_oldx3.so $B > $O/x3_step_old_$i.json 2>/dev/null; $B > $O/x3_step_new_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/x3_step_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], 'timed avg us', r['avg_launch_us'], 'alone', r['alone']['avg_launch_us'], 'parity', d['parity']['max_abs_err_vs_reference'])
PY
