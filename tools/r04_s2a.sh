#!/bin/bash
# round 4, session 2: variational-dropout masks read inside the trunk GEMMs' operand loads (byte masks) - tests, per-product times, step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 -m pytest tests/test_gpu_kernels.py -x -q -k "x3 or gemm_x or linear or mm" > $O/s2a_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/s2a_tests.log
python3 tools/mask_bench.py > $O/s2a_mask_bench.log 2>&1; cat $O/s2a_mask_bench.log
python3 tools/op_sites2.py --top 120 > $O/s2a_op_sites2.log 2>&1; tail -5 $O/s2a_op_sites2.log
B="python3 bench.py --no-cpu-baseline --no-bert512"
for i in 1 2 3; do
  RUART_FUSE_MASK=0 $B > $O/s2a_step_unfused_$i.json 2>/dev/null; $B > $O/s2a_step_fused_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/s2a_step_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], 'timed avg us', r['avg_launch_us'], 'alone', r['alone']['avg_launch_us'], 'parity', d['parity']['max_abs_err_vs_reference'])
PY
