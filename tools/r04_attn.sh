#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_sdnet.py -x -q > $O/attn_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/attn_tests.log
B="python3 bench.py --no-cpu-baseline --no-bert512"
for i in 1 2; do
  RUART_HIP_LIB=build/libruart_hip_oldx3.so $B > $O/at_step_old_$i.json 2>/dev/null; $B > $O/at_step_new_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/at_step_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], 'timed avg us', r['avg_launch_us'], 'alone', r['alone']['avg_launch_us'], 'parity', d['parity']['max_abs_err_vs_reference'])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inl -o p -- python3 bench.py --no-cpu-baseline --no-roofline --no-parity --no-bert512 --no-prefetch --steps 5 --warmup 2 > $O/inl.log 2>&1
cp $O/inl/*/p_kernel_stats.csv $O/inline_kernel_stats.csv 2>/dev/null || cp $O/inl/p_kernel_stats.csv $O/inline_kernel_stats.csv; rm -rf $O/inl
head -40 $O/inline_kernel_stats.csv | cut -c1-150
