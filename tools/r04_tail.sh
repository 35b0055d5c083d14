#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/tail_*.json
python3 -m pytest tests/test_gpu_kernels.py -x -q -k "tail" > $O/tail_ktests.log 2>&1; echo "kernel tests rc $?"; tail -4 $O/tail_ktests.log
B="python3 bench.py --no-cpu-baseline"
for i in 1 2 3; do
  RUART_TAIL_CUS=0 $B > $O/tail_off_$i.json 2>/dev/null; $B > $O/tail_on_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/tail_o*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; b=d['bert512']
    acc[f.split('/')[-1].rsplit('_',1)[0]].append((d['ms_per_step'], r['avg_launch_us'], r['alone']['avg_launch_us'], d['parity']['max_abs_err_vs_reference'], b['ms'], b['one_pass_ms'], b['gemm_only_tflops']))
for k,v in acc.items():
    print("%-9s ms/step %s | timed GEMM us %s | alone %s | parity %s | bert512 ms %s one-pass %s gemm TF %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v), ' '.join('%.0f'%x[2] for x in v), v[0][3], ' '.join('%.2f'%x[4] for x in v), ' '.join('%.2f'%x[5] for x in v), ' '.join('%.0f'%x[6] for x in v)))
PY
python3 -m pytest tests -m gpu -x -q > $O/tail_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tail_tests.log
