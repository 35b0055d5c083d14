#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/mask_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
for i in 1 2; do
  for v in 240 0 252 248 232 224; do
    RUART_PREFETCH_CUS=$v $B > $O/mask_${v}_$i.json 2>/dev/null
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/mask_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    acc[f.split('/')[-1].rsplit('_',1)[0][5:]].append((d['ms_per_step'], r['avg_launch_us'], r['alone']['avg_launch_us']))
for k,v in acc.items():
    print("mask %-5s ms/step %s | timed GEMM us %s | alone %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v), ' '.join('%.0f'%x[2] for x in v)))
PY
python3 tools/gemm_corr_order_sweep.py --hidden 1024 --inter 4096 --rows 121344 --iters 3 --orders 2,4,6,8,12,16 --auto > $O/order_sweep_stress.log 2>&1; cat $O/order_sweep_stress.log | tail -8
python3 tools/gemm_corr_order_sweep.py --iters 3 --orders 4,6,8,16 --auto > $O/order_sweep_base.log 2>&1; tail -5 $O/order_sweep_base.log
