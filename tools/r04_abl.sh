#!/bin/bash
# round 4, session 2: marginal cost of the trunk's kernel classes in the pipelined step (RUART_ABL_SKIP: launches left out, results wrong)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/abl_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
for i in 1 2; do
  for v in none lstm x3 attn lstm,x3 lstm,x3,attn; do
    RUART_DIAGNOSTICS=1 RUART_ABL_SKIP=$v $B > $O/abl_${v//,/+}_$i.json 2> $O/abl_${v//,/+}_$i.err || tail -3 $O/abl_${v//,/+}_$i.err
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/abl_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    except Exception as e:
        print(f, 'failed', e); continue
    acc[f.split('/')[-1].rsplit('_',1)[0][4:]].append((d['ms_per_step'], r['avg_launch_us']))
for k,v in acc.items():
    print("skip %-14s ms/step %s | timed GEMM us %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v)))
PY
