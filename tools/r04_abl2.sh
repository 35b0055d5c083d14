#!/bin/bash
# round 4, session 2: the step with NO trunk work (the encoder pass of the next batch, the optimizer, the host) and without pooling
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/abl2_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
for i in 1 2; do
  for v in none trunk pool lstm,x3,attn lstm,x3,attn,pool; do
    RUART_DIAGNOSTICS=1 RUART_ABL_SKIP=$v $B > $O/abl2_${v//,/+}_$i.json 2> $O/abl2_${v//,/+}_$i.err || tail -3 $O/abl2_${v//,/+}_$i.err
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/abl2_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    except Exception as e:
        print(f, 'failed', e); continue
    acc[f.split('/')[-1].rsplit('_',1)[0][5:]].append((d['ms_per_step'], r['avg_launch_us']))
for k,v in acc.items():
    print("skip %-20s ms/step %s | timed GEMM us %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v)))
PY
