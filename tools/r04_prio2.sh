#!/bin/bash
# round 4, session 2: encoder stream {masked 240, unmasked, unmasked + HIGH} x trunk {normal, LOW}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/p2_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512"
for i in 1 2; do
  for enc in "240 0" "0 0" "0 -1" "248 0" "224 0"; do
    set -- $enc
    for tp in 0 1; do
      RUART_PREFETCH_CUS=$1 RUART_ENCODER_PRIORITY=$2 RUART_TRUNK_PRIORITY=$tp $B > $O/p2_c$1_e$2_t${tp}_$i.json 2> $O/p2.err || tail -3 $O/p2.err
    done
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/p2_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    except Exception as e:
        print(f, 'failed', e); continue
    acc[f.split('/')[-1].rsplit('_',1)[0][3:]].append((d['ms_per_step'], r['avg_launch_us']))
for k,v in acc.items():
    print("%-16s ms/step %s | timed GEMM us %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v)))
PY
