#!/bin/bash
# round 4, session 2: mask size with the trunk at LOW priority (baseline: 240 CUs, trunk normal)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/p3_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
$B > /dev/null 2>&1     # warm the box
for i in 1 2 3 4; do
  for cfg in "240 0" "240 1" "248 1" "252 1" "244 1" "232 1"; do
    set -- $cfg
    RUART_PREFETCH_CUS=$1 RUART_TRUNK_PRIORITY=$2 $B > $O/p3_c$1_t$2_$i.json 2> $O/p3.err || tail -3 $O/p3.err
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/p3_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    acc[f.split('/')[-1].rsplit('_',1)[0][3:]].append((d['ms_per_step'], r['avg_launch_us']))
for k,v in acc.items():
    print("%-12s ms/step %s | timed GEMM us %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v)))
PY
