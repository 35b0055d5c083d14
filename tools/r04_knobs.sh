#!/bin/bash
# round 4, session 2: the older trunk knobs again, now that the trunk runs at LOW priority (split-K fill target, one trunk stream)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/kn_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
$B > /dev/null 2>&1
for i in 1 2; do
  for cfg in "448 160 1" "224 160 1" "128 80 1" "64 40 1" "448 160 0" "896 320 1"; do
    set -- $cfg
    RUART_X3_FILL=$1 RUART_X3_NOSPLIT_TILES=$2 RUART_STREAMS=$3 $B > $O/kn_f$1_n$2_s$3_$i.json 2> $O/kn.err || tail -3 $O/kn.err
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/kn_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    acc[f.split('/')[-1].rsplit('_',1)[0][3:]].append((d['ms_per_step'], r['avg_launch_us']))
for k,v in acc.items():
    print("%-18s ms/step %s | timed GEMM us %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v)))
PY
