#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/dpab_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
for i in 1 2 3; do
  for cfg in "0 " "1 " "0 --force-dp" "1 --force-dp"; do
    set -- $cfg
    RUART_TRUNK_PRIORITY=$1 $B $2 > $O/dpab_t$1_${2:-nodp}_$i.json 2> $O/dpab.err || tail -3 $O/dpab.err
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/dpab_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    acc[f.split('/')[-1].rsplit('_',1)[0][5:]].append(d['ms_per_step'])
for k,v in acc.items():
    print("%-16s ms/step %s" % (k, ' '.join('%.2f'%x for x in v)))
PY
