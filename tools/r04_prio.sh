#!/bin/bash
# round 4, session 2: trunk streams at LOW priority (hipStreamCreateWithPriority 1) beside the CU-masked encoder stream
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/prio_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512"
for i in 1 2 3; do
  for v in 0 1 -1; do
    RUART_TRUNK_PRIORITY=$v $B > $O/prio_${v}_$i.json 2> $O/prio_${v}_$i.err || tail -3 $O/prio_${v}_$i.err
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/prio_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    except Exception as e:
        print(f, 'failed', e); continue
    acc[f.split('/')[-1].rsplit('_',1)[0][5:]].append((d['ms_per_step'], r['avg_launch_us'], d['parity']['max_abs_err_vs_reference']))
for k,v in acc.items():
    print("trunk priority %-3s ms/step %s | timed GEMM us %s | parity %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v), v[0][2]))
PY
