#!/bin/bash
# round 4, session 2: is the host's enqueue time of the trunk forward on the step's critical path?  A busy-wait of N us at the start of the trunk
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/hd_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
$B > /dev/null 2>&1
for i in 1 2; do
  for d in 0 1000 2000 4000; do
    RUART_DIAGNOSTICS=1 RUART_ABL_HOST_DELAY_US=$d $B > $O/hd_${d}_$i.json 2> $O/hd.err || tail -3 $O/hd.err
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/hd_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    acc[int(f.split('/')[-1].split('_')[1])].append((d['ms_per_step'], r['avg_launch_us']))
for k in sorted(acc):
    v=acc[k]; print("host delay %5d us: ms/step %s | timed GEMM us %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v)))
PY
