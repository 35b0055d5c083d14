#!/bin/bash
# A/B of library variants in the pipelined step: tools/r04_ab.sh name1 name2 ...  (build/libruart_hip_<name>.so; "cur" = the product library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O; rm -f $O/ab_*.json
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
for i in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = cur ]; then $B > $O/ab_${v}_$i.json 2>/dev/null; else RUART_HIP_LIB=build/libruart_hip_$v.so $B > $O/ab_${v}_$i.json 2>/dev/null; fi
  done
done
python3 - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r04/ab_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    acc[f.split('/')[-1].rsplit('_',1)[0][3:]].append((d['ms_per_step'], r['avg_launch_us'], r['alone']['avg_launch_us']))
for k,v in acc.items():
    print("%-8s ms/step %s | timed GEMM us %s | alone %s" % (k, ' '.join('%.2f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v), ' '.join('%.0f'%x[2] for x in v)))
PY
