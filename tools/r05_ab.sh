#!/bin/bash
# round 5: interleaved A/B of environment knobs on the pipelined step; 60 timed steps per run, ROUNDS rounds, median of the runs' medians
#   tools/r05_ab.sh "name1:VAR=val VAR2=val" "name2:" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/ab; mkdir -p $O
ROUNDS=${ROUNDS:-3}
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline --steps ${STEPS:-60}"
for i in $(seq 1 $ROUNDS); do
  for cfg in "$@"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    env $envs $B $EXTRA > $O/${name}_$i.json 2> $O/${name}_$i.err || { echo "$name run $i failed"; tail -3 $O/${name}_$i.err; }
  done
done
python3 - "$@" <<'PY'
import json, glob, sys, statistics as st
for cfg in sys.argv[1:]:
    name = cfg.split(':')[0]
    med, mean = [], []
    for f in sorted(glob.glob('gpurun_out/r05/ab/%s_[0-9].json' % name)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        med.append(d['step_ms']['median']); mean.append(d['ms_per_step'])
    if med:
        print("%-28s median-of-medians %.2f  (medians %s | means %s)" % (name, st.median(med), ' '.join('%.2f' % x for x in med), ' '.join('%.2f' % x for x in mean)))
PY
