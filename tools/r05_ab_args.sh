#!/bin/bash
# round 5: interleaved A/B of bench ARGUMENTS (and environment) on the pipelined step:  tools/r05_ab_args.sh "name|ENV=val ...|--flag ..." ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/ab; mkdir -p $O
ROUNDS=${ROUNDS:-3}
for i in $(seq 1 $ROUNDS); do
  for cfg in "$@"; do
    IFS='|' read -r name envs args <<< "$cfg"
    env $envs python3 bench.py --no-cpu-baseline --no-bert512 --steps ${STEPS:-60} $args > $O/${name}_$i.json 2> $O/${name}_$i.err || { echo "$name run $i failed"; tail -3 $O/${name}_$i.err; }
  done
done
python3 - "$@" <<'PY'
import json, glob, sys, statistics as st
for cfg in sys.argv[1:]:
    name = cfg.split('|')[0]
    med, mean, tl = [], [], []
    for f in sorted(glob.glob('gpurun_out/r05/ab/%s_[0-9].json' % name)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        med.append(d['step_ms']['median']); mean.append(d['ms_per_step'])
        t = d.get('timeline_ms') or {}
        tl.append("%s/%s" % (t.get('optimizer_end'), t.get('encoder_pass_end')))
    if med:
        print("%-22s median-of-medians %.2f  (medians %s | means %s | trunk/encoder end %s)" % (name, st.median(med), ' '.join('%.2f' % x for x in med), ' '.join('%.2f' % x for x in mean), ' '.join(tl)))
PY
