#!/bin/bash
# round 5: per-kernel times of one frozen-encoder pass alone on the device, LayerNorm fold on / off (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/fold; mkdir -p $O
for f in 1 0; do
  export RUART_LN_FOLD=$f
  python3 tools/encoder_kernel_times.py > $O/pass_$f.log 2>&1; tail -1 $O/pass_$f.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$f -o p -- python3 tools/encoder_kernel_times.py > $O/prof_$f.log 2>&1
  python3 - $O/prof_$f <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:9]:
    print("%5d  avg %8.1f us  total %9.1f us  %s" % (int(r['Calls']), float(r['AverageNs']) / 1e3, int(r['TotalDurationNs']) / 1e3, r['Name'][:90]))
PY
done
