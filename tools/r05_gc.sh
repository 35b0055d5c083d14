#!/bin/bash
# round 5: is the one-off ~37 ms step a full garbage collection of the interpreter?  bench with / without gc.freeze() after set-up
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline"
for i in 1 2 3; do
  for f in 0 1; do
    RUART_BENCH_GC_FREEZE=$f $B --steps 60 > $O/gc_f${f}_$i.json 2> $O/gc_f${f}_$i.err
    python3 -c "
import json
d=json.loads(open('$O/gc_f${f}_$i.json').read().strip().splitlines()[-1]); s=d['step_ms']
print('freeze=$f run $i: %.2f ms/step  median %.2f p90 %.2f max %.2f first %.2f  gc %s' % (d['ms_per_step'], s['median'], s['p90'], s['max'], s['first'], s['gc']))"
  done
done
