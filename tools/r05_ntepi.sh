#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for r in 1 2; do for v in "" ntepi1 ntepi3 ntepi8 ntepi11; do
  L=""; [ -n "$v" ] && L="build/libruart_hip_$v.so"
  echo -n "[${v:-default}] "; RUART_HIP_LIB=$L python3 tools/encoder_kernel_times.py 2>&1 | grep "encoder pass"
done; done
python3 bench.py --no-cpu-baseline --no-bert512 > $O/bench_tl.json 2> $O/bench_tl.err; python3 -c "
import json
d=json.loads(open('$O/bench_tl.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:v for k,v in d['step_ms'].items() if k not in ('what','gc')}); print({k:v for k,v in d['timeline_ms'].items() if k!='what'}); print(d['roofline']['timed_gemm_us'], d['roofline']['avg_launch_us'])"
