#!/bin/bash
# round 5: non-temporal loads of the once-read streams (attention Q/K/V rows, pooling layers; LN variant): encoder pass alone and the step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for v in nt0 "" ntln; do
  L=""; [ -n "$v" ] && L="build/libruart_hip_$v.so"
  echo "== library [${v:-default}]"
  RUART_HIP_LIB=$L python3 tools/encoder_kernel_times.py 2>&1 | grep "encoder pass"
done
B="python3 bench.py --no-cpu-baseline --no-bert512"
for i in 1 2 3; do
  for v in nt0 "" ntln; do
    L=""; [ -n "$v" ] && L="build/libruart_hip_$v.so"
    RUART_HIP_LIB=$L $B > $O/nt_${v:-default}_$i.json 2> $O/nt_${v:-default}_$i.err
    python3 -c "
import json
d=json.loads(open('$O/nt_${v:-default}_$i.json').read().strip().splitlines()[-1]); s=d['step_ms']; t=d['timeline_ms']
print('%-8s run $i: %.2f ms/step median %.2f | parity %s | gemm %s/%s enc_end %s bwd_end %s' % ('${v:-default}', d['ms_per_step'], s['median'], d['parity']['max_abs_err_vs_reference'], t.get('gemm_us_beside_trunk'), t.get('gemm_us_after_trunk'), t.get('encoder_last_gemm_end'), t.get('trunk_backward_end')))"
  done
done
