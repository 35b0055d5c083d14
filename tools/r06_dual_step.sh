#!/bin/bash
# round 6: the dual tile form inside the training step (interleaved bench runs) + its kernel tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/dual_step.log; : > $L
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "dual_form or gemm_16c" 2>&1 | tail -3 | tee -a $L || exit 1
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline --steps 40"
for i in 1 2 3; do
  for c in 0 1; do
    RUART_CORR_DUAL=$c timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dual $c: ms/step %.2f median %.2f' % (d['ms_per_step'], d['step_ms']['median']))" | tee -a $L || exit 1
  done
done
