#!/bin/bash
# round 6: the plain 16-bit folded pass - kernel tests, the encoder against the reference's layer outputs, bert512 folded against unfolded
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_16_fold or (bert_encoder_vs_reference_golden and fold)" > $O/pytest_fold16.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_fold16.log
for i in 1 2 3; do
  for f in 1 0; do
    RUART_LN_FOLD=$f timeout -k 10 120 python3 bench.py --mode bert512 --precision fp16 --steps 30 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fold $f: %.3f ms frac %.4f one-pass %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['one_pass']['ms_per_step']))"
  done
done
