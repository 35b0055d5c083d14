#!/bin/bash
# round 6: north-star shape (64, 512), plain f16 - stream parts and the re-timed waits
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
for i in 1 2; do
for p in 1 2 3 4; do
  for v in "" build/libruart_hip_w1.so; do
    RUART_HIP_LIB=$v timeout -k 10 120 python3 bench.py --mode bert512 --precision fp16 --steps 30 --warmup 10 --parts $p 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('parts $p lib ${v:-default}: %.3f ms frac %.4f one-pass %s gemm %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['one_pass']['ms_per_step'], d['roofline'].get('gemm_only_tflops')))"
  done
done
done
