#!/bin/bash
# round 6: occupancy of the long-sequence attention kernel (RUART_ATTN_LONG_WPS = 2 default / 3 / 4) on the north-star shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for v in "" build/libruart_hip_al3.so build/libruart_hip_al4.so; do
    RUART_HIP_LIB=$v timeout -k 10 120 python3 bench.py --mode bert512 --precision fp16 --steps 30 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib ${v:-default}: %.3f ms frac %.4f one-pass %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['one_pass']['ms_per_step']))"
  done
done
