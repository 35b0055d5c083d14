#!/bin/bash
# round 6: the trained-encoder step (f3) with library variants, interleaved (LIBS="name name ..." under build/, "product" = the tree's library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/f3_${TAG:-ab}.log; : > $L
for i in $(seq 1 ${ROUNDS:-2}); do
  for v in ${LIBS:-product ln768 ln1024}; do
    lib=ruart_amd/libruart_hip.so; [ $v != product ] && lib=build/libruart_hip_$v.so
    RUART_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --unlock-bert --train-gemm 16 --steps 8 --warmup 3 --no-cpu-baseline --no-bert512 --no-parity 2>$O/f3.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: %.1f samples/s %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a $L || { tail -5 $O/f3.err; exit 1; }
  done
done
