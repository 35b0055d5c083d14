#!/bin/bash
# round 6 closing check of the data-parallel control flow on a one-GPU box: (a) one rank with a real RCCL group (GradSync on RCCL, world 1),
# (b) two ranks on device 0 with gradients over gloo (bench.py's N-rank control flow; its numbers mean nothing)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/dp_check.log; : > $L
timeout -k 10 300 python3 bench.py --force-dp --steps 10 --warmup 3 --no-cpu-baseline --no-bert512 --no-roofline 2>$O/dp_a.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-dp (RCCL, 1 rank): %.1f samples/s %.2f ms parity %s' % (d['value'], d['ms_per_step'], (d.get('parity') or {}).get('max_abs_err_vs_reference')))" | tee -a $L || { tail -5 $O/dp_a.err; exit 1; }
RUART_BENCH_REHEARSE_ONE_GPU=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-bert512 --no-roofline 2>$O/dp_b.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks on one GPU (gloo rehearsal): n_gpus %d ranks_seen %d value %.1f ms %.2f' % (d['n_gpus'], d['ranks_seen'], d['value'], d['ms_per_step']))" | tee -a $L || { tail -8 $O/dp_b.err; exit 1; }
