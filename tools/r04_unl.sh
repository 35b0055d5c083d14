#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_sdnet.py -x -q -k "attn or attention or unlocked or sdnet_forward" 2>&1 | tail -3
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
$B --unlock-bert --train-gemm x3 --steps 4 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('unlock x3', d['value'], d['ms_per_step'])"
$B --unlock-bert --train-gemm 16gemm --steps 4 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('unlock 16gemm', d['value'], d['ms_per_step'])"
$B --unlock-bert --steps 8 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('unlock 16', d['value'], d['ms_per_step'])"
$B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
