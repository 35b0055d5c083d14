#!/bin/bash
# round 6: the persistent work-queue form of the split attention kernel - bit equality test, then time per call for every form
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/attn_pw.log; : > $L
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "attention_split" 2>&1 | tail -3 | tee -a $L
grep -q passed $L || exit 1
timeout -k 10 300 python3 tools/attn_split_bench.py --heads=2,-2,-3,-4,-6,2,-2 --reps 30 2>&1 | grep -v "Warning\|amdgpu" | tee -a $L
