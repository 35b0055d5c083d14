#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
python3 tools/gemm_bench.py --variants 5,7 --orders 8 --iters 20 2>&1 | grep -v amdgpu
RUART_HIP_LIB=build/libruart_hip_w4e.so python3 tools/gemm_bench.py --variants 7 --orders 8 --iters 20 2>&1 | grep -v amdgpu | sed 's/variant 7/v7 early/'
done
python3 -m pytest tests/test_gpu_kernels.py -x -q -k "one_wave" 2>&1 | tail -2
