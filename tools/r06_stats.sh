#!/bin/bash
# round 6: row statistics finished ahead of the K loop (product) against at the head of the epilogue (build/libruart_hip_late.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python3 tools/r06_stats_ab.py build/libruart_hip_late.so ruart_amd/libruart_hip.so --rounds ${ROUNDS:-10} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/stats_ab.log &&
timeout -k 10 300 python3 tools/r06_dual_ab.py --rounds 6 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/dual_ab.log
