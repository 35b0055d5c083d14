#!/bin/bash
# round 6: the dual (256 x 128, two workgroups per CU) form of the fp16c projections - stamps and in-process A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
if [ -f build/libruart_hip_stamps.so ]; then
  RUART_HIP_LIB=build/libruart_hip_stamps.so timeout -k 10 300 python3 tools/r06_dual_stamps.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/dual_stamps.log
fi &&
timeout -k 10 300 python3 tools/r06_dual_ab.py --rounds ${ROUNDS:-6} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/dual_ab.log
