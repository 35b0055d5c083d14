#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
RUART_UNLOCK_X3=1 RUART_HIP_LIB=build/libruart_hip_oldx3.so python3 tools/x3_step_shapes.py 2>&1 | grep -v amdgpu > $O/x3u_old.log; tail -1 $O/x3u_old.log
RUART_UNLOCK_X3=1 python3 tools/x3_step_shapes.py 2>&1 | grep -v amdgpu > $O/x3u_new.log; tail -1 $O/x3u_new.log
head -16 $O/x3u_old.log | cut -c1-150; echo; head -16 $O/x3u_new.log | cut -c1-150
