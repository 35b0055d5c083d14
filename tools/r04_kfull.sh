#!/bin/bash
# round 4, session 2: gemm_x3 with the KFULL fast loops (no bounds select inside K) against the previous library: tests, per-shape times, step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 -m pytest tests/test_gpu_kernels.py -x -q -k "x3 or gemm_x or linear or mm or lstm or attention" > $O/kf_tests.log 2>&1; echo "tests rc $?"; tail -2 $O/kf_tests.log
RUART_HIP_LIB=build/libruart_hip_prekfull.so python3 tools/x3_step_shapes.py > $O/kf_shapes_old.log 2>&1; tail -1 $O/kf_shapes_old.log
python3 tools/x3_step_shapes.py > $O/kf_shapes_new.log 2>&1; tail -1 $O/kf_shapes_new.log
bash tools/r04_ab.sh prekfull cur
