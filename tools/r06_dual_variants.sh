#!/bin/bash
# round 6: stamps of diagnostic builds of the dual form (build/libruart_hip_st_*.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; : > $O/dual_variants.log
for v in ${VARIANTS:-stamps st_p1 st_p2 st_b2}; do
  echo "== $v" | tee -a $O/dual_variants.log
  RUART_HIP_LIB=build/libruart_hip_$v.so timeout -k 10 200 python3 tools/r06_dual_stamps.py 2>&1 | grep -A2 "dual:" | tee -a $O/dual_variants.log || exit 1
done
