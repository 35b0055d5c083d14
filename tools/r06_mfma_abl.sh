#!/bin/bash
# round 6: which property of the f16 run makes its K-tile longer than the fp8 run's - stamps of builds with one MFMA instruction in both runs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; : > $O/mfma_abl.log
for v in stamps st_allf8 st_allf16 stamps; do
  echo "== $v" | tee -a $O/mfma_abl.log
  RUART_HIP_LIB=build/libruart_hip_$v.so timeout -k 10 200 python3 tools/r06_corr_stamps.py 2>&1 | grep "tiles\|K loop of wave" | tee -a $O/mfma_abl.log || exit 1
done
