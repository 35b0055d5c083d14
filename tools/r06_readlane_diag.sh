#!/bin/bash
# round 6: the removed readlane form of the pooling kernel under the builder's own probe (tools/r05_race_probe.py, TRAIN_FIRST=8: a second
# trainer after a first one has trained; three-stream against one-stream forward), ONE run of N = 8 comparisons per form:
#   POOLVAR=1 the shipped scalar-load form | 10 the removed form | 11 + a full vmcnt(0) behind the statistics load
#   12 row loads with the default cache policy | 13 the pairs through ds_bpermute instead of v_readlane
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
export RUART_HIP_LIB=build/libruart_hip_rl.so
for v in ${FORMS:-1 10 11 12 13}; do
  echo "== POOLVAR=$v" | tee -a $O/readlane_diag.log
  TRAIN_FIRST=8 N=8 POOLVAR=$v timeout -k 10 200 python3 tools/r05_race_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/readlane_diag.log
done
