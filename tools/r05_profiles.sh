#!/bin/bash
# round 5 closing profiles: kernel stats (pipelined / inline / trained encoder), HBM counters, op table (outputs under gpurun_out/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export R=r05
mkdir -p gpurun_out
echo "== refresh"; bash tools/refresh_profiles.sh 2>&1 | tail -20
python3 tools/op_table.py > gpurun_out/r05_op_table_per_step.log 2>&1; grep -n "hipLaunchKernel" gpurun_out/r05_op_table_per_step.log | head -2
for d in pipe inl un; do cp $(find gpurun_out/r05f_$d -name "*kernel_stats.csv" | head -1) gpurun_out/r05_$d.kernel_stats.csv; done
rm -rf gpurun_out/r05f_pipe gpurun_out/r05f_inl gpurun_out/r05f_un gpurun_out/r05f_pf gpurun_out/r05f_pw
ls -la gpurun_out | grep r05 | tail -20
