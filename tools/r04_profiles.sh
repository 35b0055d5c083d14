#!/bin/bash
# round 4 closing profiles: secondary lines, kernel stats, HBM counters (progress goes to stdout; everything else under gpurun_out/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export R=r04
mkdir -p gpurun_out
bash tools/secondary_lines.sh 2>&1 | grep -v "^+" | tail -30
echo "== refresh"; bash tools/refresh_profiles.sh 2>&1 | tail -20
python3 tools/op_table.py > gpurun_out/r04_op_table_per_step.log 2>&1; grep -n "hipLaunchKernel" gpurun_out/r04_op_table_per_step.log | head -2
for d in pipe inl un; do cp $(find gpurun_out/r04f_$d -name "*kernel_stats.csv" | head -1) gpurun_out/r04_$d.kernel_stats.csv; done
rm -rf gpurun_out/r04f_pipe gpurun_out/r04f_inl gpurun_out/r04f_un gpurun_out/r04f_pf gpurun_out/r04f_pw
ls -la gpurun_out | tail -20
