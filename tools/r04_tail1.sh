#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 tools/tail_split_one.py 2>&1 | grep -v amdgpu.ids
python3 tools/tail_split_one.py --K 768 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t1 -o p -- python3 tools/tail_split_one.py > $O/t1.log 2>&1
find $O/t1 -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-60,170-330
find $O/t1 -name "*kernel_trace.csv" -exec python3 - {} \; <<'PY'
PY
rm -rf $O/t1
