#!/bin/bash
# round 6: kernel trace of the pipelined step (timed schedule) for offline critical-path analysis (tools/r06_trace_chain.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-roofline --no-parity --no-bert512"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pipe -o p -- $B --steps 8 --warmup 3 > $O/trace_pipe.log 2>&1; echo "rc $?"
grep -h ms_per_step $O/trace_pipe.log | cut -c1-160
T=$(find $O/trace_pipe -name "*kernel_trace.csv" | head -1); ls -la $T
python3 - "$T" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("Start_Timestamp", "End_Timestamp", "Kernel_Name", "Queue_Id", "Stream_Id", "Grid_Size", "Workgroup_Size")
with gzip.open("gpurun_out/r06/trace_pipe_kernels.csv.gz", "wt") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows: w.writerow([r.get(k, "")[:90] for k in keep])
print("rows", len(rows))
PY
cp $(find $O/trace_pipe -name "*kernel_stats.csv" | head -1) $O/r06_bench_pipelined_kernel_stats.csv; rm -rf $O/trace_pipe
