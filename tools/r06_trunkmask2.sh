#!/bin/bash
# round 6: the trunk's suffix CU mask as the schedule default - sessions with close() and an evaluation in between (hardware queue slots,
# HISTORY round 5 (9)), the stream tests, and a profiler-attached bench (teardown with the masked streams destroyed by close(final=True))
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O; L=$O/trunk_mask_sessions.log; : > $L
echo "== default (trunk on the LAST 160 CUs), evaluation between sessions" | tee -a $L
EVAL_BETWEEN=1 timeout -k 10 300 python3 tools/r05_two_sessions.py 2>&1 | grep -v Warning | tee -a $L &&
echo "== RUART_TRUNK_CUS=0 (unmasked trunk), evaluation between sessions" | tee -a $L &&
RUART_TRUNK_CUS=0 EVAL_BETWEEN=1 timeout -k 10 300 python3 tools/r05_two_sessions.py 2>&1 | grep -v Warning | tee -a $L &&
echo "== default, evaluation FIRST" | tee -a $L &&
EVAL_FIRST=1 SESSIONS=1 timeout -k 10 300 python3 tools/r05_two_sessions.py 2>&1 | grep -v Warning | tee -a $L &&
timeout -k 10 600 python3 -m pytest tests/test_gpu_streams.py -x -q 2>&1 | tail -3 | tee -a $L &&
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_mask -o mask -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-bert512 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_mask_bench.json 2> $GRAFT_REPO_ROOT/$O/prof_mask_bench.err; echo "rocprof bench rc $?" | tee -a $GRAFT_REPO_ROOT/$L; tail -c 600 $GRAFT_REPO_ROOT/$O/prof_mask_bench.json)
