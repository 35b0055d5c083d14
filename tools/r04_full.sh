#!/bin/bash
# full GPU suite with live progress (a silent run is killed after 7 minutes), then the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tee $O/full_tests.log | grep -v "^$" | tail -15
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
