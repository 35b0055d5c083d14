#!/bin/bash
# round 4, first GPU call: counter list, GPU tests, baseline bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
rocprofv3 -L > gpurun_out/r04/counters.txt 2>&1
python3 -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest.log 2>&1; echo "pytest rc $?" &&
tail -3 gpurun_out/r04/pytest.log &&
python3 bench.py > gpurun_out/r04/bench_base.json 2> gpurun_out/r04/bench_base.err && cut -c1-600 gpurun_out/r04/bench_base.json
