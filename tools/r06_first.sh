#!/bin/bash
# round 6, first call: the three-waits-per-K-tile schedule (RUART_P8_WAITS=1, the default build) against the one-wait form
# (build/libruart_hip_w0.so): correctness, race screen, kernel A/B, step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
set -o pipefail
timeout -k 10 300 python3 tools/gemm_check.py > $O/gemm_check.log 2>&1; echo "gemm_check rc $?"; tail -3 $O/gemm_check.log
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc $?"; tail -3 $O/pytest_gemm.log
timeout -k 10 300 python3 tools/gemm_race_screen.py 40 > $O/race_screen.log 2>&1; echo "race screen rc $?"; grep -c OK $O/race_screen.log; grep BAD $O/race_screen.log
for i in 1 2 3; do
  timeout -k 10 120 python3 tools/gemm_corr_bench.py --rows 42752 > $O/corr_new_$i.log 2>&1; tail -1 $O/corr_new_$i.log
  RUART_HIP_LIB=build/libruart_hip_w0.so timeout -k 10 120 python3 tools/gemm_corr_bench.py --rows 42752 > $O/corr_w0_$i.log 2>&1; tail -1 $O/corr_w0_$i.log
done
cat $O/corr_new_2.log; cat $O/corr_w0_2.log
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity --no-roofline --steps 60"
for i in 1 2 3; do
  timeout -k 10 200 $B > $O/step_new_$i.json 2> $O/step_new_$i.err; python3 -c "import json;d=json.loads(open('$O/step_new_$i.json').read().strip().splitlines()[-1]);print('new',d['ms_per_step'],d['step_ms']['median'])"
  RUART_HIP_LIB=build/libruart_hip_w0.so timeout -k 10 200 $B > $O/step_w0_$i.json 2> $O/step_w0_$i.err; python3 -c "import json;d=json.loads(open('$O/step_w0_$i.json').read().strip().splitlines()[-1]);print('w0 ',d['ms_per_step'],d['step_ms']['median'])"
done
