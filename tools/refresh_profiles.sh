#!/bin/bash
# Regenerate the round's committed profiler summaries from the current tree (run on the GPU box; outputs under gpurun_out/${R}f_*).
R=${R:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-roofline --no-parity --no-bert512"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}f_pipe -o p -- $B --steps 8 --warmup 3 > gpurun_out/${R}f_pipe.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}f_inl -o p -- $B --steps 5 --warmup 2 --no-prefetch > gpurun_out/${R}f_inl.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}f_un -o p -- $B --unlock-bert --steps 6 --warmup 3 > gpurun_out/${R}f_un.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}f_pf -o p -- $B --steps 3 --warmup 1 --no-prefetch > gpurun_out/${R}f_pf.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}f_pw -o p -- $B --steps 3 --warmup 1 --no-prefetch > gpurun_out/${R}f_pw.log 2>&1 &&
python3 tools/pmc_hbm_table.py gpurun_out/${R}f_pf/p_counter_collection.csv gpurun_out/${R}f_pw/p_counter_collection.csv gpurun_out/${R}f_inl/p_kernel_stats.csv > gpurun_out/${R}f_pmc_hbm_per_kernel.csv &&
python3 tools/gemm_traffic2.py gpurun_out/${R}f_pmc_hbm_per_kernel.csv gpurun_out/${R}f_gemm_traffic.json 42752 &&
grep -h ms_per_step gpurun_out/${R}f_pipe.log gpurun_out/${R}f_inl.log gpurun_out/${R}f_un.log | cut -c1-200 && head -12 gpurun_out/${R}f_pmc_hbm_per_kernel.csv && cat gpurun_out/${R}f_gemm_traffic.json
