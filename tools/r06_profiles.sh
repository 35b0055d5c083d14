#!/bin/bash
# round 6 closing profiles: kernel stats (pipelined / inline), HBM counters per kernel, GEMM traffic, op table (outputs under gpurun_out/r06p/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06p; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-roofline --no-parity --no-bert512"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -o p -- $B --steps 8 --warmup 3 > $O/pipe.log 2>&1 && echo pipe ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inl -o p -- $B --steps 5 --warmup 2 --no-prefetch > $O/inl.log 2>&1 && echo inl ok &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -o p -- $B --steps 3 --warmup 1 --no-prefetch > $O/pf.log 2>&1 && echo pf ok &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -o p -- $B --steps 3 --warmup 1 --no-prefetch > $O/pw.log 2>&1 && echo pw ok &&
python3 tools/pmc_hbm_table.py $O/pf/p_counter_collection.csv $O/pw/p_counter_collection.csv $O/inl/p_kernel_stats.csv > $O/r06_pmc_hbm_per_kernel.csv &&
python3 tools/gemm_traffic2.py $O/r06_pmc_hbm_per_kernel.csv $O/r06_gemm_traffic.json 42752
cp $(find $O/pipe -name "*kernel_stats.csv" | head -1) $O/r06_bench_pipelined_kernel_stats.csv
cp $(find $O/inl -name "*kernel_stats.csv" | head -1) $O/r06_bench_inline_kernel_stats.csv
P="python3 bench.py --no-cpu-baseline --no-roofline --no-parity --no-bert512 --no-prefetch --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -o p -- $P > $O/pmc1.log 2>&1 && echo pass1 &&
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc4 -o p -- $P > $O/pmc4.log 2>&1 && echo pass4
python3 tools/pmc_kernel_table.py $O/r06_pmc_per_kernel.csv $O/pmc1 $O/pmc4
grep -h ms_per_step $O/pipe.log $O/inl.log | cut -c1-170
head -8 $O/r06_pmc_hbm_per_kernel.csv | cut -c1-200; cat $O/r06_gemm_traffic.json | head -40
rm -rf $O/pipe $O/inl $O/pf $O/pw $O/pmc1 $O/pmc4; du -sh $O
