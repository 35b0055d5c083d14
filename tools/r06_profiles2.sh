#!/bin/bash
# round 6 closing: per-kernel stats of the north-star (64, 512) forward as one pass (folded plain f16), and of the timed / inline training step on the closing tree
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06p; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b512 -o p -- python3 bench.py --mode bert512 --precision fp16 --parts 1 --steps 10 --warmup 3 > $O/b512.log 2>&1 && echo b512 ok &&
cp $(find $O/b512 -name "*kernel_stats.csv" | head -1) $O/r06_bert512_onepass_kernel_stats.csv &&
B="python3 bench.py --no-cpu-baseline --no-roofline --no-parity --no-bert512" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -o p -- $B --steps 8 --warmup 3 > $O/pipe.log 2>&1 && echo pipe ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inl -o p -- $B --steps 5 --warmup 2 --no-prefetch > $O/inl.log 2>&1 && echo inl ok &&
cp $(find $O/pipe -name "*kernel_stats.csv" | head -1) $O/r06_bench_pipelined_kernel_stats.csv &&
cp $(find $O/inl -name "*kernel_stats.csv" | head -1) $O/r06_bench_inline_kernel_stats.csv
grep -h "ms_per_step" $O/b512.log $O/pipe.log $O/inl.log | cut -c1-200
rm -rf $O/b512 $O/pipe $O/inl; ls -la $O
