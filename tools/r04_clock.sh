#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 tools/encoder_mask_clock.py 2>&1 | grep -v amdgpu.ids | tee $O/encoder_mask_clock.log
for c in 128 192 256; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/clk$c -o p -- python3 tools/encoder_mask_clock.py --cus $c --passes 2 > $O/clk$c.log 2>&1
  python3 tools/pmc_kernel_table.py $O/clk_$c.csv $O/clk$c --match=gemm_16c > /dev/null
  python3 - <<PY
import csv
for r in csv.DictReader(open('$O/clk_$c.csv')):
    print('$c CUs', r['kernel'][:28], 'avg_us', r['avg_us(profiled)'], 'clock_GHz', r['clock_GHz'], 'mfma_busy(of 256 CUs)', r['mfma_busy'])
PY
  rm -rf $O/clk$c
done 2>&1 | tee -a $O/encoder_mask_clock.log
