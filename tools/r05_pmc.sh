#!/bin/bash
# round 5: SQ / TCC / TCP counter passes over the inline step (one kernel at a time on the device) + the raw hipGraph node cost
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05pmc; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-roofline --no-parity --no-bert512 --no-prefetch --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -o p -- $B > $O/pmc1.log 2>&1 && echo pass1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -o p -- $B > $O/pmc2.log 2>&1 && echo pass2 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc3 -o p -- $B > $O/pmc3.log 2>&1 && echo pass3 &&
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc4 -o p -- $B > $O/pmc4.log 2>&1 && echo pass4 &&
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr --output-format csv -d $O/pmc5 -o p -- $B > $O/pmc5.log 2>&1; echo "pass5 rc $?"
python3 tools/pmc_kernel_table.py $O/r05_pmc_per_kernel.csv $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5; du -sh $O
