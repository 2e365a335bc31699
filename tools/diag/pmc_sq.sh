#!/bin/bash
# SQ stall counters for the bilinear kernels (diagnostic); two separate passes
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d $OUT -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --graph off --steps 3 --warmup 1 --profile-steps 1 --no-secondary --no-cpu-baseline > $OUT/run1.log 2>&1
echo rc=$?
rocprofv3 --kernel-trace --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS -d $OUT -o p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --graph off --steps 3 --warmup 1 --profile-steps 1 --no-secondary --no-cpu-baseline > $OUT/run2.log 2>&1
echo rc=$?
ls $OUT
