#!/bin/bash
# SQ / clock counters for the concat-MLP kernels (diagnostic)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_concat
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $GRAFT_REPO_ROOT/bench.py --critic concat_mlp --graph off --steps 2 --warmup 1 --profile-steps 1 --no-secondary --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT -o p1 --output-format csv -- $CMD > $OUT/run1.log 2>&1
echo rc=$?
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD -d $OUT -o p2 --output-format csv -- $CMD > $OUT/run2.log 2>&1
echo rc=$?
