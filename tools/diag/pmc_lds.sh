#!/bin/bash
# LDS bank-conflict counters for the bilinear kernels (diagnostic)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $OUT -o lds --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --graph off --steps 3 --warmup 1 --profile-steps 1 --no-secondary --no-cpu-baseline > $OUT/run.log 2>&1
echo rc=$?
ls -R $OUT | head -20
