#!/bin/bash
# SQ counters of the concat-MLP kernels (diagnostic): three separate --pmc passes, then a per-kernel summary
#   gpurun -- 'bash tools/diag/pmc_concat.sh'   ->   gpurun_out/pmc_concat/summary.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_concat
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $GRAFT_REPO_ROOT/bench.py --critic concat_mlp --precision ${MI_PMC_PREC:-bf16} --clock-warmup-ms 0 --graph off --steps 2 --warmup 1 --profile-steps 1 --no-secondary --no-cpu-baseline --no-parity-mode --no-fp8"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY -d $OUT -o p1 --output-format csv -- $CMD > $OUT/run1.log 2>&1
echo rc=$?
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT -o p2 --output-format csv -- $CMD > $OUT/run2.log 2>&1
echo rc=$?
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $OUT -o p3 --output-format csv -- $CMD > $OUT/run3.log 2>&1
echo rc=$?
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "concat_fwd" in k or "concat_bwd_dw2" in k or "concat_bwd_duv" in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/summary.txt", "w") as fh:
    for k, c in acc.items():
        fh.write(k + "\n")
        m = {n: sum(v) / len(v) for n, v in c.items()}
        for n in sorted(m):
            fh.write(f"  {n:28s} {m[n]:.4g}\n")
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"):
                if n in m:
                    fh.write(f"  {n} / SQ_WAVE_CYCLES = {m[n] / wc:.3f}\n")
        if "SQ_INSTS_MFMA" in m:
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
                if n in m:
                    fh.write(f"  {n} per MFMA = {m[n] / m['SQ_INSTS_MFMA']:.2f}\n")
        if m.get("SQ_LDS_IDX_ACTIVE"):
            fh.write(f"  LDS bank conflict share = {m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.3f}\n")
print(open(f"{out}/summary.txt").read())
PY
