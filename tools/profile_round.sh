#!/bin/bash
# Produces the profiles/ artefacts of one round on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh r2_a'
# then copy gpurun_out/profile_<tag>/<tag>_* into profiles/.
#   <tag>_bench_line.json         the JSON line of the profiled bench.py run (eager launches: one row per kernel)
#   <tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary of that same run
#   <tag>_pmc_traffic.json        HBM-side bytes per launch from FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
# rocprofv3 runs from /tmp with the program directly after `--` (profiling recipe of the GPU pool).
set -u
TAG=${1:-r1_x}
COMMIT=${2:-unknown}   # the snapshot on the GPU box has no .git: pass `git rev-parse --short HEAD` as the second argument
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/profile_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --graph off --steps 20 --warmup 5 --clock-warmup-ms 0 --no-cpu-baseline --no-parity-mode --no-fp8"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/bench.log" 2>&1 || exit 1
grep "^{\"metric\"" "$OUT/bench.log" | tail -1 > "$OUT/${TAG}_bench_line.json"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
# (the counter passes include the fp8 leg -- B = 8192, d = 1024 -- so that fp8_mode.roofline gets its bytes too)
SHORT="python3 $REPO/bench.py --graph off --steps 5 --warmup 2 --clock-warmup-ms 0 --profile-steps 1 --secondary-steps 2 --no-cpu-baseline --no-parity-mode --timed-iters 50"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $SHORT > "$OUT/fetch.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $SHORT > "$OUT/write.log" 2>&1 || exit 1
# matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES counts 32 per v_mfma_f32_32x32x16_bf16 (summed over all SIMDs);
# GRBM_GUI_ACTIVE / 8 = the kernel's duration in shader cycles (sum over the 8 XCDs); 1024 SIMDs on the chip
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o sq -- $SHORT > "$OUT/sq.log" 2>&1 || echo "SQ pass failed (kept going)"
CSRC_SHA=$(python3 -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.csrc_sha())" 2>/dev/null | tail -1)
python3 - "$OUT" "$TAG" "$CSRC_SHA" "$COMMIT" <<'PY'
import csv, glob, json, sys, collections
out, tag, csrc_sha = sys.argv[1], sys.argv[2], sys.argv[3]
def load(sub, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
fetch, write = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
res = {"note": "HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes). FETCH_SIZE is "
               "in KB and on gfx950 reports half of the bytes of wide coalesced reads: fetch_bytes = 2 * FETCH_SIZE * 1024 "
               "(MI355X_MICROARCH.md, HBM section); write_bytes = WRITE_SIZE * 1024. Kernels launched with several shapes "
               "(the bf16 GEMM templates) are split by grid size.", "kernels": {}}
# split by (kernel, grid) through the raw rows again
def load2(sub, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc
f2, w2 = load2("fetch", "FETCH_SIZE"), load2("write", "WRITE_SIZE")
mf, ga = load2("sq", "SQ_VALU_MFMA_BUSY_CYCLES"), load2("sq", "GRBM_GUI_ACTIVE")
res["csrc_sha"] = csrc_sha  # bench.py attaches this file's numbers only to a build of the same kernel sources
res["commit"] = sys.argv[4] if len(sys.argv) > 4 else None
res["mfma_note"] = ("mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): busy cycles of the matrix pipes "
                    "over (kernel duration in shader cycles x 1024 SIMDs)")
for key in sorted(set(f2) | set(w2)):
    name, grid = key
    if not name.startswith(("mi::", "void mi::", "_ZN2mi")):
        continue
    fb = 2 * 1024 * sum(f2.get(key, [0])) / max(len(f2.get(key, [0])), 1)
    wb = 1024 * sum(w2.get(key, [0])) / max(len(w2.get(key, [0])), 1)
    entry = {"fetch_bytes": fb, "write_bytes": wb, "total_bytes": fb + wb, "launches_sampled": len(f2.get(key, []))}
    if mf.get(key) and ga.get(key):
        busy = sum(mf[key]) / len(mf[key])
        act = sum(ga[key]) / len(ga[key])
        if act > 0:
            entry["mfma_busy_frac"] = round(busy / (act / 8.0 * 1024.0), 4)
            entry["mfma_busy_cycles"] = busy
            entry["gui_active"] = act
    res["kernels"][f"{name[:110]} grid={grid}"] = entry
json.dump(res, open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
print("kernels with traffic:", len(res["kernels"]))
PY
ls -la "$OUT"
