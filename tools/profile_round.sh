#!/bin/bash
# Produces the profiles/ artefacts of one round on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh r1_c'
# then copy gpurun_out/profile_<tag>/<tag>_* into profiles/.
#   <tag>_bench_line.json         the JSON line of the profiled bench.py run (eager launches: one row per kernel)
#   <tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary of that same run
#   <tag>_pmc_traffic.json        HBM-side bytes per launch from FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
# rocprofv3 runs from /tmp with the program directly after `--` (profiling recipe of the GPU pool).
set -u
TAG=${1:-r1_x}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/profile_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --graph off --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/bench.log" 2>&1 || exit 1
grep "^{\"metric\"" "$OUT/bench.log" | tail -1 > "$OUT/${TAG}_bench_line.json"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
SHORT="python3 $REPO/bench.py --graph off --steps 5 --warmup 2 --profile-steps 1 --secondary-steps 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $SHORT > "$OUT/fetch.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $SHORT > "$OUT/write.log" 2>&1 || exit 1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
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
for key in sorted(set(f2) | set(w2)):
    name, grid = key
    if not name.startswith(("mi::", "void mi::", "_ZN2mi")):
        continue
    fb = 2 * 1024 * sum(f2.get(key, [0])) / max(len(f2.get(key, [0])), 1)
    wb = 1024 * sum(w2.get(key, [0])) / max(len(w2.get(key, [0])), 1)
    res["kernels"][f"{name[:110]} grid={grid}"] = {"fetch_bytes": fb, "write_bytes": wb, "total_bytes": fb + wb,
                                                  "launches_sampled": len(f2.get(key, []))}
json.dump(res, open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
print("kernels with traffic:", len(res["kernels"]))
PY
ls -la "$OUT"
