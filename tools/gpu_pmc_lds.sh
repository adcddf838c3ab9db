#!/bin/bash
# LDS bank-conflict counters per kernel over configs 3 and 5 (short runs) and the batched mesh paths
TAG=${1:-pmclds}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/$TAG/c35 -- python3 $R/tools/bench_configs.py --cfg3 --cfg5x --reps 1 --iters 60 > $R/gpurun_out/$TAG/c35.log 2>&1; echo "c35 rc=$?"
for fr in 32 256; do
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/$TAG/b$fr -- python3 $R/bench.py --no-cpu-baseline --no-extra --repeats 1 --steps 3 --warmup 1 --prewarm-s 0 --events --resident --frames-per-gpu $fr > $R/gpurun_out/$TAG/b$fr.log 2>&1; echo "b$fr rc=$?"
done
cd $R
find gpurun_out/$TAG -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/$TAG/*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    g = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        g[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1].get("SQ_LDS_BANK_CONFLICT", [0]))):
        n = len(v["SQ_WAVES"]); m = {c: sum(x) / n for c, x in v.items()}
        if m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            print("%-46s n %5d  conflict %10.0f  lds_active %10.0f  frac %.2f  busy %10.0f  waves %6.0f" % (k, n, m["SQ_LDS_BANK_CONFLICT"], m["SQ_LDS_IDX_ACTIVE"], m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m["SQ_BUSY_CYCLES"], m["SQ_WAVES"]))
PY
