#!/bin/bash
# On the GPU box: VALU / SALU / VMEM instruction counts per wave of the dense-path kernels (config 5, 30 iterations).
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_nearest_$TAG -- python3 $R/tools/bench_configs.py --cfg5x --reps 1 --iters 120 > $R/gpurun_out/pmc_nearest_$TAG.log 2>&1
cd $R
find gpurun_out/pmc_nearest_$TAG -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_nearest_$TAG/**/*counter_collection.csv", recursive=True)[0]
g = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    g[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in g.items():
    if "nearest" in k:
        w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
        print(k, {c: round(sum(x) / len(x) / w, 1) for c, x in v.items()}, "per wave; launches", len(v["SQ_WAVES"]))
PY
