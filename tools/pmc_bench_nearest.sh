#!/bin/bash
# On the GPU box: VALU / SALU / VMEM instructions per query of every launch of tools/bench_nearest.py (one launch per scenario).
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_bn_$TAG -- python3 $R/tools/bench_nearest.py --reps 1 > $R/gpurun_out/pmc_bn_$TAG.log 2>&1
cd $R
find gpurun_out/pmc_bn_$TAG -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_bn_$TAG/**/*counter_collection.csv", recursive=True)[0]
g = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "nearest" in r["Kernel_Name"]:
        g.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["cold", "exact", "moved_1mm", "moved_5mm", "too_close", "nan", "far"]
for i, (k, v) in enumerate(sorted(g.items())):
    w = v["SQ_WAVES"]
    print("%-10s" % names[i % 7], {c: round(x / w, 1) for c, x in v.items() if c != "SQ_WAVES"}, "waves", int(w))
PY
