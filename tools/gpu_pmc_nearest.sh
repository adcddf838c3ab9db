#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/r3f/pmc_nearest -- python3 $R/tools/bench_configs.py --cfg5x --reps 1 --iters 120 > $R/gpurun_out/r3f/pmc_nearest.log 2>&1
cd $R
find gpurun_out/r3f -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/r3f/pmc_nearest/**/*counter_collection.csv", recursive=True)[0]
g = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    g[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in g.items():
    if "nearest" in k:
        w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
        n = len(v["SQ_WAVES"])
        print(k, {c: round(sum(x) / len(x) / w, 1) for c, x in v.items()}, "per wave; launches", n)
        # first 20 launches (fit, far queries) vs the last 20 (SMPL+D)
        for name, sl in (("first 20", slice(0, 20)), ("fit tail", slice(max(0, n // 2 - 25), n // 2 - 5)), ("last 20", slice(n - 20, n))):
            print("  ", name, {c: round(sum(x[sl]) / max(len(x[sl]), 1) / w, 1) for c, x in v.items()})
PY
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/trace_cfg5 -- python3 $R/tools/bench_configs.py --cfg5x --reps 1 > $R/gpurun_out/r3f/trace_cfg5.log 2>&1
cd $R
find gpurun_out/r3f -name "*.db" -delete
grep config gpurun_out/r3f/trace_cfg5.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/r3f/trace_cfg5/**/*kernel_trace.csv", recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g[(r["Kernel_Name"][:34], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:6]:
    v2 = sorted(v)
    print("%-36s grid %-8s n %5d  total %8.1f ms  median %7.1f us  max %7.1f" % (k[0], k[1], len(v), sum(v) / 1e3, v2[len(v) // 2], v2[-1]))
PY
