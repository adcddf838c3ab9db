#!/bin/bash
# On the GPU box: scan tests, then a kernel trace of the config-5 timings.  usage: tools/prof_cfg5.sh <tag>
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_scan_$TAG.log 2>&1; tail -3 gpurun_out/pytest_scan_$TAG.log
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg5_$TAG -- python3 $R/tools/bench_configs.py --cfg5x --cfg5 --cfg3 --reps 1 > $R/gpurun_out/prof_cfg5_$TAG.log 2>&1
cd $R
find gpurun_out/prof_cfg5_$TAG -name "*.db" -delete
grep config gpurun_out/prof_cfg5_$TAG.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_cfg5_$TAG/**/*kernel_trace.csv", recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g[(r["Kernel_Name"][:28], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v2 = sorted(v)
    print("%-30s grid %-8s n %5d  total %8.1f ms  median %7.1f us  max %7.1f" % (k[0], k[1], len(v), sum(v) / 1e3, v2[len(v) // 2], v2[-1]))
PY
