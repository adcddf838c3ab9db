#!/bin/bash
# On the GPU box: kernel trace of config 5 (short) with the closest-point rule named by $1 (reference | fast).  usage: tools/prof_nearest.sh <rule> <tag>
RULE=${1:-reference}; TAG=${2:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp BF_NEAREST_RULE=$RULE
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nearest_$TAG -- python3 $R/tools/bench_configs.py --cfg5x --reps 1 --iters 120 > $R/gpurun_out/prof_nearest_$TAG.log 2>&1
cd $R
find gpurun_out/prof_nearest_$TAG -name "*.db" -delete
grep config gpurun_out/prof_nearest_$TAG.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_nearest_$TAG/**/*kernel_trace.csv", recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g[(r["Kernel_Name"][:30], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v2 = sorted(v)
    print("%-32s grid %-8s n %5d  total %8.1f ms  median %7.1f us  max %7.1f" % (k[0], k[1], len(v), sum(v) / 1e3, v2[len(v) // 2], v2[-1]))
PY
