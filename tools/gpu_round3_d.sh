#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3i
python -m pytest tests/test_gpu_scan.py -m gpu -q > gpurun_out/r3i/pytest_scan.log 2>&1; echo "rc=$?" >> gpurun_out/r3i/pytest_scan.log
python -m pytest tests/test_gpu_configs_full.py -m gpu -q -s > gpurun_out/r3i/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/r3i/pytest_full.log
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3i/trace_cfg5 -- python3 $R/tools/bench_configs.py --cfg5x --reps 1 > $R/gpurun_out/r3i/trace_cfg5.log 2>&1
cd $R
find gpurun_out/r3i -name "*.db" -delete
grep config gpurun_out/r3i/trace_cfg5.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/r3i/trace_cfg5/**/*kernel_trace.csv", recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g[(r["Kernel_Name"][:34], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v2 = sorted(v)
    print("%-36s grid %-8s n %5d  total %8.1f ms  median %7.1f us  max %7.1f" % (k[0], k[1], len(v), sum(v) / 1e3, v2[len(v) // 2], v2[-1]))
PY
tail -n 5 gpurun_out/r3i/pytest_scan.log; tail -n 14 gpurun_out/r3i/pytest_full.log | cut -c1-300
