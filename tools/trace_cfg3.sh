#!/bin/bash
# kernel timeline of config 3's dense iterations (rocprofv3 kernel trace)   usage: tools/trace_cfg3.sh <tag>
TAG=${1:-t3}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -- python3 $R/tools/bench_configs.py --cfg3 --reps 1 > $R/gpurun_out/trace_$TAG.log 2>&1
f=$(find $R/gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('bf_kp_contour')]
k=idx[-20]
t0=int(rows[k-6]['Start_Timestamp'])
for r in rows[k-7:k+9]:
    s=(int(r['Start_Timestamp'])-t0)/1000; e=(int(r['End_Timestamp'])-t0)/1000
    print(f"{r['Kernel_Name'][:40]:40s} start {s:8.1f} end {e:8.1f} dur {e-s:6.1f}")
PY
find $R/gpurun_out/trace_$TAG -name "*.db" -delete; find $R/gpurun_out/trace_$TAG -name "*trace.csv" -delete
