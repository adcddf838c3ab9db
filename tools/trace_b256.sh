#!/bin/bash
# kernel timeline of the 256-frame step with resident inputs (config 4's whole job on one GPU)   usage: tools/trace_b256.sh <tag>
TAG=${1:-b256}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-extra --repeats 1 --steps 8 --warmup 3 --frames-per-gpu 256 --resident > $R/gpurun_out/trace_$TAG.log 2>&1
f=$(find $R/gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:44]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx=[i for i,r in enumerate(rows) if r[2].startswith('void fit_kernel')]
k=idx[len(idx)//2]
t0=rows[k][0]
for s,e,n in rows[k-1:idx[len(idx)//2+3]+6]:
    print(f"{n:46s} start {(s-t0)/1000:9.1f} end {(e-t0)/1000:9.1f} dur {(e-s)/1000:7.1f}")
PY
find $R/gpurun_out/trace_$TAG -name "*.db" -delete; find $R/gpurun_out/trace_$TAG -name "*trace.csv" -delete
tail -n 1 $R/gpurun_out/trace_$TAG.log | cut -c1-200
