#!/bin/bash
# On the GPU box: where bf_nearest_kernel's time goes - wave-cycle breakdown, cache hit rates and memory-side traffic of every launch of
# tools/bench_nearest.py (config 5's scan and query count; one launch per scenario).  usage: tools/gpu_pmc_nearest6.sh <tag> [lib]
TAG=${1:-x}; LIB=${2:-libbodyfit.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp BODYFIT_LIB=$R/bodyfitting_amd/$LIB
mkdir -p $R/gpurun_out/pmc6_$TAG
cd /tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc6_$TAG/p$i -- python3 $R/tools/bench_nearest.py --reps 1 > $R/gpurun_out/pmc6_$TAG/p$i.log 2>&1; echo "pass $i rc=$?"
done
cd $R
find gpurun_out/pmc6_$TAG -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
g = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc6_$TAG/**/*counter_collection.csv", recursive=True)):
    k = 0; seen = {}
    for r in csv.DictReader(open(f)):
        if "bf_nearest" in r["Kernel_Name"]:
            d = int(r["Dispatch_Id"])
            if d not in seen: seen[d] = len(seen)
            g.setdefault(seen[d], {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["cold", "exact", "moved_1mm", "moved_5mm", "too_close", "nan", "far"]
for i, v in g.items():
    w = v.get("SQ_WAVES", 1.0)
    out = {c: (round(x / w, 1) if c.startswith("SQ_") and c != "SQ_BUSY_CYCLES" else x) for c, x in v.items() if c != "SQ_WAVES"}
    print("sigma#%d %-10s waves %d" % (i // 7, names[i % 7], int(w)), out)
PY
