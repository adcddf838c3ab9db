#!/bin/bash
# PMC passes over the headline fit kernel (bench.py default workload, a few steps): instruction mix, LDS bank conflicts, wait cycles
TAG=${1:-pmcfit}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
P="python3 $R/bench.py --no-cpu-baseline --no-extra --repeats 1 --steps 3 --warmup 1 --prewarm-s 0 --events --resident"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$TAG/p$i -- $P > $R/gpurun_out/$TAG/p$i.log 2>&1; echo "p$i rc=$?"
done
cd $R
find gpurun_out/$TAG -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/$TAG/p*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    g = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"].startswith("void fit_kernel") or "fit_kernel" in r["Kernel_Name"]:
            g[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in g.items():
        print(k, {c: round(max(x)) for c, x in v.items()}, "(max over launches = a 100-iteration launch)")
PY
