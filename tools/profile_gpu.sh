#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + stats, then HBM PMC passes, of the bench command.
# usage: tools/profile_gpu.sh <tag>     -> gpurun_out/prof_<tag>/ and a summary in gpurun_out/prof_<tag>/summary.*
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
# the dense schedule too (mesh kernel every iteration): gives the mesh kernel thousands of launches
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_dense" -- $BENCH --dense > "$OUT/trace_dense.log" 2>&1
echo "trace_dense rc=$?"
# PMC passes on their own (no trace domains), one counter group per run
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH --dense > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH --dense > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summary.err"
echo "summary rc=$?"
# keep what comes back small
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
