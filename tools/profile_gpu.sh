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
# (counter collection serialises every dispatch: keep these runs tiny; the sparse leg gives the fit kernel's
#  100-iteration launches, the dense leg many mesh-kernel launches)
PMCB="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
for leg in sparse dense; do
  flag=""; [ $leg = dense ] && flag="--dense --iters 10"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$leg" -- $PMCB $flag > "$OUT/pmc_fetch_$leg.log" 2>&1
  echo "pmc_fetch_$leg rc=$?"
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$leg" -- $PMCB $flag > "$OUT/pmc_write_$leg.log" 2>&1
  echo "pmc_write_$leg rc=$?"
done
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summary.err"
echo "summary rc=$?"
# keep what comes back small
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
