#!/bin/bash
# Run on the GPU box (through gpurun): round-6 evidence - kernel traces, PMC passes (MFMA, FETCH / WRITE of config 2, the batched mesh path, configs 3 and 5), the closest-point kernel's wave-cycle breakdown.   usage: tools/profile_round6.sh <tag>
#   1. kernel trace + stats of the headline bench command (config 2) and of the 32- / 256-frame batches (config 4 shard / whole)
#   2. PMC passes (each on its own, no trace domains): MFMA instruction / busy counters of the batched mesh kernels,
#      FETCH_SIZE and WRITE_SIZE of the same launches, and of the headline fit kernel
#   3. kernel trace of config 3 and config 5 (tools/bench_configs.py)
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra --repeats 2"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg2" -- $B --steps 30 --warmup 5 > "$OUT/trace_cfg2.log" 2>&1; echo "trace_cfg2 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_b32" -- $B --steps 10 --warmup 2 --frames-per-gpu 32 > "$OUT/trace_b32.log" 2>&1; echo "trace_b32 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_b256" -- $B --steps 10 --warmup 2 --frames-per-gpu 256 > "$OUT/trace_b256.log" 2>&1; echo "trace_b256 rc=$?"
# the batched mesh path UN-overlapped (--events --resident: every call on the batch's one stream, nothing of the next call's fit kernel
# beside it): the durations the MFMA / HBM fractions below are formed with
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_b32_serial" -- $B --steps 10 --warmup 2 --frames-per-gpu 32 --events --resident > "$OUT/trace_b32_serial.log" 2>&1; echo "trace_b32_serial rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_b256_serial" -- $B --steps 10 --warmup 2 --frames-per-gpu 256 --events --resident > "$OUT/trace_b256_serial.log" 2>&1; echo "trace_b256_serial rc=$?"
P="python3 $ROOT/bench.py --no-cpu-baseline --no-extra --repeats 1 --steps 3 --warmup 1 --prewarm-s 0 --events --resident"
for fr in 32 256; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/pmc_mfma_b$fr" -- $P --frames-per-gpu $fr > "$OUT/pmc_mfma_b$fr.log" 2>&1; echo "pmc_mfma_b$fr rc=$?"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_b$fr" -- $P --frames-per-gpu $fr > "$OUT/pmc_fetch_b$fr.log" 2>&1; echo "pmc_fetch_b$fr rc=$?"
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_b$fr" -- $P --frames-per-gpu $fr > "$OUT/pmc_write_b$fr.log" 2>&1; echo "pmc_write_b$fr rc=$?"
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_cfg2" -- $P > "$OUT/pmc_fetch_cfg2.log" 2>&1; echo "pmc_fetch_cfg2 rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_cfg2" -- $P > "$OUT/pmc_write_cfg2.log" 2>&1; echo "pmc_write_cfg2 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg35" -- python3 $ROOT/tools/bench_configs.py --cfg3 --cfg5x --reps 1 > "$OUT/trace_cfg35.log" 2>&1; echo "trace_cfg35 rc=$?"
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_nearest" -- python3 $ROOT/tools/bench_configs.py --cfg5x --reps 1 --iters 120 > "$OUT/pmc_nearest.log" 2>&1; echo "pmc_nearest rc=$?"
# the closest-point kernel with the 2 x 2 rule (BF_NEAREST_RULE=fast) beside the default (the reference's arithmetic)
export BF_NEAREST_RULE=fast
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg5_fast_rule" -- python3 $ROOT/tools/bench_configs.py --cfg5x --reps 1 > "$OUT/trace_cfg5_fast_rule.log" 2>&1; echo "trace_cfg5_fast_rule rc=$?"
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_nearest_fast_rule" -- python3 $ROOT/tools/bench_configs.py --cfg5x --reps 1 --iters 120 > "$OUT/pmc_nearest_fast_rule.log" 2>&1; echo "pmc_nearest_fast_rule rc=$?"
unset BF_NEAREST_RULE
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_texfit" -- python3 $ROOT/tools/bench_texfit.py > "$OUT/trace_texfit.log" 2>&1; echo "trace_texfit rc=$?"
# the dense kernels WITHOUT the resident fit launch (BF_DENSE_PERSISTENT=0: one fit launch per iteration, no doorbells): in the product
# schedule the forward mesh kernels request their posedirs slice and then WAIT for the fit launch's bell inside the kernel, so their
# durations in trace_cfg35 include that wait (~16-20 us); here they are the kernels' own
export BF_DENSE_PERSISTENT=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg35_nodoor" -- python3 $ROOT/tools/bench_configs.py --cfg3 --cfg5x --reps 1 > "$OUT/trace_cfg35_nodoor.log" 2>&1; echo "trace_cfg35_nodoor rc=$?"
unset BF_DENSE_PERSISTENT
# the closest-point kernel's wave-cycle breakdown, cache hit rates and traffic on config 5's scan and query count (tools/bench_nearest.py)
BODYFIT_LIB=$ROOT/bodyfitting_amd/libbodyfit.so bash $ROOT/tools/gpu_pmc_nearest6.sh $TAG libbodyfit.so > "$OUT/nearest_pmc.txt" 2>&1; echo "nearest pmc rc=$?"
cd /tmp
# memory-side traffic of the dense kernels (config 3 and config 5 apart: bf_mesh_multi_kernel<1> / <8>), one counter per pass
for c in 3 5; do
  A=$([ $c = 3 ] && echo "--cfg3" || echo "--cfg5x")
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_cfg$c" -- python3 $ROOT/tools/bench_configs.py $A --reps 1 > "$OUT/pmc_fetch_cfg$c.log" 2>&1; echo "pmc_fetch_cfg$c rc=$?"
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_cfg$c" -- python3 $ROOT/tools/bench_configs.py $A --reps 1 > "$OUT/pmc_write_cfg$c.log" 2>&1; echo "pmc_write_cfg$c rc=$?"
done
cd "$ROOT"
find "$OUT" -name "*.db" -delete
python3 tools/summarize_dense_traffic.py "$OUT" "$OUT/pmc_traffic_dense.json" > "$OUT/summary_dense_traffic.md" 2> "$OUT/summary_dense_traffic.err"; echo "dense traffic rc=$?"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summary.err"; echo "summary rc=$?"
du -sh "$OUT"
