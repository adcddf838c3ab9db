#!/bin/bash
# first GPU pass of round 3: streaming API, parity on the new template, bench in the three staging modes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_parity.py tests/test_gpu_batches.py -m gpu -x -q > gpurun_out/r3a/pytest_parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/r3a/pytest_parity.log
python -m pytest tests/test_gpu_mask.py tests/test_gpu_smplx.py tests/test_gpu_group.py -m gpu -q > gpurun_out/r3a/pytest_dense.log 2>&1
echo "dense rc=$?" >> gpurun_out/r3a/pytest_dense.log
for mode in kernel memcpy zerocopy; do
  BF_STAGE_MODE=$mode python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/r3a/bench_$mode.json 2> gpurun_out/r3a/bench_$mode.err
done
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline --resident > gpurun_out/r3a/bench_resident.json 2> gpurun_out/r3a/bench_resident.err
for mode in kernel zerocopy; do
  BF_STAGE_MODE=$mode python bench.py --steps 50 --warmup 5 --frames-per-gpu 32 --no-extra --no-cpu-baseline > gpurun_out/r3a/bench32_$mode.json 2> gpurun_out/r3a/bench32_$mode.err
  BF_STAGE_MODE=$mode python bench.py --steps 20 --warmup 3 --frames-per-gpu 256 --no-extra --no-cpu-baseline > gpurun_out/r3a/bench256_$mode.json 2> gpurun_out/r3a/bench256_$mode.err
done
python bench.py --steps 50 --warmup 5 --frames-per-gpu 32 --no-extra --no-cpu-baseline --resident > gpurun_out/r3a/bench32_resident.json 2>/dev/null
python bench.py --steps 20 --warmup 3 --frames-per-gpu 256 --no-extra --no-cpu-baseline --resident > gpurun_out/r3a/bench256_resident.json 2>/dev/null
tail -n 5 gpurun_out/r3a/pytest_parity.log gpurun_out/r3a/pytest_dense.log
for f in gpurun_out/r3a/bench*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config']['workload'][:60])" 2>/dev/null || tail -n 3 ${f%.json}.err; done
