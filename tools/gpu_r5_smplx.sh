#!/bin/bash
# SMPL-X / dense-loop suites + configs 3 and 5 (bench).  usage: tools/gpu_r5_smplx.sh <tag>
TAG=${1:-sx}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_smplx.py tests/test_gpu_configs_full.py tests/test_gpu_mask.py -m gpu -q -x > gpurun_out/$TAG/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/pytest.log
BF_POISON=255 python -m pytest tests/test_gpu_smplx.py -m gpu -q -x -k "sub_model or forward or loss" > gpurun_out/$TAG/pytest_poison.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/pytest_poison.log
python bench.py --config 3 > gpurun_out/$TAG/bench_cfg3.json 2> gpurun_out/$TAG/bench_cfg3.err
python bench.py --config 5 > gpurun_out/$TAG/bench_cfg5.json 2> gpurun_out/$TAG/bench_cfg5.err
tail -n 3 gpurun_out/$TAG/pytest.log; tail -n 3 gpurun_out/$TAG/pytest_poison.log
for c in 3 5; do python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_cfg$c.json')); print('cfg$c', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))" 2>/dev/null || tail -n 3 gpurun_out/$TAG/bench_cfg$c.err; done
