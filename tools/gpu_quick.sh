#!/bin/bash
# quick loop: parity of the keypoint path + the headline bench.  usage: tools/gpu_quick.sh <tag> [pytest -k expression]
TAG=${1:-q}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py tests/test_gpu_batches.py tests/test_gpu_smplx.py -m gpu -q ${2:+-k "$2"} > gpurun_out/$TAG/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/pytest.log
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -n 4 gpurun_out/$TAG/pytest.log
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench.json')); print('value', d['value'], 'ms', d['ms_per_step'], 'cycles/it', d['roofline']['latency']['cycles_per_iteration'], 'fit_ms', d['device_ms_per_step']['fit_ms'])"
