#!/bin/bash
# one turn of the fit-kernel loop: keypoint parity + bit-equality tests, the headline bench, per-wave stamps.  usage: tools/gpu_iter.sh <tag> [-k expr]
TAG=${1:-it}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py tests/test_gpu_batches.py -m gpu -q -x ${2:+-k "$2"} > gpurun_out/$TAG/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/pytest.log
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
BODYFIT_LIB=bodyfitting_amd/libbodyfit_stamp.so python tests/gpu_stamps.py > gpurun_out/$TAG/stamps.txt 2>&1
tail -n 4 gpurun_out/$TAG/pytest.log
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench.json')); print('value', d['value'], 'ms', d['ms_per_step'], 'cycles/it', d['roofline']['latency']['cycles_per_iteration'], 'fit_ms', d['device_ms_per_step']['fit_ms'])"
tail -n 7 gpurun_out/$TAG/stamps.txt
