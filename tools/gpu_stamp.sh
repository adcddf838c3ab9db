#!/bin/bash
# headline bench (short) + per-wave stamps.  usage: tools/gpu_stamp.sh <tag>
TAG=${1:-st}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
BODYFIT_LIB=bodyfitting_amd/libbodyfit_stamp.so python tests/gpu_stamps.py > gpurun_out/$TAG/stamps.txt 2>&1
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench.json')); print('value', d['value'], 'ms', d['ms_per_step'], 'cycles/it', d['roofline']['latency']['cycles_per_iteration'], 'fit_ms', d['device_ms_per_step']['fit_ms'])"
tail -n 8 gpurun_out/$TAG/stamps.txt
