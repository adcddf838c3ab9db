#!/bin/bash
TAG=${1:-c5ab}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
for kp in 1 0; do
  BF_DENSE_SUBMODEL_KP=$kp python bench.py --config 5 > gpurun_out/$TAG/cfg5_kp$kp.json 2> gpurun_out/$TAG/cfg5_kp$kp.err
  python -c "import json; d=json.load(open('gpurun_out/$TAG/cfg5_kp$kp.json')); print('kp$kp', d['value'], d['ms_per_step'], d.get('ms_per_step_parts_rank0'))"
done
BF_DENSE_SUBMODEL_KP=1 python bench.py --config 3 > gpurun_out/$TAG/cfg3_kp1.json 2>/dev/null; BF_DENSE_SUBMODEL_KP=0 python bench.py --config 3 > gpurun_out/$TAG/cfg3_kp0.json 2>/dev/null
for kp in 1 0; do python -c "import json; d=json.load(open('gpurun_out/$TAG/cfg3_kp$kp.json')); print('cfg3 kp$kp', d['value'], d['ms_per_step'], d.get('ms_per_step_parts_rank0'))"; done
