#!/bin/bash
# dense configs: silhouette / scan / group tests + the config 3 and 5 bench lines.  usage: tools/gpu_dense.sh <tag>
TAG=${1:-d}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_mask.py tests/test_gpu_smplx.py tests/test_gpu_group.py tests/test_gpu_scan.py tests/test_gpu_configs_full.py -m gpu -q > gpurun_out/$TAG/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/pytest.log
tail -n 6 gpurun_out/$TAG/pytest.log
for c in 3 5; do
  python bench.py --config $c --no-cpu-baseline > gpurun_out/$TAG/bench_cfg$c.json 2> gpurun_out/$TAG/bench_cfg$c.err
  python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_cfg$c.json')); print('cfg$c', d['value'], d['ms_per_step'], d.get('extra'))"
done
