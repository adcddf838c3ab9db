#!/bin/bash
# the whole -m gpu suite, smoke, the default bench line, configs 3 / 5
TAG=${1:-validate}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -q -s > gpurun_out/$TAG/pytest_all.log 2>&1; echo "all rc=$?" >> gpurun_out/$TAG/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/$TAG/smoke.log
python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
python bench.py --config 3 > gpurun_out/$TAG/bench_cfg3.json 2> gpurun_out/$TAG/bench_cfg3.err
python bench.py --config 5 > gpurun_out/$TAG/bench_cfg5.json 2> gpurun_out/$TAG/bench_cfg5.err
grep -E "passed|failed|rc=" gpurun_out/$TAG/pytest_all.log | tail -3; tail -n 2 gpurun_out/$TAG/smoke.log
for f in gpurun_out/$TAG/bench_default.json gpurun_out/$TAG/bench_cfg3.json gpurun_out/$TAG/bench_cfg5.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d.get('device_ms_last_iteration'))" 2>/dev/null || tail -n 3 ${f%.json}.err; done
