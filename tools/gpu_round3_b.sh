#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3g
python -m pytest tests -m gpu -q --deselect tests/test_gpu_configs_full.py > gpurun_out/r3g/pytest_all.log 2>&1
echo "all rc=$?" >> gpurun_out/r3g/pytest_all.log
python -m pytest tests/test_gpu_configs_full.py -m gpu -q -s > gpurun_out/r3g/pytest_full.log 2>&1
echo "full rc=$?" >> gpurun_out/r3g/pytest_full.log
python bench.py > gpurun_out/r3g/bench_default.json 2> gpurun_out/r3g/bench_default.err
tail -n 6 gpurun_out/r3g/pytest_all.log; tail -n 16 gpurun_out/r3g/pytest_full.log | cut -c1-260; grep -E "FAILED|ERROR" gpurun_out/r3g/pytest_all.log | head -20
for f in gpurun_out/r3g/bench*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d.get('ms_per_step_parts_rank0'))" 2>/dev/null || tail -n 3 ${f%.json}.err; done
