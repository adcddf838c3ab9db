#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/c5s
for v in 1 0; do
  BF_SCAN_BUILD_STREAM=$v python bench.py --config 5 2>gpurun_out/c5s/err$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('own_stream=$v', d['value'], d['ms_per_step_parts_rank0'])"
done 2>&1 | tee gpurun_out/c5s/out.txt
