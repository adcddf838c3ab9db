#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/bis5
for d in build/b_2f3ffb7 build/b_91cd70e; do
  (cd $d && python bench.py --config 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step_parts_rank0'])")
done 2>&1 | tee gpurun_out/bis5/out.txt
