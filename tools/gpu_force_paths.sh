#!/bin/bash
# the N-GPU code paths on one GPU: bench.py --force-group (bf_group, one process) and --force-ranks (bf_comm, rank per GPU)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/${1:-force}
for fl in --force-group --force-ranks; do
  timeout 200 python bench.py $fl --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2> gpurun_out/${1:-force}/err$fl.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$fl', 'value %.1f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], d['config'].get('parallelism'), d['config'].get('launch_mode'))" || tail -3 gpurun_out/${1:-force}/err$fl.txt
done
