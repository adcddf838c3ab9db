#!/bin/bash
# headline bench with an environment variable off / on, interleaved, on one box.  usage: tools/gpu_env_ab.sh <tag> <VAR> <off-value> <on-value>
TAG=$1; VAR=$2; OFF=$3; ON=$4
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/$TAG
for rep in 1 2 3; do for v in $OFF $ON; do
  env $VAR=$v python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', 'ms/step %.4f' % d['ms_per_step'], 'value %.1f' % d['value'])"
done; done | tee gpurun_out/$TAG/ab.txt
