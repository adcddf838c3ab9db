#!/bin/bash
# several libraries on ONE box: headline bench, twice each, interleaved.  usage: tools/gpu_abn.sh <tag> <lib>...
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
  for L in "$@"; do
    BODYFIT_LIB=bodyfitting_amd/$L python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'cycles/it %.0f' % d['roofline']['latency']['cycles_per_iteration'], 'value %.1f' % d['value'])"
  done
done | tee gpurun_out/$TAG/ab.txt
