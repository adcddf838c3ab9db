#!/bin/bash
# A/B of two libraries on ONE box (boxes differ by ~1 %): headline bench alternately, three times each.  usage: tools/gpu_ab.sh <tag> <libA> <libB>
TAG=${1:-ab}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
for rep in 1 2 3; do
  for L in "$2" "$3"; do
    BODYFIT_LIB=bodyfitting_amd/$L python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'cycles/it %.0f' % d['roofline']['latency']['cycles_per_iteration'], 'value %.1f' % d['value'])"
  done
done | tee gpurun_out/$TAG/ab.txt
