#!/bin/bash
# closest-point kernel alone (tools/bench_nearest.py) for several libraries on one box.  usage: tools/gpu_nearest_abn.sh <tag> <lib>...
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
for L in "$@"; do
  BODYFIT_LIB=bodyfitting_amd/$L python tools/bench_nearest.py --reps 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$L', 'sigma', d['sigma_m'], 'cold', d['cold_us'], 'exact', d['exact_us'], 'moved_5mm', d['moved_5mm_us'], 'same', all(v for k,v in d.items() if k.endswith('_same')))"
done | tee gpurun_out/$TAG/out.txt
