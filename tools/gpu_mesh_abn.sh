#!/bin/bash
# the full-mesh kernels (single frame span, 32- and 256-frame batches) for several libraries on one box
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for L in "$@"; do
  BODYFIT_LIB=bodyfitting_amd/$L python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']; print('$L', 'mesh span us %.2f' % (d['roofline_mesh']['avg_launch_ms']*1e3), 'b32 ms %.4f' % e['batch_32_frames']['ms_per_step'], 'b256 ms %.4f' % e['batch_256_frames']['ms_per_step'], 'b1024 %.3f' % e['batch_1024_frames']['ms_per_step'])"
done; done | tee gpurun_out/$TAG/ab.txt
