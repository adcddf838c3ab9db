#!/bin/bash
# configs 3 and 5 (bench.py --config) for several libraries on one box.  usage: tools/gpu_cfg_abn.sh <tag> <lib>...
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for L in "$@"; do for c in 3 5; do
  BODYFIT_LIB=bodyfitting_amd/$L timeout 300 python bench.py --config $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['ms_per_step_parts_rank0']; print('$L cfg$c', '%.2f fps' % d['value'], 'fit %.3f ms' % p['fit_ms'], 'disp', p['displacement_ms'])"
done; done; done | tee gpurun_out/$TAG/ab.txt
