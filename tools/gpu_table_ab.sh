#!/bin/bash
# the table-driven fit instance (bench.py's extra leg) for several libraries on one box
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for L in "$@"; do
  BODYFIT_LIB=bodyfitting_amd/$L python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']; print('$L', 'table-driven %.1f fps' % e['table_driven_instance']['value'], 'headline %.1f' % d['value'], 'b256 ms %.4f' % e['batch_256_frames']['ms_per_step'])"
done; done | tee gpurun_out/$TAG/ab.txt
