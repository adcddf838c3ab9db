#!/bin/bash
# config 5 (and 3) from several checkouts (build/b_<sha>, each with its own library) and the working tree, on one box.  usage: tools/gpu_bisect5.sh <tag> <dir>...
TAG=$1; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for d in "$@"; do for c in 3 5; do
  (cd $d && python bench.py --config $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['ms_per_step_parts_rank0']; print('$d cfg$c', '%.2f fps' % d['value'], 'fit %.3f' % p['fit_ms'], 'disp %.3f' % (p['displacement_ms'] or 0))")
done; done; done 2>&1 | tee gpurun_out/$TAG/out.txt
