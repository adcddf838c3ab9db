#!/bin/bash
# configs 3 and 5 (bench.py --config) with one environment variable at several values, interleaved, twice, on one box.
#   usage: tools/gpu_env_cfg_ab.sh <tag> <VAR> <value>... [-- <config>...]      (configs default to 3 5)
TAG=$1; VAR=$2; shift 2
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" = "--" ] && shift
CFGS=${@:-3 5}
cd "${GRAFT_REPO_ROOT:-$(git -C "$(dirname "$0")" rev-parse --show-toplevel)}"
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for c in $CFGS; do for v in "${VALS[@]}"; do
  env $VAR=$v timeout 300 python bench.py --config $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['ms_per_step_parts_rank0']; print('$VAR=$v cfg$c', '%.2f fps' % d['value'], 'fit %.3f ms' % p['fit_ms'], 'disp', p['displacement_ms'], {k: round(x * 1e3, 1) for k, x in (d.get('device_ms_last_iteration') or {}).items() if isinstance(x, float)})"
done; done; done | tee gpurun_out/$TAG/ab.txt
