#!/bin/bash
# kernel stats of config 5 (tools/bench_configs.py --cfg5x) from several checkouts on one box.  usage: tools/gpu_trace_dirs.sh <tag> <dir>...
TAG=$1; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/$TAG
for d in "$@"; do
  n=$(echo $d | tr '/.' '__')
  cd /tmp
  (cd $R/$d && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/$n -- python3 tools/bench_configs.py --cfg5x --reps 1 > $R/gpurun_out/$TAG/$n.log 2>&1)
  cd $R; find gpurun_out/$TAG -name "*.db" -delete
  echo "== $d"; f=$(ls gpurun_out/$TAG/$n/*/*_kernel_stats.csv | head -1); head -16 $f | cut -d, -f1-4 | cut -c1-100
done 2>&1 | tee gpurun_out/$TAG/out.txt
