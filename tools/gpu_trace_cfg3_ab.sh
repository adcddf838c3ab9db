#!/bin/bash
# kernel durations of config 3 (rocprofv3 trace) for several libraries.  usage: tools/gpu_trace_cfg3_ab.sh <tag> <lib>...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/$TAG
for L in "$@"; do
  cd /tmp
  BODYFIT_LIB=$R/bodyfitting_amd/$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/${L%.so} -- python3 $R/tools/bench_configs.py --cfg3 --reps 1 > $R/gpurun_out/$TAG/${L%.so}.log 2>&1
  cd $R; find gpurun_out/$TAG -name "*.db" -delete
  echo "== $L"; grep -h "config\|ms_per" gpurun_out/$TAG/${L%.so}.log | tail -2 | cut -c1-200
  f=$(ls gpurun_out/$TAG/${L%.so}/*/*_kernel_stats.csv | head -1); head -8 $f | cut -d, -f1-4 | cut -c1-110
done 2>&1 | tee gpurun_out/$TAG/ab.txt
