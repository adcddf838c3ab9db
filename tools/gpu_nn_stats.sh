#!/bin/bash
# closest-point counters per query over config 5 (a -DBF_NEAREST_STATS library).  usage: tools/gpu_nn_stats.sh <tag> <lib>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/$1
BODYFIT_LIB=bodyfitting_amd/$2 timeout 600 python tools/nearest_stats_cfg5.py --slice 20 --iters 120 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tee gpurun_out/$1/stats.txt
