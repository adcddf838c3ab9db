#!/bin/bash
# instruction counters per wave of tools/bench_nearest.py's launches for several libraries.  usage: tools/gpu_pmc_ab.sh <tag> <lib>...
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in "$@"; do echo "== $L"; BODYFIT_LIB=$PWD/bodyfitting_amd/$L tools/pmc_bench_nearest.sh ${TAG}_${L%.so} 2>&1 | tail -8; done | tee gpurun_out/pmc_ab_$TAG.txt
