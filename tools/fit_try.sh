#!/bin/bash
# build both libraries (stop on a compiler error), then one turn of the fit-kernel loop on the GPU.  usage: tools/fit_try.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(git -C "$(dirname "$0")" rev-parse --show-toplevel)}
cd "$ROOT/bodyfitting_amd/csrc" || exit 1
make -j8 2>&1 | grep -E "error|warning|Illegal" && { echo "PRODUCT BUILD FAILED"; exit 1; }
make stamp 2>&1 | grep -E "error|Illegal" && { echo "STAMP BUILD FAILED"; exit 1; }
make resource-usage 2>&1 | grep -A5 "Function Name: _Z10fit_kernelILi24ELi10ELi11ELi25ELb0" | grep -E "Scratch|VGPRs" | tr '\n' ' '; echo
cd "$ROOT"
tools/gpurun_retry.sh 900 "timeout 600 tools/gpu_iter.sh $1" 2>&1 | grep -v mark | tail -3
grep -E "phase|one iteration" gpurun_out/$1/stamps.txt | tail -5
