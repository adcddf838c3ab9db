#!/bin/bash
# per-wave stamps of two stamp libraries side by side.  usage: tools/gpu_stamp2.sh <tag> <libA> <libB>
TAG=${1:-st2}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
for L in "$2" "$3"; do
  echo "=== $L" >> gpurun_out/$TAG/stamps.txt
  BODYFIT_LIB=bodyfitting_amd/$L python tests/gpu_stamps.py 2>&1 | tail -n 7 >> gpurun_out/$TAG/stamps.txt
done
cat gpurun_out/$TAG/stamps.txt
