#!/bin/bash
# the closest-point / scan / SMPL-X GPU tests, then configs 3 and 5 with the product library.  usage: tools/gpu_scan_tests.sh <tag>
TAG=${1:-scantests}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
python -m pytest tests -q -m gpu -x -k "nearest or scan or smplx or searcher" 2>&1 | grep -E "passed|failed|rror" | tail -8 | tee gpurun_out/$TAG/tests.txt
tools/gpu_cfg_abn.sh $TAG libbodyfit.so
