#!/bin/bash
# mesh kernels + configs 3 / 5 for several libraries on one box, then the mesh / SMPL-X / scan tests with the product library
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tools/gpu_mesh_abn.sh $TAG "$@"
tools/gpu_cfg_abn.sh ${TAG}c "$@"
python -m pytest tests -q -m gpu -x -k "mesh or forward or smplx or batch or parity" 2>&1 | grep -E "passed|failed|rror" | tail -5
