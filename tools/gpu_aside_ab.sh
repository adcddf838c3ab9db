#!/bin/bash
# headline A/B for several libraries on one box, then the frame-loop / staging / parity tests with the product library
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tools/gpu_abn.sh $TAG "$@"
python -m pytest tests -q -m gpu -x -k "stream or stage or previous or pipeline or parity or mirror or group or shard" 2>&1 | grep -E "passed|failed|rror" | tail -5
