#!/bin/bash
# a -k selection of the GPU tests.  usage: tools/gpu_tests_k.sh "<expr>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -q -m gpu -x -k "$1" 2>&1 | grep -E "passed|failed|rror" | tail -5
