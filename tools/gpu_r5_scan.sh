#!/bin/bash
TAG=${1:-r5scan}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_scan.py tests/test_gpu_dropin_files.py tests/test_gpu_smplx.py -m gpu -q -s -k "destroyed or in_flight or two_frames or stream or config5 or dropin or golden" > gpurun_out/$TAG/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/pytest.log
grep -E "scan build|passed|failed|rc=|Error|error" gpurun_out/$TAG/pytest.log | tail -20
