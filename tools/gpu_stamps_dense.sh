#!/bin/bash
# stamps of the resident dense-schedule fit launch (stamp library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/${1:-sd}
BODYFIT_LIB=bodyfitting_amd/libbodyfit_stamp.so python tools/gpu_stamps_smplx.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tee gpurun_out/${1:-sd}/stamps_dense.txt
