#!/bin/bash
# build libbodyfit_V_<tag>.so from a patched copy of fit_kernels.hip.  usage: tools/build_variant.sh <tag> <python-patch-file>
# the patch file defines patch(s) -> s on the source text.
set -e
TAG=$1; PATCH=$2
cd /root/repo/bodyfitting_amd/csrc
python3 - "$PATCH" <<'PY'
import sys, importlib.util
spec = importlib.util.spec_from_file_location("p", sys.argv[1]); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
s = open("fit_kernels.hip").read()
t = m.patch(s)
assert t != s, "patch did not change the source"
open("_variant.hip", "w").write(t)
PY
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -mllvm -amdgpu-sched-strategy=max-ilp -c _variant.hip -o _variant.o 2>&1 | grep -E "error|Illegal|warning" && { echo "BUILD FAILED $TAG"; rm -f _variant.hip _variant.o; exit 1; }
OBJ=$(ls *.o | grep -v "fit_kernels.o\|_variant.o\|stamp" | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../libbodyfit_V_$TAG.so $OBJ _variant.o -ldl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -mllvm -amdgpu-sched-strategy=max-ilp -Rpass-analysis=kernel-resource-usage -c _variant.hip -o /dev/null 2>&1 | grep -A5 "Function Name: _Z10fit_kernelILi24ELi10ELi11ELi25ELb0" | grep -E "Scratch|VGPRs" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' '; echo " <- $TAG"
rm -f _variant.hip _variant.o
