#!/bin/bash
# build bodyfitting_amd/libbodyfit_V_<tag>.so from a patched copy of ONE kernel source, with the product's own flags for that file
# (the Makefile's `variant` target: nothing is duplicated here).   usage: tools/build_variant.sh <tag> <python-patch-file> [source-stem]
# The patch file defines patch(s) -> s on the source text; source-stem defaults to fit_kernels.  The build fails on any warning.
set -e
TAG=$1; PATCH=$2; STEM=${3:-fit_kernels}
ROOT=${GRAFT_REPO_ROOT:-$(git -C "$(dirname "$0")" rev-parse --show-toplevel)}
cd "$ROOT/bodyfitting_amd/csrc"
python3 - "$PATCH" "$STEM" <<'PY'
import sys, importlib.util
spec = importlib.util.spec_from_file_location("p", sys.argv[1]); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
s = open(sys.argv[2] + ".hip").read()
t = m.patch(s)
assert t != s, "patch did not change the source"
open("_variant_" + sys.argv[2] + ".hip", "w").write(t)
PY
make variant TAG=$TAG VSRC=$STEM VFILE=_variant_$STEM.hip 2>&1 | grep -E "error|Illegal|warning:" && { echo "BUILD FAILED $TAG"; rm -f _variant_$STEM.hip; exit 1; }
rm -f _variant_$STEM.hip
echo "built bodyfitting_amd/libbodyfit_V_$TAG.so"
