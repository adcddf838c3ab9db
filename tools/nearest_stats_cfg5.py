#!/usr/bin/env python3
"""GPU box tool for a -DBF_NEAREST_STATS build of csrc/scan_kernels.hip: config 5's fit and SMPL+D stage in slices of `--slice` iterations,
the closest-point kernel's counters per query after every slice (searches = 1 + guesses that failed, groups, trips, screen passes, -,
rule passes, records through the rule)."""
import argparse, ctypes as C, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bodyfitting_amd import native as N, synthetic as S, _lib as L   # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--slice", type=int, default=20); ap.add_argument("--iters", type=int, default=120); a = ap.parse_args()
fn = getattr(L.load(), "bf_nearest_stats_read", None)
if fn is None:
    raise SystemExit("not a -DBF_NEAREST_STATS build")
model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
items = [S.make_scan_problem_smplx(model, frame=f, n_views=48) for f in range(8)]
scans = [N.Scan(sv, sf) for _, sv, sf in items]
c2w, K, kp, ndiv, betas, pose = N.pack_problem([p for p, _, _ in items])
b = N.FrameBatch(dev, 8, 48)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans(scans)
def read():
    st = (C.c_ulonglong * 8)(); fn(st, 1)
    return [round(x / max(st[0], 1), 2) for x in st] + [int(st[0])]
b.reset()
for i in range(0, a.iters, a.slice):
    b.fit(a.slice, flags=0 if i else L.FIT_RESET); b.sync()
    print("fit  %3d.." % i, read(), flush=True)
# (fit_displacement restarts its own Adam every call: slices are not a continuation - only the first slice is what the stage does)
b.fit_displacement(a.slice); b.sync(); print("disp first %d" % a.slice, read(), flush=True)
b.fit_displacement(a.iters); b.sync(); print("disp all %d" % a.iters, read(), flush=True)
