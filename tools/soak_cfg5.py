#!/usr/bin/env python3
"""Soak of config 5's fit + SMPL+D stage (keypoint workgroups beside the closest-point search, joined through a doorbell count): the same
eight frames fitted again and again - parameters and displacements must be the same bits every time.   usage: tools/soak_cfg5.py [--fits N]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--fits", type=int, default=25); ap.add_argument("--frames", type=int, default=8); a = ap.parse_args()
model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
items = [S.make_scan_problem_smplx(model, frame=f, n_views=48) for f in range(a.frames)]
scans = [N.Scan(sv, sf) for _, sv, sf in items]
c2w, K, kp, ndiv, betas, pose = N.pack_problem([p for p, _, _ in items])
b = N.FrameBatch(dev, a.frames, 48)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans(scans)
first, bad = None, 0
for i in range(a.fits):
    b.reset(); b.fit(300); b.fit_displacement(60); b.sync()
    cur = (b.get_params().copy(), b.get_displacement().copy())
    if first is None: first = cur
    elif not (np.array_equal(cur[0], first[0]) and np.array_equal(cur[1], first[1])):
        bad += 1
        if bad < 4: print("fit", i, "differs: max |d params|", float(np.abs(cur[0] - first[0]).max()), "max |d disp|", float(np.abs(cur[1] - first[1]).max()), flush=True)
print("frames", a.frames, "fits", a.fits, "differing from the first:", bad, flush=True)
sys.exit(1 if bad else 0)
