#!/usr/bin/env python3
"""Soak of config 3's fit (silhouette gradients through the contour scan's fixed-point atomic sums): the same problem fitted again and again -
the fitted parameters must be the same bits every time, whatever order the atomics arrived in.   usage: tools/soak_cfg3.py [--fits N] [--frames F]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--fits", type=int, default=150); ap.add_argument("--frames", type=int, default=1); a = ap.parse_args()
model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
mask_frames = list(range(0, 48, 6))[:8]
probs = [S.make_problem_smplx(model, frame=f, n_views=48, mask_frames=mask_frames) for f in range(a.frames)]
c2w, K, kp, ndiv, betas, pose = N.pack_problem(probs)
b = N.FrameBatch(dev, a.frames, 48)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
b.set_masks(np.stack([np.array(p["masks"]) for p in probs]), mask_frames, None)
first, bad = None, 0
for i in range(a.fits):
    b.reset(); b.fit(200); b.sync()
    p = b.get_params().copy()
    if first is None: first = p
    elif not np.array_equal(p, first):
        bad += 1
        if bad < 4: print("fit", i, "differs from fit 0: max |d|", float(np.abs(p - first).max()), flush=True)
print("frames", a.frames, "fits", a.fits, "differing from the first:", bad, flush=True)
sys.exit(1 if bad else 0)
