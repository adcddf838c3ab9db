"""Bring-up helper (not a test): per-phase cycle stamps of one fit iteration from the -DBF_STAMP build.
    BODYFIT_LIB=bodyfitting_amd/libbodyfit_stamp.so python tests/gpu_stamps.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402

model, gmm = S.make_model("smpl"), S.make_gmm()
dev = N.DeviceModel(model, gmm)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([S.make_problem(model, 0, 48)])
b = N.FrameBatch(dev, 1, 48)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
for rep in range(3):
    b.reset(); b.fit(100); b.sync()
    d = np.nan_to_num(b.debug_dump(4096 + 96)[4096:], nan=0.0, posinf=0.0, neginf=0.0)
    d = np.where(np.abs(d) > 1e7, 0.0, d)          # (uninitialised stamp slots)
    n = int(np.max(np.nonzero(d[:32])[0])) + 1 if np.any(d[:32]) else 0
    st = d[:n]
    print("rep", rep, "timing", b.last_timing())
    print("  cumulative cycles after each barrier:", [int(x) for x in st])
    print("  per-phase:", [int(x) for x in np.diff(np.concatenate([[0], st]))])
    print("  phase A inner: local transforms loaded, after chain levels:", [int(x) for x in d[40:42]])
    print("  phase D inner: start, after view loop, after reduce+route:", [int(d[45]), int(d[43]), int(d[44])])
    print("  GMM wave 4: done with its pose blend (phase A), with its chunks of phase B, with the prior (phase D):", int(d[42]), int(d[49]), int(d[55]))
    print("  phase F inner: wave 0 done, wave 3 done (d pose feature), GMM wave 4 done:", [int(d[46]), int(d[47]), int(d[48])])
    print("  IK inner: wave0 after rodrigues_bwd, wave0 end | wave3 after g_beta partials, after beta Adam, end:", [int(d[50]), int(d[51]), int(d[52]), int(d[53]), int(d[54])])
