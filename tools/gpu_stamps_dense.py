"""Bring-up helper (not a test): cycle stamps of the resident dense-schedule fit launch (SMPL-X, keypoints only) from the -DBF_STAMP build.
    BODYFIT_LIB=bodyfitting_amd/libbodyfit_stamp.so python tools/gpu_stamps_dense.py
Stamps are those of the LAST iteration of the launch: cumulative cycles after every barrier of the iteration (phase A first; the wait
for the dense kernels sits between the first and the second), the pose state's inner marks, and the three marks of the hand-over."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402

model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm)
prob = S.make_problem_smplx(model, frame=0, n_views=48)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
b = N.FrameBatch(dev, 1, 48)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
for rep in range(2):
    b.reset(); b.fit(40); b.sync()
    d = np.nan_to_num(b.debug_dump(4096 + 96)[4096:], nan=0.0, posinf=0.0, neginf=0.0)
    d = np.where(np.abs(d) > 1e8, 0.0, d)
    st = [int(x) for x in d[:12]]
    print("rep", rep, "timing", b.last_timing())
    print("  cumulative cycles after each barrier of the last iteration:", st)
    print("  pose state (wave 3, from its start): theta, rotations, rest joints, chain:", [int(x) for x in d[49:53]], " wave 3 back at:", int(d[59]))
    print("  hand-over (from its start): state written + released, bell heard (= the dense kernels' time), ext staged:", [int(d[56]), int(d[57]), int(d[58])])
    print("  phase A chain waves: transforms ready, chain done:", [int(d[40]), int(d[41])])
