"""Bring-up helper: cycle stamps of the dense schedule's one-iteration fit launch (SMPL-X + masks), -DBF_STAMP build.
    BODYFIT_LIB=bodyfitting_amd/libbodyfit_stamp.so python tools/stamps_dense.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402

model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
mask_frames = list(range(0, 48, 6))[:8]
prob = S.make_problem_smplx(model, frame=0, n_views=48, mask_frames=mask_frames)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
b = N.FrameBatch(dev, 1, 48)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
b.set_masks(np.array(prob["masks"])[None], mask_frames, None)
for rep in range(2):
    b.reset(); b.fit(30); b.sync()
    d = np.nan_to_num(b.debug_dump(4096 + 96)[4096:], nan=0.0, posinf=0.0, neginf=0.0)
    d = np.where(np.abs(d) > 1e7, 0.0, d)
    st = d[:12]
    print("rep", rep, "per-barrier cumulative cycles of the iteration:", [int(x) for x in st])
    print("  door marks (cycles since phase A ended): state written", int(d[56]), " ext bell seen", int(d[57]), " ext staged", int(d[58]), "| wave 7 pose state done at", int(d[59]), "cycles of the iteration")
    print("  pose state on wave 3 (cycles from its start): theta assembled", int(d[49]), " rotations", int(d[50]), " rest joints", int(d[51]), " chain", int(d[52]))
    k = d[64:72]
    print("  kernel entry -> prologue done (t0, t256):", int(k[0]), int(k[1]), " after its barrier:", int(k[2]))
    k2 = d[72:80]
    print("  prologue inner (t0): after constant staging / image", int(k2[0]), " after proj+params", int(k2[1]), " after role registers", int(k2[2]),
          " after keypoint table", int(k2[3]), " after Adam state", int(k2[4]))
    print("  loop done (t0, t256):", int(k[4]), int(k[5]), " tail start:", int(k[6]), " tail end:", int(k[7]))
