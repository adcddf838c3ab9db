"""Bring-up helper: per-wave arrive / leave stamps (BF_T / BF_SYNC) of the LAST iteration of the resident dense-schedule fit launch
(SMPL-X, keypoints only; `--masks` adds silhouettes = config 3's shape).   BODYFIT_LIB=bodyfitting_amd/libbodyfit_stamp.so python tools/gpu_stamps_smplx.py"""
import os, sys
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
    raw = b.debug_dump(4352 + 192)[4352:].view(np.int32).astype(np.int64).reshape(-1, 8)
    print("rep", rep, "timing", b.last_timing())
    rows = [k for k in range(24) if raw[k].any()]
    t0 = min(int(raw[k][raw[k] != 0].min()) for k in rows)
    for k in rows:
        print("  row %2d (%s %d): %s" % (k, "arrive" if k % 2 == 0 else "leave ", k // 2, " ".join("%7d" % (((x - t0) & 0xFFFFFFFF) if x else -1) for x in raw[k])))
