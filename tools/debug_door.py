import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N, synthetic as S
model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
prob = S.make_problem_smplx(model, frame=0, n_views=8)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
b = N.FrameBatch(dev, 1, 8)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
for n in (1, 1, 8, 30, 5, 2, 3):
    import time; t0 = time.time()
    try:
        b.fit(n); p = b.get_params(); print("fit", n, "ok", float(np.abs(p).sum()), "%.1f ms" % (1e3 * (time.time() - t0)))
    except Exception as e:
        print("fit", n, "FAILED after %.1f ms:" % (1e3 * (time.time() - t0)), e)
