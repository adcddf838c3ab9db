#!/usr/bin/env python3
"""GPU box diagnostic: end state of the SMPL+D stage (reduced model, 300 + 300 iterations) for both closest-point rules and for
one-ulp nudges of the initial pose - how far the HIP stage moves from ITSELF (the stage is chaotic)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bodyfitting_amd import native as N, synthetic as S
from oracle import mesh_oracle as MO
model = S.make_model("smpl", seed=0, nv=690); dev = N.DeviceModel(model, S.make_gmm(seed=0), device=0)
prob0, sv, sf = S.make_scan_problem(model, frame=0, n_views=8)
srch = MO.ReferenceSearcher(sv, sf)
def met(P):
    ids, cp, _ = srch.nearest(P.astype(np.float32)); d = np.linalg.norm(P - cp, axis=1)
    return "mean %.3f med %.3f p95 %.3f max %.2f icp %.4f" % (d.mean() * 1e3, np.median(d) * 1e3, np.percentile(d, 95) * 1e3, d.max() * 1e3, np.linalg.norm(P - cp))
for rule in ("reference", "fast"):
    N.set_nearest_rule(rule)
    for nudge in range(4):
        prob = dict(prob0)
        ip = prob0["init_pose"].astype(np.float32)
        for _ in range(nudge):
            ip = np.nextafter(ip, np.float32(np.inf)).astype(np.float32)
        prob["init_pose"] = ip
        scan = N.Scan(sv, sf)
        c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
        b = N.FrameBatch(dev, 1, 8); b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans([scan])
        b.fit(300)
        verts = b.get_result()[0][0]
        b.fit_displacement(300)
        disp = b.get_displacement()[0]
        print(rule, "nudge", nudge, "| after fit", met(verts), "| after SMPL+D", met(verts + disp), "max|disp| %.3f" % np.abs(disp).max(), flush=True)
        b.close(); scan.close()
