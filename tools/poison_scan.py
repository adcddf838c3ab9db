"""Bring-up helper: which allocation is read before anybody wrote it?  Runs a small dense SMPL-X fit once per allocation index with
only that allocation filled with NaN bytes (BF_POISON=255 BF_POISON_ONLY=k) and reports the indices that turn the result non-finite."""
import os, subprocess, sys
code = r'''
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from bodyfitting_amd import native as N, synthetic as S
model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
prob = S.make_problem_smplx(model, frame=0, n_views=8)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
b = N.FrameBatch(dev, 1, 8)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
b.fit(12)
print("FINITE", bool(np.isfinite(b.get_params()).all()))
'''
env = dict(os.environ, BF_POISON="255", BF_POISON_LOG="1", BF_POISON_ONLY="100000")
out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
allocs = [l for l in out.stderr.splitlines() if l.startswith("alloc ")]
print(len(allocs), "allocations;", out.stdout.strip().splitlines()[-1])
o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BF_POISON="255"), capture_output=True, text=True)
print("everything poisoned:", o.stdout.strip().splitlines()[-1])
bad = []
for k in range(len(allocs)):
    env = dict(os.environ, BF_POISON="255", BF_POISON_ONLY=str(k))
    o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if "FINITE True" not in o.stdout:
        bad.append(k); print("NOT FINITE with", [a for a in allocs if a.startswith("alloc %d:" % k)], flush=True)
print("bad:", bad)
