#!/usr/bin/env python3
"""GPU box experiment: config 5's shard (8 SMPL-X frames with their scans, 300 + 300 iterations) as ONE batch of 8 against `--parts`
batches of 8 / parts frames driven by a host thread each on the same device (one part's closest-point search then runs under the
other's mesh passes)."""
import argparse, json, os, sys, threading, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--parts", type=int, default=2); ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--frames", type=int, default=8); ap.add_argument("--reps", type=int, default=3); a = ap.parse_args()
model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
items = [S.make_scan_problem_smplx(model, frame=f, n_views=48) for f in range(a.frames)]
scans = [N.Scan(sv, sf) for _, sv, sf in items]
per = a.frames // a.parts
batches = []
for p in range(a.parts):
    sub = items[p * per:(p + 1) * per]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([q for q, _, _ in sub])
    b = N.FrameBatch(dev, per, 48)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans(scans[p * per:(p + 1) * per])
    batches.append(b)

def work(b, out, i):
    b.reset(); b.fit(a.iters); b.fit_displacement(a.iters); b.sync()
    out[i] = b.get_params().copy()

def once():
    out = [None] * a.parts
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(b, out, i)) for i, b in enumerate(batches)]
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0, np.concatenate(out)
once()
ts = []
for _ in range(a.reps):
    dt, params = once(); ts.append(dt)
print(json.dumps({"parts": a.parts, "frames": a.frames, "iters": a.iters, "s_per_step": sorted(ts)[len(ts) // 2], "frames_per_s": a.frames / sorted(ts)[len(ts) // 2],
                  "all": [round(t, 4) for t in ts], "params_digest": float(np.abs(params).sum())}), flush=True)
