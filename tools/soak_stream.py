#!/usr/bin/env python3
"""Soak of the frame loop (bf_batch_stage_inputs on the second stream, deferred mesh tail, completion event on the fit's own dispatch): thousands
of frames through ONE batch, every result compared bit for bit with the same frame set fitted alone.   usage: tools/soak_stream.py [--steps N] [--frames F]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N, synthetic as S, _lib   # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20000); ap.add_argument("--frames", type=int, default=1); ap.add_argument("--sets", type=int, default=6)
a = ap.parse_args()
model, gmm = S.make_model("smpl"), S.make_gmm()
dev = N.DeviceModel(model, gmm)
F = a.frames
sets = [N.pack_problem([S.make_problem(model, frame=7 * s + f, n_views=48) for f in range(F)]) for s in range(a.sets)]
want = []
for c2w, K, kp, ndiv, betas, pose in sets:
    ref = N.FrameBatch(dev, F, 48)
    ref.set_cameras(sets[0][0], sets[0][1]); ref.set_keypoints(kp, ndiv); ref.set_init(betas, pose)
    ref.fit(100); want.append((ref.get_params().copy(), ref.get_result()[0].copy())); ref.close()
flags = _lib.FIT_RESET | _lib.FIT_FETCH | _lib.FIT_NOTIME
b = N.FrameBatch(dev, F, 48)
b.set_cameras(sets[0][0], sets[0][1])
rng = np.random.default_rng(0)
order = rng.integers(0, a.sets, a.steps)
bad = 0
t0 = time.time()
for i, s in enumerate(order):
    _, _, kp, ndiv, betas, pose = sets[s]
    b.stage_inputs(kp, ndiv, betas, pose)
    b.fit(100, flags=flags)
    if i > 0:
        p = b.get_previous()
        ps = order[i - 1]
        if not (np.array_equal(p[0], want[ps][0]) and np.array_equal(p[1], want[ps][1])):
            bad += 1
            if bad < 5: print("step", i - 1, "set", ps, "differs: max |d params|", float(np.abs(p[0] - want[ps][0]).max()), flush=True)
b.sync()
dt = time.time() - t0
print("frames", F, "steps", a.steps, "mismatches", bad, "%.1f steps/s" % (a.steps / dt), flush=True)
b.close()
sys.exit(1 if bad else 0)
