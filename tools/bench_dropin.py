#!/usr/bin/env python3
"""Latency of the drop-in entry point as apps/genebody_fitting.py uses it: `SMPLify(...)(net_output, c2ws, Ks, keypoints, ...)`
once per frame (reference smplify/smplify.py:84-254), synthetic data, 1 GPU.  usage: python tools/bench_dropin.py [--frames N]"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bodyfitting_amd import assets, synthetic as S          # noqa: E402
from bodyfitting_amd.smplify import SMPLify                  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--iters", type=int, default=100)
    a = ap.parse_args()
    model, gmm = S.make_model("smpl", seed=0), S.make_gmm(seed=0)
    assets.register_model(model, "smpl", "neutral")
    assets.register_gmm(gmm)
    probs = [S.make_problem(model, frame=f, n_views=48) for f in range(a.frames)]
    fitter = SMPLify(smpl_type="smpl", num_iters=a.iters, gender="neutral", device=0, debug=False)
    call = lambda p: fitter((p["init_betas"], p["init_pose"]), p["c2ws"], p["Ks"], p["keypoints"], use_frames=p["use_frames"], imsize=p["imsize"])
    call(probs[0])
    t0 = time.perf_counter()
    for p in probs:
        call(p)
    dt = (time.perf_counter() - t0) / a.frames
    print(json.dumps({"entry": "SMPLify.__call__ per frame (48 views, %d iterations, result dict incl. vertices)" % a.iters,
                      "ms_per_call": dt * 1e3, "frames_per_s": 1.0 / dt}))
    # the same frames through SMPLify.stream: the next frame's upload + fit are issued before the previous frame's result is read
    frames = lambda: (((p["init_betas"], p["init_pose"]), p["keypoints"]) for p in probs)      # noqa: E731
    for _ in fitter.stream(frames(), probs[0]["c2ws"], probs[0]["Ks"]):
        pass
    t0 = time.perf_counter()
    n = sum(1 for _ in fitter.stream(frames(), probs[0]["c2ws"], probs[0]["Ks"]))
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"entry": "SMPLify.stream per frame (the capture's frame loop as a two-deep pipeline; same result dicts)",
                      "ms_per_frame": dt * 1e3, "frames_per_s": 1.0 / dt}))
