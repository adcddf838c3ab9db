#!/usr/bin/env python3
"""GPU box tool: the closest-point kernel alone on config 5's scan (83,784 triangles, 10,475 queries = the body's vertices displaced by
sigma), cold / with the exact answer as the hint / with a hint `move` away / with garbage hints; checks every variant returns the cold
search's answer bit for bit.   usage: python tools/bench_nearest.py [--reps 20]"""
import argparse, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--copies", type=int, default=8); a = ap.parse_args()
model = S.make_model("smplx", seed=0)
prob, sv, sf = S.make_scan_problem_smplx(model, frame=0, n_views=8)
scan = N.Scan(sv, sf)
rng = np.random.default_rng(0)
base = np.concatenate([sv[rng.choice(len(sv), 10475, replace=False)] for _ in range(a.copies)])     # (config 5 launches 8 frames' queries at once)
import ctypes as C
from bodyfitting_amd import _lib as L
stats_fn = getattr(L.load(), "bf_nearest_stats_read", None)       # (only in a -DBF_NEAREST_STATS build)
for sigma in (0.003, 0.01, 0.03):
    q = (base + rng.normal(0, sigma, base.shape)).astype(np.float32)
    pts, ids, bary, us_cold = scan.nearest_points_hinted(q, None, reps=a.reps)
    out = {"sigma_m": sigma, "queries": len(q), "cold_us": round(us_cold, 1)}
    if stats_fn is not None:
        st = (C.c_ulonglong * 8)(); stats_fn(st, 1)
        out["cold_stats_per_query"] = [round(x / max(st[0], 1), 2) for x in st]
    for name, hint in (("exact", pts), ("moved_1mm", pts + rng.normal(0, 0.001, pts.shape).astype(np.float32)),
                       ("moved_5mm", pts + rng.normal(0, 0.005, pts.shape).astype(np.float32)),
                       ("too_close", q + (pts - q) * 0.5), ("nan", np.full_like(pts, np.nan)), ("far", pts + 1.0)):
        p2, i2, b2, us = scan.nearest_points_hinted(q, hint, reps=a.reps)
        same = bool(np.array_equal(i2, ids) and np.array_equal(p2.view(np.uint32), pts.view(np.uint32)) and np.array_equal(b2.view(np.uint32), bary.view(np.uint32)))
        out[name + "_us"] = round(us, 1); out[name + "_same"] = same
        if stats_fn is not None:
            st = (C.c_ulonglong * 8)(); stats_fn(st, 1)
            out[name + "_stats_per_query"] = [round(x / max(st[0], 1), 2) for x in st]
    print(json.dumps(out), flush=True)
