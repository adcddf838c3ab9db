"""Texture-fitting iteration rate at the reference's sizes (render 512 x 512 with 2 x 2 super-sampling, texture size 4,
SMPL+D mesh 13,776 faces, scan --scan-faces faces; smplify/texture_fitting.py:174,240-275).  Synthetic meshes."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from bodyfitting_amd import texture_fitting as TF     # noqa: E402
from texfit_cases import icosphere                     # noqa: E402


def blob(level, seed, ts):
    rng = np.random.default_rng(seed)
    v, f = icosphere(level)
    v = (v * np.array([0.45, 0.8, 0.4], np.float32) * (1 + 0.05 * np.sin(9 * v[:, 1:2])) + np.array([0, 0.9, 0], np.float32)).astype(np.float32)
    return v, f, rng.uniform(0, 1, (len(f), ts, ts, ts, 3)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=90)
    ap.add_argument("--image", type=int, default=512)
    ap.add_argument("--scan-level", type=int, default=6)      # 81,920 faces
    ap.add_argument("--fit-level", type=int, default=5)       # 20,480 faces (SMPL: 13,776)
    a = ap.parse_args()
    scan, fit = blob(a.scan_level, 0, 4), blob(a.fit_level, 1, 4)
    center, dist = TF.scene_bound(scan[0])
    r = TF.Renderer(a.image, 4, near=0.0, far=2 * dist)
    r.set_mesh(r.TARGET, scan); r.set_mesh(r.FITTED, fit)
    ring = TF.gen_cam_views(center, 18, dist, gl=True)
    for i in range(5):
        r.step(ring[i], 1e-2)
    t0 = time.perf_counter()
    losses = [r.step(ring[i % 18], 1e-2) for i in range(a.iters)]
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "texture_fitting_iterations_per_s", "value": a.iters / dt, "ms_per_iteration": 1e3 * dt / a.iters,
                      "image": a.image, "scan_faces": len(scan[1]), "fit_faces": len(fit[1]), "loss_first": losses[0], "loss_last": losses[-1]}))


if __name__ == "__main__":
    main()
