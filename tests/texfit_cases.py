"""Synthetic meshes for the texture-fitting tests: a closed, outward-wound blob as the 'scan' and a perturbed copy as the
mesh whose textures are fitted."""
import numpy as np


def icosphere(level):
    g = (1 + 5 ** 0.5) / 2
    v = [(-1, g, 0), (1, g, 0), (-1, -g, 0), (1, -g, 0), (0, -1, g), (0, 1, g), (0, -1, -g), (0, 1, -g), (g, 0, -1), (g, 0, 1), (-g, 0, -1), (-g, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def blob_pair(level=2, ts=4, seed=0, wind=1):
    """-> (scan mesh, fitted mesh), each (verts, faces, textures[NF,ts,ts,ts,3]).  wind=-1 reverses the faces."""
    rng = np.random.default_rng(seed)
    v, f = icosphere(level)
    if wind < 0:
        f = f[:, ::-1].copy()
    scan_v = (v * np.array([0.45, 0.8, 0.4], np.float32) + np.array([0.1, 0.9, -0.05], np.float32)).astype(np.float32)
    cen = scan_v[f].mean(1)
    base = 0.5 + 0.5 * np.stack([np.sin(5 * cen[:, 0]), np.cos(4 * cen[:, 1]), np.sin(3 * cen[:, 2] + 1)], 1)
    scan_t = np.clip(base[:, None, None, None, :] + 0.15 * rng.standard_normal((len(f), ts, ts, ts, 3)), 0, 1).astype(np.float32)
    fit_v = (scan_v + 0.01 * rng.standard_normal(scan_v.shape)).astype(np.float32)
    fit_t = np.full((len(f), ts, ts, ts, 3), 0.5, np.float32)
    return (scan_v, f, scan_t), (fit_v, f, fit_t)


def uv_atlas(n_faces, cols=None, margin=0.08, seed=0):
    """A UV layout for a mesh of n_faces triangles, the way an OBJ's `vt` / `f v/vt` lines give it: every face its own triangle in a
    grid cell of the unit square (half of them wound the other way, so front and back sides both occur).
    -> (uv[3 n_faces, 2] in [0, 1], uv_faces[n_faces, 3] 0-based)"""
    rng = np.random.default_rng(seed)
    cols = cols or int(np.ceil(np.sqrt(n_faces)))
    rows = int(np.ceil(n_faces / cols))
    uv = np.zeros((3 * n_faces, 2))
    for i in range(n_faces):
        cx, cy = i % cols, i // cols
        x0, y0, w, h = cx / cols, cy / rows, 1.0 / cols, 1.0 / rows
        tri = np.array([[margin, margin], [1 - margin, margin + 0.1 * rng.uniform()], [margin + 0.2 * rng.uniform(), 1 - margin]])
        if i % 2:
            tri = tri[::-1]
        uv[3 * i:3 * i + 3] = [x0, y0] + tri * [w, h]
    return np.round(uv, 6), np.arange(3 * n_faces, dtype=np.int32).reshape(-1, 3)
