"""Adversarial geometry for the closest-point rule of the reference's mesh_grid extension.  TEST INFRASTRUCTURE ONLY.

thirdparty/mesh_grid/mesh_grid_kernel.cu:12-109 (`search_nearest_proj`) solves the KKT system of the projection onto the
triangle's plane and, when a barycentric coefficient comes out negative, falls back to the edge OPPOSITE THE MOST NEGATIVE
coefficient, clamped to its end points (kernel.cu:74-101).  That is the exact closest point for every non-obtuse triangle
(measured: 0 of 150,000 random acute cases differ from Ericson's exact routine) but not for obtuse ones: beyond the obtuse
corner the fallback edge can be the wrong one, and the rule then returns a VERTEX although another edge - or another
vertex - is closer (about 4 % of random queries around random obtuse triangles, by up to a factor of 18 in squared
distance).  The reference extension cannot be built in this container, so the restatement (oracle/mesh_oracle.closest_rule)
and the HIP kernel are pinned on exactly these inputs by independent exact geometry: `voronoi_region` below classifies a
query by Ericson's region tests, which says where the rule must agree with the exact answer and where it may not.

`soup(...)` builds a scan made of isolated needle / obtuse / sliver / regular triangles on a lattice, with queries spread
over all seven Voronoi regions of each, and drops the queries whose branch decisions (which coefficient is most negative,
clamped or not) are too close to call between float32 and float64.
"""
from __future__ import annotations

import numpy as np

from . import mesh_oracle as MO

SHAPES = ("regular", "right", "needle", "obtuse150", "obtuse175", "sliver")


def shape(name, rng):
    """one triangle [3,3] in its own plane (z = 0), longest edge ~1"""
    if name == "regular":
        t = np.array([[0, 0], [1, 0], [0.5, 0.8660254]])
    elif name == "right":
        t = np.array([[0, 0], [1, 0], [0, 0.6]])
    elif name == "needle":                      # aspect 1 : 400, all angles acute or right
        t = np.array([[0, 0], [1, 0], [0.5, 0.0025]]) if rng.random() < 0.5 else np.array([[0, 0], [1, 0], [1.0, 0.0025]])
    elif name == "obtuse150":
        t = np.array([[0, 0], [1, 0], [0.3, 0.08]])
    elif name == "obtuse175":
        t = np.array([[0, 0], [1, 0], [0.25, 0.011]])
    elif name == "sliver":                      # two long edges, one very short
        t = np.array([[0, 0], [1, 0.004], [1, -0.004]])
    else:
        raise ValueError(name)
    return np.concatenate([t, np.zeros((3, 1))], 1)


def _rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def voronoi_region(tri, q):
    """Ericson (Real-Time Collision Detection 5.1.5) region of q w.r.t. triangle tri[3,3]: 'A' 'B' 'C' (vertex regions),
    'AB' 'BC' 'CA' (edge regions), 'F' (face).  float64."""
    a, b, c = (np.asarray(t, np.float64) for t in tri)
    p = np.asarray(q, np.float64)
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = ab @ ap, ac @ ap
    if d1 <= 0 and d2 <= 0:
        return "A"
    bp = p - b
    d3, d4 = ab @ bp, ac @ bp
    if d3 >= 0 and d4 <= d3:
        return "B"
    if d1 * d4 - d3 * d2 <= 0 and d1 >= 0 and d3 <= 0:
        return "AB"
    cp = p - c
    d5, d6 = ab @ cp, ac @ cp
    if d6 >= 0 and d5 <= d6:
        return "C"
    if d5 * d2 - d1 * d6 <= 0 and d2 >= 0 and d6 <= 0:
        return "CA"
    if d3 * d6 - d5 * d4 <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0:
        return "BC"
    return "F"


def decision_margin(tri, q):
    """how far (relative) the rule's branch decisions for (tri, q) are from flipping: min over |smallest coefficient| (inside or
    not), the gap between the two smallest coefficients (which edge), and the distance of the edge parameter from 0 and 1
    (clamped or not).  Queries with a tiny margin are dropped from float32-vs-float64 comparisons."""
    p = np.asarray(tri, np.float64) - np.asarray(q, np.float64)
    e1, e2 = p[1] - p[0], p[2] - p[0]
    a11, a12, a22 = e1 @ e1, e1 @ e2, e2 @ e2
    det = a11 * a22 - a12 * a12
    u = (-(p[0] @ e1) * a22 + (p[0] @ e2) * a12) / det
    v = (-a11 * (p[0] @ e2) + a12 * (p[0] @ e1)) / det
    c = np.array([1 - u - v, u, v])
    s = np.sort(c)
    m = abs(s[0]) / max(1.0, np.abs(c).max())
    if s[0] < 0:
        m = min(m, (s[1] - s[0]) / max(1.0, np.abs(c).max()))
        i = int(np.argmin(c))
        j, k = (i + 1) % 3, (i + 2) % 3
        d = p[k] - p[j]
        t = -(p[j] @ d) / (d @ d)
        m = min(m, abs(t), abs(t - 1))
    return m


def soup(seed=0, per_shape=6, queries_per_triangle=40, spacing=4.0, min_margin=2e-3):
    """-> dict(verts f32[3T,3], faces i32[T,3], queries f32[Q,3], owner i32[Q] (the triangle a query was placed around),
    kind [T] shape names, region [Q] Ericson region w.r.t. the owner)"""
    rng = np.random.default_rng(seed)
    tris, kinds = [], []
    for name in SHAPES:
        for _ in range(per_shape):
            tris.append(shape(name, rng) * rng.uniform(0.3, 1.0) @ _rotation(rng).T)
            kinds.append(name)
    T = len(tris)
    side = int(np.ceil(T ** (1 / 3)))
    verts, queries, owner, region = [], [], [], []
    for t, tri in enumerate(tris):
        centre = np.array([t % side, (t // side) % side, t // (side * side)], np.float64) * spacing
        tri = (tri - tri.mean(0) + centre).astype(np.float32)          # the scan stores float32 vertices
        verts.append(tri)
        t64 = tri.astype(np.float64)
        e1, e2 = t64[1] - t64[0], t64[2] - t64[0]
        n = np.cross(e1, e2)
        n /= np.linalg.norm(n)
        size = max(np.linalg.norm(e1), np.linalg.norm(e2))
        made = 0
        while made < queries_per_triangle:
            # in-plane position around the triangle (barycentric blob stretched beyond every edge and corner) + a lift
            if kinds[t].startswith("obtuse") and rng.random() < 0.5:
                # around the obtuse corner (vertex 2), where the fallback edge can be the wrong one
                ang = rng.uniform(0, 2 * np.pi)
                u = e1 / np.linalg.norm(e1)
                q = t64[2] + size * rng.uniform(0.05, 1.2) * (np.cos(ang) * u + np.sin(ang) * np.cross(n, u))
            else:
                w = rng.dirichlet([0.6, 0.6, 0.6]) * rng.uniform(1.0, 3.0) - rng.uniform(0.0, 0.7, 3)
                q = (w / w.sum()) @ t64 if abs(w.sum()) > 0.2 else t64.mean(0)
            q = q + n * rng.normal(0, 0.3) * size * (rng.random() < 0.7)
            if np.linalg.norm(q - centre) > 0.45 * spacing:
                continue
            q = q.astype(np.float32)
            if decision_margin(tri, q) < min_margin:
                continue
            queries.append(q)
            owner.append(t)
            region.append(voronoi_region(tri, q))
            made += 1
    return {"verts": np.concatenate(verts).astype(np.float32), "faces": np.arange(3 * T, dtype=np.int32).reshape(T, 3),
            "queries": np.stack(queries).astype(np.float32), "owner": np.array(owner, np.int32), "kind": kinds,
            "region": np.array(region)}


def exact_point(tri, q):
    """Ericson's closest point itself (float64) for one triangle / query"""
    a, b, c = (np.asarray(t, np.float64) for t in tri)
    r = voronoi_region(tri, q)
    p = np.asarray(q, np.float64)
    if r in "ABC" and len(r) == 1:
        return {"A": a, "B": b, "C": c}[r]
    if r == "F":
        n = np.cross(b - a, c - a)
        return p - n * ((p - a) @ n) / (n @ n)
    x, y = {"AB": (a, b), "BC": (b, c), "CA": (c, a)}[r]
    d = y - x
    return x + d * ((p - x) @ d) / (d @ d)


def rule_vs_exact(data):
    """per query of a soup: (rule squared distance, exact squared distance) to the OWNER triangle, float64"""
    V = data["verts"].astype(np.float64)
    tri = V[data["faces"][data["owner"]]]
    q = data["queries"].astype(np.float64)
    _, d_rule = MO.closest_rule(tri[:, 0] - q, tri[:, 1] - q, tri[:, 2] - q)
    d_exact = MO.closest_exact(tri[:, 0] - q, tri[:, 1] - q, tri[:, 2] - q)
    return d_rule, d_exact
