#!/usr/bin/env python3
"""Closed genus-0 template surfaces for the synthetic body models (bodyfitting_amd/synthetic.py).

No SMPL / SMPL-X model file exists in the build or GPU containers, so the synthetic models need a template mesh of their own.
The reference ships the real topology only as UV templates (smpl_uv/smpl_uv.obj 6890 v / 13776 f, smplx_uv.obj 10475 v /
20908 f); what matters for the kernels is that the synthetic stand-in is a SURFACE of the same
kind: one closed 2-manifold of about a human's area (~1.8 m^2) with near-uniform triangles (median edge ~1.4 cm at 10,475
vertices), because the closest-point search, the silhouette and the SMPL+D stage are all sensitive to triangle size and to how
many triangles lie within reach of a query.  (Rounds 1-2 used disconnected tubes around the bones: 9.9 m^2 of nested sheets
with 5-19 cm edges.)

Construction, all numpy and deterministic:
  1. implicit body = smooth union of round cones around the bones of the rest skeleton (+ skull, heels, toes, finger tips);
  2. marching tetrahedra (Kuhn's 6-tetrahedra split of every grid cube: consistent across cubes, so the result is a closed
     2-manifold) on a uniform grid;
  3. shortest-edge collapses (link condition, no normal flips) down to EXACTLY the requested vertex count, with the vertices
     pulled back onto the implicit surface every time the count has halved, then a few tangential smoothing + projection sweeps;
  4. vertices in Morton order (so every 4th vertex - the silhouette loss's sample, loss.py:99 - covers the body evenly and
     neighbouring vertices are neighbours in memory), faces by smallest vertex, outward orientation.

Output: bodyfitting_amd/data/template_<type>_<nv>.npz (verts float32 [NV,3], faces int32 [2NV-4,3]).  Committed as data; the
statistics are asserted by tests/test_template.py.

    python tools/make_template.py            # the four templates the tests and the bench use
    python tools/make_template.py smplx 10475
"""
from __future__ import annotations

import heapq
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bodyfitting_amd import synthetic as S   # noqa: E402

OUT_DIR = os.path.join(REPO, "bodyfitting_amd", "data")


# ------------------------------------------------------------------------------------------------------------------
# implicit body
# ------------------------------------------------------------------------------------------------------------------

body_primitives = S.body_primitives


def _cone_dist(p, a, b, ra, rb):
    ab = b - a
    L2 = float(ab @ ab)
    if L2 < 1e-12:
        return np.linalg.norm(p - a, axis=-1) - ra
    t = np.clip(((p - a) @ ab) / L2, 0.0, 1.0)
    return np.linalg.norm(p - (a + t[..., None] * ab), axis=-1) - (ra + t * (rb - ra))


def _smin(d, e, k):
    h = np.maximum(k - np.abs(d - e), 0.0) / k
    return np.minimum(d, e) - h * h * (k * 0.25)


def sdf(points, prims, chunk=40000):
    points = np.asarray(points, np.float64)
    out = np.empty(len(points))
    for lo in range(0, len(points), chunk):
        p = points[lo:lo + chunk]
        d = np.full(len(p), 1.0e3)
        for a, b, ra, rb, k, _j in prims:
            d = _smin(d, _cone_dist(p, a, b, ra, rb), k)
        out[lo:lo + chunk] = d
    return out


def sizing(points, prims, r_full=0.02, s_min=0.5, chunk=40000):
    """Relative target edge length in (s_min .. 1]: 1 where the body is thicker than r_full, proportionally smaller on thin parts
    (fingers, nose) so that a tube keeps a handful of vertices around - the real SMPL-X template also spends a large share of its
    vertices on the hands."""
    points = np.asarray(points, np.float64)
    out = np.empty(len(points))
    for lo in range(0, len(points), chunk):
        p = points[lo:lo + chunk]
        best = np.full(len(p), 1.0e3)
        rad = np.full(len(p), r_full)
        for a, b, ra, rb, _, _j in prims:
            ab = b - a
            L2 = float(ab @ ab)
            t = np.clip(((p - a) @ ab) / L2, 0.0, 1.0) if L2 > 1e-12 else np.zeros(len(p))
            r = ra + t * (rb - ra)
            d = np.linalg.norm(p - (a + t[:, None] * ab), axis=1) - r
            take = d < best
            best = np.where(take, d, best)
            rad = np.where(take, r, rad)
        out[lo:lo + chunk] = np.clip(rad / r_full, s_min, 1.0)
    return out


def sdf_grad(points, prims, eps=1e-4):
    g = np.empty((len(points), 3))
    for ax in range(3):
        e = np.zeros(3)
        e[ax] = eps
        g[:, ax] = (sdf(points + e, prims) - sdf(points - e, prims)) / (2 * eps)
    return g


def project(points, prims, sweeps=3):
    p = np.array(points, np.float64)
    for _ in range(sweeps):
        d = sdf(p, prims)
        g = sdf_grad(p, prims)
        n2 = np.maximum((g * g).sum(1), 1e-12)
        gl = np.sqrt(n2)
        step = np.clip(d / gl, -0.006, 0.006)               # (bounded: the gradient is unreliable on the medial axis of thin parts)
        p -= (step / gl)[:, None] * g
    return p


def sdf_grid(prims, h):
    lo = np.min([np.minimum(a, b) - max(ra, rb) for a, b, ra, rb, _, _j in prims], 0) - 3 * h
    hi = np.max([np.maximum(a, b) + max(ra, rb) for a, b, ra, rb, _, _j in prims], 0) + 3 * h
    n = np.ceil((hi - lo) / h).astype(int) + 1
    d = np.full(tuple(n), 1.0e3, np.float64)
    margin = 4 * h
    for a, b, ra, rb, k, _j in prims:
        r = max(ra, rb) + k + margin
        i0 = np.maximum(np.floor((np.minimum(a, b) - r - lo) / h).astype(int), 0)
        i1 = np.minimum(np.ceil((np.maximum(a, b) + r - lo) / h).astype(int) + 1, n)
        ax = [lo[c] + h * np.arange(i0[c], i1[c]) for c in range(3)]
        X, Y, Z = np.meshgrid(*ax, indexing="ij")
        p = np.stack([X, Y, Z], -1)
        sl = tuple(slice(i0[c], i1[c]) for c in range(3))
        d[sl] = _smin(d[sl], _cone_dist(p, a, b, ra, rb), k)
    d[d == 0.0] = 1e-9
    return d, lo


# ------------------------------------------------------------------------------------------------------------------
# marching tetrahedra
# ------------------------------------------------------------------------------------------------------------------

_CORNER = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]])


def _kuhn_tets():
    tets = []
    for perm in ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)):
        c = np.zeros(3, int)
        path = [0]
        for ax in perm:
            c[ax] = 1
            path.append(int(c[0] + 2 * c[1] + 4 * c[2]))
        tets.append(path)
    return np.array(tets)          # [6,4] cube-corner ids


def _tet_table():
    """per 4-bit inside mask: up to two triangles, each three tetrahedron edges (pairs of tet-vertex ids)"""
    cnt = np.zeros(16, int)
    tab = np.zeros((16, 2, 3, 2), int)
    for c in range(1, 15):
        ins = [i for i in range(4) if c >> i & 1]
        out = [i for i in range(4) if not c >> i & 1]
        if len(ins) == 1 or len(ins) == 3:
            lone = ins[0] if len(ins) == 1 else out[0]
            oth = [i for i in range(4) if i != lone]
            cnt[c] = 1
            tab[c, 0] = [[lone, oth[0]], [lone, oth[1]], [lone, oth[2]]]
        else:
            a, b = ins
            p, q = out
            cnt[c] = 2
            tab[c, 0] = [[a, p], [a, q], [b, q]]
            tab[c, 1] = [[a, p], [b, q], [b, p]]
    return cnt, tab


def marching_tets(d, lo, h):
    nx, ny, nz = d.shape
    inside = d < 0
    # cubes whose corners straddle the level
    s = np.zeros((nx - 1, ny - 1, nz - 1), np.int8)
    for cx, cy, cz in _CORNER:
        s += inside[cx:nx - 1 + cx, cy:ny - 1 + cy, cz:nz - 1 + cz]
    act = np.argwhere((s > 0) & (s < 8))                         # [C,3]
    lin = lambda ijk: (ijk[..., 0] * ny + ijk[..., 1]) * nz + ijk[..., 2]
    corner_ids = lin(act[:, None, :] + _CORNER[None])            # [C,8] linear grid ids
    dflat = d.reshape(-1)
    cnt, tab = _tet_table()
    tris_a, tris_b, pos_c, neg_c = [], [], [], []
    for tet in _kuhn_tets():
        ids = corner_ids[:, tet]                                 # [C,4]
        val = dflat[ids]
        mask = ((val < 0) * np.array([1, 2, 4, 8])).sum(1)
        for t in range(2):
            sel = np.nonzero(cnt[mask] > t)[0]
            if not len(sel):
                continue
            e = tab[mask[sel], t]                                # [n,3,2] tet-vertex ids
            ga = np.take_along_axis(ids[sel], e[:, :, 0], 1)     # [n,3] grid ids of edge ends
            gb = np.take_along_axis(ids[sel], e[:, :, 1], 1)
            tris_a.append(ga)
            tris_b.append(gb)
            # orientation reference: from the inside corners towards the outside corners of the tetrahedron
            v = val[sel]
            pts = _grid_pos(ids[sel], ny, nz, lo, h)             # [n,4,3]
            wneg = (v < 0)[..., None]
            neg_c.append((pts * wneg).sum(1) / wneg.sum(1))
            pos_c.append((pts * ~wneg).sum(1) / (~wneg).sum(1))
    ga, gb = np.concatenate(tris_a), np.concatenate(tris_b)
    out_dir = np.concatenate(pos_c) - np.concatenate(neg_c)
    ea, eb = np.minimum(ga, gb), np.maximum(ga, gb)
    key = ea.astype(np.int64) * d.size + eb
    uniq, inv = np.unique(key.reshape(-1), return_inverse=True)
    ua, ub = uniq // d.size, uniq % d.size
    va, vb = dflat[ua], dflat[ub]
    t = va / (va - vb)
    pa = _grid_pos(ua, ny, nz, lo, h)
    pb = _grid_pos(ub, ny, nz, lo, h)
    verts = pa + t[:, None] * (pb - pa)
    faces = inv.reshape(-1, 3)
    n = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
    flip = (n * out_dir).sum(1) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]
    # drop degenerate triangles (two corners on the same grid edge cannot happen; zero-area ones can when t hits 0 / 1)
    return verts, faces


def _grid_pos(ids, ny, nz, lo, h):
    ids = np.asarray(ids)
    z = ids % nz
    y = (ids // nz) % ny
    x = ids // (nz * ny)
    return lo + h * np.stack([x, y, z], -1).astype(np.float64)


def largest_component(verts, faces):
    parent = np.arange(len(verts))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    # union by iterating edges with numpy-assisted label propagation
    lab = np.arange(len(verts))
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    while True:
        m = np.minimum(lab[e[:, 0]], lab[e[:, 1]])
        new = lab.copy()
        np.minimum.at(new, e[:, 0], m)
        np.minimum.at(new, e[:, 1], m)
        if np.array_equal(new, lab):
            break
        lab = new
    ids, counts = np.unique(lab, return_counts=True)
    keep_lab = ids[np.argmax(counts)]
    keepv = lab == keep_lab
    remap = -np.ones(len(verts), int)
    remap[keepv] = np.arange(keepv.sum())
    f = faces[keepv[faces[:, 0]]]
    return verts[keepv], remap[f], len(ids)


def mesh_stats(verts, faces, prims=None):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    es = np.sort(e, 1)
    uniq, counts = np.unique(es[:, 0].astype(np.int64) * len(verts) + es[:, 1], return_counts=True)
    # every directed edge once <=> consistently oriented
    dkey = e[:, 0].astype(np.int64) * len(verts) + e[:, 1]
    elen = np.linalg.norm(verts[uniq // len(verts)] - verts[uniq % len(verts)], axis=1)
    n = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
    area = 0.5 * np.linalg.norm(n, axis=1)
    vol = (verts[faces[:, 0]] * n).sum() / 6.0
    rel = {}
    if prims is not None:
        sz = sizing(verts, prims)
        r = elen / (0.5 * (sz[uniq // len(verts)] + sz[uniq % len(verts)]))
        rel = {"edge_rel_cm": {q: float(np.quantile(r, q / 100) * 100) for q in (5, 25, 50, 75, 95, 100)},
               "verts_on_thin_parts": float((sz < 0.999).mean())}
    return {**rel, "n_verts": len(verts), "n_faces": len(faces), "n_edges": len(uniq),
            "closed_manifold": bool((counts == 2).all()), "oriented": len(np.unique(dkey)) == len(dkey),
            "euler": len(verts) - len(uniq) + len(faces), "area_m2": float(area.sum()), "volume_m3": float(vol),
            "edge_cm": {q: float(np.quantile(elen, q / 100) * 100) for q in (5, 25, 50, 75, 95, 100)},
            "min_face_area_cm2": float(area.min() * 1e4), "min_quality": float(_quality(verts[faces], 2 * area).min()),
            "height_m": float(verts[:, 1].max() - verts[:, 1].min())}


# ------------------------------------------------------------------------------------------------------------------
# decimation to an exact vertex count
# ------------------------------------------------------------------------------------------------------------------

class Decimator:
    def __init__(self, verts, faces, prims):
        self.prims = prims
        self.size = sizing(verts, prims)
        self.v = np.array(verts, np.float64)
        self.f = np.array(faces, np.int64)
        self.alive_f = np.ones(len(faces), bool)
        self.alive_v = np.ones(len(verts), bool)
        self.vf = [set() for _ in range(len(verts))]
        for i, (a, b, c) in enumerate(self.f):
            self.vf[a].add(i); self.vf[b].add(i); self.vf[c].add(i)
        self.ver = np.zeros(len(verts), np.int64)
        self.n_alive = len(verts)
        self.n_v, self.n_f = len(verts), len(faces)          # used rows of the (growable) arrays
        self.tiny = 0.0             # relative edge length below which the sliver guard is off (set per target in run)
        self.qmin = 0.02            # a collapse may not CREATE a (near-)degenerate face; halved when a full pass finds nothing to collapse
        self.heap = []
        self.rebuild_heap()

    def neighbours(self, u):
        out = set()
        for fi in self.vf[u]:
            out.update(self.f[fi].tolist())
        out.discard(u)
        return out

    def rebuild_heap(self):
        fa = self.f[:self.n_f][self.alive_f[:self.n_f]]
        e = np.concatenate([fa[:, [0, 1]], fa[:, [1, 2]], fa[:, [2, 0]]])
        e = np.unique(np.sort(e, 1), axis=0)
        L = np.linalg.norm(self.v[e[:, 0]] - self.v[e[:, 1]], axis=1) / (0.5 * (self.size[e[:, 0]] + self.size[e[:, 1]]))
        self.heap = [(float(l), int(a), int(b), int(self.ver[a]), int(self.ver[b])) for l, (a, b) in zip(L, e)]
        heapq.heapify(self.heap)

    def try_collapse(self, u, w):
        """merge w into u at the midpoint; False if it would break the manifold or flip a face"""
        fu, fw = self.vf[u], self.vf[w]
        shared = fu & fw
        if len(shared) != 2:
            return False
        nu, nw = self.neighbours(u), self.neighbours(w)
        opp = set()
        for fi in shared:
            opp.update(x for x in self.f[fi].tolist() if x != u and x != w)
        if (nu & nw) != opp or len(opp) != 2:
            return False
        if len(nu | nw) - 2 < 3:
            return False
        p = 0.5 * (self.v[u] + self.v[w])
        coincident = float(np.abs(self.v[u] - self.v[w]).max()) < 1e-9       # (merging them changes no geometry: only the link condition counts)
        ring = np.array(sorted((fu | fw) - shared))
        tri = self.f[ring]
        before = self.v[tri]
        after = before.copy()
        moved = (tri == u) | (tri == w)
        after[moved] = p
        n0 = np.cross(before[:, 1] - before[:, 0], before[:, 2] - before[:, 0])
        n1 = np.cross(after[:, 1] - after[:, 0], after[:, 2] - after[:, 0])
        l0 = np.linalg.norm(n0, axis=1)
        l1 = np.linalg.norm(n1, axis=1)
        q0, q1 = _quality(before, l0), _quality(after, l1)
        # (a face that is a sliver already has no normal worth protecting)
        if not coincident:
            if (l1 < 1e-14).any() or (((n0 * n1).sum(1) < 0.2 * l0 * l1) & (q0 > 0.05)).any():
                return False
            # never create a sliver or make one worse - except when the edge itself is tiny: the faces that turn into slivers then
            # do so because they hold ANOTHER tiny edge of the same cluster of near-coincident vertices, which goes next
            rel = float(np.linalg.norm(self.v[u] - self.v[w])) / (0.5 * (self.size[u] + self.size[w]))
            if rel > self.tiny and (q1 < np.minimum(self.qmin, q0)).any():
                return False
        # commit
        for fi in shared:
            self.alive_f[fi] = False
            for x in self.f[fi].tolist():
                self.vf[x].discard(fi)
        for fi in list(fw):
            row = self.f[fi]
            row[row == w] = u
            fu.add(fi)
        self.vf[w] = set()
        self.alive_v[w] = False
        self.v[u] = p
        self.size[u] = 0.5 * (self.size[u] + self.size[w])
        self.ver[u] += 1
        self.ver[w] += 1
        self.n_alive -= 1
        self.offer_around(u)
        return True

    def offer_around(self, u):
        """(re-)offer every edge of the faces around u: the spokes and the rim (a rim edge rejected earlier may be fine now)"""
        seen = set()
        for fi in self.vf[u]:
            a, b, c = self.f[fi].tolist()
            for x, y in ((a, b), (b, c), (c, a)):
                e = (x, y) if x < y else (y, x)
                if e not in seen:
                    seen.add(e)
                    heapq.heappush(self.heap, (float(np.linalg.norm(self.v[x] - self.v[y]) / (0.5 * (self.size[x] + self.size[y]))),
                                               e[0], e[1], int(self.ver[e[0]]), int(self.ver[e[1]])))

    def _grow(self, nv, nf):
        if self.n_v + nv > len(self.v):
            extra = max(1024, nv)
            self.v = np.concatenate([self.v, np.zeros((extra, 3))])
            self.size = np.concatenate([self.size, np.ones(extra)])
            self.alive_v = np.concatenate([self.alive_v, np.zeros(extra, bool)])
            self.ver = np.concatenate([self.ver, np.zeros(extra, np.int64)])
            self.vf.extend(set() for _ in range(extra))
        if self.n_f + nf > len(self.f):
            extra = max(2048, nf)
            self.f = np.concatenate([self.f, np.zeros((extra, 3), np.int64)])
            self.alive_f = np.concatenate([self.alive_f, np.zeros(extra, bool)])

    def split(self, a, b, prims):
        """midpoint split of edge (a, b): +1 vertex, +2 faces; always keeps the manifold"""
        shared = sorted(self.vf[a] & self.vf[b])
        if len(shared) != 2:
            return False
        self._grow(1, 2)
        m = self.n_v
        self.n_v += 1
        self.v[m] = project(0.5 * (self.v[a] + self.v[b])[None], prims, sweeps=2)[0]
        self.size[m] = sizing(self.v[m][None], prims)[0]
        self.alive_v[m] = True
        self.n_alive += 1
        for fi in shared:
            row = self.f[fi].copy()
            # (.., a, b, ..) in this face's cyclic order, or (.., b, a, ..)
            ia = int(np.nonzero(row == a)[0][0])
            fwd = row[(ia + 1) % 3] == b
            x, y = (a, b) if fwd else (b, a)
            c = int([v for v in row if v != a and v != b][0])
            # (x, y, c) -> (x, m, c) + (m, y, c)
            self.f[fi] = [x, m, c]
            nf = self.n_f
            self.n_f += 1
            self.f[nf] = [m, y, c]
            self.alive_f[nf] = True
            self.vf[y].discard(fi)
            self.vf[y].add(nf); self.vf[c].add(nf); self.vf[m].add(fi); self.vf[m].add(nf)
        self.ver[a] += 1; self.ver[b] += 1
        self.offer_around(m)
        return True

    def edge_lengths(self):
        """edges and their lengths RELATIVE to the local target size"""
        fa = self.f[:self.n_f][self.alive_f[:self.n_f]]
        e = np.concatenate([fa[:, [0, 1]], fa[:, [1, 2]], fa[:, [2, 0]]])
        e = np.unique(np.sort(e, 1), axis=0)
        return e, np.linalg.norm(self.v[e[:, 0]] - self.v[e[:, 1]], axis=1) / (0.5 * (self.size[e[:, 0]] + self.size[e[:, 1]]))

    def equalize(self, prims, ratio=0.45, max_pairs=4000):
        """At a fixed vertex count: collapse the shortest edge, split the longest, until the shortest edge is at least `ratio` of the
        median (the classic split / collapse remeshing step; every pair keeps the count)."""
        target = self.n_alive
        pairs = 0
        while pairs < max_pairs:
            e, L = self.edge_lengths()
            med = float(np.median(L))
            short = np.argsort(L)
            if L[short[0]] >= ratio * med:
                break
            n_short = int((L < ratio * med).sum())
            done = 0
            for i in short[:n_short]:
                a, b = int(e[i, 0]), int(e[i, 1])
                if self.alive_v[a] and self.alive_v[b] and len(self.vf[a] & self.vf[b]) == 2 and self.try_collapse(a, b):
                    done += 1
            if not done:
                break
            pairs += done
            while self.n_alive < target:
                e, L = self.edge_lengths()
                for i in np.argsort(-L)[:max(1, (target - self.n_alive))]:
                    if self.n_alive >= target:
                        break
                    self.split(int(e[i, 0]), int(e[i, 1]), prims)
        self.rebuild_heap()
        return pairs

    def run(self, target, prims):
        fa = self.f[:self.n_f][self.alive_f[:self.n_f]]
        area = 0.5 * np.linalg.norm(np.cross(self.v[fa[:, 1]] - self.v[fa[:, 0]], self.v[fa[:, 2]] - self.v[fa[:, 0]]), axis=1).sum()
        self.tiny = 0.3 * np.sqrt(area / (2 * target) * 4 / np.sqrt(3.0))      # 0.3 x the edge of the target's average triangle
        next_project = max(target, self.n_alive // 2)
        rejected = []
        at_rebuild = self.n_alive
        while self.n_alive > target:
            if not self.heap:
                if self.n_alive == at_rebuild:          # a whole pass without one collapse: loosen the sliver guard
                    self.qmin *= 0.5
                    if self.qmin < 1e-4:
                        raise RuntimeError(f"decimation stuck at {self.n_alive} vertices")
                self.rebuild_heap()
                at_rebuild = self.n_alive
                rejected = []
            l, a, b, va, vb = heapq.heappop(self.heap)
            if not (self.alive_v[a] and self.alive_v[b]) or self.ver[a] != va or self.ver[b] != vb:
                continue
            if not self.try_collapse(a, b):
                rejected.append((a, b))
                continue
            if self.n_alive <= next_project and self.n_alive > target:
                idx = np.nonzero(self.alive_v)[0]
                remap = -np.ones(len(self.v), int)
                remap[idx] = np.arange(len(idx))
                self.size[idx] = sizing(self.v[idx], prims)
                self.v[idx] = guarded_move(self.v[idx], project(self.v[idx], prims, sweeps=2), remap[self.f[:self.n_f][self.alive_f[:self.n_f]]])
                self.rebuild_heap()
                next_project = max(target, self.n_alive // 2)
        self.qmin = 0.02
        pairs = self.equalize(prims)
        print(f"  equalize at {self.n_alive}: {pairs} collapse / split pairs", flush=True)
        idx = np.nonzero(self.alive_v)[0]
        remap = -np.ones(len(self.v), int)
        remap[idx] = np.arange(len(idx))
        return self.v[idx].copy(), remap[self.f[:self.n_f][self.alive_f[:self.n_f]]]


def _quality(tri, twice_area):
    """2 sqrt(3) |n| / sum of squared edges: 1 for an equilateral triangle, 0 for a degenerate one"""
    e2 = ((tri[:, 1] - tri[:, 0]) ** 2).sum(1) + ((tri[:, 2] - tri[:, 1]) ** 2).sum(1) + ((tri[:, 0] - tri[:, 2]) ** 2).sum(1)
    return 2.0 * np.sqrt(3.0) * twice_area / np.maximum(e2, 1e-30)


def relax(verts, faces, prims, sweeps=6, lam=0.5):
    """tangential Laplacian smoothing with re-projection: evens the triangles out without moving the surface"""
    v = verts.copy()
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    deg = np.bincount(e[:, 0], minlength=len(v)).astype(float)
    for _ in range(sweeps):
        acc = np.zeros_like(v)
        np.add.at(acc, e[:, 0], v[e[:, 1]])
        cen = acc / deg[:, None]
        g = sdf_grad(v, prims)
        g /= np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-12)
        d = cen - v
        d -= (d * g).sum(1, keepdims=True) * g
        cand = project(v + lam * d, prims, sweeps=2)
        # keep a move only if no incident triangle flips
        n0 = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
        n1 = np.cross(cand[faces[:, 1]] - cand[faces[:, 0]], cand[faces[:, 2]] - cand[faces[:, 0]])
        l0, l1 = np.linalg.norm(n0, axis=1), np.linalg.norm(n1, axis=1)
        bad = ((n0 * n1).sum(1) < 0.3 * l0 * l1) | (_quality(cand[faces], l1) < np.minimum(0.15, _quality(v[faces], l0)))
        freeze = np.zeros(len(v), bool)
        freeze[faces[bad].reshape(-1)] = True
        cand[freeze] = v[freeze]
        v = cand
    return v


def flip_slivers(verts, faces, qbad=0.25, rounds=6):
    """Edge flips around badly shaped faces: the longest edge (a, b) of a bad face (a, b, c), shared with (b, a, d), becomes (c, d)
    when that edge does not exist yet, the two new faces are better shaped than the worse of the old pair and keep the
    orientation.  A flip keeps the surface a closed 2-manifold of the same genus."""
    v = verts
    f = faces.copy()
    for _ in range(rounds):
        edge_face = {}
        for i, (a, b, c) in enumerate(f):
            edge_face[(a, b)] = i; edge_face[(b, c)] = i; edge_face[(c, a)] = i
        tri = v[f]
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        q = _quality(tri, np.linalg.norm(n, axis=1))
        touched = set()
        flips = 0
        for i in np.argsort(q):
            if q[i] >= qbad:
                break
            if i in touched:
                continue
            fi = f[i]
            L = [np.linalg.norm(v[fi[(k + 1) % 3]] - v[fi[k]]) for k in range(3)]
            k = int(np.argmax(L))
            a, b, c = int(fi[k]), int(fi[(k + 1) % 3]), int(fi[(k + 2) % 3])
            j = edge_face.get((b, a))
            if j is None or j in touched:
                continue
            fj = f[j]
            d = int([x for x in fj if x != a and x != b][0])
            if (c, d) in edge_face or (d, c) in edge_face or c == d:
                continue
            new = np.array([[c, d, b], [d, c, a]])
            tn = v[new]
            nn = np.cross(tn[:, 1] - tn[:, 0], tn[:, 2] - tn[:, 0])
            qn = _quality(tn, np.linalg.norm(nn, axis=1))
            nref = n[i] + n[j]
            if qn.min() <= min(q[i], q[j]) + 1e-9 or (nn @ nref).min() <= 0:
                continue
            f[i], f[j] = new[0], new[1]
            touched.update((i, j))
            # (edges of the touched pair are stale in edge_face: their faces are skipped until the next round)
            for e in ((a, b), (b, a)):
                edge_face.pop(e, None)
            edge_face[(c, d)] = i; edge_face[(d, c)] = j
            flips += 1
        if not flips:
            break
    return f


def guarded_move(v, cand, faces):
    """move v -> cand except where an incident triangle would flip or degenerate"""
    cand = cand.copy()
    for _ in range(4):
        n0 = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
        n1 = np.cross(cand[faces[:, 1]] - cand[faces[:, 0]], cand[faces[:, 2]] - cand[faces[:, 0]])
        l0, l1 = np.linalg.norm(n0, axis=1), np.linalg.norm(n1, axis=1)
        bad = ((n0 * n1).sum(1) < 0.3 * l0 * l1) | (_quality(cand[faces], l1) < np.minimum(0.12, _quality(v[faces], l0)))
        if not bad.any():
            break
        idx = np.unique(faces[bad].reshape(-1))
        cand[idx] = v[idx]
    return cand


def morton_order(verts):
    lo, hi = verts.min(0), verts.max(0)
    q = np.minimum(((verts - lo) / (hi - lo).max() * 1023.0).astype(np.int64), 1023)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return np.argsort(code, kind="stable")


def make_templates(model_type, counts, verbose=True):
    """templates of one model type for several vertex counts (descending): one marching pass, one decimation run"""
    t0 = time.perf_counter()
    prims = body_primitives(model_type)
    # grid fine enough for the thinnest feature and for ~4x the wanted vertex count before decimation
    h = 0.004 if model_type == "smplx" else 0.006
    d, lo = sdf_grid(prims, h)
    verts, faces = marching_tets(d, lo, h)
    verts, faces, ncomp = largest_component(verts, faces)
    st = mesh_stats(verts, faces)
    if verbose:
        print(f"[{model_type}] marching tetrahedra h={h}: {st['n_verts']} v, {st['n_faces']} f, components {ncomp}, "
              f"euler {st['euler']}, closed {st['closed_manifold']}, area {st['area_m2']:.3f} m^2, {time.perf_counter() - t0:.1f} s")
    assert st["closed_manifold"] and st["euler"] == 2, st
    dec = Decimator(verts, faces, prims)
    out = []
    for nv in sorted(counts, reverse=True):
        verts, faces = dec.run(nv, prims)
        out.append(_finish(model_type, nv, verts, faces, prims, verbose, t0))
    return out


def _finish(model_type, nv, verts, faces, prims, verbose, t0):
    verts = guarded_move(verts, project(verts, prims, sweeps=3), faces)
    for _ in range(3):
        faces = flip_slivers(verts, faces)
        verts = relax(verts, faces, prims, sweeps=3)
    order = morton_order(verts)
    inv = np.empty(len(order), int)
    inv[order] = np.arange(len(order))
    verts, faces = verts[order], inv[faces]
    # rotate every face so its smallest vertex comes first, then sort the faces
    r = np.argmin(faces, 1)
    faces = np.stack([np.take_along_axis(faces, ((r + k) % 3)[:, None], 1)[:, 0] for k in range(3)], 1)
    faces = faces[np.lexsort((faces[:, 2], faces[:, 1], faces[:, 0]))]
    st = mesh_stats(verts, faces, prims)
    if verbose:
        print(f"[{model_type} {nv}] final: {st}  ({time.perf_counter() - t0:.1f} s)")
    assert st["closed_manifold"] and st["oriented"] and st["euler"] == 2 and st["n_verts"] == nv and st["volume_m3"] > 0, st
    return verts.astype(np.float32), faces.astype(np.int32), st


def main():
    jobs = {"smpl": [6890, 690], "smplx": [10475, 1200]}
    if len(sys.argv) >= 3:
        jobs = {sys.argv[1]: [int(x) for x in sys.argv[2:]]}
    os.makedirs(OUT_DIR, exist_ok=True)
    for mt, counts in jobs.items():
        for v, f, st in make_templates(mt, counts):
            np.savez_compressed(os.path.join(OUT_DIR, f"template_{mt}_{len(v)}.npz"), verts=v, faces=f)


if __name__ == "__main__":
    main()
