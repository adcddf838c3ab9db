"""CPU oracle for the scan closest-point path (config 5).  TEST INFRASTRUCTURE ONLY.

Restates, in numpy / torch:
  * MeshGridSearcher.set_mesh grid parameters      reference utils/mesh_grid_searcher.py:56-79
  * insert_grid_surface (cell lists of the grid)    thirdparty/mesh_grid/mesh_grid_kernel.cu:110-157,178-236
  * MeshGridSearcher.inside_mesh (axis-ray parity)   thirdparty/mesh_grid/mesh_grid_kernel.cu:461-641
  * MeshGridSearcher.intersects_any (any-hit rays)   thirdparty/mesh_grid/mesh_grid_kernel.cu:742-1026,1029-1231
  * the per-triangle closest-point rule             thirdparty/mesh_grid/mesh_grid_kernel.cu:12-109
    (KKT solve for the barycentric coefficients; if one is negative, fall back to the edge opposite
    the MOST NEGATIVE coefficient and clamp to its end points - which is not the exact closest point
    in some obtuse configurations; reproduced on purpose)
  * nearest face / point / barycentrics per query   mesh_grid_kernel.cu:239-353 (here: brute force
    over all faces - the grid only prunes, it does not change the argmin except on exact ties)
  * point_cloud_loss_mesh_grid, normal_loss_mesh_grid, normal_laplacian_smoothness
                                                    smplify/loss.py:233-242,260-288
  * compute_normal_torch                            utils/io_utils.py:406-428

PARITY PINNING: the reference's `mesh_grid` CUDA extension cannot be built here (needs the CUDA
toolkit headers / ATen extension build; writing stand-in headers is not allowed), so the search
itself is pinned by (i) an independent exact closest-point-on-triangle routine (Ericson) - equal
wherever the rule's fallback is exact, and never closer - and (ii) goldens of the *imported* reference
losses (loss.py / io_utils.py run unmodified on torch CPU) fed with this module's nearest points
through a stand-in MeshGridSearcher (oracle/gen_golden.py).
"""
from __future__ import annotations

import numpy as np
import torch


def grid_params(verts):
    """step, cell counts l[3], origin[3] exactly as set_mesh computes them (float32 torch ops)."""
    v = torch.as_tensor(np.asarray(verts), dtype=torch.float32)
    _min, _max = v.min(0)[0], v.max(0)[0]
    step = (torch.cumprod(_max - _min, 0)[-1] / len(v)) ** (1.0 / 3.0)
    l = _max - _min
    c = (_max + _min) / 2
    l = torch.max(torch.floor(l / step), torch.zeros_like(l)) + 1
    origin = c - step * l / 2
    return float(step), l.numpy().astype(np.int64), origin.numpy().astype(np.float32)


def insert_grid_surface(verts, faces, step, origin, num):
    """insert_grid_surface_kernel + the cumulative sum between its two passes (mesh_grid_kernel.cu:110-157,209):
    every triangle enters each cell its axis-aligned bounding box touches (float32 `(x - min) / step`, clamped to
    the grid, kernel.cu:127-141).  -> (tri_num int32[cells] inclusive cumsum, tri_idx int32[entries] = face id + 1,
    ascending inside a cell - the reference's own order inside a cell is decided by an atomicCAS race)."""
    v = np.asarray(verts, np.float32)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    num = np.asarray(num, np.int64)
    org = np.asarray(origin, np.float32)
    step = np.float32(step)
    tri = v[f]                                                        # [F,3 corners,3]
    lo_f = (tri.min(1) - org) / step
    hi_f = (tri.max(1) - org) / step

    def cell(x):
        c = np.floor(x).astype(np.int64)
        c = np.where(x < 0, 0, c)
        return np.where(x >= num[None].astype(np.float32), num[None] - 1, c)
    lo, hi = cell(lo_f), cell(hi_f) + 1
    cells, ids = [], []
    for t in range(len(f)):
        xs, ys, zs = (np.arange(lo[t, d], hi[t, d]) for d in range(3))
        c = ((xs[:, None, None] * num[1] + ys[None, :, None]) * num[2] + zs[None, None, :]).reshape(-1)
        cells.append(c)
        ids.append(np.full(len(c), t + 1, np.int64))
    cells, ids = np.concatenate(cells), np.concatenate(ids)
    order = np.lexsort((ids, cells))
    tri_num = np.cumsum(np.bincount(cells, minlength=int(num.prod()))).astype(np.int32)
    return tri_num, ids[order].astype(np.int32)


def axis_ray_hits(q, axis, plus, tri):
    """intersect_tri for three dimensions (mesh_grid_kernel.cu:461-567): does the ray from q along +-axis cross the
    triangle?  q float32[3], tri float32[3,3].  (1) a corner strictly ahead; (2) crossing parity of the projected
    edges with the 2-D ray towards -u; (3) sign pattern of the cofactors of the corners' axis coordinate."""
    f = np.float32
    q = np.asarray(q, f)
    tri = np.asarray(tri, f)
    a, u, w = axis, (axis + 1) % 3, (axis + 2) % 3
    if not any((tri[i, a] > q[a]) if plus else (tri[i, a] < q[a]) for i in range(3)):
        return False
    crossings = 0
    for d in range(3):
        A, B = tri[(d + 1) % 3], tri[(d + 2) % 3]
        if not (A[u] < q[u] or B[u] < q[u]):
            continue
        au, aw, bu, bw = f(A[u] - q[u]), f(A[w] - q[w]), f(B[u] - q[u]), f(B[w] - q[w])
        det = f(f(au * bw) - f(aw * bu))
        if det == 0:
            continue
        if det > 0:
            crossings += int((not bw >= 0) and (not -aw >= 0))
        else:
            crossings += int(bw >= 0 and -aw >= 0)
    if crossings % 2 == 0:
        return False
    r = tri - q
    c = [f(f(r[(i + 1) % 3, u] * r[(i + 2) % 3, w]) - f(r[(i + 1) % 3, w] * r[(i + 2) % 3, u])) for i in range(3)]
    det = f(f(f(c[0] * r[0, a]) + f(c[1] * r[1, a])) + f(c[2] * r[2, a]))
    if det == 0:
        return False
    want_negative = (det > 0) != bool(plus)
    return all(want_negative == bool(ci < 0) for ci in c)


def inside_mesh(verts, faces, queries, step, origin, num, tri_num, tri_idx):
    """search_inside_mesh_kernel (mesh_grid_kernel.cu:569-641): +1 when the axis ray towards the nearest grid wall crosses
    an odd number of distinct triangles (distinct among the last 15 hits, the kernel's `visited[16]`), -1 otherwise and
    for queries off the grid.  tri_num / tri_idx as insert_grid_surface returns them.  Pure-Python loops: small cases."""
    f = np.float32
    v = np.asarray(verts, f)
    fa = np.asarray(faces, np.int64).reshape(-1, 3)
    num = [int(n) for n in num]
    org = np.asarray(origin, f)
    start = np.concatenate([[0], np.asarray(tri_num, np.int64)])
    out = np.empty(len(queries), f)
    for qi, q in enumerate(np.asarray(queries, f)):
        xf = (q - org) / f(step)
        if np.any(xf < 0) or np.any(xf >= np.asarray(num, f)):
            out[qi] = -1
            continue
        x = [int(t) for t in xf]
        to_end = [x[0], num[0] - 1 - x[0], x[1], num[1] - 1 - x[1], x[2], num[2] - 1 - x[2]]
        direction = int(np.argmin(to_end))                    # first minimum, as the strict `<` scan of the kernel
        a, plus = direction // 2, direction % 2 == 1
        seen, hits = [], 0
        for _ in range(to_end[direction] + 1):
            cell = (x[0] * num[1] + x[1]) * num[2] + x[2]
            for t in np.asarray(tri_idx[start[cell]:start[cell + 1]], np.int64) - 1:
                if not axis_ray_hits(q, a, plus, v[fa[t]]) or t in seen:
                    continue
                seen = (seen + [t])[-15:]
                hits += 1
            x[a] += 1 if plus else -1
        out[qi] = 1 if hits % 2 else -1
    return out


def intersect_tri2(src, direction, va, vb, vc, both_direction=False, precision=1e-9):
    """intersect_tri2 (mesh_grid_kernel.cu:742-1026) for one ray and one triangle, float32 scalars in the kernel's operation order
    (every product and sum rounded: numpy scalars do not fuse).  -> bool.  `coeff` is not returned, but the two places where the
    kernel's coeff != NULL path rewrites a numerator before the final test (:900-901, :1004-1005) are followed, as
    search_ray_grid_kernel always passes coeff (:1077-1081)."""
    F = np.float32
    p = F(precision)
    src, direction, va, vb, vc = (np.asarray(x, F) for x in (src, direction, va, vb, vc))
    A = [va[0] - src[0], vb[0] - src[0], vc[0] - src[0], -direction[0],
         va[1] - src[1], vb[1] - src[1], vc[1] - src[1], -direction[1],
         va[2] - src[2], vb[2] - src[2], vc[2] - src[2], -direction[2]]
    I = [A[5] * A[10] - A[6] * A[9], A[2] * A[9] - A[1] * A[10], A[1] * A[6] - A[2] * A[5],
         A[6] * A[8] - A[4] * A[10], A[0] * A[10] - A[2] * A[8], A[2] * A[4] - A[0] * A[6],
         A[4] * A[9] - A[5] * A[8], A[1] * A[8] - A[0] * A[9], A[0] * A[5] - A[1] * A[4]]
    N = [-A[3] * I[0] - A[7] * I[1] - A[11] * I[2], -A[3] * I[3] - A[7] * I[4] - A[11] * I[5],
         -A[3] * I[6] - A[7] * I[7] - A[11] * I[8], A[0] * I[0] + A[4] * I[1] + A[8] * I[2]]
    det = N[0] + N[1] + N[2]
    if det > p or det < -p:                                               # :768-780
        if det < 0:
            N = [-x for x in N]
        return bool(N[0] >= -p and N[1] >= -p and N[2] >= -p and (both_direction or N[3] >= -p))
    norm = A[3] * A[3] + A[7] * A[7] + A[11] * A[11]                      # :782
    S = [I[0] + I[3] + I[6], I[1] + I[4] + I[7], I[2] + I[5] + I[8]]
    area = S[0] * S[0] + S[1] * S[1] + S[2] * S[2]

    def edges():
        e = [vc[0] - vb[0], vc[1] - vb[1], vc[2] - vb[2], va[0] - vc[0], va[1] - vc[1], va[2] - vc[2],
             vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]]
        l = [e[0] * e[0] + e[1] * e[1] + e[2] * e[2], e[3] * e[3] + e[4] * e[4] + e[5] * e[5], e[6] * e[6] + e[7] * e[7] + e[8] * e[8]]
        i = 1 if l[0] < l[1] else 0
        i = 2 if l[i] < l[2] else i
        return e, l, i, (i + 1) % 3, (i + 2) % 3

    def cross_d(c):                                                        # (column c of A) x (-direction), :862-865 and friends
        return [A[c + 4] * A[11] - A[c + 8] * A[7], A[c + 8] * A[3] - A[c] * A[11], A[c] * A[7] - A[c + 4] * A[3]]

    def dot3(u, w):
        return u[0] * w[0] + u[1] * w[1] + u[2] * w[2]

    if norm <= p:                                                          # :789-848 direction degenerate to a point
        if area > p:
            B = [dot3(I[0:3], S), dot3(I[3:6], S), dot3(I[6:9], S)]
            return bool(B[0] >= -p and B[1] >= -p and B[2] >= -p and -p <= N[3] <= p)
        e, l, i, j, k = edges()
        if l[i] > p:
            ni = dot3(I[3 * i:3 * i + 3], I[3 * i:3 * i + 3])
            nj = A[k] * e[3 * i] + A[k + 4] * e[3 * i + 1] + A[k + 8] * e[3 * i + 2]
            nk = -A[j] * e[3 * i] - A[j + 4] * e[3 * i + 1] - A[j + 8] * e[3 * i + 2]
            return bool(ni <= p and nj >= -p and nk >= -p and -p <= N[3] <= p)
        ni = A[i] * A[i] + A[i + 4] * A[i + 4] + A[i + 8] * A[i + 8]
        return bool(ni <= p and -p <= N[3] <= p)
    if area <= p:                                                          # :850-911 degenerate triangle
        e, l, i, j, k = edges()
        if l[i] <= p:                                                      # a point
            cr = cross_d(i)
            ni = dot3(cr, cr)
            n3 = -A[i] * A[3] - A[i + 4] * A[7] - A[i + 8] * A[11]
            return bool(ni <= p and (both_direction or n3 >= -p))
        norm_ = dot3(I[3 * i:3 * i + 3], I[3 * i:3 * i + 3])               # a segment
        if norm_ > p:
            cj, ck = cross_d(j), cross_d(k)
            nj = dot3(I[3 * i:3 * i + 3], ck)
            nk = -I[3 * i] * cj[0] - I[3 * i + 1] * cj[1] - I[3 * i + 2] * cj[2]
            n3 = nj + nk
        else:
            nj = A[k] * e[3 * i] + A[k + 4] * e[3 * i + 1] + A[k + 8] * e[3 * i + 2]
            nk = -A[j] * e[3 * i] - A[j + 4] * e[3 * i + 1] - A[j + 8] * e[3 * i + 2]
            n3 = l[i]
        if -p <= n3 <= p:                                                  # (coeff != NULL, :900-901)
            n3 = p
        return bool(-p <= N[i] <= p and nj >= -p and nk >= -p and (both_direction or n3 > p))
    # :912-1023 direction parallel to the triangle
    B = [dot3(I[0:3], S), dot3(I[3:6], S), dot3(I[6:9], S)]
    i = 0 if B[0] < B[1] else 1
    i = i if B[i] < B[2] else 2
    j, k = (i + 1) % 3, (i + 2) % 3
    if B[k] < -p:
        k = j; j = i; i = 3 - j - k
    in_plane = -p <= N[3] <= p
    if B[j] < -p:
        ci, cj, ck = cross_d(i), cross_d(j), cross_d(k)
        Ii, Ij = I[3 * i:3 * i + 3], I[3 * j:3 * j + 3]
        d0 = dot3(Ii, ck)
        d1 = -Ii[0] * cj[0] - Ii[1] * cj[1] - Ii[2] * cj[2]
        d2 = dot3(Ij, ci)
        d3 = -Ij[0] * ck[0] - Ij[1] * ck[1] - Ij[2] * ck[2]
        v0 = d0 >= -p and d1 >= -p and (both_direction or dot3(Ii, Ii) > p)
        v1 = d2 >= -p and d3 >= -p and (both_direction or dot3(Ij, Ij) > p)
        return bool((v0 or v1) and in_plane)
    if B[i] < -p:
        cj, ck = cross_d(j), cross_d(k)
        Ii = I[3 * i:3 * i + 3]
        nj = dot3(Ii, ck)
        nk = -Ii[0] * cj[0] - Ii[1] * cj[1] - Ii[2] * cj[2]
        ni = nj + nk
        if -p <= ni <= p:                                                  # (coeff != NULL, :1004-1005)
            ni = p
        return bool(nj >= -p and nk >= -p and in_plane and (both_direction or ni > p))
    return bool(B[i] >= -p and in_plane)


def intersects_any(verts, faces, origins, directions):
    """search_intersect's answer (mesh_grid_kernel.cu:1029-1231) by brute force: a ray hits iff SOME triangle passes
    intersect_tri2 (above).  The regular branch is evaluated for all triangles at once (float32, the kernel's order); the pairs with
    |det| <= 1e-9 go through the scalar restatement of the degenerate branches.  The OR over triangles does not depend on the
    order of the reference's cell walk."""
    f32 = np.float32
    v = np.asarray(verts, f32)
    fc = np.asarray(faces, np.int64).reshape(-1, 3)
    tri = v[fc]                                                           # [T,3,3]
    out = np.zeros(len(origins), bool)
    prec = f32(1e-9)
    for r, (o, d) in enumerate(zip(np.asarray(origins, f32), np.asarray(directions, f32))):
        if f32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < prec:           # kernel.cu:1062-1065
            continue
        a = tri - o                                                       # rows: va-o, vb-o, vc-o
        A0, A1, A2 = a[:, 0, 0], a[:, 1, 0], a[:, 2, 0]
        A4, A5, A6 = a[:, 0, 1], a[:, 1, 1], a[:, 2, 1]
        A8, A9, A10 = a[:, 0, 2], a[:, 1, 2], a[:, 2, 2]
        A3, A7, A11 = -d[0], -d[1], -d[2]
        i0, i1, i2 = A5 * A10 - A6 * A9, A2 * A9 - A1 * A10, A1 * A6 - A2 * A5
        i3, i4, i5 = A6 * A8 - A4 * A10, A0 * A10 - A2 * A8, A2 * A4 - A0 * A6
        i6, i7, i8 = A4 * A9 - A5 * A8, A1 * A8 - A0 * A9, A0 * A5 - A1 * A4
        n0 = -A3 * i0 - A7 * i1 - A11 * i2
        n1 = -A3 * i3 - A7 * i4 - A11 * i5
        n2 = -A3 * i6 - A7 * i7 - A11 * i8
        n3 = A0 * i0 + A4 * i1 + A8 * i2
        det = n0 + n1 + n2
        ok = (det > prec) | (det < -prec)
        sg = np.where(det < 0, f32(-1), f32(1))
        hit = bool(np.any(ok & (n0 * sg >= -prec) & (n1 * sg >= -prec) & (n2 * sg >= -prec) & (n3 * sg >= -prec)))
        if not hit:
            for t in np.nonzero(~ok)[0]:
                if intersect_tri2(o, d, tri[t, 0], tri[t, 1], tri[t, 2]):
                    hit = True
                    break
        out[r] = hit
    return out


def closest_rule(p0, p1, p2):
    """Rule of search_nearest_proj for triangles given RELATIVE to the query (p_i = vertex_i - q).

    p0,p1,p2: [N,3] float64.  Returns (coeff[N,3], dist2[N])."""
    P = np.stack([p0, p1, p2], 1)                                   # [N,3,3]
    N = len(P)
    e1, e2 = p1 - p0, p2 - p0
    a11, a12, a22 = (e1 * e1).sum(1), (e1 * e2).sum(1), (e2 * e2).sum(1)
    b1, b2 = -(p0 * e1).sum(1), -(p0 * e2).sum(1)
    det = a11 * a22 - a12 * a12
    ok = det > 1e-30 * np.maximum(a11 * a22, 1e-300)
    safe = np.where(ok, det, 1.0)
    u = (b1 * a22 - b2 * a12) / safe
    v = (a11 * b2 - a12 * b1) / safe
    c = np.stack([1 - u - v, u, v], 1)
    inside = ok & (c.min(1) >= 0)
    # fallback edge: opposite the most negative coefficient; degenerate triangle: the longest edge
    i_neg = np.argmin(c, 1)
    elen = np.stack([((p1 - p2) ** 2).sum(1), ((p2 - p0) ** 2).sum(1), ((p0 - p1) ** 2).sum(1)], 1)
    i_deg = np.argmax(elen, 1)
    i = np.where(ok, i_neg, i_deg)
    j = (i + 1) % 3
    k = 3 - i - j
    ar = np.arange(N)
    pj, pk = P[ar, j], P[ar, k]
    d = pk - pj
    dd = (d * d).sum(1)
    t = np.where(dd > 0, -(pj * d).sum(1) / np.where(dd > 0, dd, 1.0), 0.5)
    cj, ck = 1 - t, t
    # order of the reference's tests: coefficient of j negative -> vertex k; then k negative -> vertex j
    at_k = cj < 0
    at_j = (~at_k) & (ck < 0)
    cj = np.where(at_k, 0.0, np.where(at_j, 1.0, cj))
    ck = np.where(at_k, 1.0, np.where(at_j, 0.0, ck))
    ce = np.zeros((N, 3))
    ce[ar, j], ce[ar, k] = cj, ck
    coeff = np.where(inside[:, None], c, ce)
    x = np.einsum("ni,nik->nk", coeff, P)
    return coeff, (x * x).sum(1)


def closest_exact(p0, p1, p2):
    """Independent exact closest point on a triangle to the origin (Ericson, Real-Time Collision
    Detection 5.1.5), vectorised; returns dist2[N]."""
    ab, ac, ap = p1 - p0, p2 - p0, -p0
    d1, d2 = (ab * ap).sum(1), (ac * ap).sum(1)
    bp = -p1
    d3, d4 = (ab * bp).sum(1), (ac * bp).sum(1)
    cp = -p2
    d5, d6 = (ab * cp).sum(1), (ac * cp).sum(1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    N = len(p0)
    out = np.zeros((N, 3))
    done = np.zeros(N, bool)

    def put(mask, pts):
        m = mask & ~done
        out[m] = pts[m]
        done[m] = True

    put((d1 <= 0) & (d2 <= 0), p0)
    put((d3 >= 0) & (d4 <= d3), p1)
    with np.errstate(divide="ignore", invalid="ignore"):
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), p0 + (d1 / (d1 - d3))[:, None] * ab)
        put((d6 >= 0) & (d5 <= d6), p2)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), p0 + (d2 / (d2 - d6))[:, None] * ac)
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), p1 + w[:, None] * (p2 - p1))
        den = 1.0 / (va + vb + vc)
        put(np.ones(N, bool), p0 + ab * (vb * den)[:, None] + ac * (vc * den)[:, None])
    return (out * out).sum(1)


def nearest_bruteforce(verts, faces, queries, chunk=256):
    """(face id int32[Q], nearest point f32[Q,3], barycentrics f32[Q,3]) by the reference rule over
    ALL faces (float64 arithmetic on the float32 inputs)."""
    V = np.asarray(verts, np.float32).astype(np.float64)
    F = np.asarray(faces, np.int64)
    Q = np.asarray(queries, np.float32).astype(np.float64)
    tri = V[F]                                                     # [F,3,3]
    ids = np.zeros(len(Q), np.int32)
    pts = np.zeros((len(Q), 3), np.float32)
    bary = np.zeros((len(Q), 3), np.float32)
    for s in range(0, len(Q), chunk):
        q = Q[s:s + chunk]
        rel = tri[None] - q[:, None, None, :]                      # [q,F,3,3]
        r = rel.reshape(-1, 3, 3)
        coeff, d2 = closest_rule(r[:, 0], r[:, 1], r[:, 2])
        d2 = d2.reshape(len(q), len(F))
        best = np.argmin(d2, 1)
        cb = coeff.reshape(len(q), len(F), 3)[np.arange(len(q)), best]
        ids[s:s + chunk] = best
        bary[s:s + chunk] = cb
        pts[s:s + chunk] = np.einsum("qi,qik->qk", cb, tri[best])
    return ids, pts, bary


class ReferenceSearcher:
    """MeshGridSearcher (utils/mesh_grid_searcher.py:51-84) in the reference's OWN float32 arithmetic: set_mesh's grid + the search of
    oracle/nearest_ref.c (mesh_grid_kernel.cu:12-109, 239-353, matrix.h).  What the loops of oracle/smplify_oracle.py and the goldens'
    stand-in searcher (oracle/gen_golden.py) use since round 4; nearest_bruteforce above is the float64 form of the same rule."""

    def __init__(self, verts, faces, fused=False):
        self.verts = np.ascontiguousarray(verts, np.float32)
        self.faces = np.ascontiguousarray(faces, np.int32)
        step, num, origin = grid_params(self.verts)
        tri_num, tri_idx = insert_grid_surface(self.verts, self.faces, step, origin, num)
        self.grid = (step, num, origin, tri_num, tri_idx)
        self.fused = fused

    def nearest(self, queries):
        """-> (face ids int32[Q], nearest points f32[Q,3], coefficients f32[Q,3])"""
        from . import nearest_ref as NR
        ids, pts, coeff, _ = NR.search_nearest(self.verts, self.faces, np.asarray(queries, np.float32).reshape(-1, 3), self.grid, fused=self.fused)
        return ids, pts, coeff


# ----------------------------------------------------------------------------------------------
# the losses built on the search (torch, differentiable): restated from loss.py / io_utils.py
# ----------------------------------------------------------------------------------------------

def compute_normal_torch(vertices, faces):
    """io_utils.py:406-428.  vertices[NV,3] float tensor, faces[F,3] long tensor."""
    va, vb, vc = vertices[faces[:, 0]], vertices[faces[:, 1]], vertices[faces[:, 2]]
    n = torch.cross(vb - va, vc - va, dim=1)
    n = n / (torch.norm(n, dim=-1, keepdim=True) + 1e-8)
    norm = torch.zeros_like(vertices)
    for j in range(3):
        norm = norm.index_add(0, faces[:, j], n)
    return norm / (torch.norm(norm, dim=-1, keepdim=True) + 1e-8)


def point_cloud_loss(points, closest):
    """loss.py:233-242: ONE Frobenius norm over all vertices (the mean of a scalar is itself)."""
    return torch.norm(points.reshape(-1, 3) - closest.detach(), p=2)


def normal_loss(closest_face_norm, point_norm):
    """loss.py:260-271 with the (un-normalised, smplify.py:149) scan face normals already gathered."""
    return torch.mean(1 - torch.sum(closest_face_norm * point_norm, dim=-1))


def normal_laplacian_smoothness(norms, faces):
    """loss.py:273-288."""
    na, nb, nc = norms[faces[:, 0]], norms[faces[:, 1]], norms[faces[:, 2]]
    mse = lambda x, y: torch.sum((x - y) ** 2, dim=-1)   # noqa: E731
    return torch.mean(mse(na, nb) + mse(nc, na) + mse(nb, nc))


def nearest_backward(verts, faces, points, face_ids, bary, dnearest):
    """dL/d(points) through the closest point, by torch.autograd (float64) of the closed form of the region each answer lies
    in - read off the zeros of `bary` like the device kernel does: face -> projection onto the plane, edge -> onto the edge's
    line, corner -> constant.  (What SurfaceNearest.backward, utils/mesh_grid_searcher.py:17-49, set out to compute; the
    reference's kernel for it is unfinished, mesh_grid_kernel.cu:371-372.)"""
    V = torch.as_tensor(np.asarray(verts, np.float64))
    tri = V[torch.as_tensor(np.asarray(faces, np.int64))[torch.as_tensor(np.asarray(face_ids, np.int64))]]      # [Q,3,3]
    p = torch.tensor(np.asarray(points, np.float64), requires_grad=True)
    b = np.asarray(bary)
    zeros = (b == 0).sum(1)
    a, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    n = torch.cross(e1, e2, dim=1)
    c_face = p - n * ((p - a) * n).sum(1, keepdim=True) / (n * n).sum(1, keepdim=True)
    z = np.where(b[:, 0] == 0, 0, np.where(b[:, 1] == 0, 1, 2))
    ar = torch.arange(len(b))
    vj, vk = tri[ar, torch.as_tensor((z + 1) % 3)], tri[ar, torch.as_tensor((z + 2) % 3)]
    d = vk - vj
    c_edge = vj + d * ((p - vj) * d).sum(1, keepdim=True) / (d * d).sum(1, keepdim=True)
    c_corner = (torch.as_tensor(b, dtype=torch.float64)[:, :, None] * tri).sum(1)
    zt = torch.as_tensor(zeros)[:, None]
    c = torch.where(zt == 0, c_face, torch.where(zt == 1, c_edge, c_corner))
    (c * torch.as_tensor(np.asarray(dnearest, np.float64))).sum().backward()
    return p.grad.numpy()
