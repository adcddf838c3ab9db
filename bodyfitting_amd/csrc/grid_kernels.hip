// Closest-point grid of a scan, built on the device (reference: insert_grid_surface_cuda,
// thirdparty/mesh_grid/mesh_grid_kernel.cu:110-157,178-236 - count pass, cumulative sum, fill pass).
//
// The reference fills a cell's list through atomicCAS in whatever order the threads arrive; here the fill uses an atomic
// cursor as well, and a last pass puts every list into ascending face order while it writes the packed 48-byte records
// the search kernel reads, so the finished grid is the same bytes run to run.
//
//   bf_grid_count_kernel   thread per triangle: +1 on every cell its bounding box covers   (kernel.cu:119-148, surf_idx == NULL)
//   bf_grid_scan_kernel    one workgroup: exclusive prefix sum over the cells              (tri_num.cumsum, kernel.cu:209)
//   bf_grid_fill_kernel    thread per triangle: face id into a free slot of every covered cell   (kernel.cu:149-155)
//   bf_grid_pack_kernel    thread per list entry: rank inside its cell -> sorted list + (corners | face id) record
//   bf_face_normal_kernel  un-normalised scan face normals, float64 cross product rounded once (smplify.py:148-149)
//   bf_inside_mesh_kernel  MeshGridSearcher.inside_mesh: parity of the triangles an axis ray crosses (kernel.cu:461-641)
#include <hip/hip_runtime.h>
#include "bf_internal.h"

struct GridBox { int lo[3], hi[3]; };

// Cell range of triangle f: the arithmetic of kernel.cu:127-141 in float32, division included.
__device__ __forceinline__ GridBox grid_box(const ScanDev &S, int f) {
    GridBox b;
    const int i0 = S.faces[f * 3], i1 = S.faces[f * 3 + 1], i2 = S.faces[f * 3 + 2];
    const float org[3] = {S.ox, S.oy, S.oz};
    const int num[3] = {S.nx, S.ny, S.nz};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float a = S.verts[i0 * 3 + d], b1 = S.verts[i1 * 3 + d], c = S.verts[i2 * 3 + d];
        const float mn = fminf(a, fminf(b1, c)), mx = fmaxf(a, fmaxf(b1, c));
        float x = __fdiv_rn(__fsub_rn(mn, org[d]), S.step);
        b.lo[d] = x < 0.f ? 0 : (x >= (float)num[d] ? num[d] - 1 : (int)floorf(x));
        x = __fdiv_rn(__fsub_rn(mx, org[d]), S.step);
        b.hi[d] = (x < 0.f ? 0 : (x >= (float)num[d] ? num[d] - 1 : (int)floorf(x))) + 1;
    }
    return b;
}

extern "C" __global__ void __launch_bounds__(256) bf_grid_count_kernel(ScanDev S, int *count /*[ncell + 1], zeroed; slot c + 1 = cell c*/) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= S.nf) return;
    const GridBox b = grid_box(S, f);
    for (int x = b.lo[0]; x < b.hi[0]; ++x)
        for (int y = b.lo[1]; y < b.hi[1]; ++y)
            for (int z = b.lo[2]; z < b.hi[2]; ++z) atomicAdd(count + ((size_t)x * S.ny + y) * S.nz + z + 1, 1);
}

// In-place inclusive prefix sum over n ints by ONE workgroup of 1024 threads (n is the number of cells + 1, a few 10^5 at
// most): chunks of 4096 (four consecutive elements per thread), wave-level DPP-free shuffle scan, carry kept in a register.
// cursor[] receives a copy of the exclusive starts for the fill pass.
extern "C" __global__ void __launch_bounds__(1024) bf_grid_scan_kernel(int *data, int *cursor, int n) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 4096) {
        const int i = base + t * 4;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i + k < n ? data[i + k] : 0;
        v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
        int s = v[3];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(s, o, 64);
            if (lane >= o) s += u;
        }
        if (lane == 63) wsum[w] = s;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int carry = carry_s;
        const int excl = carry + woff + s - v[3];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < n) {
                data[i + k] = excl + v[k];
            }
        __syncthreads();
        if (t == 1023) carry_s = carry + woff + s;
        __syncthreads();
    }
    // data[c] now = number of entries in cells < c (slot 0 was zero) = start of cell c; copy for the cursors
    for (int i = t; i < n; i += 1024) cursor[i] = data[i];
}

extern "C" __global__ void __launch_bounds__(256) bf_grid_fill_kernel(ScanDev S, int *cursor, int *tris_raw) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= S.nf) return;
    const GridBox b = grid_box(S, f);
    for (int x = b.lo[0]; x < b.hi[0]; ++x)
        for (int y = b.lo[1]; y < b.hi[1]; ++y)
            for (int z = b.lo[2]; z < b.hi[2]; ++z) tris_raw[atomicAdd(cursor + ((size_t)x * S.ny + y) * S.nz + z, 1)] = f;
}

// One thread per list entry e: its cell by bisection of cell_start, its rank = number of smaller face ids in the cell
// (a triangle enters a cell once, so the ids of a cell are distinct), then the sorted list entry and the packed record.
extern "C" __global__ void __launch_bounds__(256) bf_grid_pack_kernel(ScanDev S, const int *tris_raw, int *tris_sorted, float4 *pack,
                                                                     float4 *box, int n_entries) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_entries) return;
    int lo = 0, hi = S.nx * S.ny * S.nz;            // largest c with cell_start[c] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (S.cell_start[mid] <= e) lo = mid; else hi = mid;
    }
    const int st = S.cell_start[lo], en = S.cell_start[lo + 1];
    const int f = tris_raw[e];
    int rank = 0;
    for (int k = st; k < en; ++k) rank += tris_raw[k] < f ? 1 : 0;
    const int o = st + rank;
    tris_sorted[o] = f;
    const float *a = S.verts + (size_t)S.faces[f * 3] * 3, *b = S.verts + (size_t)S.faces[f * 3 + 1] * 3,
                *c = S.verts + (size_t)S.faces[f * 3 + 2] * 3;
    pack[(size_t)o * 3] = make_float4(a[0], a[1], a[2], b[0]);
    pack[(size_t)o * 3 + 1] = make_float4(b[1], b[2], c[0], c[1]);
    pack[(size_t)o * 3 + 2] = make_float4(c[2], __int_as_float(f), 0.f, 0.f);
    // the screen's record: the triangle's box (exact minima / maxima of its float32 coordinates) and its margin
    const float lx = fminf(a[0], fminf(b[0], c[0])), ly = fminf(a[1], fminf(b[1], c[1])), lz = fminf(a[2], fminf(b[2], c[2]));
    const float hx = fmaxf(a[0], fmaxf(b[0], c[0])), hy = fmaxf(a[1], fmaxf(b[1], c[1])), hz = fmaxf(a[2], fmaxf(b[2], c[2]));
    const float dx = hx - lx, dy = hy - ly, dz = hz - lz;
    box[(size_t)o * 2] = make_float4(lx, ly, lz, 2.1e-5f * (dx * dx + dy * dy + dz * dz));
    box[(size_t)o * 2 + 1] = make_float4(hx, hy, hz, 0.f);
}

extern "C" __global__ void __launch_bounds__(256) bf_face_normal_kernel(const float *verts, const int *faces, int nf, float *fn) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    const float *a = verts + (size_t)faces[f * 3] * 3, *b = verts + (size_t)faces[f * 3 + 1] * 3, *c = verts + (size_t)faces[f * 3 + 2] * 3;
    const double u0 = (double)b[0] - a[0], u1 = (double)b[1] - a[1], u2 = (double)b[2] - a[2];
    const double w0 = (double)c[0] - a[0], w1 = (double)c[1] - a[1], w2 = (double)c[2] - a[2];
    // no contraction: product, product, difference - each rounded to double, as the host arithmetic
    fn[f * 3] = (float)__dsub_rn(__dmul_rn(u1, w2), __dmul_rn(u2, w1));
    fn[f * 3 + 1] = (float)__dsub_rn(__dmul_rn(u2, w0), __dmul_rn(u0, w2));
    fn[f * 3 + 2] = (float)__dsub_rn(__dmul_rn(u0, w1), __dmul_rn(u1, w0));
}

// ---- MeshGridSearcher.inside_mesh (utils/mesh_grid_searcher.py:86-91 -> search_inside_mesh_kernel, kernel.cu:569-641) ----------
// Does the axis ray from q (axis a, towards + if `plus`) cross the triangle?  The reference's test (kernel.cu:461-567), for
// three dimensions: (1) some corner lies strictly ahead of q along the axis; (2) q's projection on the other two axes is
// inside the projected triangle by crossing parity - a 2-D ray towards -u crosses edge (A, B) iff the edge has a corner
// with u < 0, its cross product det = A.u B.w - A.w B.u is non-zero and (det > 0 ? A.w > 0 > B.w : A.w <= 0 <= B.w);
// (3) the cofactors c_i of the corners' axis coordinate (det = sum c_i V_i[a]) all carry the sign that puts the hit ahead.
__device__ inline bool axis_ray_hits(const float q[3], int a, bool plus, const float v[3][3]) {
    bool ahead = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) ahead = ahead || (plus ? v[i][a] > q[a] : v[i][a] < q[a]);
    if (!ahead) return false;
    const int u = (a + 1) % 3, w = (a + 2) % 3;
    int crossings = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int ia = (d + 1) % 3, ib = (d + 2) % 3;
        if (!(v[ia][u] < q[u] || v[ib][u] < q[u])) continue;
        const float au = v[ia][u] - q[u], aw = v[ia][w] - q[w], bu = v[ib][u] - q[u], bw = v[ib][w] - q[w];
        const float det = __fsub_rn(__fmul_rn(au, bw), __fmul_rn(aw, bu));
        if (det == 0.f) continue;
        crossings += det > 0.f ? (!(bw >= 0.f) && !(-aw >= 0.f)) : (bw >= 0.f && -aw >= 0.f);
    }
    if ((crossings & 1) == 0) return false;
    float r[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) r[i][k] = v[i][k] - q[k];
    // cofactor of corner i's coordinate a: (V_j x V_k)[a] over the cyclic (i, j, k)
    float c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        c[i] = __fsub_rn(__fmul_rn(r[j][u], r[k][w]), __fmul_rn(r[j][w], r[k][u]));
    }
    const float det = __fadd_rn(__fadd_rn(__fmul_rn(c[0], r[0][a]), __fmul_rn(c[1], r[1][a])), __fmul_rn(c[2], r[2][a]));
    if (det == 0.f) return false;
    const bool want_negative = (det > 0.f) != plus;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (want_negative != (c[i] < 0.f)) return false;
    return true;
}

// One thread per query.  sign = +1 when the ray towards the nearest grid wall (ties: -x, +x, -y, +y, -z, +z order) crosses an odd
// number of distinct triangles, -1 otherwise and for queries outside the grid.  A triangle listed in several cells of the walk is
// counted once as long as it is among the last 15 hits (the reference's fixed `visited[16]`).
extern "C" __global__ void __launch_bounds__(256) bf_inside_mesh_kernel(ScanDev S, const float *__restrict__ points, int n,
                                                                       float *__restrict__ sign) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const float q[3] = {points[id * 3], points[id * 3 + 1], points[id * 3 + 2]};
    const float org[3] = {S.ox, S.oy, S.oz};
    const int num[3] = {S.nx, S.ny, S.nz};
    int x[3], to_end[6];
    for (int d = 0; d < 3; ++d) {
        const float xf = __fdiv_rn(__fsub_rn(q[d], org[d]), S.step);
        if (xf < 0.f || xf >= (float)num[d]) { sign[id] = -1.f; return; }
        x[d] = (int)xf;
        to_end[2 * d] = x[d];
        to_end[2 * d + 1] = num[d] - 1 - x[d];
    }
    int dir = 0;
    for (int d = 1; d < 6; ++d) if (to_end[d] < to_end[dir]) dir = d;
    const int a = dir >> 1;
    const bool plus = dir & 1;
    int seen[15], n_seen = 0, hits = 0;
    for (int i = 0; i <= to_end[dir]; ++i) {
        const int cell = (x[0] * S.ny + x[1]) * S.nz + x[2];
        for (int e = S.cell_start[cell]; e < S.cell_start[cell + 1]; ++e) {
            const int t = S.cell_tris[e];
            float v[3][3];
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < 3; ++k) v[c][k] = S.verts[(size_t)S.faces[t * 3 + c] * 3 + k];
            if (!axis_ray_hits(q, a, plus, v)) continue;
            bool known = false;
            for (int s = 0; s < n_seen; ++s) known = known || seen[s] == t;
            if (known) continue;
            if (n_seen < 15) seen[n_seen++] = t;
            else { for (int s = 0; s + 1 < 15; ++s) seen[s] = seen[s + 1]; seen[14] = t; }
            ++hits;
        }
        x[a] += plus ? 1 : -1;
    }
    sign[id] = (hits & 1) ? 1.f : -1.f;
}

// ---- MeshGridSearcher.intersects_any (utils/mesh_grid_searcher.py:93-99 -> search_intersect, kernel.cu:1029-1231) ---------------
// Does the ray origin + t direction, t >= 0, hit any triangle?  The per-triangle test is the reference's intersect_tri2
// (kernel.cu:742-1026, both_direction = false, precision 1e-9) in float32 and in its operation order (no fused multiply-adds: the
// tests against 1e-9 are decided by the last bit):
//   regular (|det| > 1e-9)   solve [va - o | vb - o | vc - o | -d ; 1 1 1 0] by cofactors; hit iff the three barycentric numerators and
//                            the ray parameter, signed by det, are >= -1e-9                                   (:742-780)
//   ray in the triangle's plane (|det| <= 1e-9, triangle area^2 > 1e-9)   the origin must lie in the plane (|volume| <= 1e-9); then by the
//                            barycentric numerators of the origin: inside -> hit; outside one edge -> the ray must cross that edge
//                            going in; outside two edges -> it must cross one of the two                     (:912-1023)
//   degenerate triangle (area^2 <= 1e-9)   a segment (longest edge^2 > 1e-9): coplanarity + the ray crossing it; a point: the origin-to-
//                            point vector parallel to the ray and not behind it                               (:849-911)
// (A zero direction is rejected by the caller, kernel.cu:1062-1065, so the norm <= 1e-9 branches :793-848 are never reached.)
// The answer is an OR over triangles, so the order of the reference's cell walk does not matter: one thread per ray walks the cells
// its ray crosses (3-D DDA from the ray's entry into the grid) and stops at the first hit.
// (float32 throughout, no fused multiply-adds: `#pragma clang fp contract(off)` in every helper; sums run left to right)
struct Vec3 { float x, y, z; };
__device__ inline Vec3 v3_sub(const float *a, const float *b) { return Vec3{a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
__device__ inline Vec3 v3_cross(const Vec3 &u, const Vec3 &v) {
#pragma clang fp contract(off)
    return Vec3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
__device__ inline float v3_dot(const Vec3 &u, const Vec3 &v) {
#pragma clang fp contract(off)
    return u.x * v.x + u.y * v.y + u.z * v.z;
}
__device__ inline float v3_minus_dot(const Vec3 &u, const Vec3 &v) {      // -u.v, the negation inside the first product
#pragma clang fp contract(off)
    return -u.x * v.x - u.y * v.y - u.z * v.z;
}
__device__ inline Vec3 v3_pick(int i, const Vec3 &a, const Vec3 &b, const Vec3 &c) { return i == 0 ? a : (i == 1 ? b : c); }

// One ray against one triangle: the corners relative to the ray's origin, minus the direction, and the three products
// face[c] = rel[c + 1] x rel[c + 2] (the rows of the adjugate of [rel_a | rel_b | rel_c]).
struct RayTriangle {
    Vec3 rel[3], minus_dir, face[3];
    float prec;
    __device__ RayTriangle(const float o[3], const float d[3], const float *va, const float *vb, const float *vc) : prec(1e-9f) {
        rel[0] = v3_sub(va, o); rel[1] = v3_sub(vb, o); rel[2] = v3_sub(vc, o);
        minus_dir = Vec3{-d[0], -d[1], -d[2]};
        face[0] = v3_cross(rel[1], rel[2]); face[1] = v3_cross(rel[2], rel[0]); face[2] = v3_cross(rel[0], rel[1]);
    }
    __device__ Vec3 rel_of(int c) const { return v3_pick(c, rel[0], rel[1], rel[2]); }
    __device__ Vec3 face_of(int c) const { return v3_pick(c, face[0], face[1], face[2]); }
    // does the ray, seen in the plane whose normal is face_of(i), pass between corners j and k - and on which side of each?
    // (the two numerators of kernel.cu:868-871 / 960-963: n . (rel_k x -d) and -n . (rel_j x -d))
    __device__ void between(int i, int j, int k, float &nj, float &nk) const {
        const Vec3 n = face_of(i);
        nj = v3_dot(n, v3_cross(rel_of(k), minus_dir));
        nk = v3_minus_dot(n, v3_cross(rel_of(j), minus_dir));
    }
    __device__ bool within(float x) const { return x >= -prec && x <= prec; }

    // |det| > 1e-9: Cramer's rule, numerators signed by the determinant (kernel.cu:742-780)
    __device__ bool regular(float num[4], float det) const {
        const bool flip = det < 0.f;
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 4; ++c) ok = ok && (flip ? -num[c] : num[c]) >= -prec;
        return ok;
    }
    // area^2 <= 1e-9: the triangle is a segment or a point (kernel.cu:849-911)
    __device__ bool degenerate(const float num[4], const float *va, const float *vb, const float *vc) const {
#pragma clang fp contract(off)
        const Vec3 edge[3] = {v3_sub(vc, vb), v3_sub(va, vc), v3_sub(vb, va)};          // edge c is opposite corner c
        const float len[3] = {v3_dot(edge[0], edge[0]), v3_dot(edge[1], edge[1]), v3_dot(edge[2], edge[2])};
        int i = len[0] < len[1] ? 1 : 0;                                                // the longest edge
        i = len[i] < len[2] ? 2 : i;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        if (len[i] <= prec) {                                                           // a point: origin-to-point parallel to the ray, not behind it
            const Vec3 c = v3_cross(rel_of(i), minus_dir);
            return v3_dot(c, c) <= prec && v3_minus_dot(rel_of(i), minus_dir) >= -prec;
        }
        const Vec3 n = face_of(i), e = v3_pick(i, edge[0], edge[1], edge[2]);
        float nj, nk, ahead;
        if (v3_dot(n, n) > prec) { between(i, j, k, nj, nk); ahead = nj + nk; }
        else { nj = v3_dot(rel_of(k), e); nk = v3_minus_dot(rel_of(j), e); ahead = len[i]; }     // the origin is on the segment's line
        return within(num[i]) && nj >= -prec && nk >= -prec && ahead > prec;
    }
    // |det| <= 1e-9, area^2 > 1e-9: the ray is parallel to the triangle's plane (kernel.cu:912-1023)
    __device__ bool coplanar(const float num[4], const Vec3 &normal) const {
        const float inside[3] = {v3_dot(face[0], normal), v3_dot(face[1], normal), v3_dot(face[2], normal)};   // the origin's barycentric numerators
        int i = inside[0] < inside[1] ? 0 : 1;                                          // the smallest
        i = inside[i] < inside[2] ? i : 2;
        int j = (i + 1) % 3, k = (i + 2) % 3;
        if (inside[k] < -prec) { k = j; j = i; i = 3 - j - k; }
        const bool in_plane = within(num[3]);
        if (inside[j] < -prec) {                                                        // outside two edges: in through either
            float a0, a1, b0, b1;
            between(i, j, k, a0, a1);
            between(j, k, i, b0, b1);
            const Vec3 ni = face_of(i), nj = face_of(j);
            const bool through_i = a0 >= -prec && a1 >= -prec && v3_dot(ni, ni) > prec;
            const bool through_j = b0 >= -prec && b1 >= -prec && v3_dot(nj, nj) > prec;
            return (through_i || through_j) && in_plane;
        }
        if (inside[i] < -prec) {                                                        // outside one edge: in through that one
            float nj, nk;
            between(i, j, k, nj, nk);
            return nj >= -prec && nk >= -prec && in_plane && nj + nk > prec;
        }
        return inside[i] >= -prec && in_plane;                                          // the origin is inside the triangle
    }
};

__device__ inline bool ray_hits_triangle(const float o[3], const float d[3], const float *va, const float *vb, const float *vc) {
#pragma clang fp contract(off)
    const RayTriangle T(o, d, va, vb, vc);
    float num[4] = {v3_minus_dot(T.minus_dir, T.face[0]), v3_minus_dot(T.minus_dir, T.face[1]), v3_minus_dot(T.minus_dir, T.face[2]),
                    v3_dot(T.rel[0], T.face[0])};
    const float det = num[0] + num[1] + num[2];
    if (det > T.prec || det < -T.prec) return T.regular(num, det);
    const Vec3 normal{T.face[0].x + T.face[1].x + T.face[2].x, T.face[0].y + T.face[1].y + T.face[2].y, T.face[0].z + T.face[1].z + T.face[2].z};
    return v3_dot(normal, normal) <= T.prec ? T.degenerate(num, va, vb, vc) : T.coplanar(num, normal);
}

extern "C" __global__ void __launch_bounds__(256) bf_intersect_kernel(ScanDev S, const float *__restrict__ origins,
                                                                     const float *__restrict__ directions, int n,
                                                                     unsigned char *__restrict__ hit) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const float o[3] = {origins[id * 3], origins[id * 3 + 1], origins[id * 3 + 2]};
    const float d[3] = {directions[id * 3], directions[id * 3 + 1], directions[id * 3 + 2]};
    hit[id] = 0;
    if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < 1e-9f) return;             // (kernel.cu:1062-1065: a zero direction hits nothing)
    const float org[3] = {S.ox, S.oy, S.oz};
    const int num[3] = {S.nx, S.ny, S.nz};
    // entry parameter into the grid's box (slab test); t0 = 0 when the origin is inside
    float t0 = 0.f, t1 = 3.0e38f;
    for (int k = 0; k < 3; ++k) {
        const float lo = org[k], hi = org[k] + S.step * num[k];
        if (d[k] == 0.f) { if (o[k] < lo || o[k] > hi) return; }
        else {
            float a = (lo - o[k]) / d[k], b = (hi - o[k]) / d[k];
            if (a > b) { const float t = a; a = b; b = t; }
            t0 = fmaxf(t0, a); t1 = fminf(t1, b);
        }
    }
    if (t0 > t1) return;
    int c[3], stp[3];
    float tmax[3], tdel[3];
    for (int k = 0; k < 3; ++k) {
        const float p = o[k] + d[k] * t0;
        c[k] = min(max((int)floorf((p - org[k]) / S.step), 0), num[k] - 1);
        stp[k] = d[k] > 0.f ? 1 : (d[k] < 0.f ? -1 : 0);
        const float edge = org[k] + S.step * (c[k] + (stp[k] > 0 ? 1 : 0));
        tmax[k] = stp[k] ? (edge - o[k]) / d[k] : 3.0e38f;
        tdel[k] = stp[k] ? S.step / fabsf(d[k]) : 3.0e38f;
    }
    for (int guard = 0; guard < S.nx + S.ny + S.nz + 3; ++guard) {
        const int cell = (c[0] * S.ny + c[1]) * S.nz + c[2];
        for (int e = S.cell_start[cell]; e < S.cell_start[cell + 1]; ++e) {
            const int t = S.cell_tris[e];
            if (ray_hits_triangle(o, d, S.verts + (size_t)S.faces[t * 3] * 3, S.verts + (size_t)S.faces[t * 3 + 1] * 3,
                                  S.verts + (size_t)S.faces[t * 3 + 2] * 3)) { hit[id] = 1; return; }
        }
        const int k = tmax[0] <= tmax[1] ? (tmax[0] <= tmax[2] ? 0 : 2) : (tmax[1] <= tmax[2] ? 1 : 2);
        c[k] += stp[k];
        if (stp[k] == 0 || c[k] < 0 || c[k] >= num[k]) return;
        tmax[k] += tdel[k];
    }
}


// d(nearest point)/d(query), applied to an incoming gradient: thread per query.  The region the closest point lies in is read
// off the barycentric coefficients the search returned (the per-triangle rule sets clamped coefficients to exact zeros): none
// zero = the triangle's face, the closest point is the projection onto its plane, J = I - n n^T; one zero = the edge between the
// other two corners, J = d d^T / |d|^2; two zeros = a corner, J = 0.  dpoints = J^T dnearest (J is symmetric).
// (The reference's own backward, mesh_grid_kernel.cu:354-382 + the commented-out SurfaceNearest.backward of
// utils/mesh_grid_searcher.py:17-49, was never finished - its kernel leaves the KKT inverse as a TODO and multiplies by the
// un-inverted matrix - so this is the gradient it set out to compute, not a restatement of what it computes.)
extern "C" __global__ void __launch_bounds__(256) bf_nearest_backward_kernel(ScanDev S, int n, const int *__restrict__ face_ids,
                                                                            const float *__restrict__ bary, const float *__restrict__ dnearest,
                                                                            float *__restrict__ dpoints) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int f = face_ids[i];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (f >= 0 && f < S.nf) {
        const float *v[3] = {S.verts + (size_t)S.faces[f * 3] * 3, S.verts + (size_t)S.faces[f * 3 + 1] * 3, S.verts + (size_t)S.faces[f * 3 + 2] * 3};
        const float b[3] = {bary[i * 3], bary[i * 3 + 1], bary[i * 3 + 2]};
        const float d0 = dnearest[i * 3], d1 = dnearest[i * 3 + 1], d2 = dnearest[i * 3 + 2];
        const int zeros = (b[0] == 0.f) + (b[1] == 0.f) + (b[2] == 0.f);
        if (zeros == 0) {
            const float ux = v[1][0] - v[0][0], uy = v[1][1] - v[0][1], uz = v[1][2] - v[0][2];
            const float wx = v[2][0] - v[0][0], wy = v[2][1] - v[0][1], wz = v[2][2] - v[0][2];
            const float nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx, nn = nx * nx + ny * ny + nz * nz;
            const float k = nn > 0.f ? (nx * d0 + ny * d1 + nz * d2) / nn : 0.f;
            g0 = d0 - nx * k; g1 = d1 - ny * k; g2 = d2 - nz * k;
        } else if (zeros == 1) {
            const int z = b[0] == 0.f ? 0 : (b[1] == 0.f ? 1 : 2), j = (z + 1) % 3, k2 = (z + 2) % 3;
            const float ex = v[k2][0] - v[j][0], ey = v[k2][1] - v[j][1], ez = v[k2][2] - v[j][2], ee = ex * ex + ey * ey + ez * ez;
            const float k = ee > 0.f ? (ex * d0 + ey * d1 + ez * d2) / ee : 0.f;
            g0 = ex * k; g1 = ey * k; g2 = ez * k;
        }
    }
    dpoints[i * 3] = g0; dpoints[i * 3 + 1] = g1; dpoints[i * 3 + 2] = g2;
}
