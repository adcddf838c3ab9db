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
                                                                     int n_entries) {
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
__device__ inline bool ray_hits_triangle(const float o[3], const float d[3], const float *va, const float *vb, const float *vc) {
#pragma clang fp contract(off)
    const float prec = 1e-9f;
    // A = [va - o | vb - o | vc - o | -d] by rows x, y, z: A[r * 4 + c]
    const float A[12] = {va[0] - o[0], vb[0] - o[0], vc[0] - o[0], -d[0],
                         va[1] - o[1], vb[1] - o[1], vc[1] - o[1], -d[1],
                         va[2] - o[2], vb[2] - o[2], vc[2] - o[2], -d[2]};
    const float I[9] = {A[5] * A[10] - A[6] * A[9], A[2] * A[9] - A[1] * A[10], A[1] * A[6] - A[2] * A[5],
                        A[6] * A[8] - A[4] * A[10], A[0] * A[10] - A[2] * A[8], A[2] * A[4] - A[0] * A[6],
                        A[4] * A[9] - A[5] * A[8], A[1] * A[8] - A[0] * A[9], A[0] * A[5] - A[1] * A[4]};
    float N[4] = {-A[3] * I[0] - A[7] * I[1] - A[11] * I[2], -A[3] * I[3] - A[7] * I[4] - A[11] * I[5],
                  -A[3] * I[6] - A[7] * I[7] - A[11] * I[8], A[0] * I[0] + A[4] * I[1] + A[8] * I[2]};
    float det = N[0] + N[1] + N[2];
    if (det > prec || det < -prec) {
        if (det < 0.f) { N[0] = -N[0]; N[1] = -N[1]; N[2] = -N[2]; N[3] = -N[3]; }
        return N[0] >= -prec && N[1] >= -prec && N[2] >= -prec && N[3] >= -prec;
    }
    // ---- |det| <= 1e-9 (the direction is not degenerate here)
    const float Sx = I[0] + I[3] + I[6], Sy = I[1] + I[4] + I[7], Sz = I[2] + I[5] + I[8];
    const float area = Sx * Sx + Sy * Sy + Sz * Sz;
    auto col = [&](int c, int r) { return A[r * 4 + c]; };                       // component r of column c
    auto cross_d = [&](int c, float out[3]) {                                    // (column c) x (-d)
        out[0] = col(c, 1) * A[11] - col(c, 2) * A[7];
        out[1] = col(c, 2) * A[3] - col(c, 0) * A[11];
        out[2] = col(c, 0) * A[7] - col(c, 1) * A[3];
    };
    if (area <= prec) {
        const float e[9] = {vc[0] - vb[0], vc[1] - vb[1], vc[2] - vb[2], va[0] - vc[0], va[1] - vc[1], va[2] - vc[2],
                            vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]};
        const float l[3] = {e[0] * e[0] + e[1] * e[1] + e[2] * e[2], e[3] * e[3] + e[4] * e[4] + e[5] * e[5], e[6] * e[6] + e[7] * e[7] + e[8] * e[8]};
        int i = l[0] < l[1] ? 1 : 0;
        i = l[i] < l[2] ? 2 : i;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        if (l[i] <= prec) {                                  // the triangle is a point
            float cr[3];
            cross_d(i, cr);
            const float n_i = cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2];
            const float n3 = -col(i, 0) * A[3] - col(i, 1) * A[7] - col(i, 2) * A[11];
            return n_i <= prec && n3 >= -prec;
        }
        // the triangle is a segment (its longest edge e_i, from corner j to corner k)
        const float norm_ = I[3 * i] * I[3 * i] + I[3 * i + 1] * I[3 * i + 1] + I[3 * i + 2] * I[3 * i + 2];
        float nj, nk, n3;
        if (norm_ > prec) {
            float cj[3], ck[3];
            cross_d(j, cj); cross_d(k, ck);
            nj = I[3 * i] * ck[0] + I[3 * i + 1] * ck[1] + I[3 * i + 2] * ck[2];
            nk = -I[3 * i] * cj[0] - I[3 * i + 1] * cj[1] - I[3 * i + 2] * cj[2];
            n3 = nj + nk;
        } else {                                             // the origin is on the segment's line
            nj = col(k, 0) * e[3 * i] + col(k, 1) * e[3 * i + 1] + col(k, 2) * e[3 * i + 2];
            nk = -col(j, 0) * e[3 * i] - col(j, 1) * e[3 * i + 1] - col(j, 2) * e[3 * i + 2];
            n3 = l[i];
        }
        return N[i] >= -prec && N[i] <= prec && nj >= -prec && nk >= -prec && n3 > prec;
    }
    // ---- the ray is parallel to the triangle's plane
    float B[3] = {I[0] * Sx + I[1] * Sy + I[2] * Sz, I[3] * Sx + I[4] * Sy + I[5] * Sz, I[6] * Sx + I[7] * Sy + I[8] * Sz};
    int i = B[0] < B[1] ? 0 : 1;
    i = B[i] < B[2] ? i : 2;
    int j = (i + 1) % 3, k = (i + 2) % 3;
    if (B[k] < -prec) { k = j; j = i; i = 3 - j - k; }
    const bool in_plane = N[3] >= -prec && N[3] <= prec;
    if (B[j] < -prec) {                                      // outside two edges: the ray has to come in through one of them
        float ci[3], cj[3], ck[3];
        cross_d(i, ci); cross_d(j, cj); cross_d(k, ck);
        const float d0 = I[3 * i] * ck[0] + I[3 * i + 1] * ck[1] + I[3 * i + 2] * ck[2];
        const float d1 = -I[3 * i] * cj[0] - I[3 * i + 1] * cj[1] - I[3 * i + 2] * cj[2];
        const float d2 = I[3 * j] * ci[0] + I[3 * j + 1] * ci[1] + I[3 * j + 2] * ci[2];
        const float d3 = -I[3 * j] * ck[0] - I[3 * j + 1] * ck[1] - I[3 * j + 2] * ck[2];
        const float ni = I[3 * i] * I[3 * i] + I[3 * i + 1] * I[3 * i + 1] + I[3 * i + 2] * I[3 * i + 2];
        const float nj = I[3 * j] * I[3 * j] + I[3 * j + 1] * I[3 * j + 1] + I[3 * j + 2] * I[3 * j + 2];
        const bool v0 = d0 >= -prec && d1 >= -prec && ni > prec, v1 = d2 >= -prec && d3 >= -prec && nj > prec;
        return (v0 || v1) && in_plane;
    }
    if (B[i] < -prec) {                                      // outside one edge
        float cj[3], ck[3];
        cross_d(j, cj); cross_d(k, ck);
        const float nj = I[3 * i] * ck[0] + I[3 * i + 1] * ck[1] + I[3 * i + 2] * ck[2];
        const float nk = -I[3 * i] * cj[0] - I[3 * i + 1] * cj[1] - I[3 * i + 2] * cj[2];
        const float ni = nj + nk;
        return nj >= -prec && nk >= -prec && in_plane && ni > prec;
    }
    return B[i] >= -prec && in_plane;                        // the origin is inside the triangle
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
