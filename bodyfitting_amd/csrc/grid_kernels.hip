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
