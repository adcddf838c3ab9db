// SMPL+D stage for gfx950: Adam on a per-vertex displacement of the fitted mesh towards the scan
// (reference smplify/smplify.py:228-247): loss = icp + (normal_loss + laplacian) * constant_scale * 0.1 with
//   compute_normal_torch          utils/io_utils.py:406-428  (face normals / (|n|+1e-8), summed per vertex
//                                 corner by corner, normalised again)
//   point_cloud_loss_mesh_grid    smplify/loss.py:233-242    (one Frobenius norm)
//   normal_loss_mesh_grid         smplify/loss.py:260-271    (un-normalised scan face normals, smplify.py:149)
//   normal_laplacian_smoothness   smplify/loss.py:273-288
// and their hand-derived reverse.  All sums are gathers over a vertex -> (corner, face) CSR list in
// the order the reference's sparse products add them, so the stage is deterministic.
// Kernels per iteration: faces -> vertices(+P) -> [nearest, pc_partial from scan_kernels.hip] ->
// vertex gradient -> face gradient -> vertex gather + Adam.
#include "bf_internal.h"
#ifndef BF_ADJ_BATCH
#define BF_ADJ_BATCH 8        // incident faces of a vertex walked together (a closed triangle mesh averages six)
#endif

// grid (ceil(NF/256), F)
extern "C" __global__ void __launch_bounds__(256)
bf_disp_face_kernel(const int *__restrict__ faces, int nf, int nv, const float *__restrict__ base,
                    const float *__restrict__ disp, float *__restrict__ fnorm /*[F][nf][4]: unit normal, |n|*/) {
    const int f = blockIdx.x * 256 + threadIdx.x, fr = blockIdx.y;
    if (f >= nf) return;
    const float *b = base + (size_t)fr * nv * 3, *d = disp + (size_t)fr * nv * 3;
    float p[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v = faces[f * 3 + c];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[c * 3 + k] = b[v * 3 + k] + d[v * 3 + k];
    }
    float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
    float n0 = e1[1] * e2[2] - e1[2] * e2[1], n1 = e1[2] * e2[0] - e1[0] * e2[2], n2 = e1[0] * e2[1] - e1[1] * e2[0];
    float len = sqrtf(n0 * n0 + n1 * n1 + n2 * n2), s = len + 1e-8f;
    float4 o = {n0 / s, n1 / s, n2 / s, len};
    ((float4 *)fnorm)[(size_t)fr * nf + f] = o;
}

// grid (ceil(NV/256), F): deformed vertex P, vertex normal (unit) and |sum of face normals|
extern "C" __global__ void __launch_bounds__(256)
bf_disp_vertex_kernel(const int *__restrict__ adj_start, const int *__restrict__ adj /*face*4 + corner*/, int nf, int nv,
                      const float *__restrict__ base, const float *__restrict__ disp, const float *__restrict__ fnorm,
                      float *__restrict__ P, float *__restrict__ vnorm /*[F][nv][4]*/) {
    const int v = blockIdx.x * 256 + threadIdx.x, fr = blockIdx.y;
    if (v >= nv) return;
    const size_t o = ((size_t)fr * nv + v) * 3;
    P[o] = base[o] + disp[o]; P[o + 1] = base[o + 1] + disp[o + 1]; P[o + 2] = base[o + 2] + disp[o + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    // (the incident faces eight at a time: list entries, then normals, each level's loads in flight together - entry by entry the walk
    //  was a chain of dependent round trips per face; the additions stay in list order)
    const int i0 = adj_start[v], i1 = adj_start[v + 1];
    for (int base = i0; base < i1; base += BF_ADJ_BATCH) {
        int a[BF_ADJ_BATCH];
        float4 n[BF_ADJ_BATCH];
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) a[e] = base + e < i1 ? adj[base + e] : -1;
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) n[e] = a[e] >= 0 ? ((const float4 *)fnorm)[(size_t)fr * nf + (a[e] >> 2)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) if (a[e] >= 0) { a0 += n[e].x; a1 += n[e].y; a2 += n[e].z; }
    }
    float len = sqrtf(a0 * a0 + a1 * a1 + a2 * a2), s = len + 1e-8f;
    float4 r = {a0 / s, a1 / s, a2 / s, len};
    ((float4 *)vnorm)[(size_t)fr * nv + v] = r;
}

// grid (ceil(NV/256), F): dL/d(sum of face normals at v) from the normal and laplacian terms
extern "C" __global__ void __launch_bounds__(256)
bf_disp_vgrad_kernel(const int *__restrict__ faces, const int *__restrict__ adj_start, const int *__restrict__ adj, int nf, int nv,
                     const float *__restrict__ vnorm, const float *const *__restrict__ scan_fn, const int *__restrict__ cface, const float *__restrict__ cscale,
                     float *__restrict__ dvraw /*[F][nv][3]*/, const float *__restrict__ P, const float *__restrict__ C,
                     float *__restrict__ pc_partial) {
    const int v = blockIdx.x * 256 + threadIdx.x, fr = blockIdx.y;
    if (pc_partial) {
        // this block's share of |P - C|^2 while it is here (bf_pc_partial_kernel's sums in its order: one launch less per iteration)
        __shared__ float s_pc[4];
        float a = 0.f;
        if (v < nv) {
            const float *p = P + ((size_t)fr * nv + v) * 3, *c = C + ((size_t)fr * nv + v) * 3;
            const float d0 = p[0] - c[0], d1 = p[1] - c[1], d2 = p[2] - c[2];
            a = d0 * d0 + d1 * d1 + d2 * d2;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if ((threadIdx.x & 63) == 0) s_pc[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) pc_partial[(size_t)fr * gridDim.x + blockIdx.x] = (s_pc[0] + s_pc[1]) + (s_pc[2] + s_pc[3]);
    }
    if (v >= nv) return;
    const float w = cscale[fr] * 0.1f;                                  // smplify.py:242
    const float4 *vn = (const float4 *)vnorm + (size_t)fr * nv;
    const float4 me = vn[v];
    // normal loss: mean_v (1 - fn_scan[closest face] . n_v)
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    int cf = cface[(size_t)fr * nv + v];
    if (cf >= 0) {
        const float *fn = scan_fn[fr] + (size_t)cf * 3;
        float k = -w / (float)nv;
        g0 = k * fn[0]; g1 = k * fn[1]; g2 = k * fn[2];
    }
    // laplacian: mean_f (|na-nb|^2 + |nc-na|^2 + |nb-nc|^2)  ->  d/dn_v = 2 (2 n_v - n_o1 - n_o2) / NF per incident face
    const float k2 = 2.f * w / (float)nf;
    const int i0 = adj_start[v], i1 = adj_start[v + 1];
    for (int base = i0; base < i1; base += BF_ADJ_BATCH) {            // (eight incident faces at a time, level by level: see bf_disp_vertex_kernel)
        int a[BF_ADJ_BATCH], v1[BF_ADJ_BATCH], v2[BF_ADJ_BATCH];
        float4 o1[BF_ADJ_BATCH], o2[BF_ADJ_BATCH];
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) a[e] = base + e < i1 ? adj[base + e] : -1;
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) {
            const int f = a[e] >> 2, c = a[e] & 3;
            v1[e] = a[e] >= 0 ? faces[f * 3 + (c + 1) % 3] : 0; v2[e] = a[e] >= 0 ? faces[f * 3 + (c + 2) % 3] : 0;
        }
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) { o1[e] = vn[v1[e]]; o2[e] = vn[v2[e]]; }
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e)
            if (a[e] >= 0) { g0 += k2 * (2.f * me.x - o1[e].x - o2[e].x); g1 += k2 * (2.f * me.y - o1[e].y - o2[e].y); g2 += k2 * (2.f * me.z - o1[e].z - o2[e].z); }
    }
    // through n = x / (|x| + 1e-8):  dx = dn / s - n (n . dn) / |x|
    float len = me.w, s = len + 1e-8f, dot = me.x * g0 + me.y * g1 + me.z * g2;
    float q = len > 0.f ? dot / len : 0.f;
    const size_t o = ((size_t)fr * nv + v) * 3;
    dvraw[o] = g0 / s - me.x * q; dvraw[o + 1] = g1 / s - me.y * q; dvraw[o + 2] = g2 / s - me.z * q;
}

// grid (ceil(NF/256), F): per-face corner gradients dL/dP from the normal terms
extern "C" __global__ void __launch_bounds__(256)
bf_disp_fgrad_kernel(const int *__restrict__ faces, int nf, int nv, const float *__restrict__ P,
                     const float *__restrict__ fnorm, const float *__restrict__ dvraw, float *__restrict__ dPf /*[F][nf][9]*/) {
    const int f = blockIdx.x * 256 + threadIdx.x, fr = blockIdx.y;
    if (f >= nf) return;
    const float *pp = P + (size_t)fr * nv * 3, *dv = dvraw + (size_t)fr * nv * 3;
    int va = faces[f * 3], vb = faces[f * 3 + 1], vc = faces[f * 3 + 2];
    float d0 = dv[va * 3] + dv[vb * 3] + dv[vc * 3], d1 = dv[va * 3 + 1] + dv[vb * 3 + 1] + dv[vc * 3 + 1],
          d2 = dv[va * 3 + 2] + dv[vb * 3 + 2] + dv[vc * 3 + 2];
    float4 n = ((const float4 *)fnorm)[(size_t)fr * nf + f];
    float len = n.w, s = len + 1e-8f, dot = n.x * d0 + n.y * d1 + n.z * d2;
    float q = len > 0.f ? dot / len : 0.f;
    float g[3] = {d0 / s - n.x * q, d1 / s - n.y * q, d2 / s - n.z * q};        // dL/d(e1 x e2)
    float e1[3] = {pp[vb * 3] - pp[va * 3], pp[vb * 3 + 1] - pp[va * 3 + 1], pp[vb * 3 + 2] - pp[va * 3 + 2]};
    float e2[3] = {pp[vc * 3] - pp[va * 3], pp[vc * 3 + 1] - pp[va * 3 + 1], pp[vc * 3 + 2] - pp[va * 3 + 2]};
    float de1[3] = {e2[1] * g[2] - e2[2] * g[1], e2[2] * g[0] - e2[0] * g[2], e2[0] * g[1] - e2[1] * g[0]};   // e2 x g
    float de2[3] = {g[1] * e1[2] - g[2] * e1[1], g[2] * e1[0] - g[0] * e1[2], g[0] * e1[1] - g[1] * e1[0]};   // g x e1
    float *o = dPf + ((size_t)fr * nf + f) * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = -de1[k] - de2[k]; o[3 + k] = de1[k]; o[6 + k] = de2[k]; }
}

// grid (ceil(NV/256), F): total gradient of vertex v, then torch-semantics Adam on its 3 displacement scalars
extern "C" __global__ void __launch_bounds__(256)
bf_disp_adam_kernel(const int *__restrict__ adj_start, const int *__restrict__ adj, int nf, int nv, const float *__restrict__ P,
                    const float *__restrict__ C, const float *__restrict__ pc_partial, int n_partial,
                    const float *__restrict__ dPf, float *__restrict__ disp, float *__restrict__ am, float *__restrict__ av,
                    float step_size, float bc2_sqrt, float beta1, float beta2, float eps) {
    const int v = blockIdx.x * 256 + threadIdx.x, fr = blockIdx.y;
    if (v >= nv) return;
    float tot = 0.f;
    for (int b = 0; b < n_partial; ++b) tot += pc_partial[(size_t)fr * n_partial + b];     // fixed order
    const float inorm = 1.0f / sqrtf(tot);
    const size_t o = ((size_t)fr * nv + v) * 3;
    float g[3] = {(P[o] - C[o]) * inorm, (P[o + 1] - C[o + 1]) * inorm, (P[o + 2] - C[o + 2]) * inorm};
    const int i0 = adj_start[v], i1 = adj_start[v + 1];
    for (int base = i0; base < i1; base += BF_ADJ_BATCH) {            // (eight incident faces at a time, level by level: see bf_disp_vertex_kernel)
        int a[BF_ADJ_BATCH];
        float q[BF_ADJ_BATCH][3];
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) a[e] = base + e < i1 ? adj[base + e] : -1;
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) {
            const float *qp = dPf + ((size_t)fr * nf + (a[e] >= 0 ? a[e] >> 2 : 0)) * 9 + (a[e] >= 0 ? a[e] & 3 : 0) * 3;
            q[e][0] = qp[0]; q[e][1] = qp[1]; q[e][2] = qp[2];
        }
#pragma unroll
        for (int e = 0; e < BF_ADJ_BATCH; ++e) if (a[e] >= 0) { g[0] += q[e][0]; g[1] += q[e][1]; g[2] += q[e][2]; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float m = am[o + k], vv = av[o + k];
        m = m + (g[k] - m) * (1.0f - beta1);
        vv = vv * beta2 + (1.0f - beta2) * g[k] * g[k];
        am[o + k] = m; av[o + k] = vv;
        disp[o + k] = disp[o + k] - step_size * (m / (sqrtf(vv) / bc2_sqrt + eps));
    }
}
