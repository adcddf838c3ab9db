// Device bodies shared by the stand-alone loss kernels and the fused launch (bf_kp_contour_kernel, scan_kernels.hip) that runs
// the dense keypoint loss beside the silhouette loss's nearest-vertex scan: both only read the projected mesh, so one launch
// hides the keypoint workgroup's ~20 us latency chain under the scan's arithmetic.
#pragma once
#include "bf_internal.h"

namespace {
__device__ inline float lb_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
}  // namespace

// One sampled vertex X (world space, similarity applied) in mask view m of frame f: uvi = (u, v, inside, 1 / pix_z), duvb =
// d(binary term) / duv * weight; returns the vertex's share of the binary term (loss.py:95-104,119-128).  Shared by
// bf_mask_project_kernel and the forward mesh pass, which projects its tile's sampled vertices while it has them.
__device__ __forceinline__ float bf_mask_project_one(const MaskIO &K, float X0, float X1, float X2, const float *__restrict__ proj_all,
                                                     int f, int m, int s, float *__restrict__ uvi, float *__restrict__ duvb) {
    const float *P = proj_all + ((size_t)f * K.n_views + K.view_index[m]) * 12;
    float p0 = P[0] * X0 + P[1] * X1 + P[2] * X2 + P[3];
    float p1 = P[4] * X0 + P[5] * X1 + P[6] * X2 + P[7];
    float p2 = P[8] * X0 + P[9] * X1 + P[10] * X2 + P[11];
    float u = p0 / p2, v = p1 / p2;
    bool inside = u < K.imsize && u >= 0.f && v < K.imsize && v >= 0.f;
    // grid_sample(1 - mask, uv / imsize * 2 - 1): ix = ((x + 1) W - 1) / 2
    const float sx = (float)K.W / K.imsize, sy = (float)K.H / K.imsize;
    float ix = ((u / K.imsize * 2.f - 1.f + 1.f) * K.W - 1.f) * 0.5f, iy = ((v / K.imsize * 2.f - 1.f + 1.f) * K.H - 1.f) * 0.5f;
    float fx = floorf(ix), fy = floorf(iy);
    int x0 = (int)fx, y0 = (int)fy;
    float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const unsigned char *mk = K.masks + ((size_t)f * K.n_masks + m) * K.H * K.W;
    auto at = [&](int y, int x) -> float {
        return (x >= 0 && x < K.W && y >= 0 && y < K.H) ? 1.f - (float)mk[(size_t)y * K.W + x] : 0.f;   // zeros padding
    };
    float v00 = at(y0, x0), v01 = at(y0, x0 + 1), v10 = at(y0 + 1, x0), v11 = at(y0 + 1, x0 + 1);
    const float lval = K.eps * (v00 * wx0 * wy0 + v01 * wx1 * wy0 + v10 * wx0 * wy1 + v11 * wx1 * wy1);
    float gx = ((v01 - v00) * wy0 + (v11 - v10) * wy1) * sx, gy = ((v10 - v00) * wx0 + (v11 - v01) * wx1) * sy;
    const size_t o = ((size_t)f * K.n_masks + m) * K.ns + s;
    float4 rec = {u, v, inside ? 1.f : 0.f, 1.f / p2};
    ((float4 *)uvi)[o] = rec;
    duvb[o * 2] = K.weight * K.eps * gx;
    duvb[o * 2 + 1] = K.weight * K.eps * gy;
    if (K.acc) { K.acc[o * 2] = 0ull; K.acc[o * 2 + 1] = 0ull; }      // (this iteration's contour sums start here)
    return lval;
}

// Dense keypoint loss (more than 32 loss joints, i.e. SMPL-X with hands + face): multiview_keypoint_loss
// (smplify/loss.py:139-203) over nl joints x V views from the all-joints array of bf_joints_kernel, and the
// routing of its gradient: chain joints -> ext's dGt / dt / ds blocks, vertex-based joints (selector vertices,
// barycentric face landmarks) -> dL/dvout, added in joint order by one workgroup per frame (deterministic).
// grid (F), 512 threads.
__device__ __forceinline__ void bf_kp_loss_body(int f, float *sm, KpIO Q, const float *__restrict__ jraw, const float *__restrict__ state, const float *__restrict__ proj_all,
                  const float *__restrict__ keypoints, const int *__restrict__ ndiv, const int *__restrict__ lmk_vid,
                  const float *__restrict__ lmk_w, float *__restrict__ ext, float *__restrict__ dvout, float *__restrict__ terms) {
    const int tid = threadIdx.x, nl = Q.nl, V = Q.n_views;
    const int NLP = (nl + 31) & ~31, slots = max(1, 512 / NLP);
    float *s_part = sm;                       // [slots][NLP][4]
    float *s_g = s_part + slots * NLP * 4;    // [nl][4]  dL/dXw and the loss share
    float *s_x = s_g + nl * 4;                // [nl][3]  model-space joint
    StateView st = bf_state_view(const_cast<float *>(state) + (size_t)f * bf_state_stride(Q.nj, Q.npf, Q.nb), Q.nj, Q.npf, Q.nb);
    const float t0 = st.t[0], t1 = st.t[1], t2 = st.t[2], cs = st.sc[1], sc = st.sc[0] * cs;
    const float ndiv_f = (float)ndiv[f], icoeff = 1.0f / Q.coeff, kscale = -1.0f / (Q.coeff * ndiv_f), s2 = Q.sigma2;
    const int j = tid % NLP, vs = tid / NLP;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, ls = 0.f;
    // The views' projection matrices are staged in LDS once (the routing keys' slot: free until the loss is summed) and a thread's
    // keypoints are requested four views ahead: rolled, with everything read from global memory inside it, the loop paid one
    // memory latency per view - 16 views per thread, 21 k of this workgroup's 49 k cycles (round 3 stamps).  The arithmetic and its
    // order per (joint, view slot) are unchanged.
    float *s_P = sm + ((slots * NLP * 4 + nl * 4 + nl * 3 + 8 + 3) & ~3);   // [V][12], 16-byte aligned (three b128 per view), while it fits the keys' + weights' slot
    const bool p_lds = V * 12 + 3 <= 1024 + nl * 3;
    // ... and the model's small index tables (joint map, the chain joints' CSR lists, selector vertex ids) go to LDS with them: every
    // one of them was the first half of a dependent pair of global loads somewhere down this workgroup's chain
    int *s_jm = (int *)(s_x + nl * 3 + 8) + 1024 + nl * 3 + 16;      // [nl] joint_map | [nj + 1] cj_start | [nl] cj_list | [n_selector] (bf_kp_tab_ints)
    int *s_cs = s_jm + nl, *s_cl = s_cs + Q.nj + 1, *s_sel = s_cl + nl;
    for (int i = tid; i < nl; i += 512) { s_jm[i] = Q.joint_map[i]; if (i < Q.n_cj_list) s_cl[i] = Q.cj_list[i]; }
    for (int i = tid; i <= Q.nj; i += 512) s_cs[i] = Q.cj_start[i];
    for (int i = tid; i < Q.n_selector; i += 512) s_sel[i] = Q.selector_ids[i];
    if (p_lds)
        for (int i = tid; i < V * 12; i += 512) s_P[i] = proj_all[(size_t)f * V * 12 + i];
    // The routing stage far below needs, per (loss joint, corner) item, the vertex a barycentric landmark's corner sits on and its weight
    // (per frame: the contour landmarks move with the neck).  They depend on nothing this workgroup computes, so they are requested
    // NOW - joint map entry, then the pair - and wait in registers: down there the pair was a global round trip (~1.5 us with the
    // caches as cold as a kernel start leaves them) in the middle of the chain.
    const int n_ori = Q.nj + Q.n_selector;
    int pre_vid[2] = {-1, -1};
    float pre_w[2] = {1.f, 1.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = tid + h * 512;
        if (i < nl * 3) {
            const int q = i / 3, c = i - q * 3, src = Q.joint_map[q];
            if (src >= n_ori + Q.n_extra) {
                const size_t l = ((size_t)f * Q.n_lmk + (src - n_ori - Q.n_extra)) * 3 + c;
                pre_vid[h] = lmk_vid[l];
                pre_w[h] = lmk_w[l];
            }
        }
    }
    __syncthreads();
    if (vs < slots && j < nl) {
        const float *x = jraw + ((size_t)f * Q.n_all + s_jm[j]) * 3;
        const float y0 = x[0] + t0, y1 = x[1] + t1, y2 = x[2] + t2;
        const float x0 = y0 * sc, x1 = y1 * sc, x2 = y2 * sc;
        if (vs == 0) { s_x[j * 3] = x[0]; s_x[j * 3 + 1] = x[1]; s_x[j * 3 + 2] = x[2]; }
        auto one_view = [&](const float *Pm, float kx, float ky, float kc) {
            const float4 Pr0 = ((const float4 *)Pm)[0], Pr1 = ((const float4 *)Pm)[1], Pr2 = ((const float4 *)Pm)[2];       // (rows of K [R | t]; 16-byte aligned in both homes)
            const float P[12] = {Pr0.x, Pr0.y, Pr0.z, Pr0.w, Pr1.x, Pr1.y, Pr1.z, Pr1.w, Pr2.x, Pr2.y, Pr2.z, Pr2.w};
            float c2 = kc * kc;
            float p0 = P[0] * x0 + P[1] * x1 + P[2] * x2 + P[3];
            float p1 = P[4] * x0 + P[5] * x1 + P[6] * x2 + P[7];
            float p2 = P[8] * x0 + P[9] * x1 + P[10] * x2 + P[11];
            // (v_rcp_f32, 1 ulp, like the sparse fit kernel's projection: the three correctly rounded divisions were a quarter of this
            //  loop's instructions, and the loop is issue-bound - eight waves of one workgroup share four SIMDs)
            float ip2 = __builtin_amdgcn_rcpf(p2), u = p0 * ip2, w = p1 * ip2;
            float rx = (kx - u) * icoeff, ry = (ky - w) * icoeff;
            float ix = __builtin_amdgcn_rcpf(s2 + rx * rx), iy = __builtin_amdgcn_rcpf(s2 + ry * ry);
            ls += c2 * (s2 * rx * rx * ix + s2 * ry * ry * iy);
            float k = c2 * kscale;
            float du = k * (2.f * s2 * s2 * rx * ix * ix), dw = k * (2.f * s2 * s2 * ry * iy * iy);
            float q0 = du * ip2, q1 = dw * ip2, q2 = -(du * u + dw * w) * ip2;
            g0 += P[0] * q0 + P[4] * q1 + P[8] * q2;
            g1 += P[1] * q0 + P[5] * q1 + P[9] * q2;
            g2 += P[2] * q0 + P[6] * q1 + P[10] * q2;
        };
        const float *kp0 = keypoints + ((size_t)f * V * nl + j) * 3;             // + v * nl * 3
        auto fetch = [&](int v, float *k3) {
            if (v < V) { const float *kp = kp0 + (size_t)v * nl * 3; k3[0] = kp[0]; k3[1] = kp[1]; k3[2] = kp[2]; }
            else { k3[0] = 0.f; k3[1] = 0.f; k3[2] = 0.f; }
        };
        float kn[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) fetch(vs + q * slots, kn[q]);
        for (int v = vs; v < V; v += 4 * slots) {
            float kc_[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) { kc_[q][0] = kn[q][0]; kc_[q][1] = kn[q][1]; kc_[q][2] = kn[q][2]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) fetch(v + (4 + q) * slots, kn[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vv = v + q * slots;
                // (two calls, not one pointer chosen between LDS and global memory: that pointer would be a flat one)
                if (vv < V) { if (p_lds) one_view(s_P + vv * 12, kc_[q][0], kc_[q][1], kc_[q][2]); else one_view(proj_all + ((size_t)f * V + vv) * 12, kc_[q][0], kc_[q][1], kc_[q][2]); }
            }
        }
    }
    if (vs < slots && j < NLP) { float4 pr = {g0, g1, g2, ls}; ((float4 *)s_part)[vs * NLP + j] = pr; }
    __syncthreads();
    if (tid < nl) {
        float4 a = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < slots; ++q) { float4 p = ((float4 *)s_part)[q * NLP + tid]; a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
        ((float4 *)s_g)[tid] = a;
    }
    __syncthreads();
    const int EXT_T = Q.npf + Q.nj * 12 + Q.nb, EXT_G = EXT_T + 4, EXT_K = EXT_G + Q.nj * 3, EXT = EXT_K + 4;
    float *e = ext + (size_t)f * EXT;
    // chain joints: pull the loss joints that map to each (CSR), in loss-joint order
    for (int i = tid; i < Q.nj * 3; i += 512) {
        int cj = i / 3, k = i - cj * 3;
        float acc = 0.f;
        for (int q = s_cs[cj]; q < s_cs[cj + 1]; ++q) acc += s_g[s_cl[q] * 4 + k];
        e[EXT_G + i] = acc * sc;
    }
    {
        // d/dt, d/ds through the chain-joint-based loss joints only (the vertex-based ones go through dvout), and the loss value:
        // five sums over the loss joints, one WAVE each (waves 3..7; lane l takes joints l, l + 64, ... in order, then a fixed
        // xor tree) - a single thread walking 135 LDS entries per sum was a third of this kernel's time
        const int wv = tid >> 6, lane = tid & 63, which = wv - 3;
        if (which >= 0 && which < 5) {
            float acc = 0.f;
            for (int q = lane; q < nl; q += 64) {
                const bool chain = s_jm[q] < Q.nj;
                if (which < 3) acc += chain ? s_g[q * 4 + which] : 0.f;
                else if (which == 3) acc += chain ? s_g[q * 4] * (s_x[q * 3] + t0) + s_g[q * 4 + 1] * (s_x[q * 3 + 1] + t1) + s_g[q * 4 + 2] * (s_x[q * 3 + 2] + t2) : 0.f;
                else acc += s_g[q * 4 + 3];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) {
                if (which < 3) e[EXT_K + which] = acc * sc;
                else if (which == 3) e[EXT_K + 3] = acc * cs;
                else terms[(size_t)f * 4] = acc / ndiv_f;
            }
        }
    }
    // vertex-based joints: their vertices may coincide, so the order of the additions matters.  The (joint, corner) items
    // are SORTED by (vertex, item index) - a bitonic network over keys in LDS - so the items of a vertex become one run in
    // joint order; the thread at the start of a run adds them up in that order and applies the total with a single
    // read-modify-write (dL/dvertices is zero when this kernel starts, so the bits are those of adding item by item).
    // (The first version ranked every item against all earlier ones: O(n^2) LDS reads, 15 of this kernel's 26 us.  A uniform walk -
    //  every item reads the whole id list once, four ids per broadcast b128, and the lowest item of a vertex sums the later ones -
    //  was measured in round 3: no barriers, but some lane of a wave matches in nearly every step, so every step pays the
    //  accumulate body: config 3 went from 0.081 to 0.091 ms per iteration.  The sort's 45 short barrier stages are cheaper.)
    float *dv = dvout + (size_t)f * Q.nv * 3;
    int *s_key = (int *)(s_x + nl * 3 + 8);               // [N] (vertex << 10 | item), 0x7fffffff = no vertex
    float *s_w = (float *)(s_key + 1024);                 // [n_items] weight of the item
    const int n_items = nl * 3, N = n_items <= 512 ? 512 : 1024;
    __syncthreads();
    for (int i = tid; i < N; i += 512) {
        int key = 0x7fffffff;
        if (i < n_items) {
            const int q = i / 3, c = i - q * 3, src = s_jm[q];
            int vid = -1;
            float w = 1.f;
            if (src >= Q.nj) {
                if (src < n_ori) { if (c == 0) vid = s_sel[src - Q.nj]; }
                else if (src < n_ori + Q.n_extra) { }                  // a regressed joint: every vertex of its row, below
                else { vid = pre_vid[i >= 512 ? 1 : 0]; w = pre_w[i >= 512 ? 1 : 0]; }       // (requested at the top of the kernel)
            }
            s_w[i] = w;
            if (vid >= 0) key = (vid << 10) | i;
        }
        s_key[i] = key;
    }
    __syncthreads();
    if (N == 512) {
        // one key per thread: a compare-exchange with a partner less than 64 positions away is a wave shuffle (39 of the 45 stages);
        // the six others cross waves through LDS.  (All in LDS with a barrier per stage this sort was 20 k of the workgroup's 49 k cycles.)
        int key = s_key[tid];
        for (int k = 2; k <= 512; k <<= 1)
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                int other;
                if (jj < 64) other = __shfl_xor(key, jj);
                else {
                    __syncthreads();
                    s_key[tid] = key;
                    __syncthreads();
                    other = s_key[tid ^ jj];
                }
                const bool up = (tid & k) == 0, low = (tid & jj) == 0;            // (ascending block; this thread holds the lower position)
                const int lo = key < other ? key : other, hi = key < other ? other : key;
                key = (up == low) ? lo : hi;
            }
        __syncthreads();
        s_key[tid] = key;
        __syncthreads();
    } else
    for (int k = 2; k <= N; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int t = tid; t < N / 2; t += 512) {
                const int lo = ((t / jj) * jj * 2) + (t % jj), hi = lo + jj;
                const int a = s_key[lo], b2 = s_key[hi];
                const bool up = (lo & k) == 0;
                if ((a > b2) == up) { s_key[lo] = b2; s_key[hi] = a; }
            }
            __syncthreads();
        }
    for (int p = tid; p < N; p += 512) {
        const int key = s_key[p];
        if (key == 0x7fffffff) continue;
        const int vid = key >> 10;
        if (p > 0 && (s_key[p - 1] >> 10) == vid) continue;            // not the start of its vertex's run
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int r = p; r < N && (s_key[r] >> 10) == vid && s_key[r] != 0x7fffffff; ++r) {
            const int item = s_key[r] & 1023, q = item / 3;
            const float w = s_w[item];
            a0 += w * s_g[q * 4]; a1 += w * s_g[q * 4 + 1]; a2 += w * s_g[q * 4 + 2];
        }
        // (dL/dvertices is zero when this workgroup starts - the forward mesh pass or a memset left it so, and nothing else writes it
        //  before the keypoint loss - and a vertex's run has exactly one owner: a plain store, not a load + add + store whose load
        //  was the last global round trip of the chain)
        float *o = dv + (size_t)vid * 3;
        o[0] = a0; o[1] = a1; o[2] = a2;
    }
    // loss joints that come from J_regressor_extra (models/smpl.py:72: joint = row . vertices): every vertex of the row gets its
    // weight times the joint's gradient.  Joint after joint in loss-joint order, a vertex always by the same thread, after the
    // runs above: a fixed order of additions.  (No reference model has such a joint among the SMPL-X loss joints; the ABI allows it.)
    if (Q.n_extra > 0) {
        __syncthreads();
        for (int q = 0; q < nl; ++q) {
            const int src = s_jm[q];
            if (src < n_ori || src >= n_ori + Q.n_extra) continue;
            const float *row = Q.j_extra + (size_t)(src - n_ori) * Q.nv;
            const float gq0 = s_g[q * 4], gq1 = s_g[q * 4 + 1], gq2 = s_g[q * 4 + 2];
            for (int v = tid; v < Q.nv; v += 512) {
                const float w = row[v];
                if (w != 0.f) { float *o = dv + (size_t)v * 3; o[0] += w * gq0; o[1] += w * gq1; o[2] += w * gq2; }
            }
        }
    }
}

// grid (ceil(16 Cmax/256), M, F).  For contour point c: choice[F][M][Cmax] = sampled vertex (or -1),
// cgrad[F][M][Cmax][2] = weight * coeff * (uv - c) / |uv - c|.  SIXTEEN lanes per contour point, each scanning every 16th
// sampled vertex; the group is merged with the lexicographic (distance, index) minimum = torch.min's first minimum.
template <int NT>
__device__ __forceinline__ void bf_mask_contour_body(int bx, int m, int f, float4 *tile, float *sred, MaskIO K, const float *__restrict__ uvi, int *__restrict__ choice, float *__restrict__ cgrad,
                       float *__restrict__ loss_part) {
    const int gid = bx * NT + threadIdx.x, c = gid >> 4, sub = gid & 15;
    const int vm = f * K.n_masks + m;
    const int cnt = K.contour_count[vm];
    const float *cp = K.contour_xy + ((size_t)K.contour_start[vm] + (c < cnt ? c : 0)) * 2;
    const float cx = cp[0], cy = cp[1];
    const float4 *rec = (const float4 *)uvi + (size_t)vm * K.ns;
    float best = 3.0e38f;
    int bidx = -1;
    // (the next tile's record is requested before this tile is scanned; a vertex outside the image is parked at u = 3e19, so
    //  its squared distance overflows past `best` and the scan needs no inside test; the winner's coordinates are re-read at the end)
    // cdist form (torch.cdist for more than 25 points, loss.py:108): dist^2 = x1_ . x2_ with x1_ = (-2u, -2v, |uv|^2, 1) and
    // x2_ = (cx, cy, 1, |c|^2), accumulated k = 0..3 as ONE fma chain (what the CPU sgemm does for K = 4), clamped at 0; the
    // norms are sums of individually rounded squares (pow(2).sum(-1)).  The tile carries |uv|^2 in .z.
    const bool cdist = K.cdist != 0;
    const float n2 = __fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy));
    auto fetch = [&](int s) {
        float4 r = s < K.ns ? rec[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(r.z > 0.5f)) r.x = 3.0e19f;
        r.z = __fadd_rn(__fmul_rn(r.x, r.x), __fmul_rn(r.y, r.y));
        return r;
    };
    // The tile sits in LDS as three arrays - X = -2 u, Y = -2 v (cdist form; u, v in the exact form), Z = |uv|^2 - and lane `sub` takes the
    // records 2 sub, 2 sub + 1 (+ 32 t): a pair is three b64 reads and the distance of both is five packed instructions (the factor
    // -2 is applied once per record when the tile is stored, not once per evaluation; a product by 2 is exact, so the bits are those of
    // the one-record form).  Each lane still walks its records in ascending order with a strict <, and the lanes are merged on
    // (distance, index): torch.min's first minimum, whatever the partition.
    typedef float c2f __attribute__((ext_vector_type(2)));
    float *tX = (float *)tile, *tY = tX + NT, *tZ = tY + NT;
    float4 nxt = fetch(threadIdx.x);
    for (int base = 0; base < K.ns; base += NT) {
        tX[threadIdx.x] = cdist ? -2.f * nxt.x : nxt.x;
        tY[threadIdx.x] = cdist ? -2.f * nxt.y : nxt.y;
        tZ[threadIdx.x] = nxt.z;
        nxt = fetch(base + NT + threadIdx.x);
        __syncthreads();
        const int lim = min(NT, K.ns - base);
        if (cdist) {
            const c2f cx2 = {cx, cx}, cy2 = {cy, cy}, n22 = {n2, n2};
            for (int i = 2 * sub; i < lim; i += 32) {
                const c2f x2 = *(const c2f *)(tX + i), y2 = *(const c2f *)(tY + i), z2 = *(const c2f *)(tZ + i);
                c2f acc = cx2 * x2;                                            // (separate statements: no contraction across them, as in the scalar form)
                acc = __builtin_elementwise_fma(cy2, y2, acc);
                acc = acc + z2;
                acc = acc + n22;
                const float d0 = fmaxf(acc.x, 0.f), d1 = fmaxf(acc.y, 0.f);  // clamp_min(0); a parked record gives +inf (so does one past the end)
                if (d0 < best) { best = d0; bidx = base + i; }
                if (d1 < best) { best = d1; bidx = base + i + 1; }
            }
        } else {
            for (int i = 2 * sub; i < lim; i += 32) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float dx = tX[i + e] - cx, dy = tY[i + e] - cy;
                    const float d2 = dx * dx + dy * dy;
                    if (d2 < best) { best = d2; bidx = base + i + e; }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 1; x < 16; x <<= 1) {
        const float ob = __shfl_xor(best, x);
        const int oi = __shfl_xor(bidx, x);
        if (oi >= 0 && (bidx < 0 || ob < best || (ob == best && oi < bidx))) { best = ob; bidx = oi; }
    }
    float bu = 0.f, bv = 0.f;
    if (bidx >= 0 && sub == 0) { const float4 r = rec[bidx]; bu = r.x; bv = r.y; }
    float lval = 0.f;
    if (c < cnt && sub == 0) {
        const size_t o = (size_t)vm * K.cmax + c;
        float gx = 0.f, gy = 0.f;
        if (bidx >= 0) {
            float d = sqrtf(best);
            int px = (int)bu, py = (int)bv;                                    // .long() truncation (loss.py:114)
            const unsigned char *mk = K.masks + (size_t)vm * K.H * K.W;
            float mval = (px >= 0 && px < K.W && py >= 0 && py < K.H) ? (float)mk[(size_t)py * K.W + px] : 0.f;
            float coeff = mval < 0.1f ? K.eps : 1.f;                            // (eps - 1) * outside + 1
            lval = coeff * d;
            if (d > 0.f) { gx = K.weight * coeff * (bu - cx) / d; gy = K.weight * coeff * (bv - cy) / d; }
            if (K.acc && d > 0.f) {
                unsigned long long *a = K.acc + ((size_t)vm * K.ns + bidx) * 2;
                (void)__hip_atomic_fetch_add(a, bf_acc_fixed(gx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_fetch_add(a + 1, bf_acc_fixed(gy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!K.acc) { choice[o] = bidx; cgrad[o * 2] = gx; cgrad[o * 2 + 1] = gy; }      // (the gather kernel's inputs: not in sums mode)
    }
    lval = lb_wave_sum(lval);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = lval;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < NT / 64; ++w) tot += sred[w];
        loss_part[(size_t)vm * K.part_stride + K.proj_blocks + bx] = tot;
    }
}
