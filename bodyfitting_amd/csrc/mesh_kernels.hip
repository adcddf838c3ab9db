// Full-mesh SMPL forward for gfx950: the HBM-bound part of the path.
//
//   bf_pose_state_kernel  theta, beta -> per-frame pose state (chain matrices A_j, pose feature)
//   bf_mesh_kernel        streams posedirs / shapedirs / lbs_weights once and emits all NV vertices
//   bf_joints_kernel      chain + selector + extra-regressor joints, gathered by joint_map
//
// Reference being replaced: smplx 0.1.13 `lbs()` as called from models/smpl.py:69-83 and
// smplify/smplify.py:179-190; per frame-iteration it streams
// posedirs 17,114,760 B + shapedirs 826,800 B + lbs_weights 661,440 B + v_template 82,680 B.
//
// Layout: posedirs is [P, 3NV] row-major, i.e. for one pose-feature row consecutive lanes read
// consecutive (vertex, xyz) columns - already coalesced.  A workgroup owns a tile of 32 vertices
// (96 columns) and splits the P rows over 8 row groups (split-K inside the workgroup, reduced through
// LDS), so the grid is ceil(NV/32) = 216 workgroups of 12 waves: every load of a wave is
// independent, the whole 17 MB matrix is in flight at once and all 256 CUs have work.  The 24 chain
// matrices (1.1 KB) and the 207-float pose feature sit in LDS.
#include "bf_internal.h"

namespace {
__device__ inline void m_rodrigues(float tx, float ty, float tz, float *R) {
    float ux = tx + 1e-8f, uy = ty + 1e-8f, uz = tz + 1e-8f;
    float a = sqrtf(ux * ux + uy * uy + uz * uz);
    float nx = tx / a, ny = ty / a, nz = tz / a;
    float s = sinf(a), c = cosf(a), oc = 1.0f - c;
    R[0] = 1.0f + oc * (-nz * nz - ny * ny);
    R[1] = s * (-nz) + oc * (nx * ny);
    R[2] = s * ny + oc * (nx * nz);
    R[3] = s * nz + oc * (nx * ny);
    R[4] = 1.0f + oc * (-nz * nz - nx * nx);
    R[5] = s * (-nx) + oc * (ny * nz);
    R[6] = s * (-ny) + oc * (nx * nz);
    R[7] = s * nx + oc * (ny * nz);
    R[8] = 1.0f + oc * (-ny * ny - nx * nx);
}
__device__ inline float m_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
}  // namespace

// One 128-thread workgroup per parameter set.  betas[n][nb], orient[n][3], body_pose[n][3(nj-1)];
// transl / scale are taken as (0,0,0) / 1 / 1 when `sim` is null, else sim[n][5] = t, s, c.
// `packed` != null: read everything from the optimiser-order parameter block packed[n][np] instead
// (transl, scale from it; constant scale from cscale[n] or cscale_all).
extern "C" __global__ void __launch_bounds__(128)
bf_pose_state_kernel(FitTab T, const float *__restrict__ betas, const float *__restrict__ orient,
                     const float *__restrict__ body_pose, const float *__restrict__ sim, float *state,
                     const float *__restrict__ packed, const float *__restrict__ cscale, float cscale_all) {
    __shared__ float R[64 * 9], J[64 * 3], GR[64 * 9], Gt[64 * 3];
    const int tid = threadIdx.x, nt = 128, f = blockIdx.x;
    const int nj = T.nj, nb = T.nb, npf = T.npf;
    const float *pk = packed ? packed + (size_t)f * T.np : nullptr;
    const float *beta = pk ? pk + T.off_beta : betas + (size_t)f * nb;
    StateView st = bf_state_view(state + (size_t)f * bf_state_stride(nj, npf, nb), nj, npf, nb);
    if (tid < nj) {
        float th[3];
        if (pk) {
#pragma unroll
            for (int k = 0; k < 3; ++k) th[k] = bf_theta(pk, tid, k, T.th_kind, T.th_off, T.pose_mean, T.hand_comp, T.n_pca, T.off_lh, T.off_rh);
        } else {
            const float *src = tid == 0 ? orient + (size_t)f * 3 : body_pose + (size_t)f * 3 * (nj - 1) + 3 * (tid - 1);
            th[0] = src[0]; th[1] = src[1]; th[2] = src[2];
        }
        m_rodrigues(th[0], th[1], th[2], R + tid * 9);
        st.theta[tid * 3] = th[0]; st.theta[tid * 3 + 1] = th[1]; st.theta[tid * 3 + 2] = th[2];
    }
    for (int i = tid; i < nj * 3; i += nt) {
        float acc = 0.f;
        for (int l = 0; l < nb; ++l) acc += T.Jd[i * nb + l] * beta[l];
        J[i] = T.Jt[i] + acc;
    }
    __syncthreads();
    if (tid < 9) GR[tid] = R[tid];
    if (tid >= 64 && tid < 67) Gt[tid - 64] = J[tid - 64];
    __syncthreads();
    for (int lev = 1; lev < T.n_levels; ++lev) {
        int ls = T.level_start[lev], cnt = (T.level_start[lev + 1] - ls) * 3;
        for (int idx = tid; idx < cnt; idx += nt) {
            int i = T.level_joints[ls + idx / 3], r = idx % 3, p = T.parents[i];
            float g0 = GR[p * 9 + r * 3], g1 = GR[p * 9 + r * 3 + 1], g2 = GR[p * 9 + r * 3 + 2];
            const float *Ri = R + i * 9;
            GR[i * 9 + r * 3 + 0] = g0 * Ri[0] + g1 * Ri[3] + g2 * Ri[6];
            GR[i * 9 + r * 3 + 1] = g0 * Ri[1] + g1 * Ri[4] + g2 * Ri[7];
            GR[i * 9 + r * 3 + 2] = g0 * Ri[2] + g1 * Ri[5] + g2 * Ri[8];
            float r0 = J[i * 3] - J[p * 3], r1 = J[i * 3 + 1] - J[p * 3 + 1], r2 = J[i * 3 + 2] - J[p * 3 + 2];
            Gt[i * 3 + r] = g0 * r0 + g1 * r1 + g2 * r2 + Gt[p * 3 + r];
        }
        __syncthreads();
    }
    for (int i = tid; i < nj * 9; i += nt) st.GR[i] = GR[i];
    for (int i = tid; i < nj * 3; i += nt) {
        int j = i / 3, a = i % 3;
        const float *g = GR + j * 9 + a * 3;
        st.At[i] = Gt[i] - (g[0] * J[j * 3] + g[1] * J[j * 3 + 1] + g[2] * J[j * 3 + 2]);
        st.Gt[i] = Gt[i];
    }
    for (int p = tid; p < npf; p += nt) {
        int j = 1 + p / 9, e = p % 9;
        st.feat[p] = R[j * 9 + e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
    }
    if (tid < nb) st.beta[tid] = beta[tid];
    if (pk) {
        if (tid < 3) st.t[tid] = pk[tid];
        if (tid == 3) { st.sc[0] = pk[3]; st.sc[1] = cscale ? cscale[f] : cscale_all; }
    } else {
        if (tid < 3) st.t[tid] = sim ? sim[(size_t)f * 5 + tid] : 0.f;
        if (tid == 3) { st.sc[0] = sim ? sim[(size_t)f * 5 + 3] : 1.f; st.sc[1] = sim ? sim[(size_t)f * 5 + 4] : 1.f; }
    }
}

// grid (ceil(NV/32), F), block 96 x 8.  vraw = model-space vertices (lbs output), vout = (v + t) s c.
// Every operand a thread will need is requested before the first barrier (posedirs slice by all
// threads; lbs_weights row by row-group 0, shapedirs + template by row-group 1, the extra-regressor
// slice by row-group 2), so the kernel pays one memory round trip instead of one per phase.
#define BF_MESH_PF 26      // posedirs rows prefetched per thread (= ceil(207 / 8) for SMPL); more are streamed
#define BF_MESH_WPF 24     // lbs weights prefetched per vertex thread
extern "C" __global__ void __launch_bounds__(BF_MESH_TILE * 3 * BF_MESH_RG)
bf_mesh_kernel(MeshTab M, const float *__restrict__ state, float *__restrict__ vraw, float *__restrict__ vout,
               float *__restrict__ xpart, float *__restrict__ vposed, const float *__restrict__ pose_off) {
    constexpr int COLS = BF_MESH_TILE * 3;
    extern __shared__ __align__(16) float sm[];
    const int nj = M.nj, nb = M.nb, npf = M.npf, nv = M.nv;
    float *s_feat = sm;                         // [npf]
    float *s_A = s_feat + ((npf + 3) & ~3);     // [nj][12]  rows of (GR | At)
    float *s_red = s_A + nj * 12;               // [RG][COLS]
    float *s_vp = s_red + BF_MESH_RG * COLS;    // [COLS]
    float *s_beta = s_vp + COLS;                // [nb] + t[3] + sc[2]
    const int tid = threadIdx.x, nt = COLS * BF_MESH_RG;
    const int col = tid % COLS, rg = tid / COLS;
    const int frame = blockIdx.y;
    StateView st = bf_state_view(const_cast<float *>(state) + (size_t)frame * bf_state_stride(nj, npf, nb), nj, npf, nb);
    const int ncols = 3 * nv;
    const int gcol = blockIdx.x * COLS + col;
    const bool ok = gcol < ncols;
    const int vl = col / 3, k = col - vl * 3, v = blockIdx.x * BF_MESH_TILE + vl;

    // ---- request everything ------------------------------------------------------------------
    const int rows = (npf + BF_MESH_RG - 1) / BF_MESH_RG;
    const int p0 = rg * rows, p1 = min(npf, p0 + rows);
    const float *pd = M.posedirs + (size_t)p0 * ncols + (ok ? gcol : 0);
    float pv[BF_MESH_PF];
#pragma unroll
    for (int i = 0; i < BF_MESH_PF; ++i) pv[i] = (!pose_off && p0 + i < p1) ? pd[(size_t)i * ncols] : 0.f;
    // (pose_off: the pose blend of this frame was already formed by bf_poseblend_mfma_kernel for the whole batch)
    const float poff = (pose_off && rg == 0 && ok) ? pose_off[(size_t)frame * ncols + gcol] : 0.f;
    float wreg[BF_MESH_WPF], sdreg[12], vt = 0.f, jx[BF_MESH_TILE];
#pragma unroll
    for (int j = 0; j < BF_MESH_WPF; ++j) wreg[j] = 0.f;
#pragma unroll
    for (int l = 0; l < 12; ++l) sdreg[l] = 0.f;
#pragma unroll
    for (int i = 0; i < BF_MESH_TILE; ++i) jx[i] = 0.f;
    const int ne3 = M.n_extra * 3;
    if (rg == 0 && ok) {
        const float *w = M.lbs_weights + (size_t)v * nj;
#pragma unroll
        for (int j = 0; j < BF_MESH_WPF; ++j) if (j < nj) wreg[j] = w[j];
    } else if (rg == 1 && ok) {
        const float *sd = M.shapedirs + (size_t)gcol * nb;
#pragma unroll
        for (int l = 0; l < 12; ++l) if (l < nb) sdreg[l] = sd[l];
        vt = M.v_template[gcol];
    } else if (rg == 2 && col < ne3 && xpart) {
        const int e = col / 3, v0 = blockIdx.x * BF_MESH_TILE;
        const float *row = M.j_extra + (size_t)e * nv + v0;
#pragma unroll
        for (int i = 0; i < BF_MESH_TILE; ++i) if (v0 + i < nv) jx[i] = row[i];
    }
    for (int i = tid; i < npf; i += nt) s_feat[i] = st.feat[i];
    for (int i = tid; i < nj * 12; i += nt) {
        int j = i / 12, e = i % 12, a = e / 4, b = e % 4;
        s_A[i] = b < 3 ? st.GR[j * 9 + a * 3 + b] : st.At[j * 3 + a];
    }
    if (tid < nb + 5) s_beta[tid] = st.beta[tid];   // beta, t, sc are contiguous in the state record
    __syncthreads();

    // ---- pose blend: this row group's share, reduced through LDS -----------------------------------
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < BF_MESH_PF; ++i) acc += (p0 + i < p1 ? s_feat[p0 + i] : 0.f) * pv[i];
    if (!pose_off) for (int p = p0 + BF_MESH_PF; p < p1; ++p) acc += s_feat[p] * pd[(size_t)(p - p0) * ncols];
    s_red[rg * COLS + col] = pose_off ? (rg == 0 ? poff : 0.f) : acc;
    if (rg == 1) {
        float a2 = 0.f;
#pragma unroll
        for (int l = 0; l < 12; ++l) if (l < nb) a2 += sdreg[l] * s_beta[l];
        for (int l = 12; l < nb; ++l) a2 += M.shapedirs[(size_t)gcol * nb + l] * s_beta[l];
        s_vp[col] = ok ? vt + a2 : 0.f;          // shaped vertex; the pose offset is added below
    }
    __syncthreads();
    if (rg == 0) {
        float off = 0.f;
#pragma unroll
        for (int q = 0; q < BF_MESH_RG; ++q) off += s_red[q * COLS + col];
        s_vp[col] += off;
    }
    __syncthreads();
    // ---- skinning ---------------------------------------------------------------------------------
    if (rg == 0) {
        float r = 0.f;
        if (ok) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, tt = 0.f;
#pragma unroll
            for (int j = 0; j < BF_MESH_WPF; ++j) {
                if (j < nj) {
                    const float4 a = *(const float4 *)(s_A + j * 12 + k * 4);
                    t0 += wreg[j] * a.x; t1 += wreg[j] * a.y; t2 += wreg[j] * a.z; tt += wreg[j] * a.w;
                }
            }
            for (int j = BF_MESH_WPF; j < nj; ++j) {
                float wj = M.lbs_weights[(size_t)v * nj + j];
                const float4 a = *(const float4 *)(s_A + j * 12 + k * 4);
                t0 += wj * a.x; t1 += wj * a.y; t2 += wj * a.z; tt += wj * a.w;
            }
            r = t0 * s_vp[vl * 3] + t1 * s_vp[vl * 3 + 1] + t2 * s_vp[vl * 3 + 2] + tt;
            size_t o = (size_t)frame * ncols + gcol;
            if (vraw) vraw[o] = r;
            if (vout) vout[o] = (r + s_beta[nb + k]) * s_beta[nb + 3] * s_beta[nb + 4];
            if (vposed) vposed[o] = s_vp[col];       // kept for the dense reverse pass
        }
        s_red[col] = r;                          // (own column of the reduce buffer: no hazard)
    }
    if (xpart) {
        // this tile's share of J_regressor_extra . vertices (models/smpl.py:72); summed over tiles, in tile
        // order, by bf_joints_kernel
        __syncthreads();
        if (rg == 2 && col < ne3) {
            float a3 = 0.f;
#pragma unroll
            for (int i = 0; i < BF_MESH_TILE; ++i) a3 += jx[i] * s_red[i * 3 + k];
            xpart[((size_t)frame * gridDim.x + blockIdx.x) * ne3 + col] = a3;
        }
    }
}

extern "C" size_t bf_mesh_smem_bytes(int nj, int npf, int nb) {
    constexpr int COLS = BF_MESH_TILE * 3;
    return sizeof(float) * (((npf + 3) & ~3) + nj * 12 + BF_MESH_RG * COLS + COLS + nb + 8);
}

// One 256-thread workgroup per frame.  All joints in smplx order: chain joints | selector vertices |
// J_regressor_extra rows (SMPL wrapper, models/smpl.py:72-75) | face landmarks (SMPL-X: 51 static + 17 contour
// landmarks chosen by the neck's yaw), then gathered by joint_map; similarity of smplify.py:189
// applied to the outputs.  `jraw` (optional) receives ALL joints in model space and `lmk_vid` / `lmk_w` the
// vertex ids / barycentric weights of the landmarks actually used, for the dense keypoint loss.
extern "C" __global__ void __launch_bounds__(256)
bf_joints_kernel(MeshTab M, const float *__restrict__ state, const float *__restrict__ vraw,
                 const float *__restrict__ xpart, float *__restrict__ joints, float *__restrict__ joints_ori,
                 float *__restrict__ jraw, int *__restrict__ lmk_vid, float *__restrict__ lmk_w) {
    __shared__ float s_extra[32 * 3];
    __shared__ float s_all[256 * 3];
    __shared__ int s_row;
    const int tid = threadIdx.x, frame = blockIdx.x;
    const int nj = M.nj, nb = M.nb, npf = M.npf, nv = M.nv, ne = M.n_extra, nsel = M.n_selector;
    const int nlm = M.n_lmk_static + M.n_lmk_dyn;
    StateView st = bf_state_view(const_cast<float *>(state) + (size_t)frame * bf_state_stride(nj, npf, nb), nj, npf, nb);
    const float *vr = vraw + (size_t)frame * nv * 3;
    const float t0 = st.t[0], t1 = st.t[1], t2 = st.t[2], sc = st.sc[0] * st.sc[1];
    const int ne3 = ne * 3, nt8 = M.n_tiles;
    for (int base = 0; base < ne3 * 8; base += 256) {
        int idx = base + tid, o = idx >> 3, sl = idx & 7;
        float acc = 0.f;
        if (o < ne3) {
            int per = (nt8 + 7) / 8, a = sl * per, b = min(nt8, a + per);
            const float *p = xpart + ((size_t)frame * nt8 + a) * ne3 + o;
            for (int t = a; t < b; ++t, p += ne3) acc += *p;
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
        if (o < ne3 && sl == 0) s_extra[o] = acc;
    }
    if (tid == 0 && M.n_lmk_dyn > 0) {
        // find_dynamic_lmk_idx_and_bcoords: y = round(clamp(-yaw * 180 / pi, max = 39)), negatives folded to 39 - y / 78
        const float *G = st.GR + M.neck_joint * 9;
        float yaw = atan2f(-G[6], sqrtf(G[0] * G[0] + G[3] * G[3]));
        int y = (int)rintf(fminf(-yaw * 180.0f / 3.14159265358979323846f, 39.f));
        if (y < 0) y = y < -39 ? 78 : 39 - y;
        s_row = y;
    }
    __syncthreads();
    const int n_ori = nj + nsel, n_all = n_ori + ne + nlm;
    for (int i = tid; i < n_all * 3; i += 256) {
        int j = i / 3, k = i - j * 3;
        float x;
        if (j < nj) x = st.Gt[j * 3 + k];
        else if (j < n_ori) x = vr[(size_t)M.selector_ids[j - nj] * 3 + k];
        else if (j < n_ori + ne) x = s_extra[(j - n_ori) * 3 + k];
        else {
            int l = j - n_ori - ne;
            int face = l < M.n_lmk_static ? M.lmk_faces[l] : M.dyn_faces[s_row * M.n_lmk_dyn + (l - M.n_lmk_static)];
            const float *bw = l < M.n_lmk_static ? M.lmk_bary + l * 3 : M.dyn_bary + ((size_t)s_row * M.n_lmk_dyn + (l - M.n_lmk_static)) * 3;
            const int *fv = M.faces + (size_t)face * 3;
            x = bw[0] * vr[(size_t)fv[0] * 3 + k] + bw[1] * vr[(size_t)fv[1] * 3 + k] + bw[2] * vr[(size_t)fv[2] * 3 + k];
            if (k == 0 && lmk_vid) {
                int *vo = lmk_vid + ((size_t)frame * nlm + l) * 3;
                float *wo = lmk_w + ((size_t)frame * nlm + l) * 3;
                vo[0] = fv[0]; vo[1] = fv[1]; vo[2] = fv[2]; wo[0] = bw[0]; wo[1] = bw[1]; wo[2] = bw[2];
            }
        }
        if (jraw) jraw[(size_t)frame * n_all * 3 + i] = x;
        float tk = k == 0 ? t0 : (k == 1 ? t1 : t2);
        s_all[i] = (x + tk) * sc;
    }
    __syncthreads();
    if (joints_ori)
        for (int i = tid; i < n_ori * 3; i += 256) joints_ori[(size_t)frame * n_ori * 3 + i] = s_all[i];
    if (joints)
        for (int i = tid; i < M.n_joint_map * 3; i += 256)
            joints[(size_t)frame * M.n_joint_map * 3 + i] = s_all[M.joint_map[i / 3] * 3 + i % 3];
}


// Batched pose blend on the matrix cores: OFF[f][col] = sum_p feat[f][p] * posedirs[p][col] for a whole batch of
// frames at once, i.e. the GEMM [F x P] . [P x 3NV] that the reference evaluates frame by frame as
// `torch.matmul(pose_feature, posedirs)` (smplx lbs).  fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain in k order), so parity is unchanged.
//   workgroup = 4 waves = 128 columns; wave = one 32-column strip x ALL frames (FT tiles of 32 frames, 16
//   accumulator VGPRs each), so every posedirs element is read from HBM exactly once per launch and reused
//   FT*32 times from registers; the pose features of the batch are staged through LDS in 16-deep k chunks
//   (row stride 17 floats: conflict-free for the A-fragment pattern lane -> (frame, k)).
// A fragment: lane l holds A[i = l & 31][k = l >> 5]; B fragment: lane l holds B[k = l >> 5][j = l & 31];
// accumulator: column = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define BF_PB_KC 16
#define BF_PB_LD 17
template <int FT>
__global__ void __launch_bounds__(256)
poseblend_mfma_kernel(MeshTab M, const float *__restrict__ state, int n_frames, float *__restrict__ pose_off) {
    __shared__ float s_a[FT * 32 * BF_PB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npf = M.npf, ncols = 3 * M.nv;
    const int col0 = (blockIdx.x * 4 + wave) * 32, col = col0 + (lane & 31), kh = lane >> 5;
    const bool col_ok = col < ncols;
    const int stride = bf_state_stride(M.nj, npf, M.nb), feat_off = M.nj * 15;      // GR 9 + At 3 + Gt 3 per joint, then feat
    f32x16 acc[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int p0 = 0; p0 < npf; p0 += BF_PB_KC) {
        __syncthreads();
        for (int i = tid; i < FT * 32 * BF_PB_KC; i += 256) {
            int f = i / BF_PB_KC, kk = i - f * BF_PB_KC;
            s_a[f * BF_PB_LD + kk] = (f < n_frames && p0 + kk < npf) ? state[(size_t)f * stride + feat_off + p0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BF_PB_KC; kk += 2) {
            const int p = p0 + kk + kh;
            const float b = (col_ok && p < npf) ? M.posedirs[(size_t)p * ncols + col] : 0.f;
#pragma unroll
            for (int t = 0; t < FT; ++t) {
                const float a = s_a[(t * 32 + (lane & 31)) * BF_PB_LD + kk + kh];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    if (col_ok) {
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int f = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (f < n_frames) pose_off[(size_t)f * ncols + col] = acc[t][r];
            }
    }
}

// frames [f0, f0 + n) of the batch; `state` and `pose_off` already point at frame f0
extern "C" hipError_t bf_poseblend_launch(const MeshTab *M, const float *state, int n, float *pose_off, hipStream_t stream) {
    const int ncols = 3 * M->nv;
    dim3 grid((ncols + 127) / 128);
    if (n <= 32) hipLaunchKernelGGL(poseblend_mfma_kernel<1>, grid, dim3(256), 0, stream, *M, state, n, pose_off);
    else if (n <= 64) hipLaunchKernelGGL(poseblend_mfma_kernel<2>, grid, dim3(256), 0, stream, *M, state, n, pose_off);
    else if (n <= 128) hipLaunchKernelGGL(poseblend_mfma_kernel<4>, grid, dim3(256), 0, stream, *M, state, n, pose_off);
    else hipLaunchKernelGGL(poseblend_mfma_kernel<8>, grid, dim3(256), 0, stream, *M, state, n, pose_off);
    return hipGetLastError();
}


// Per-frame part of the batched path: shaped vertex + pose offset (from the MFMA GEMM) -> skinning, for one
// 32-vertex tile.  grid (n_tiles, F), 128 threads (96 = vertex x coordinate, the last 32 take the extra-joint
// partials), nothing idle: the batched path launches F x 216 of these.
extern "C" __global__ void __launch_bounds__(128)
bf_mesh_epilogue_kernel(MeshTab M, const float *__restrict__ state, const float *__restrict__ pose_off,
                        float *__restrict__ vraw, float *__restrict__ vout, float *__restrict__ xpart, float *__restrict__ vposed) {
    constexpr int COLS = BF_MESH_TILE * 3;
    __shared__ __align__(16) float s_A[64 * 12];
    __shared__ float s_vp[COLS], s_raw[COLS], s_beta[24];
    const int nj = M.nj, nb = M.nb, npf = M.npf, nv = M.nv, ncols = 3 * nv;
    const int tid = threadIdx.x, frame = blockIdx.y, col = tid;
    StateView st = bf_state_view(const_cast<float *>(state) + (size_t)frame * bf_state_stride(nj, npf, nb), nj, npf, nb);
    const int gcol = blockIdx.x * COLS + col;
    const bool ok = col < COLS && gcol < ncols;
    const int vl = col / 3, k = col - vl * 3, v = blockIdx.x * BF_MESH_TILE + vl;
    float wreg[BF_MESH_WPF], sdreg[12], vt = 0.f, off = 0.f;
#pragma unroll
    for (int j = 0; j < BF_MESH_WPF; ++j) wreg[j] = 0.f;
#pragma unroll
    for (int l = 0; l < 12; ++l) sdreg[l] = 0.f;
    if (ok) {
        const float *w = M.lbs_weights + (size_t)v * nj;
#pragma unroll
        for (int j = 0; j < BF_MESH_WPF; ++j) if (j < nj) wreg[j] = w[j];
        const float *sd = M.shapedirs + (size_t)gcol * nb;
#pragma unroll
        for (int l = 0; l < 12; ++l) if (l < nb) sdreg[l] = sd[l];
        vt = M.v_template[gcol];
        off = pose_off[(size_t)frame * ncols + gcol];
    }
    for (int i = tid; i < nj * 12; i += 128) {
        int j = i / 12, e = i % 12, a = e / 4, b = e % 4;
        s_A[i] = b < 3 ? st.GR[j * 9 + a * 3 + b] : st.At[j * 3 + a];
    }
    if (tid < nb + 5) s_beta[tid] = st.beta[tid];
    __syncthreads();
    if (col < COLS) {
        float a2 = 0.f;
#pragma unroll
        for (int l = 0; l < 12; ++l) if (l < nb) a2 += sdreg[l] * s_beta[l];
        for (int l = 12; l < nb; ++l) a2 += M.shapedirs[(size_t)gcol * nb + l] * s_beta[l];
        s_vp[col] = ok ? vt + a2 + off : 0.f;
    }
    __syncthreads();
    if (col < COLS) {
        float r = 0.f;
        if (ok) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, tt = 0.f;
#pragma unroll
            for (int j = 0; j < BF_MESH_WPF; ++j) {
                if (j < nj) {
                    const float4 a = *(const float4 *)(s_A + j * 12 + k * 4);
                    t0 += wreg[j] * a.x; t1 += wreg[j] * a.y; t2 += wreg[j] * a.z; tt += wreg[j] * a.w;
                }
            }
            for (int j = BF_MESH_WPF; j < nj; ++j) {
                float wj = M.lbs_weights[(size_t)v * nj + j];
                const float4 a = *(const float4 *)(s_A + j * 12 + k * 4);
                t0 += wj * a.x; t1 += wj * a.y; t2 += wj * a.z; tt += wj * a.w;
            }
            r = t0 * s_vp[vl * 3] + t1 * s_vp[vl * 3 + 1] + t2 * s_vp[vl * 3 + 2] + tt;
            size_t o = (size_t)frame * ncols + gcol;
            if (vraw) vraw[o] = r;
            if (vout) vout[o] = (r + s_beta[nb + k]) * s_beta[nb + 3] * s_beta[nb + 4];
            if (vposed) vposed[o] = s_vp[col];
        }
        s_raw[col] = r;
    }
    if (xpart) {
        __syncthreads();
        const int ne3 = M.n_extra * 3, q = tid - COLS;
        if (q >= 0 && q < ne3) {
            int e = q / 3, kk = q - e * 3, v0 = blockIdx.x * BF_MESH_TILE;
            const float *row = M.j_extra + (size_t)e * nv + v0;
            float a3 = 0.f;
            for (int i = 0; i < BF_MESH_TILE; ++i) if (v0 + i < nv) a3 += row[i] * s_raw[i * 3 + kk];
            xpart[((size_t)frame * gridDim.x + blockIdx.x) * ne3 + q] = a3;
        }
    }
}
