// The pose state of one parameter set (theta, beta -> chain matrices A_j, pose feature): the body of bf_pose_state_kernel, also run
// by one wave of the resident dense-schedule fit launch for the parameters of every iteration (fit_kernels.hip: door_state), so that
// both produce the same bits.
#pragma once
#include "bf_internal.h"

#define BF_POSE_STATE_LDS (64 * 9 + 64 * 3 + 64 * 9 + 64 * 3 + 66 + 64 + 64 + 2)
#define BF_POSE_STATE_FLAG (64 * 9 + 64 * 3 + 64 * 9 + 64 * 3 + 66 + 64 + 64)      // (int) "R and J of iteration <value> are complete": bf_pose_chain_row's cue

namespace {
__device__ inline void m_rodrigues(float tx, float ty, float tz, float *R) {
    float ux = tx + 1e-8f, uy = ty + 1e-8f, uz = tz + 1e-8f;
    float a = sqrtf(ux * ux + uy * uy + uz * uz);
    float nx = tx / a, ny = ty / a, nz = tz / a;
    float s, c;
    sincosf(a, &s, &c);            // (one argument reduction for both; the same values as sinf / cosf)
    const float oc = 1.0f - c;
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
}  // namespace

// Where the body reads the model's small tables from: the FitTab's global arrays, or copies a caller already holds in LDS
// (the resident dense-schedule fit launch: a dependent global load costs ~700 cycles, there are a dozen of them in a row here)
struct PoseTabs {
    const int *th_kind, *th_off, *parents;
    const float *pose_mean, *hand_comp, *Jd, *Jt;
    int jd_stride;                  // row stride of Jd (nb in the FitTab, padded in the fit kernel's LDS copy)
    const int *level_joints, *level_start, *depth;
};
__device__ __forceinline__ PoseTabs bf_pose_tabs(const FitTab &T) {
    return PoseTabs{T.th_kind, T.th_off, T.parents, T.pose_mean, T.hand_comp, T.Jd, T.Jt, T.nb, T.level_joints, T.level_start, T.depth};
}

// The outputs of bf_pose_state_body from what it left in `lds` (R, J, GR, Gt): GR, A_j translations, Gt, the pose feature, betas
// and the similarity.  Separate so that a caller whose body ran on one wave can write the record with all its threads.
struct PoseNoHook { __device__ __forceinline__ void operator()(const float *) const {} };
template <bool PACKED>
__device__ __forceinline__ void bf_pose_state_emit(const FitTab &T, const float *__restrict__ sim, float *state, const float *__restrict__ packed,
                                                   const float *__restrict__ cscale, float cscale_all, const int f, const int tid, const int nt,
                                                   const float *lds, const float *__restrict__ betas, const float *packed_lds, const bool skip_feat = false) {
    const float *R = lds, *J = R + 64 * 9, *GR = J + 64 * 3, *Gt = GR + 64 * 9;
    const int nj = T.nj, nb = T.nb, npf = T.npf;
    const float *pk = PACKED ? (packed_lds ? packed_lds : packed + (size_t)f * T.np) : nullptr;
    const float *beta;
    if constexpr (PACKED) beta = pk + T.off_beta; else beta = betas + (size_t)f * nb;
    StateView st = bf_state_view(state + (size_t)f * bf_state_stride(nj, npf, nb), nj, npf, nb);
    for (int i = tid; i < nj * 9; i += nt) st.GR[i] = GR[i];
    for (int i = tid; i < nj * 3; i += nt) {
        int j = i / 3, a = i % 3;
        const float *g = GR + j * 9 + a * 3;
        st.At[i] = Gt[i] - (g[0] * J[j * 3] + g[1] * J[j * 3 + 1] + g[2] * J[j * 3 + 2]);
        st.Gt[i] = Gt[i];
    }
    if (!skip_feat)                  // (the resident dense launch published the pose feature early: bf_pose_state_body's hook)
    for (int p = tid; p < npf; p += nt) {
        int j = 1 + p / 9, e = p % 9;
        st.feat[p] = R[j * 9 + e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
    }
    if (tid < nb) st.beta[tid] = beta[tid];
    if constexpr (PACKED) {
        if (tid < 3) st.t[tid] = pk[tid];
        if (tid == 3) { st.sc[0] = pk[3]; st.sc[1] = cscale ? cscale[f] : cscale_all; }
    } else {
        if (tid < 3) st.t[tid] = sim ? sim[(size_t)f * 5 + tid] : 0.f;
        if (tid == 3) { st.sc[0] = sim ? sim[(size_t)f * 5 + 3] : 1.f; st.sc[1] = sim ? sim[(size_t)f * 5 + 4] : 1.f; }
    }
}

// One 128-thread workgroup per parameter set.  betas[n][nb], orient[n][3], body_pose[n][3(nj-1)];
// transl / scale are taken as (0,0,0) / 1 / 1 when `sim` is null, else sim[n][5] = t, s, c.
// `packed` != null: read everything from the optimiser-order parameter block packed[n][np] instead
// (transl, scale from it; constant scale from cscale[n] or cscale_all).
// Called by EVERY thread of the workgroup (it synchronises); `nt` = the workgroup's thread count.
// WAVE: the caller is ONE wavefront (tid = lane, nt = 64) with `lds` to itself: barriers become wave fences (a wave's LDS
// operations execute in order), everything else - the arithmetic included - is the same code.
// `after_rotations(R)`: called by every thread once the rotations R[nj][9] are complete in LDS (behind the first fence / barrier).
template <bool PACKED, bool WAVE = false, bool EMIT = true, class HOOK = PoseNoHook, bool CHAIN = true>
__device__ __forceinline__ void bf_pose_state_body(const FitTab &T, const float *__restrict__ betas, const float *__restrict__ orient,
                                                   const float *__restrict__ body_pose, const float *__restrict__ sim, float *state,
                                                   const float *__restrict__ packed, const float *__restrict__ cscale, float cscale_all,
                                                   const int f, const int tid, const int nt, float *lds, const PoseTabs P,
                                                   const float *packed_lds = nullptr, const float *J_pre = nullptr, HOOK after_rotations = HOOK()) {
    // lds: BF_POSE_STATE_LDS floats of workgroup-shared scratch
#ifdef BF_STAMP
    long long *bf_marks = (long long *)(lds + 1740);
#define BF_PMARK(k) do { if (WAVE && tid == 0) bf_marks[k] = clock64(); } while (0)
#else
#define BF_PMARK(k) do { } while (0)
#endif
    BF_PMARK(0);
    float *R = lds, *J = R + 64 * 9, *GR = J + 64 * 3, *Gt = GR + 64 * 9;
    int *s_ls = (int *)(Gt + 64 * 3), *s_lj = s_ls + 66, *s_par = s_lj + 64;   // tree levels and parents: read once, not once per level (dependent global loads)
    const int nj = T.nj, nb = T.nb, npf = T.npf;
    if (tid < nj) { s_lj[tid] = P.level_joints[tid]; s_par[tid] = P.parents[tid]; }
    if (tid <= T.n_levels && tid < 66) s_ls[tid] = P.level_start[tid];
    // (packed_lds: this frame's parameter block, already in LDS)
    const float *pk = PACKED ? (packed_lds ? packed_lds : packed + (size_t)f * T.np) : nullptr;
    const float *beta;
    if constexpr (PACKED) beta = pk + T.off_beta; else beta = betas + (size_t)f * nb;
    StateView st = bf_state_view(state + (size_t)f * bf_state_stride(nj, npf, nb), nj, npf, nb);
    if (tid < nj) {
        float th[3];
        if constexpr (PACKED) {
            bf_theta3(pk, tid, th, P.th_kind, P.th_off, P.pose_mean, P.hand_comp, T.n_pca, T.off_lh, T.off_rh);
        } else {
            const float *src = tid == 0 ? orient + (size_t)f * 3 : body_pose + (size_t)f * 3 * (nj - 1) + 3 * (tid - 1);
            th[0] = src[0]; th[1] = src[1]; th[2] = src[2];
        }
        BF_PMARK(1);
        m_rodrigues(th[0], th[1], th[2], R + tid * 9);
        BF_PMARK(2);
        st.theta[tid * 3] = th[0]; st.theta[tid * 3 + 1] = th[1]; st.theta[tid * 3 + 2] = th[2];
    }
    // Rest joints J = Jt + (Jd . beta, the products added l ascending from 0): the fit kernel's table-driven beta_dependent
    // (fit_kernels.hip) forms its own J with exactly this arithmetic, so a caller that already holds those values (J_pre: the resident
    // dense-schedule launch, whose wave 3 formed them right behind the betas' Adam step) hands them over instead of repeating
    // 165 ten-term sums on the critical wave of every iteration - same bits either way.
    if (J_pre) {
        for (int i = tid; i < nj * 3; i += nt) J[i] = J_pre[i];
    } else if (nb <= 16) {
        // (all of a row's operands requested before the first multiply-add: the chain below is the same sum, l ascending)
        float bb[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) bb[l] = l < nb ? beta[l] : 0.f;
        for (int i = tid; i < nj * 3; i += nt) {
            float jd[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) jd[l] = l < nb ? P.Jd[i * P.jd_stride + l] : 0.f;
            const float jt = P.Jt[i];
            float acc = 0.f;
#pragma unroll
            for (int l = 0; l < 16; ++l) if (l < nb) acc += jd[l] * bb[l];
            J[i] = jt + acc;
        }
    } else
    for (int i = tid; i < nj * 3; i += nt) {
        float acc = 0.f;
        for (int l = 0; l < nb; ++l) acc += P.Jd[i * P.jd_stride + l] * beta[l];
        J[i] = P.Jt[i] + acc;
    }
    if constexpr (WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); else __syncthreads();
    BF_PMARK(3);
    after_rotations(R);
    if constexpr (!CHAIN) return;          // (the caller's other waves form the chain, row by row: bf_pose_chain_row)
    if (tid < 9) GR[tid] = R[tid];
    if (tid >= 9 && tid < 12) Gt[tid - 9] = J[tid - 9];
    if constexpr (WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); else __syncthreads();
    if constexpr (WAVE) {
        // one wave: lane = joint, all three rows of its transform when its level comes up (the parent's rows through LDS, its own
        // rotation straight from the Rodrigues registers' LDS copy, requested once) - per element the same expressions as below
        const int dep = tid < nj ? P.depth[tid] : 0;
        const int i = tid < nj ? tid : 0, p = s_par[i];
        float Ri[9], rj[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) Ri[e] = R[i * 9 + e];
#pragma unroll
        for (int e = 0; e < 3; ++e) rj[e] = J[i * 3 + e] - J[p * 3 + e];
        for (int lev = 1; lev < T.n_levels; ++lev) {
            if (tid < nj && dep == lev) {
                float g[9], gt[3];
#pragma unroll
                for (int e = 0; e < 9; ++e) g[e] = GR[p * 9 + e];
#pragma unroll
                for (int e = 0; e < 3; ++e) gt[e] = Gt[p * 3 + e];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float g0 = g[r * 3], g1 = g[r * 3 + 1], g2 = g[r * 3 + 2];
                    GR[i * 9 + r * 3 + 0] = g0 * Ri[0] + g1 * Ri[3] + g2 * Ri[6];
                    GR[i * 9 + r * 3 + 1] = g0 * Ri[1] + g1 * Ri[4] + g2 * Ri[7];
                    GR[i * 9 + r * 3 + 2] = g0 * Ri[2] + g1 * Ri[5] + g2 * Ri[8];
                    const float r0 = rj[0], r1 = rj[1], r2 = rj[2];
                    Gt[i * 3 + r] = g0 * r0 + g1 * r1 + g2 * r2 + gt[r];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    } else
    for (int lev = 1; lev < T.n_levels; ++lev) {
        int ls = s_ls[lev], cnt = (s_ls[lev + 1] - ls) * 3;
        for (int idx = tid; idx < cnt; idx += nt) {
            int i = s_lj[ls + idx / 3], r = idx % 3, p = s_par[i];
            float g0 = GR[p * 9 + r * 3], g1 = GR[p * 9 + r * 3 + 1], g2 = GR[p * 9 + r * 3 + 2];
            const float *Ri = R + i * 9;
            GR[i * 9 + r * 3 + 0] = g0 * Ri[0] + g1 * Ri[3] + g2 * Ri[6];
            GR[i * 9 + r * 3 + 1] = g0 * Ri[1] + g1 * Ri[4] + g2 * Ri[7];
            GR[i * 9 + r * 3 + 2] = g0 * Ri[2] + g1 * Ri[5] + g2 * Ri[8];
            float r0 = J[i * 3] - J[p * 3], r1 = J[i * 3 + 1] - J[p * 3 + 1], r2 = J[i * 3 + 2] - J[p * 3 + 2];
            Gt[i * 3 + r] = g0 * r0 + g1 * r1 + g2 * r2 + Gt[p * 3 + r];
        }
        if constexpr (WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); else __syncthreads();
    }
    BF_PMARK(4);
    if constexpr (EMIT) bf_pose_state_emit<PACKED>(T, sim, state, packed, cscale, cscale_all, f, tid, nt, lds, betas, packed_lds);
}

// Row r of every joint's chain transform, by ONE wave (lane = joint), from the rotations R and rest joints J that another wave left
// in `lds` (bf_pose_state_body<.., CHAIN = false>): row r of G_i depends on row r of G_parent only, so three waves form the three
// rows side by side without ever reading each other's results.  Per element the expressions of bf_pose_state_body's own chain.
__device__ __forceinline__ void bf_pose_chain_row(const int nj, const int n_levels, const int r, const int lane, float *lds,
                                                  const int *parents, const int *depth) {
    const float *R = lds, *J = R + 64 * 9;
    float *GR = lds + 64 * 9 + 64 * 3, *Gt = GR + 64 * 9;
    const int i = lane < nj ? lane : 0, p = i > 0 ? parents[i] : 0;
    const int dep = lane < nj ? depth[i] : 0;
    float Ri[9], rj[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) Ri[e] = R[i * 9 + e];
#pragma unroll
    for (int e = 0; e < 3; ++e) rj[e] = J[i * 3 + e] - J[p * 3 + e];
    if (lane == 0) {
        GR[r * 3] = Ri[r * 3]; GR[r * 3 + 1] = Ri[r * 3 + 1]; GR[r * 3 + 2] = Ri[r * 3 + 2];
        Gt[r] = J[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int lev = 1; lev < n_levels; ++lev) {
        if (lane < nj && dep == lev) {
            const float g0 = GR[p * 9 + r * 3], g1 = GR[p * 9 + r * 3 + 1], g2 = GR[p * 9 + r * 3 + 2], gt = Gt[p * 3 + r];
            GR[i * 9 + r * 3 + 0] = g0 * Ri[0] + g1 * Ri[3] + g2 * Ri[6];
            GR[i * 9 + r * 3 + 1] = g0 * Ri[1] + g1 * Ri[4] + g2 * Ri[7];
            GR[i * 9 + r * 3 + 2] = g0 * Ri[2] + g1 * Ri[5] + g2 * Ri[8];
            const float r0 = rj[0], r1 = rj[1], r2 = rj[2];
            Gt[i * 3 + r] = g0 * r0 + g1 * r1 + g2 * r2 + gt;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}
