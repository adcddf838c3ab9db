// Persistent per-frame SMPLify fit kernel for gfx950 (MI355X).
//
// One 512-thread workgroup (8 wave64, two per SIMD) owns one frame for ALL Adam iterations of
// reference smplify/smplify.py:177-213: parameters, kinematic chain, gradients and Adam moments stay
// in LDS / registers for the whole fit; the only HBM traffic is the one-off table load (~190 KB, L2
// resident across frames) and the final write-back.  No kernel boundary, host sync or host->device
// copy inside the loop (the reference pays 48 keypoint copies + 4 syncs per iteration, loss.py:160,219).
//
// Work per iteration (keypoint-only objective, loss.py:139-230): only the first 25 of the 49 joints
// enter the loss (loss.py:163) = 14 chain joints + 11 selector vertices, so an iteration evaluates
// Rodrigues x NJ (smplx quirk angle=||theta+1e-8||), J(beta) from the pre-contracted regressor, the
// kinematic chain, LBS of the selector vertices only (their 33 posedirs columns live in LDS), the
// 48-view projection + GMoF, the merged 8x69x69 GMM prior, the angle / shape priors, the exact
// reverse sweep of all of it, and the torch-semantics Adam update.  Derivation: oracle/analytic.py.
//
// The kernel is bound by instruction issue on its critical waves and by the LDS round trips between its phases, so the
// design keeps the number of workgroup barriers at 5 per iteration, every index in registers, and deals the work out by
// instruction count:
//   waves 0-2  chain specialists: wave r carries matrix row r of every joint (row r of G_i depends only on row r of G_parent),
//              lane = joint; the local transforms [R_j | rel_j] are in LDS when the iteration starts (the rotation formed by the
//              lane that stepped the joint's dofs, rel_j by wave 3 with the betas), two tree levels per round through ds_bpermute;
//              then skinning of the selector vertices, the 48-view projection, the reverse skinning + subtree sums, and on wave 0
//              the reverse Rodrigues + Adam for the pose;
//   wave 3     skinning / projection like the others; the betas' gradient, step and everything the next forward pass derives
//              from them; the priors' share of dL/dtheta (arg-min component, angle prior) for the Adam phase;
//   waves 4-7  GMM specialists: the symmetrised precision rows of components 2g, 2g+1 live in VGPRs for the whole launch; they
//              also carry what needs few registers and only LDS inputs: the pose blend of the selector vertices (phase A) and
//              d(pose feature) (phase F).
// (That is the compile-time-sized SMPL instance.  Other models - and SMPL-X in the dense schedule, which has its own sized
// instance - take the table-driven phases: Rodrigues in phase A on the chain waves, a two-phase pose blend over all eight waves,
// masked subtree sums, Adam one parameter per thread.)
// The reverse chain is flattened: with t_i = sum over subtree(i) of dL/dGt and
// N_i = D_i GR_i^T + t_i (Gt_i - Gt_parent)^T, dL/dGR_p = (D_p GR_p^T + sum_{i in strict subtree} N_i) GR_p,
// i.e. two subtree sums instead of one barrier per tree level.  All reductions have a fixed order.
#include "bf_internal.h"
#include <hip/hip_ext.h>
#include "pose_state_body.h"
#include <type_traits>

// Diagnostic build only (-DBF_STAMP, libbodyfit_stamp.so; the product library has no stamps).  Round 5: the stamps are LIGHT - EVERY
// wave stores the low word of the shader clock as it arrives at a barrier and as it leaves it, every iteration (the last one
// stays), raw, into S.stamp[64 + (2 b + {0: arrive, 1: leave}) * 8 + wave] (BF_T(k): slot k, k < 24; 22 / 23 = the tops of the last
// two iterations; 12 .. 21 free for marks inside a phase); the differences are taken on the host (tests/gpu_stamps.py).  The earlier
// form (64-bit differences converted to float under `it == 2`, extra marks inside the phases) cost the stamp build 352 B of scratch
// per lane and 60 % more cycles per iteration than the product build: it measured itself.  (Branch-free - every lane stores the same
// word to the same slot, the wave index in a scalar register: an `if (lane == 0)` around the store split the scheduling regions
// of the GMM loop and brought 250 B of the spills back.  The stamp array stays at 64 + 192 floats: at 64 + 256 the compiler
// emitted "Illegal instruction detected: Operand has incorrect register class".)
#ifdef BF_STAMP
#define BF_T(k) do { S.stamp[64 + (k) * 8 + bf_wave_s] = __int_as_float((int)clock64()); } while (0)
#define BF_SYNC() do { BF_T(2 * sidx); __syncthreads(); BF_T(2 * sidx + 1); ++sidx; } while (0)
#else
#define BF_T(k) do { } while (0)
#define BF_SYNC() __syncthreads()
#endif
#if defined(BF_STAMP) && defined(BF_STAMP_MARKS)
// (make stamp STAMPFLAGS=-DBF_STAMP_MARKS: the marks 50 .. 54 inside the Adam phase - wave 0's Rodrigues reverse / Adam, wave 3's
//  geometric d(beta) / Adam / beta-dependent tables - go to the free slots 12 .. 16; they perturb the build a little)
#define BF_MARK(k, who, itv, t0v) do { (void)(itv); (void)(t0v); if ((k) >= 50 && (k) <= 54) BF_T(12 + (k) - 50); } while (0)
#else
#define BF_MARK(k, who, itv, t0v) do { (void)(itv); (void)(t0v); } while (0)
#endif
// (stamp build) cycles since kernel entry at a few points of a dense-schedule launch: thread `who` of frame 0 -> io.debug[4160 + k]
#ifdef BF_STAMP
#define BF_KMARK(k, who) do { if (EXT && tid == (who) && frame == 0 && io.debug) io.debug[4160 + (k)] = (float)(long long)(clock64() - bf_k0); } while (0)
#else
#define BF_KMARK(k, who) do { } while (0)
#endif
// orders this wave's LDS traffic for the compiler; the hardware executes a wave's LDS ops in order
#define BF_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

namespace {

struct FitSmem {
    float *pa, *pb, *R, *L, *rc, *J, *G, *At, *vs, *vp, *TR, *vsel, *part, *dvsel, *dvp;
    float *dGR, *dGt, *tt, *N, *dAt, *dJ, *dR, *drel, *dfeat, *gth, *g, *gd, *gy, *gq, *gtail, *scal, *feat, *vpp;
    float *Jtrel, *Dg, *Jt, *Jd, *Jdrel, *rel, *sel_vt, *sel_sd, *sel_pd, *sel_w, *means, *proj, *nzw, *theta, *pmean, *hcomp, *kp, *stamp, *sel_pd2;
    int *nzj, *thk, *tho, *par, *pk, *pa_, *pb_;
    float *am, *av;
    float *ext;          // the dense schedule's outside gradient blocks of this frame (EXT launches)
    int *lvl;            // level_joints[nj] | level_start[n_levels + 1] (<= 67) | depth[nj], for the pose state a dense launch publishes
};

__host__ __device__ inline int pad4(int n) { return (n + 3) & ~3; }
#define BF_PDT_LD 210      // row stride of the transposed selector posedirs: 7 slices of 30 rows

// Carve the dynamic LDS segment; the same function sizes it on the host (base == nullptr).
// (`placed`: false = sizes only.  Said by the caller, NOT read off `base`: in the kernel `base` is the LDS segment, a generic pointer whose
//  test against null the compiler sometimes failed to fold - "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base", a
//  compare of the segment's aperture with zero in a form the instruction set does not have - and sometimes not, with unrelated edits
//  deciding which: the defect that cost round 5 six builds before `llc` on the whole module, not on one kernel, reproduced it.)
__host__ __device__ inline size_t fit_smem_carve(FitSmem &s, float *base, int nj, int nb, int npf, int ns,
                                                  int nl, int np, int nviews, bool placed = true) {
    size_t o = 0;
    auto take = [&](int n) { float *p = placed ? base + o : nullptr; o += pad4(n); return p; };
    // arrays whose size follows from (nj, nb, npf, ns) first: in the compile-time-sized instance their offsets are
    // immediates of the ds instructions; the ones sized by np (a launch value) and by the number of views come last
    s.R = take(nj * 9);    s.rc = take(nj * 4);   s.J = take(nj * 3);
    s.L = take(nj * 12);   // local transforms [R_j | rel_j], row-major 3 x 4: what a chain lane reads at the top of phase A (3 x b128)
    s.G = take(nj * 12);   s.At = take(nj * 3);
    s.vs = take(ns * 3);   s.vp = take(ns * 3);   s.TR = take(ns * 9);   s.vsel = take(ns * 3);
    s.part = take(BF_VSUB * 32 * 4);
    s.dvsel = take(ns * 3); s.dvp = take(ns * 3);
    s.dGR = take(nj * 12); s.dGt = take(nj * 3);  s.tt = take(nj * 3);   s.N = take(nj * 12);
    s.dAt = take(nj * 3);  s.dJ = take(nj * 3);
    s.dR = take(nj * 9);   s.drel = take(nj * 3); s.dfeat = take(npf);   s.gth = take(nj * 3);
    s.gd = take(BF_GMM_M * BF_GMM_LD); s.gy = take(BF_GMM_M * BF_GMM_LD);
    s.gq = take(BF_GMM_M);             s.gtail = take(256);   s.scal = take(8);
    s.feat = take(npf);                s.vpp = take(BF_FIT_THREADS + ns * 3);
    s.Jtrel = take(nj * 3); s.Dg = take(nj * 12);
    s.Jt = take(nj * 3);   s.Jd = take(nj * 3 * pad4(nb + 1));      s.Jdrel = take(nj * 3 * pad4(nb + 1));
    s.rel = take(nj * 3);
    s.sel_vt = take(ns * 3); s.sel_sd = take(ns * 3 * pad4(nb + 1)); s.sel_pd = take(npf * pad4(ns * 3));       // rows padded to float4s (the reverse pass reads a row as b128s)
    s.sel_w = take(ns * nj); s.means = take(BF_GMM_M * BF_GMM_LD);
    s.nzw = take(ns * BF_SEL_NNZ);  s.nzj = (int *)take(ns * BF_SEL_NNZ);
    s.kp = take(BF_VSUB * BF_KP_ROUNDS * 16 * 8);      // 8-float keypoint record per (view, loss-joint pair), zero padded
    s.stamp = take(64 + 192);
    s.sel_pd2 = take(npf * (ns * 3 + 1));     // sel_pd TRANSPOSED for the GMM waves' pose blend: [3 ns][BF_PDT_LD] (a lane's 30-row slice
                                              // of one output is contiguous: b64 reads), zero padded; fits: 3 ns * 210 <= npf * (3 ns + 1)
    s.theta = take(nj * 3);  s.pmean = take(nj * 3);  s.hcomp = take(2 * 6 * 45);
    s.thk = (int *)take(nj); s.tho = (int *)take(nj); s.par = (int *)take(nj);
    s.pa = take(np);       s.pb = take(np);   s.g = take(np);
    s.pk = (int *)take(np); s.pa_ = (int *)take(np); s.pb_ = (int *)take(np); s.am = take(np); s.av = take(np);
    s.proj = take(nviews * 12);
    s.ext = take(npf + nj * 12 + nb + 4 + nj * 3 + 4);
    s.lvl = (int *)take(2 * nj + 68);
    (void)nl;
    return o * sizeof(float);
}

// Cross-lane sums on the VALU's DPP path (a few cycles each) instead of ds_bpermute (an LDS round trip each):
//   quad_perm [1,0,3,2] / [2,3,0,1]  = xor 1 / xor 2 inside a quad; row_ror 4 / 8 rotate inside a 16-lane row.
template <int CTRL>
__device__ inline float dpp_add(float v) {
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(r);
}
__device__ inline int bf_launder(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ inline float quad_sum(float v) { v = dpp_add<0xB1>(v); return dpp_add<0x4E>(v); }        // all 4 lanes of a quad
__device__ inline float half8_sum(float v) { v = quad_sum(v); return dpp_add<0x141>(v); }             // 8 lanes: + row_half_mirror
__device__ inline float row16_sum(float v) { v = quad_sum(v); v = dpp_add<0x124>(v); return dpp_add<0x128>(v); }   // all 16 lanes of a row
// every lane gets the wave's total, added as ((row0 + row1) + (row2 + row3)): fixed order
__device__ inline float wave_sum(float v) {
    v = row16_sum(v);
    float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}

// sin and cos of a non-negative angle of moderate size (|a| < ~1e4): Cody-Waite reduction by pi/2 in three
// parts, then the classic degree-7 / degree-8 minimax kernels on [-pi/4, pi/4]; about 1 ulp, ~30 instructions
// (the OCML sinf + cosf pair costs several hundred cycles on the critical path of every iteration)
__device__ inline float bf_launder_f(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ inline void sincos_small(float a, float *sn, float *cs) {
    const float n = rintf(a * 0.636619772367581343f);
    float r = fmaf(n, -1.57079625129699707031e+00f, a);
    r = fmaf(n, -7.54978941586159635335e-08f, r);
    r = fmaf(n, -5.39030285815811905290e-15f, r);
    const float z = r * r;
    // (the leading coefficients are materialised here each time: hoisted out of the persistent loop as a VGPR pair for
    //  the packed fma they end up spilled, and the reload sits on the critical path of every iteration)
    const float s4 = bf_launder_f(2.7557314297e-06f), c4 = bf_launder_f(-2.7557314297e-07f);
    const float ps = fmaf(z, fmaf(z, fmaf(z, s4, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
    const float sr = fmaf(r * z, ps, r);
    const float pc = fmaf(z, fmaf(z, fmaf(z, c4, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f);
    const float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int q = (int)n & 3;
    const float s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// smplx batch_rodrigues for one joint: angle a = |theta + 1e-8|, axis n = theta / a,
// R = I + sin(a) K(n) + (1 - cos(a)) K(n)^2.
// rc = (sinc, cosc, d sinc / du, d cosc / du) at u = a^2, with sinc = sin a / a and cosc = (1 - cos a) / a^2: what the reverse pass
// needs, in the form in which it needs no division (rodrigues_bwd).  [Measured: forming R itself from Taylor series of the four
// functions - no sqrt, rcp or range reduction on the way - shortens the dependent chain but costs 46 fused multiply-adds per lane, and
// the phase got 350 cycles LONGER; only the reverse pass keeps the division-free form.]
__device__ inline void rodrigues_fwd(float tx, float ty, float tz, float *R, float *rc) {
    float ux = tx + 1e-8f, uy = ty + 1e-8f, uz = tz + 1e-8f;
    // (v_sqrt_f32 / v_rcp_f32, 1 ulp each: the correctly rounded forms cost ~25 more instructions on the critical path)
    const float u = ux * ux + uy * uy + uz * uz;
    float a = __builtin_amdgcn_sqrtf(u);
    float ia = __builtin_amdgcn_rcpf(a);
    float nx = tx * ia, ny = ty * ia, nz = tz * ia;
    float s, c;
    sincos_small(a, &s, &c);
    float oc = 1.0f - c;
    R[0] = 1.0f + oc * (-nz * nz - ny * ny);
    R[1] = s * (-nz) + oc * (nx * ny);
    R[2] = s * ny + oc * (nx * nz);
    R[3] = s * nz + oc * (nx * ny);
    R[4] = 1.0f + oc * (-nz * nz - nx * nx);
    R[5] = s * (-nx) + oc * (ny * nz);
    R[6] = s * (-ny) + oc * (nx * nz);
    R[7] = s * nx + oc * (ny * nz);
    R[8] = 1.0f + oc * (-ny * ny - nx * nx);
    // (off the chain's critical path: stored by one wave, read in the reverse sweep)
    const float iu = ia * ia, A = s * ia, B = oc * iu;
    const bool tiny = u < 1.0e-3f;          // (c - A) and (A - 2 B) cancel there: two Taylor terms are exact to float32
    rc[0] = A; rc[1] = B;
    rc[2] = tiny ? fmaf(u, 1.6666667e-02f, -1.6666667e-01f) : (c - A) * (0.5f * iu);
    rc[3] = tiny ? fmaf(u, 2.7777778e-03f, -4.1666667e-02f) : (A - 2.0f * B) * (0.5f * iu);
}

// reverse of rodrigues_fwd: G = dL/dR (row-major 3x3) -> dL/dtheta.
//   dL/dtheta_k = sinc <G, dX/dtheta_k> + cosc <G, dY/dtheta_k> + (sinc' <G, X> + cosc' <G, Y>) du/dtheta_k
// where R = I + sinc X + cosc Y, X = [theta]x, Y = theta theta^T - |theta|^2 I, u = |theta + 1e-8|^2 (the same matrix as the forward's),
// with <G, dX/dtheta> = (G21 - G12, G02 - G20, G10 - G01), <G, dY/dtheta> = (G + G^T) theta - 2 tr(G) theta,
// <G, X> = theta . <G, dX/dtheta>, <G, Y> = theta^T G theta - |theta|^2 tr(G), du/dtheta = 2 (theta + 1e-8).  No division anywhere:
// theta = 0 (where the reference's own formula has angle = |1e-8|) is an ordinary point.
__device__ inline void rodrigues_bwd(float tx, float ty, float tz, const float *rc, const float *G, float *gth) {
    const float A = rc[0], B = rc[1], dA = rc[2], dB = rc[3];
    const float x0 = G[7] - G[5], x1 = G[2] - G[6], x2 = G[3] - G[1];
    const float tr = G[0] + G[4] + G[8];
    const float s0 = (G[0] + G[0]) * tx + (G[1] + G[3]) * ty + (G[2] + G[6]) * tz;
    const float s1 = (G[3] + G[1]) * tx + (G[4] + G[4]) * ty + (G[5] + G[7]) * tz;
    const float s2 = (G[6] + G[2]) * tx + (G[7] + G[5]) * ty + (G[8] + G[8]) * tz;
    const float tt = tx * tx + ty * ty + tz * tz;
    const float gX = tx * x0 + ty * x1 + tz * x2;
    const float gY = 0.5f * (tx * s0 + ty * s1 + tz * s2) - tt * tr;
    const float k = 2.0f * (dA * gX + dB * gY);
    const float tr2 = tr + tr;
    gth[0] = A * x0 + B * (s0 - tr2 * tx) + k * (tx + 1e-8f);
    gth[1] = A * x1 + B * (s1 - tr2 * ty) + k * (ty + 1e-8f);
    gth[2] = A * x2 + B * (s2 - tr2 * tz) + k * (tz + 1e-8f);
}


#define GR_(j, r, c) S.G[((j) * 3 + (r)) * 4 + (c)]
#define GT_(j, r) S.G[((j) * 3 + (r)) * 4 + 3]

}  // namespace


__device__ inline void copy_f(float *dst, const float *src, int n, int tid, int nt) {
    for (int i = tid; i < n; i += nt) dst[i] = src[i];
}
__device__ inline void copy_i(int *dst, const int *src, int n, int tid, int nt) {
    for (int i = tid; i < n; i += nt) dst[i] = src[i];
}


// mode: 0 = fit (n_iters Adam steps), 1 = one loss/gradient evaluation, no update.
// adam_tab[it] = {lr_transl_scale / bc1, lr / bc1, sqrt(bc2)} evaluated in double on the host exactly
// as torch's single-tensor Adam does.  NJ/NB/NS/NL > 0 fix the sizes at compile time
// (SMPL: 24, 10, 11, 25) so every inner loop unrolls; 0 = take them from the tables.
template <int NJ, int NB, int NS, int NL, bool EXT>
__global__ void __launch_bounds__(BF_FIT_THREADS)
fit_kernel(FitTab T, FrameIO io, HyperDev hp, int n_iters, int mode, const float *__restrict__ adam_tab, int adam_t0) {
    extern __shared__ __align__(16) float smem_raw[];
#ifdef BF_STAMP
    const long long bf_k0 = clock64();
#endif
    const int tid = threadIdx.x, nt = BF_FIT_THREADS;
#ifdef BF_STAMP
    const int bf_wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x) >> 6;        // (the wave index in a scalar register, for the stamps)
#endif
    constexpr int NG = 256;                    // threads of the geometry waves (0-3)
    const int lane = tid & 63, wave = tid >> 6;
    const int frame = blockIdx.x;
    const int nj = NJ ? NJ : T.nj, nb = NB ? NB : T.nb, ns = NS ? NS : T.ns, nl = NL ? NL : T.nl;
    const int npf = 9 * (nj - 1), np = T.np, V = io.n_views, ns3 = ns * 3, nj3 = nj * 3;
    FitSmem S;
    fit_smem_carve(S, smem_raw, nj, nb, npf, ns, nl, np, V);
    // persistent form of the same launch (io.door): all dense iterations of a call in ONE launch, paced by doorbells
    int *const door = EXT ? io.door : nullptr;
    if (EXT && door && io.door_resident && tid == 0) __hip_atomic_fetch_add(io.door_resident, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

    // ---- one-off loads --------------------------------------------------------------------
    const int nbp = pad4(nb + 1);
    const int sel_nnz = T.sel_nnz;
    // Dense-schedule launches run ONE iteration, so their prologue is on the critical path of every iteration (and a sparse-schedule
    // launch of 100 iterations still spends 2 % of its time in it).  Once a first
    // launch has left an image of the model-constant LDS arrays (three contiguous runs of the carve, FitTab::img_seg), everything
    // such a launch needs from global memory is issued back to back into registers - image, projection matrices, parameters,
    // Adam moments, the outside gradient blocks, the tree levels - and stored to LDS after ONE round trip, instead of two dozen
    // dependent load -> store loops at ~2,500 cycles each (a kernel starts with cold caches).
    const int n_ext = npf + nj * 12 + nb + 4 + nj * 3 + 4;
    const bool use_image = T.lds_image != nullptr && mode != 2 && V * 3 <= nt && np <= nt && n_ext <= 3 * nt &&
                           T.img_seg[0][1] + T.img_seg[1][1] + T.img_seg[2][1] <= 4 * nt && nj <= nt && T.n_levels < 66;
    if (use_image) {
        const float4 *img = (const float4 *)__builtin_assume_aligned(T.lds_image, 16);
        float4 *lds4 = (float4 *)__builtin_assume_aligned(smem_raw, 16);
        const int c0 = T.img_seg[0][1], c1 = c0 + T.img_seg[1][1], c2 = c1 + T.img_seg[2][1];
        int at[4];
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = q * nt + tid;           // flat index over the three runs -> float4 index of the LDS segment
            at[q] = i < c0 ? T.img_seg[0][0] + i : (i < c1 ? T.img_seg[1][0] + (i - c0) : (i < c2 ? T.img_seg[2][0] + (i - c1) : -1));
            if (at[q] >= 0) v[q] = img[at[q]];
        }
        const float4 *pj = (const float4 *)__builtin_assume_aligned(io.proj + (size_t)frame * V * 12, 16);
        float4 vp = {0.f, 0.f, 0.f, 0.f};
        if (tid < V * 3) vp = pj[tid];
        float r_pa = 0.f, r_am = 0.f, r_av = 0.f, r_ext[3] = {0.f, 0.f, 0.f};
        int r_lj = 0, r_ls = 0, r_dp = 0;
        if (tid < np) {
            r_pa = (io.params0 ? io.params0 : io.params)[(size_t)frame * np + tid];
            if (!io.params0) { r_am = io.adam_m[(size_t)frame * np + tid]; r_av = io.adam_v[(size_t)frame * np + tid]; }
        }
        if (EXT) {
            const float *eg = io.ext + (size_t)frame * n_ext;
#pragma unroll
            for (int q = 0; q < 3; ++q) if (!door && q * nt + tid < n_ext) r_ext[q] = eg[q * nt + tid];
            if (tid < nj) { r_lj = T.level_joints[tid]; r_dp = T.depth[tid]; }
            if (tid <= T.n_levels) r_ls = T.level_start[tid];
        }
        // zero-initialised arrays: GMM d / y, and the keypoint table when this launch has no loss joints
        for (int i = tid; i < 2 * BF_GMM_M * BF_GMM_LD; i += nt) S.gd[i] = 0.f;          // (gd and gy are adjacent in the carve)
        if (nl == 0) for (int i = tid; i < BF_VSUB * BF_KP_ROUNDS * 16 * 2; i += nt) ((float4 *)S.kp)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (at[q] >= 0) lds4[at[q]] = v[q];
        if (tid < V * 3) ((float4 *)S.proj)[tid] = vp;
        if (tid < np) { S.pa[tid] = r_pa; S.am[tid] = r_am; S.av[tid] = r_av; }
        if (EXT) {
#pragma unroll
            for (int q = 0; q < 3; ++q) if (q * nt + tid < n_ext) S.ext[q * nt + tid] = r_ext[q];
            if (tid < nj) { S.lvl[tid] = r_lj; S.lvl[nj + 67 + tid] = r_dp; }
            if (tid <= T.n_levels) S.lvl[nj + tid] = r_ls;
        }
    } else {
    copy_f(S.Jt, T.Jt, nj3, tid, nt);
    copy_f(S.Jtrel, T.Jtrel, nj3, tid, nt);
    // beta tables with rows padded to float4s (stride nbp); column nb carries the constant term (it meets a 1)
    for (int i = tid; i < nj3 * nbp; i += nt) {
        const int r = i / nbp, l = i - r * nbp;
        S.Jd[i] = l < nb ? T.Jd[r * nb + l] : (l == nb ? T.Jt[r] : 0.f);
        S.Jdrel[i] = l < nb ? T.Jdrel[r * nb + l] : (l == nb ? T.Jtrel[r] : 0.f);
    }
    for (int i = tid; i < ns3 * nbp; i += nt) {
        const int r = i / nbp, l = i - r * nbp;
        S.sel_sd[i] = l < nb ? T.sel_sd[r * nb + l] : (l == nb ? T.sel_vt[r] : 0.f);
    }
    copy_f(S.sel_vt, T.sel_vt, ns3, tid, nt);
    for (int i = tid; i < npf * pad4(ns3); i += nt) { const int pp = i / pad4(ns3), o = i - pp * pad4(ns3); S.sel_pd[i] = o < ns3 ? T.sel_pd[pp * ns3 + o] : 0.f; }
    if (ns3 * BF_PDT_LD <= npf * (ns3 + 1))
        for (int i = tid; i < ns3 * BF_PDT_LD; i += nt) { const int o = i / BF_PDT_LD, p = i - o * BF_PDT_LD; S.sel_pd2[i] = p < npf ? T.sel_pd[p * ns3 + o] : 0.f; }
    copy_f(S.sel_w, T.sel_w, ns * nj, tid, nt);
    for (int i = tid; i < ns * BF_SEL_NNZ; i += nt) { S.nzw[i] = T.sel_nzw[i]; S.nzj[i] = T.sel_nzj[i]; }
    for (int i = tid; i < nj; i += nt) { S.thk[i] = T.th_kind[i]; S.tho[i] = T.th_off[i]; S.par[i] = i > 0 ? T.parents[i] : 0; }
    for (int i = tid; i < nj * 3; i += nt) S.pmean[i] = T.pose_mean ? T.pose_mean[i] : 0.f;
    for (int i = tid; i < 2 * T.n_pca * 45 && i < 2 * 6 * 45; i += nt) S.hcomp[i] = T.hand_comp[i];
    for (int i = tid; i < BF_GMM_M * BF_GMM_LD; i += nt) {
        int m = i / BF_GMM_LD, j = i % BF_GMM_LD;
        S.means[i] = j < BF_GMM_D ? T.g_means[m * BF_GMM_D + j] : 0.f;
        S.gd[i] = 0.f;
        S.gy[i] = 0.f;
    }
    copy_f(S.proj, io.proj + (size_t)frame * V * 12, V * 12, tid, nt);
    copy_f(S.pa, (io.params0 ? io.params0 : io.params) + (size_t)frame * np, np, tid, nt);
    if (EXT) {
        if (mode != 2 && !door) copy_f(S.ext, io.ext + (size_t)frame * n_ext, n_ext, tid, nt);
        for (int i = tid; i < nj; i += nt) { S.lvl[i] = T.level_joints[i]; S.lvl[nj + 67 + i] = T.depth[i]; }
        for (int i = tid; i <= T.n_levels && i < 67; i += nt) S.lvl[nj + i] = T.level_start[i];
    }
    }
    float *Pcur = S.pa, *Pnext = S.pb;
    BF_KMARK(8, 0);

    // chain-row role (waves 0-2): lane = joint
    const bool cw_on = wave < 3 && lane < nj;
    const int wj = cw_on ? lane : 0;
    const int wp = wj > 0 ? T.parents[wj] : 0;
    const int wd = cw_on ? T.depth[wj] : -1;
    const int w_gp = wp > 0 ? T.parents[wp] : 0;          // grandparent (root for the first two levels)
    const int w_feat = (wj > 0 ? wj - 1 : 0) * 9;          // (non-negative base: the nine stores share one address register)
    const int w_kind = T.th_kind[wj], w_off = T.th_off[wj];
    const float w_pm0 = (NJ != 24 && T.pose_mean) ? T.pose_mean[wj * 3] : 0.f, w_pm1 = (NJ != 24 && T.pose_mean) ? T.pose_mean[wj * 3 + 1] : 0.f,
                w_pm2 = (NJ != 24 && T.pose_mean) ? T.pose_mean[wj * 3 + 2] : 0.f;      // (SMPL has no pose mean)
    // depth-first role of the merged reverse-skinning + subtree-sum phase (waves 0-2 = rows, lane = DFS position)
    const int fg_k = (wave < 3 && lane < nj) ? T.dfs_order[lane] : 0;
    const int fg_last4 = ((wave < 3 && lane < nj) ? T.dfs_last[lane] : 0) * 4;
    // (joint, row) role of the reverse sweep: tid < 3 nj
    const bool c_on = tid < nj3;
    const int ci = c_on ? tid / 3 : 0, cr = c_on ? tid - ci * 3 : 0;
    const int cp = ci > 0 ? T.parents[ci] : 0;
    const unsigned long long cmask = c_on ? T.desc[ci] : 0ull;       // strict descendants of ci
    (void)cr; (void)cp; (void)cmask; (void)w_feat;                    // (only the two-phase variants read them)
    // projection role (geometry waves 0-3 only, so the GMM waves keep their registers for the precision rows): a lane
    // owns a PAIR of loss joints (2 ps, 2 ps + 1) on one view lane, ps = 4 * wave + lane / 16, view lane = lane % 16.
    // Both joints see the same projection matrix, so every multiply-add of the projection, the GMoF and the
    // reverse pass is one packed v_pk_*_f32 over the pair; the 16 view lanes of a pair are one DPP row.
    // (MERGE_BD instances: WHICH pair is the deal of FitTab::pair_slot - the selector vertices a wave projects are the ones it skins)
    constexpr bool MERGE_BD = NJ == 24 && NB > 0 && NB <= 10 && NS > 0 && NS <= 12;          // (= GBLEND)
    const int pphys = (wave & 3) * 4 + (lane >> 4), vsub = lane & 15;      // the physical slot: wave, quarter
    const int pdealt = MERGE_BD ? T.pair_slot[pphys] : pphys;
    const int pslot = pdealt >= 0 ? pdealt : 0;
    const int ja = 2 * pslot, jb = 2 * pslot + 1;
    const bool ja_on = wave < 4 && pdealt >= 0 && ja < nl, jb_on = wave < 4 && pdealt >= 0 && jb < nl;
    const int lkind_a = ja_on ? T.lj_kind[ja] : 0, lidx_a = ja_on ? T.lj_index[ja] : 0;
    const int lkind_b = jb_on ? T.lj_kind[jb] : 0, lidx_b = jb_on ? T.lj_index[jb] : 0;
    const float *lsrc_a = lkind_a == 0 ? S.G + lidx_a * 12 + 3 : S.vsel + lidx_a * 3;
    const float *lsrc_b = lkind_b == 0 ? S.G + lidx_b * 12 + 3 : S.vsel + lidx_b * 3;
    const int lstride_a = lkind_a == 0 ? 4 : 1, lstride_b = lkind_b == 0 ? 4 : 1;
    float *ldst_a = (lkind_a == 0 ? S.dGt : S.dvsel) + lidx_a * 3;
    float *ldst_b = (lkind_b == 0 ? S.dGt : S.dvsel) + lidx_b * 3;

    // GMM role (waves 4-7): precision rows pinned in registers (lane-major host copies: coalesced)
    const bool gw = wave >= 4;
    const int gwi = gw ? wave - 4 : 0;
    const int ma = 2 * gwi, mb = 2 * gwi + 1;
    // d = theta - mu of a wave's two components sits interleaved in LDS, (d_a[j], d_b[j]) pairs, so one b128 read
    // feeds two packed FMAs (v_pk_fma_f32) with no register shuffling
    float *gdw = S.gd + gwi * 2 * BF_GMM_LD;
    const float logw_a = T.g_logw[ma], logw_b = T.g_logw[mb];
    float gd_mu[4];                           // mu_a[lane], mu_b[lane], mu_a[64 + lane], mu_b[64 + lane]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        const bool ok = gw && j < BF_GMM_D;
        // (theta_j itself: parameter T.off_pose + j for j < T.nbp, else the zero pad - smplx pads 63 -> 69, loss.py:207)
        gd_mu[2 * q] = ok ? T.g_means[ma * BF_GMM_D + j] : 0.f;
        gd_mu[2 * q + 1] = ok ? T.g_means[mb * BF_GMM_D + j] : 0.f;
    }

    BF_KMARK(10, 0);
    const float ndiv_f = (float)io.ndiv[frame];
    const float icoeff = 1.0f / hp.coeff;
    const float kscale = -1.0f / (hp.coeff * ndiv_f);
    const float s2 = hp.sigma2;
    const float cscale = io.cscale ? io.cscale[frame] : hp.cscale;
    // keypoints of the first BF_VSUB * BF_KP_ROUNDS views staged in LDS, one 8-float record per (view, joint pair):
    // (x_a, x_b, y_a, y_b | k_a, k_b, c_a, c_b) with k = conf^2 * kscale * 2 sigma^4 (gradient factor) and
    // c = conf^2 * sigma^2 (loss factor); records past V or nl are zero, so they add nothing
    const float *kp_frame = io.keypoints + (size_t)frame * V * nl * 3;
    // (no loss joints in this launch and an image: the table is the image's zeros)
    for (int i = tid; i < ((use_image && nl == 0) ? 0 : BF_VSUB * BF_KP_ROUNDS * 16); i += nt) {
        const int v = i >> 4, ps = i & 15;
        float4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
        if (v < V && 2 * ps < nl) {
            const float *k = kp_frame + ((size_t)v * nl + 2 * ps) * 3;
            const float c2 = k[2] * k[2];
            e0.x = k[0]; e0.z = k[1]; e1.x = c2 * kscale * (2.f * s2 * s2); e1.z = c2 * s2;
        }
        if (v < V && 2 * ps + 1 < nl) {
            const float *k = kp_frame + ((size_t)v * nl + 2 * ps + 1) * 3;
            const float c2 = k[2] * k[2];
            e0.y = k[0]; e0.w = k[1]; e1.y = c2 * kscale * (2.f * s2 * s2); e1.w = c2 * s2;
        }
        // (pair-major inside a round - the 16 view lanes of a pair read CONSECUTIVE records - and the two halves of a record swapped
        //  for view lanes 4-7 and 12-15: the eight lanes a b128 read serves per cycle then cover all 32 banks.  View-major, all 16
        //  lanes of a DPP row sat 512 bytes apart, on the same four banks: 22 % of the launch's LDS cycles were bank conflicts, PMC)
        const int rec = (v >> 4) * (BF_VSUB * 16) + ps * BF_VSUB + (v & 15), sw = (v >> 2) & 1;
        ((float4 *)S.kp)[2 * rec + sw] = e0;
        ((float4 *)S.kp)[2 * rec + 1 - sw] = e1;
    }
    BF_KMARK(11, 0);
    // the selector vertex this lane skins in the merged phase (lane / 12-th of the wave's deal; one register kept across the loop -
    // indexing the kernel argument per lane inside the loop would be a global load per iteration)
    const int skin_dealt = MERGE_BD ? T.skin_vert[BF_SKIN_PER_WAVE * (wave & 3) + min(lane / 12, BF_SKIN_PER_WAVE - 1)] : -1;
    // (a slot nobody was dealt reads pair 15's records: zeros - there are at most 15 pairs when a slot is empty)
    const int kp_sw = (vsub >> 2) & 1;
    const float4 *kp_lane = (const float4 *)__builtin_assume_aligned(S.kp, 16) + ((pdealt >= 0 ? pdealt : 15) * BF_VSUB + vsub) * 2 + kp_sw;
    const int kp_other = 1 - 2 * kp_sw;        // (float4s from a record's first half, as this lane finds it, to its second)
    const int EXT0 = npf, EXT_A = npf, EXT_B = npf + nj * 12, EXT_T = npf + nj * 12 + nb;
    const int EXT_G = npf + nj * 12 + nb + 4;
    const int EXT_K = EXT_G + nj * 3;          // dt, ds of the dense keypoint loss
    // (EXT = the dense schedule's per-iteration launch with outside gradient blocks; compiled out of the persistent loop)
    const float *ext = EXT ? S.ext : nullptr;          // (this frame's blocks, staged in LDS by the prologue)
    (void)EXT0;

    // pose-blend role: (row slice sl, output column o of the selector vertices)
    const int pd_ld = pad4(ns3);                   // row stride of S.sel_pd
    const int NSL = (ns3 > 0 && ns3 <= nt) ? nt / ns3 : 1;
    const int rows_sl = (npf + NSL - 1) / NSL;

    // Adam role: parameter `tid`; its moments and descriptor live in LDS (read once per iteration, in the Adam phase)
    if (!use_image)
    for (int i = tid; i < np; i += nt) {
        S.am[i] = io.params0 ? 0.f : io.adam_m[(size_t)frame * np + i];
        S.av[i] = io.params0 ? 0.f : io.adam_v[(size_t)frame * np + i];
        S.pk[i] = T.p_kind[i]; S.pa_[i] = T.p_a[i]; S.pb_[i] = T.p_b[i];
    }
    BF_KMARK(12, 0);
    int hand_j0_l = 0, hand_j0_r = 0;          // first joint of each hand
    for (int j = nj - 1; j >= 0; --j) { if (T.th_kind[j] == 2) hand_j0_l = j; if (T.th_kind[j] == 3) hand_j0_r = j; }
    BF_KMARK(0, 0); BF_KMARK(1, 256);
    __syncthreads();
    BF_KMARK(2, 0);
    if (mode == 2) {                                         // image builder: leave the LDS segment as it is now and stop
        if (frame == 0 && io.image_out)
            for (int i = tid; i < T.lds_image_n4 * 4; i += nt) io.image_out[i] = smem_raw[i];
        return;
    }

    int bf_it = -1;                            // (stamp build only; dead otherwise)
    long long bf_t0 = 0;
    float grad_last = 0.f;
    int pidx_last = -1;
    constexpr bool ROT_AHEAD = NJ == 24 && NB > 0 && NB <= 10 && NS > 0 && NS * 3 <= 36;       // (= MERGE_IK; see rotations_ahead)
    // A sized instance without selector vertices and without loss joints of its own (SMPL-X in the dense schedule: the keypoint loss
    // is the dense kernels') has nothing to do between the forward chain and the reverse sweep: the pose-blend, skinning and
    // projection phases are empty, and their three barriers - ~2,000 cycles of every iteration of the resident launch - are dropped
    // (in BOTH role loops: the barrier counts must match).
    constexpr bool NO_VERT = NJ > 0 && NS == 0 && NL == 0;
    // Everything that depends on the betas alone, for the NEXT forward pass (wave 3; its lanes cover the outputs):
    // shaped selector vertices, rest joints J, joint offsets rel_j = J_j - J_parent from the pre-contracted difference
    // tables, and the zeroed targets of the projection phase's routing
    // (sized instances) one selector output and two joint coordinates per lane; every LDS read issued before the arithmetic.
    // bq = (beta_0 .. beta_{nb-1}, 1, 0 ...): column nb of a table row is its constant term.  `parts`: 1 = the rest joints J (+ the zeroed
    // routing targets dGt), 2 = the joint offsets rel, 4 = the shaped selector vertices (+ zeroed dvsel) - the Adam phase deals the
    // three parts to three waves.
    auto beta_rows = [&](const float *bq, auto parts_c) {
        constexpr int NQ = NB ? (NB + 4) / 4 : 1;
        constexpr int parts = decltype(parts_c)::value;
        const int o = lane < ns3 ? lane : 0, i0 = lane, i1 = lane + 64 < nj3 ? lane + 64 : 0;
        float4 sq[NQ], ja[NQ], ra[NQ], jb[NQ], rb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (parts & 4) sq[q] = ((const float4 *)(S.sel_sd + o * nbp))[q];
            if (parts & 1) { ja[q] = ((const float4 *)(S.Jd + i0 * nbp))[q]; jb[q] = ((const float4 *)(S.Jd + i1 * nbp))[q]; }
            if (parts & 2) { ra[q] = ((const float4 *)(S.Jdrel + i0 * nbp))[q]; rb[q] = ((const float4 *)(S.Jdrel + i1 * nbp))[q]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        auto dot = [&](const float4 *t) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc += t[q].x * bq[4 * q] + t[q].y * bq[4 * q + 1] + t[q].z * bq[4 * q + 2] + t[q].w * bq[4 * q + 3];
            return acc;
        };
        if ((parts & 4) && lane < ns3) { S.vs[lane] = dot(sq); S.dvsel[lane] = 0.f; }
        // (rel_j[r], i = 3 j + r: L[12 j + 4 r + 3] = L[4 i + 3] when the rotations are formed ahead)
        if (parts & 1) { S.J[i0] = dot(ja); S.dGt[i0] = 0.f; if (lane + 64 < nj3) { S.J[i1] = dot(jb); S.dGt[i1] = 0.f; } }
        if (parts & 2) { (ROT_AHEAD ? S.L[4 * i0 + 3] : S.rel[i0]) = dot(ra); if (lane + 64 < nj3) (ROT_AHEAD ? S.L[4 * i1 + 3] : S.rel[i1]) = dot(rb); }
    };
    auto beta_dependent = [&](const float *P) {
        const float *beta = P + T.off_beta;
        constexpr int NQ = NB ? (NB + 4) / 4 : 0;          // float4s per table row (compile-time for the SMPL instance)
        if (NQ > 0 && NS > 0 && NS * 3 <= 64 && NJ > 0 && NJ * 3 <= 128) {
            float bq[NQ > 0 ? NQ * 4 : 4];
#pragma unroll
            for (int c = 0; c < NQ * 4; ++c) bq[c] = c < nb ? beta[c] : (c == nb ? 1.0f : 0.f);
            beta_rows(bq, std::integral_constant<int, 7>());
        } else {
            for (int o = lane; o < ns3; o += 64) {
                float acc = S.sel_sd[o * nbp + nb];
                for (int l = 0; l < nb; ++l) acc += S.sel_sd[o * nbp + l] * beta[l];
                S.vs[o] = acc;
                S.dvsel[o] = 0.f;
            }
            if (NB > 0 && NB <= 11) {
                // (sized instance: a row of each table is three b128 - ten coefficients, the constant term, a zero - requested before the
                //  first multiply-add; the sums themselves are the loops' below, term by term)
                float bq[12];
#pragma unroll
                for (int c = 0; c < 12; ++c) bq[c] = c < nb ? beta[c] : 0.f;
                for (int i = lane; i < nj3; i += 64) {
                    const float4 *jd4 = (const float4 *)__builtin_assume_aligned(S.Jd + i * nbp, 16), *jr4 = (const float4 *)__builtin_assume_aligned(S.Jdrel + i * nbp, 16);
                    const float4 a0 = jd4[0], a1 = jd4[1], a2 = jd4[2], r0 = jr4[0], r1 = jr4[1], r2 = jr4[2];
                    __builtin_amdgcn_sched_barrier(0);
                    const float jd[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
                    const float jr[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
                    float acc = 0.f, acr = jr[NB > 0 ? NB : 0];
#pragma unroll
                    for (int l = 0; l < (NB > 0 ? NB : 0); ++l) { acc += jd[l] * bq[l]; acr += jr[l] * bq[l]; }
                    S.J[i] = jd[NB > 0 ? NB : 0] + acc;
                    S.rel[i] = acr;
                    S.dGt[i] = 0.f;
                }
            } else
            for (int i = lane; i < nj3; i += 64) {
                // (J as Jt + (sum of the products, l ascending): bf_pose_state_body's arithmetic, which takes these values over - J_pre)
                float acc = 0.f, acr = S.Jdrel[i * nbp + nb];
                for (int l = 0; l < nb; ++l) { const float bl = beta[l]; acc += S.Jd[i * nbp + l] * bl; acr += S.Jdrel[i * nbp + l] * bl; }
                S.J[i] = S.Jd[i * nbp + nb] + acc;
                S.rel[i] = acr;
                S.dGt[i] = 0.f;
            }
        }
    };
    // The rotations left phase A: the lane that steps a joint's three dofs (phase I, wave 0) forms the joint's rotation for the NEXT
    // forward pass right away (rotations_ahead: S.R, S.theta, S.rc, S.feat), so phase A starts with the chain, and the GMM waves find
    // the pose feature in LDS.  (Not after the last step: the epilogue publishes the state of the last forward pass.)
    constexpr bool GBLEND = ROT_AHEAD && NS <= 12;
    auto rotations_ahead = [&](int j, float th0, float th1, float th2) {
        float Ri[9], rc[4];
        rodrigues_fwd(th0, th1, th2, Ri, rc);
#pragma unroll
        for (int e = 0; e < 9; ++e) S.L[j * 12 + (e / 3) * 4 + (e % 3)] = Ri[e];
        if (mode == 1) {                             // (debug dump only)
#pragma unroll
            for (int e = 0; e < 9; ++e) S.R[j * 9 + e] = Ri[e];
        }
        S.theta[j * 3] = th0; S.theta[j * 3 + 1] = th1; S.theta[j * 3 + 2] = th2;
        *(float4 *)(S.rc + j * 4) = make_float4(rc[0], rc[1], rc[2], rc[3]);
        if (j > 0) {
            float *f = S.feat + (j - 1) * 9;
            f[0] = Ri[0] - 1.f; f[1] = Ri[1]; f[2] = Ri[2]; f[3] = Ri[3]; f[4] = Ri[4] - 1.f;
            f[5] = Ri[5]; f[6] = Ri[6]; f[7] = Ri[7]; f[8] = Ri[8] - 1.f;
        }
    };
    if (EXT && tid == 0) ((int *)S.part)[BF_POSE_STATE_FLAG] = 0;          // (the chain waves' cue: no iteration's token yet)
    if (wave == 3) beta_dependent(S.pa);
    if (ROT_AHEAD && wave == 0 && lane < nj) {
        const int po = lane > 0 ? T.off_pose + 3 * (lane - 1) : T.off_orient;
        rotations_ahead(lane, 0.f + S.pa[po], 0.f + S.pa[po + 1], 0.f + S.pa[po + 2]);
    }
    __syncthreads();
    // phases shared by both wave roles
    auto pose_blend = [&](auto nbatch) {
        constexpr int NBATCH = decltype(nbatch)::value;     // 1: all reads up front; 2: two half batches (fewer registers)
        constexpr int RS = (NJ && NS) ? (9 * (NJ - 1) + (BF_FIT_THREADS / (NS * 3)) - 1) / (BF_FIT_THREADS / (NS * 3)) : 0;
        constexpr int RB = RS > 0 ? (RS + NBATCH - 1) / NBATCH : 1;
        for (int idx = tid; idx < ns3 * NSL; idx += nt) {
            int sl = idx / ns3, o = idx - sl * ns3;
            int p0 = sl * rows_sl, p1 = min(npf, p0 + rows_sl);
            float acc = 0.f;
            const float *pd = S.sel_pd + p0 * pd_ld + o;
            if (RS > 0) {
                // compile-time slice length: the LDS reads of a batch are issued before its multiply-adds (one wait instead
                // of one per pair), rows past the end are clamped and weighted 0
#pragma unroll
                for (int h = 0; h < NBATCH; ++h) {
                    float f[RB], w[RB];
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        const int p = min(p0 + h * RB + i, npf - 1);
                        f[i] = S.feat[p];
                        w[i] = S.sel_pd[p * pd_ld + o];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < RB; ++i) acc += (h * RB + i < RS && p0 + h * RB + i < p1 ? f[i] : 0.f) * w[i];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int p = p0; p < p1; ++p, pd += pd_ld) acc += S.feat[p] * pd[0];
            }
            S.vpp[idx] = acc;
        }
    };
    // d(pose feature) = sel_pd . dvp for rows first, first + step, ... < last (the GMM waves' registers are full of
    // precision rows, so the geometry waves the reverse sweep leaves idle take it: waves 2-3 in phase G)
    auto dfeat_rows = [&](const float *dvp_src, int first, int last, int step) {
        constexpr int NO = NS > 0 ? NS * 3 : 1;
        if (NS > 0) {
            float4 dq4[(NO + 3) / 4];
#pragma unroll
            for (int q = 0; q < (NO + 3) / 4; ++q) dq4[q] = ((const float4 *)__builtin_assume_aligned(dvp_src, 16))[q];
            const float *dv = (const float *)dq4;
            for (int p = first; p < last; p += step) {
                const float4 *row = (const float4 *)__builtin_assume_aligned(S.sel_pd + p * pd_ld, 16);
                float4 rq[(NO + 3) / 4];
#pragma unroll
                for (int q = 0; q < (NO + 3) / 4; ++q) rq[q] = row[q];
                __builtin_amdgcn_sched_barrier(0);
                const float *rv = (const float *)rq;
                float acc = 0.f;
#pragma unroll
                for (int o = 0; o < NO; ++o) acc += rv[o] * dv[o];
                S.dfeat[p] = ext ? acc + ext[p] : acc;
            }
        } else {
            for (int p = first; p < last; p += step) {
                float acc = 0.f;
                const float *row = S.sel_pd + p * pd_ld;
                for (int o = 0; o < ns3; ++o) acc += row[o] * dvp_src[o];
                S.dfeat[p] = ext ? acc + ext[p] : acc;
            }
        }
    };
    // The pose blend and the skinning of the selector vertices (models with at most 12 selector vertices and at most 4 bones per
    // selector vertex; the others take the two-phase path: 15 row slices over all 8 waves, then skinning).
    // (bf_fit_launch picks a GBLEND instance only for models whose selector vertices have at most BF_SEL_NNZ = 4 bones; the others
    //  take the table-driven instance, whose pose blend is the two-phase one)
    constexpr bool merge_bc = GBLEND;
    static_assert(!GBLEND || NS * 3 * BF_PDT_LD <= 9 * (NJ - 1) * (NS * 3 + 1), "the transposed selector posedirs fit the sel_pd2 slot");
    // Round 3: the pose blend LEFT the geometry waves.  The GMM waves issue ~300 instructions per iteration against the geometry waves'
    // ~1,900, and the pose blend needs nothing but the pose feature, which is in LDS when the iteration starts (rotations_ahead):
    // every GMM wave blends its three selector vertices (the lane pattern of blend_and_skin: 9 coordinates x 7 row slices, partials
    // through a wave-private strip) into S.vp under the chain waves' phase A.  Phase B on the geometry waves is then the skinning
    // alone (skin_only).
    auto gmm_blend = [&]() {
        typedef float f2 __attribute__((ext_vector_type(2)));
        constexpr int RS = BF_PDT_LD / 7, NP2 = RS / 2, NH = 3, PB = NP2 / NH;      // 30 rows = 15 row pairs in three batches of five
        constexpr int NPF = NJ > 0 ? 9 * (NJ - 1) : 8;
        static_assert(RS * 7 == BF_PDT_LD && NP2 * 2 == RS && PB * NH == NP2, "slice shape");
        const int lq = bf_launder(lane);          // (fresh per iteration: the row addresses below must not become loop invariants)
        const int ol7 = lq / 7, ol = min(ol7, 8), sl = lq - ol7 * 7;
        const int sv = gwi * 3 + ol / 3, c = ol - (ol / 3) * 3;
        const bool on = ol7 < 9 && sv < ns;
        const int o = (on ? sv : 0) * 3 + c;
        const int p0 = sl * RS;
        const f2 *fp = (const f2 *)__builtin_assume_aligned(S.feat + p0, 8);
        const f2 *wp_ = (const f2 *)__builtin_assume_aligned(S.sel_pd2 + o * BF_PDT_LD + p0, 8);
        float acc = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            f2 f[PB], w[PB];
#pragma unroll
            for (int i = 0; i < PB; ++i) { f[i] = fp[h * PB + i]; w[i] = wp_[h * PB + i]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                // (rows past the end of the pose feature: the table holds zeros there, the feature's slot whatever follows it in LDS)
                const int p = p0 + 2 * (h * PB + i);
                acc += (p < NPF ? f[i].x : 0.f) * w[i].x;
                acc += (p + 1 < NPF ? f[i].y : 0.f) * w[i].y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float *strip = S.vpp + (4 + gwi) * 64;         // this wave's 63 partials
        strip[lq] = on ? acc : 0.f;
        BF_WAVE_FENCE();
        // lanes 0-8: coordinate lq of the wave's three vertices = shaped vertex + the seven slice partials in slot order
        const int o9 = min(lq, 8), sv9 = gwi * 3 + o9 / 3;
        const bool on9 = lq < 9 && sv9 < ns;
        const int oo = (on9 ? sv9 : 0) * 3 + (o9 - (o9 / 3) * 3);
        const float *pp = strip + o9 * 7;
        float pr[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) pr[i] = pp[i];
        float vpb = S.vs[oo];
#pragma unroll
        for (int i = 0; i < 7; ++i) vpb += pr[i];
        if (on9) S.vp[oo] = vpb;
    };
    // d(pose feature) = sel_pd . dvp in phase F, a quarter of the rows per GMM wave (their prior is complete by then: phase D).  dvp =
    // T_s^T dvsel is formed by the wave itself (lanes 0..3 ns - 1, wave 3's arithmetic) and crosses lanes through the wave's strip;
    // rows and dvp are consumed in 8-column pieces (registers), in column order: the bits of dfeat_rows.
    auto gmm_dfeat = [&]() {
        constexpr int NO = NS > 0 ? NS * 3 : 1, NQ = (NO + 3) / 4, RW = NJ > 0 ? (9 * (NJ - 1) + 3) / 4 : 1;     // rows per wave
        const int lq = bf_launder(lane);
        float *strip = S.vpp + (4 + gwi) * 64;
        {
            const int o = min(lq, NO - 1), sv = o / 3, bb = o - sv * 3;
            const float dvp = S.TR[sv * 9 + bb] * S.dvsel[sv * 3] + S.TR[sv * 9 + 3 + bb] * S.dvsel[sv * 3 + 1] +
                              S.TR[sv * 9 + 6 + bb] * S.dvsel[sv * 3 + 2];
            if (lq < NQ * 4) strip[lq] = lq < NO ? dvp : 0.f;
        }
        BF_WAVE_FENCE();
        const int p = gwi * RW + lq;
        const bool on = lq < RW && p < npf;
        const float4 *row = (const float4 *)__builtin_assume_aligned(S.sel_pd + (on ? p : 0) * pd_ld, 16);
        const float4 *dq = (const float4 *)__builtin_assume_aligned(strip, 16);
        float acc = 0.f;
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            const float4 rq = row[h], dv4 = dq[h];
            __builtin_amdgcn_sched_barrier(0);
            if (h * 4 < NO) acc += rq.x * dv4.x;
            if (h * 4 + 1 < NO) acc += rq.y * dv4.y;
            if (h * 4 + 2 < NO) acc += rq.z * dv4.z;
            if (h * 4 + 3 < NO) acc += rq.w * dv4.w;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (on) S.dfeat[p] = ext ? acc + ext[p] : acc;
    };
    // What the pose priors add to dL/dtheta, per body dof, for the Adam phase (wave 3, phase F - all eight q values are in LDS
    // since the barrier behind phase D): the arg-min component (prior.py:195), w_pose * y of that component, and the angle prior's
    // term (loss.py:54-61: dofs 52, 55, 9, 12; the sign is 0 everywhere else, which zeroes it exactly).  The lane that steps a joint
    // then adds two LDS values per dof instead of taking the arg-min and three exponentials on the critical path of the iteration.
    auto gmm_prior_grad = [&]() {
        const int lq = bf_launder(lane);
        float gq[BF_GMM_M];
#pragma unroll
        for (int m = 0; m < BF_GMM_M; ++m) gq[m] = S.gq[m];
        int ms = 0;
        float qm = gq[0];
#pragma unroll
        for (int m = 1; m < BF_GMM_M; ++m) if (gq[m] < qm) { qm = gq[m]; ms = m; }
        if (lq == 0) { S.scal[1] = (float)ms; S.scal[2] = qm; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pb = lq + 64 * q;
            if (pb < T.nbp && pb < BF_GMM_D) {
                const float gyv = S.gy[ms * BF_GMM_LD + pb], pv = Pcur[T.off_pose + pb];
                const float sg = (pb == 52 ? 1.f : 0.f) - (pb == 55 ? 1.f : 0.f) - (pb == 9 ? 1.f : 0.f) - (pb == 12 ? 1.f : 0.f);
                const float ex = __expf(pv * sg);
                S.gtail[pb] = hp.w_pose * gyv;
                S.gtail[BF_GMM_LD + pb] = hp.w_angle * 2.f * ex * ex * sg;
            }
        }
    };
    // phase B when the GMM waves blend: T_s = sum_j w_sj A_j of the wave's three vertices (lanes 0-35: vertex, row k, column b),
    // skinned vertex = T_s [vp | 1]
    auto skin_only = [&]() {
        constexpr int NZ = 4;
        const int lq = bf_launder(lane);
        const int vloc = lq / 12, e12 = lq - vloc * 12, k = e12 >> 2, b = e12 & 3;
        // (MERGE_BD: the wave's vertices are the deal's - up to five, lanes 0-59 - so that it projects what it skins)
        const int sv2 = MERGE_BD ? skin_dealt : wave * 3 + vloc;
        const bool trl = MERGE_BD ? (lq < 12 * BF_SKIN_PER_WAVE && sv2 >= 0) : (lq < 36 && sv2 < ns);
        const int sv2c = trl ? sv2 : 0;
        float wq[NZ], aq[NZ];
        int jq[NZ];
#pragma unroll
        for (int q = 0; q < NZ; ++q) { wq[q] = S.nzw[sv2c * BF_SEL_NNZ + q]; jq[q] = S.nzj[sv2c * BF_SEL_NNZ + q]; }
        const float vpb = S.vp[sv2c * 3 + (b < 3 ? b : 0)];
        const float *A = b < 3 ? S.G + k * 4 + b : S.At + k;
        const int stride = b < 3 ? 12 : 3;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NZ; ++q) aq[q] = A[jq[q] * stride];
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < NZ; ++q) t += wq[q] * aq[q];
        if (trl && b < 3) S.TR[sv2c * 9 + k * 3 + b] = t;
        float contrib = trl ? (b < 3 ? t * vpb : t) : 0.f;
        contrib = quad_sum(contrib);
        if (trl && b == 0) S.vsel[sv2c * 3 + k] = contrib;
    };
    typedef float v2f __attribute__((ext_vector_type(2)));
    auto project = [&](bool want_loss) {
        BF_MARK(45, 0, bf_it, bf_t0);
        const float tX = Pcur[0], tY = Pcur[1], tZ = Pcur[2];
        const float sc = Pcur[3] * cscale;
        const v2f y0 = {lsrc_a[0] + tX, lsrc_b[0] + tX};
        const v2f y1 = {lsrc_a[lstride_a] + tY, lsrc_b[lstride_b] + tY};
        const v2f y2 = {lsrc_a[2 * lstride_a] + tZ, lsrc_b[2 * lstride_b] + tZ};
        const v2f x0 = y0 * sc, x1 = y1 * sc, x2 = y2 * sc;
        v2f g0 = {0.f, 0.f}, g1 = {0.f, 0.f}, g2 = {0.f, 0.f}, lsum = {0.f, 0.f};
        // one view for the lane's joint pair: gxy = (x_a, x_b, y_a, y_b), kc = (k_a, k_b, c_a, c_b)
        auto one_view = [&](int v, float4 gxy, float4 kc) {
            const float4 *P = (const float4 *)__builtin_assume_aligned(S.proj + v * 12, 16);
            const float4 Pa4 = P[0], Pb4 = P[1], Pc4 = P[2];
            // (the row's fourth entry seeds the multiply-add chain: three packed fma per row instead of mul, fma, fma, splat, add)
            const v2f w0 = {Pa4.w, Pa4.w}, w1 = {Pb4.w, Pb4.w}, w2 = {Pc4.w, Pc4.w};
            v2f p0 = Pa4.x * x0 + w0; p0 = Pa4.y * x1 + p0; p0 = Pa4.z * x2 + p0;
            v2f p1 = Pb4.x * x0 + w1; p1 = Pb4.y * x1 + p1; p1 = Pb4.z * x2 + p1;
            v2f p2 = Pc4.x * x0 + w2; p2 = Pc4.y * x1 + p2; p2 = Pc4.z * x2 + p2;
            const v2f ip2 = {__builtin_amdgcn_rcpf(p2.x), __builtin_amdgcn_rcpf(p2.y)};     // v_rcp_f32: 1 ulp
            const v2f u = p0 * ip2, w = p1 * ip2;
            const v2f gx = {gxy.x, gxy.y}, gy = {gxy.z, gxy.w}, kk = {kc.x, kc.y}, cc = {kc.z, kc.w};
            const v2f rx = (gx - u) * icoeff, ry = (gy - w) * icoeff;
            const v2f ax = rx * rx, ay = ry * ry;
            const v2f dx = ax + s2, dy = ay + s2;
            // (one v_rcp_f32 per joint for both GMoF denominators, 1 / dx = dy / (dx dy): a transcendental is a quarter-rate instruction;
            //  with the two regroupings below - (kk ix)(rx ix), and q2's term last in the gradient's sum - 9,93x -> 9,85x cycles, same box)
            const v2f dxy = dx * dy;
            const v2f ixy = {__builtin_amdgcn_rcpf(dxy.x), __builtin_amdgcn_rcpf(dxy.y)};
            const v2f ix = ixy * dy, iy = ixy * dx;
            if (want_loss) lsum += cc * (ax * ix + ay * iy);          // (the value only leaves with the last forward pass)
            const v2f du = (kk * ix) * (rx * ix), dw = (kk * iy) * (ry * iy);
            const v2f q0 = du * ip2, q1 = dw * ip2, q2 = -(du * u + dw * w) * ip2;
            g0 = Pc4.x * q2 + (g0 + (Pa4.x * q0 + Pb4.x * q1));
            g1 = Pc4.y * q2 + (g1 + (Pa4.y * q0 + Pb4.y * q1));
            g2 = Pc4.z * q2 + (g2 + (Pa4.z * q0 + Pb4.z * q1));
        };
        // no per-round branch: records past V carry zero factors and the view index is clamped, so the rounds are
        // independent straight-line code the scheduler interleaves
#pragma unroll
        for (int r = 0; r < BF_KP_ROUNDS; ++r) {
            const float4 *kq = kp_lane + r * BF_VSUB * 32;
            one_view(min(vsub + BF_VSUB * r, V - 1), kq[0], kq[kp_other]);
        }
        for (int v = vsub + BF_VSUB * BF_KP_ROUNDS; v < V; v += BF_VSUB) {   // V > 48: stream the rest from global memory
            float4 gxy = {0.f, 0.f, 0.f, 0.f}, kc = {0.f, 0.f, 0.f, 0.f};
            if (ja_on) { const float *k = kp_frame + ((size_t)v * nl + ja) * 3; float c2 = k[2] * k[2];
                         gxy.x = k[0]; gxy.z = k[1]; kc.x = c2 * kscale * (2.f * s2 * s2); kc.z = c2 * s2; }
            if (jb_on) { const float *k = kp_frame + ((size_t)v * nl + jb) * 3; float c2 = k[2] * k[2];
                         gxy.y = k[0]; gxy.w = k[1]; kc.y = c2 * kscale * (2.f * s2 * s2); kc.w = c2 * s2; }
            one_view(v, gxy, kc);
        }
        BF_MARK(43, 0, bf_it, bf_t0);
        // sum over the 16 view lanes of the pair (fixed DPP tree), route dL/dX to its source, and leave this wave's
        // share of d/dt, d/ds and of the loss value for the Adam phase
        // the six 16-lane sums stage by stage (a dependent DPP costs ~20 cycles on gfx950: six chains of four in a row are
        // 24 of them back to back, four stages of six independent ones are four)
        float rv[6] = {g0.x, g1.x, g2.x, g0.y, g1.y, g2.y};
#pragma unroll
        for (int q = 0; q < 6; ++q) rv[q] = dpp_add<0xB1>(rv[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q) rv[q] = dpp_add<0x4E>(rv[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q) rv[q] = dpp_add<0x124>(rv[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q) rv[q] = dpp_add<0x128>(rv[q]);
        const float ga0 = rv[0], ga1 = rv[1], ga2 = rv[2], gb0 = rv[3], gb1 = rv[4], gb2 = rv[5];
        float la = 0.f, lb = 0.f;
        if (want_loss) { la = row16_sum(lsum.x); lb = row16_sum(lsum.y); }     // (the value only leaves with the last forward pass)
        if (vsub == 0 && wave < 4) {
            if (ja_on) { atomicAdd(ldst_a + 0, ga0 * sc); atomicAdd(ldst_a + 1, ga1 * sc); atomicAdd(ldst_a + 2, ga2 * sc); }
            if (jb_on) { atomicAdd(ldst_b + 0, gb0 * sc); atomicAdd(ldst_b + 1, gb1 * sc); atomicAdd(ldst_b + 2, gb2 * sc); }
            // this pair's share of d/dt, d/ds and of the loss value, for the Adam phase (which adds the 16 pair slots in
            // slot order); joints off the end have zero records, their sums are exactly 0
            const float gsa = ga0 * y0.x + ga1 * y1.x + ga2 * y2.x, gsb = gb0 * y0.y + gb1 * y1.y + gb2 * y2.y;
            float4 q0 = {ga0 + gb0, ga1 + gb1, ga2 + gb2, jb_on ? gsa + gsb : (ja_on ? gsa : 0.f)};
            *(float4 *)(S.part + pphys * 8) = q0;               // (by physical slot: the Adam phase adds the 16 slots in that order)
            S.part[pphys * 8 + 4] = la + lb;
        }
        BF_MARK(44, 0, bf_it, bf_t0);
    };

    // ---- persistent dense-schedule launch: the two ends of an iteration (called by EVERY thread: they synchronise) ----------
    // top: wait for this iteration's outside gradient blocks and stage them in LDS
#ifdef BF_STAMP
    long long bf_d0 = 0;
#define BF_DMARK(k) do { if (tid == 0) S.stamp[k] = (float)(long long)(clock64() - bf_d0); } while (0)
#else
#define BF_DMARK(k) do { } while (0)
#endif
    // The pose state the iteration's forward mesh pass is waiting for: bf_pose_state_kernel's code (correctly rounded sqrt /
    // division, OCML sin / cos: the fit kernel's own phase A uses 1-ulp forms, and a silhouette loss turns a last-bit difference in
    // a vertex into a different nearest-vertex choice), run by wave 3 - idle in phase A - while waves 0-2 form the chain,
    // from the LDS copies of the model's tables; the view-sum slots (dead until phase D) are its scratch.
    int door_token = 0;                      // (this iteration's cue value for the chain waves: it + 1, never the value a stale flag holds)
    auto door_state = [&](const float *P) {
        const PoseTabs PT{S.thk, S.tho, S.par, S.pmean, S.hcomp, S.Jd, S.Jt, pad4(nb + 1), S.lvl, S.lvl + nj, S.lvl + nj + 67};
        // (the table-driven beta_dependent leaves exactly the rest joints the pose state wants in S.J - see pose_state_body.h; the
        //  SMPL instance's float4-row form sums in another order, so that instance lets the body form them)
        constexpr bool J_SHARED = !(NB > 0 && NS > 0 && NS * 3 <= 64 && NJ > 0 && NJ * 3 <= 128);
        // The pose feature leaves EARLY (BF_DOOR_FEAT): as soon as the rotations exist wave 3 stores it with device-scope stores (they
        // go past this XCD's L2, so no write-back is needed), waits for them and rings - the forward mesh pass that was waiting runs
        // its pose blend under the chain below and the hand-over, and waits for the chain matrices (BF_DOOR_STATE) only after that.
        auto publish_feat = [&](const float *R) {
            StateView st = bf_state_view(io.state + (size_t)frame * bf_state_stride(nj, npf, nb), nj, npf, nb);
            for (int p = lane; p < npf; p += 64) {
                const int j = 1 + p / 9, e = p - (p / 9) * 9;
                __hip_atomic_store(st.feat + p, R[j * 9 + e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the stores have reached memory at device scope
            if (lane < BF_DOOR_COPIES) __hip_atomic_fetch_add(door + BF_DOOR_FEAT + lane * BF_DOOR_COPY_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // ... and the chain of that state is formed by the three chain waves, a row each (bf_pose_chain_row), once they are through
        // with the kernel's own chain: wave 3 tells them when R and J are complete, BEFORE it publishes the feature.
        auto cue_then_publish = [&](const float *R) {
            if (lane == 0) *(volatile int *)((int *)S.part + BF_POSE_STATE_FLAG) = door_token;
            publish_feat(R);
        };
        bf_pose_state_body<true, true, false, decltype(cue_then_publish), false>(T, nullptr, nullptr, nullptr, nullptr, io.state, io.params, io.cscale, hp.cscale,
                                                                                  frame, lane, 64, S.part, PT, P, J_SHARED ? S.J : nullptr, cue_then_publish);
#ifdef BF_STAMP
        if (lane == 0) { const long long *mk = (const long long *)(S.part + 1740); for (int k = 1; k < 5; ++k) S.stamp[48 + k] = (float)(mk[k] - mk[0]); }
#endif
    };
    static_assert(BF_VSUB * 32 * 4 >= BF_POSE_STATE_LDS, "the view-sum slots are the pose-state scratch");
    // Between phase A and the rest of an iteration (called by EVERY thread: it synchronises): publish that state, then wait for the
    // outside gradient blocks the dense kernels make of it and stage them in LDS.
    // `works` = false: the GMM waves - 150 pinned precision registers and nothing to spare: taking row slices of the record and of the
    // staging made their loop spill 24 - 31 registers in every iteration of the dense instances (128 - 176 B of scratch per lane on the
    // resident launch's critical path) - only keep the barriers; the four geometry waves copy (plain copies: the same bits).
    auto door_mid = [&](int it, const float *P, auto works) {
#ifdef BF_STAMP
        bf_d0 = clock64();
#endif
        constexpr int nw = BF_FIT_THREADS / 2;
        // (wave 3 left the rotations, chain matrices and joints of the state in the scratch during phase A: the geometry waves write the record)
        if constexpr (decltype(works)::value) bf_pose_state_emit<true>(T, nullptr, io.state, io.params, io.cscale, hp.cscale, frame, tid, nw, S.part, nullptr, P, true);
        __syncthreads();                                       // the record's stores have reached the XCD's L2
        if constexpr (decltype(works)::value) {
            if (wave == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // one device-scope release (L2 write-back) for the workgroup
            BF_DMARK(56);
            if (tid < BF_DOOR_COPIES) __hip_atomic_fetch_add(door + BF_DOOR_STATE + tid * BF_DOOR_COPY_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) bf_door_wait(door, BF_DOOR_EXT, it + 1);
            BF_DMARK(57);
        }
        __syncthreads();
        // (the blocks were read through this CU's caches an iteration ago: device-scope loads go past them, no invalidate)
        if constexpr (decltype(works)::value) {
            float *eg = const_cast<float *>(io.ext) + (size_t)frame * n_ext;
            for (int i = tid; i < n_ext; i += nw) S.ext[i] = __hip_atomic_load(eg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        BF_DMARK(58);
    };

    if (gw) {
        // ================= GMM specialists (waves 4-7): the precision rows never leave their registers
        v2f P2[GBLEND ? BF_GMM_D : BF_GMM_LD];                       // (row `lane` of component a, of component b), column j (the padding columns meet d = 0: not kept)
        float Pt[12];
#pragma unroll
        for (int j = 0; j < (GBLEND ? BF_GMM_D : BF_GMM_LD); ++j) {
            P2[j].x = T.g_plane[((size_t)ma * BF_GMM_LD + j) * 64 + lane];
            P2[j].y = T.g_plane[((size_t)mb * BF_GMM_LD + j) * 64 + lane];
        }
#pragma unroll
        for (int e = 0; e < 12; ++e) Pt[e] = T.g_ptail[((size_t)gwi * 12 + e) * 64 + lane];
        const float4 *d4 = (const float4 *)__builtin_assume_aligned(gdw, 16);
        for (int it = 0; it < n_iters; ++it) {
#ifdef BF_STAMP
            int sidx = 0;
            const long long t_iter = 0;
#else
            const long long t_iter = 0;
#endif
            // The GMM prior (d = theta - mu, y = Psym d, q = 0.5 d'y - log w~ for this wave's two components) feeds only
            // the Adam phase and costs one wave ~3000 cycles, so it is cut into eight-column chunks, one or two per phase:
            // it never holds a barrier up and runs in the issue slots the geometry waves leave free.
            v2f dA = {0.f, 0.f}, dB = {0.f, 0.f};       // (d_a[lane], d_b[lane]) and (d_a[64 + lane], d_b[64 + lane])
            // (the parameter index of theta_j, j = lane and 64 + lane, is re-derived from the lane here: -1 = zero pad; kept across the
            //  loop it costs two of the registers the pinned rows need)
            const int lq0 = bf_launder(lane);
            const int src0 = lq0 < T.nbp ? T.off_pose + lq0 : -1, src1 = 64 + lq0 < T.nbp ? T.off_pose + 64 + lq0 : -1;
            if (src0 >= 0) { const float th = Pcur[src0]; dA.x = th - gd_mu[0]; dA.y = th - gd_mu[1]; }
            else { dA.x = -gd_mu[0]; dA.y = -gd_mu[1]; }
            if (lane < BF_GMM_D - 64) {
                const float th = src1 >= 0 ? Pcur[src1] : 0.f;
                dB.x = th - gd_mu[2]; dB.y = th - gd_mu[3];
            }
            ((v2f *)gdw)[lane] = dA;                   // LDS copy, interleaved (d_a[j], d_b[j]) pairs: one b128 = two columns
            if (lane < BF_GMM_LD - 64) ((v2f *)gdw)[64 + lane] = dB;
            BF_WAVE_FENCE();
            if (merge_bc) { gmm_blend(); BF_MARK(42, 256, it, t_iter); }
            v2f y0 = {0.f, 0.f}, y1 = {0.f, 0.f};
#define BF_GMM_CHUNK(c)                                                                         \
            _Pragma("unroll") for (int j2 = 4 * (c); j2 < 4 * (c) + 4; ++j2) {                  \
                const float4 t = d4[j2];                                                        \
                const v2f t0 = {t.x, t.y}, t1 = {t.z, t.w};                                     \
                if (2 * j2 < (GBLEND ? BF_GMM_D : BF_GMM_LD)) y0 += P2[2 * j2] * t0;            \
                if (2 * j2 + 1 < (GBLEND ? BF_GMM_D : BF_GMM_LD)) y1 += P2[2 * j2 + 1] * t1;    \
            }
            constexpr bool GMM_FG = NJ > 0 && NJ <= 32 && NS > 0 && NS * 3 <= 36;     // (= MERGE_FG of the geometry loop)
            // The chunk schedule.  Instances whose GMM waves also blend (GBLEND: SMPL): chunk 0 in A, 1-8 in the short phase B, the rest of
            // the prior in D (the projection phase), so that phase F finds all eight q values in LDS.  The other instances: 0-2
            // in A, 3-5 in B / C, 6-8 and the rest in F (the Adam phase K takes the arg-min itself).
            BF_GMM_CHUNK(0)
            if (!GBLEND) {
                BF_GMM_CHUNK(1)
                BF_GMM_CHUNK(2)
            }
            BF_SYNC();                 // A
            if (EXT && door) door_mid(it, Pcur, std::false_type());
            if (!merge_bc) {           // (two-phase pose blend: everybody takes row slices)
                pose_blend(std::integral_constant<int, 2>());
                if (!NO_VERT) BF_SYNC();             // B
            }
            if (GBLEND) {              // (the geometry waves wait on LDS in this phase: a chunk costs ~70 cycles here, ~300 under the projection)
                BF_GMM_CHUNK(1)
                BF_GMM_CHUNK(2)
                BF_GMM_CHUNK(3)
                BF_GMM_CHUNK(4)
                BF_GMM_CHUNK(5)
                BF_GMM_CHUNK(6)
                BF_GMM_CHUNK(7)
                BF_GMM_CHUNK(8)
                // the order of issue for the eight chunks above: the reads of two chunks in flight, not of all eight (32 b128 = 128 registers
                // next to 150 pinned ones); 0x100 = LDS reads, 0x002 = VALU
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int c = 0; c < 7; ++c) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                BF_MARK(49, 256, it, t_iter);
            } else {
                BF_GMM_CHUNK(3)
                BF_GMM_CHUNK(4)
                BF_GMM_CHUNK(5)
            }
            if (!NO_VERT && !MERGE_BD) BF_SYNC();   // C (the merged skin + projection phase of the geometry waves has no barrier here)
            // the rest of the prior once the mat-vec is complete: y -> LDS, the twelve-column tail pieces, the two quadratic forms
            auto gmm_finish = [&]() {
                const int tpq = min(bf_launder(lane), 59), tail_cq = tpq < 30 ? 0 : 1;          // (re-derived per iteration: registers)
                const float4 *tail_dq = (const float4 *)(gdw + 2 * 12 * (tpq % 6));

                const v2f y = y0 + y1;
                const float ya = y.x, yb = y.y;
                S.gy[ma * BF_GMM_LD + lane] = ya;
                S.gy[mb * BF_GMM_LD + lane] = yb;
                float yt = 0.f;
    #pragma unroll
                for (int e2 = 0; e2 < 6; ++e2) {
                    float4 t = tail_dq[e2];
                    yt += Pt[2 * e2] * (tail_cq ? t.y : t.x);
                    yt += Pt[2 * e2 + 1] * (tail_cq ? t.w : t.z);
                }
                S.gtail[gwi * 64 + lane] = yt;
                BF_WAVE_FENCE();
                float ta = dA.x * ya, tb = dA.y * yb;
                if (lane < 5) {
                    float ysa = 0.f, ysb = 0.f;
    #pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        ysa += S.gtail[gwi * 64 + 6 * lane + e];
                        ysb += S.gtail[gwi * 64 + 30 + 6 * lane + e];
                    }
                    S.gy[ma * BF_GMM_LD + 64 + lane] = ysa;
                    S.gy[mb * BF_GMM_LD + 64 + lane] = ysb;
                    ta += dB.x * ysa;
                    tb += dB.y * ysb;
                }
                ta = wave_sum(ta);
                tb = wave_sum(tb);
                if (lane == 0) {
                    S.gq[ma] = 0.5f * ta + logw_a;       // prior.py:188-189
                    S.gq[mb] = 0.5f * tb + logw_b;
                }
            };
            if (GBLEND) {              // D (+E): projection, view reduction and routing on the geometry waves: the rest of the prior
                gmm_finish();
                BF_MARK(55, 256, it, t_iter);
            }
            if (!NO_VERT) BF_SYNC();   // D
            if (!GBLEND) { BF_GMM_CHUNK(6) }
            if (!GMM_FG) BF_SYNC();    // F (two-phase path only)
            if (!GBLEND) {
                BF_GMM_CHUNK(7)
                BF_GMM_CHUNK(8)
                gmm_finish();
            } else {                   // F: this wave's quarter of d(pose feature) = sel_pd . dvp
                if (merge_bc) gmm_dfeat();
            }
#undef BF_GMM_CHUNK
            BF_MARK(48, 256, it, t_iter);
            BF_SYNC();                 // G (+H)
            if (!(NJ == 24 && NB > 0 && NB <= 10 && NS > 0 && NS * 3 <= 36)) {     // (two-phase I, K path only)
            if (tid == 256) {                        // arg-min GMM component (prior.py:195) for the Adam phase
                int ms = 0;
                float qm = S.gq[0];
#pragma unroll
                for (int m = 1; m < BF_GMM_M; ++m) { float q = S.gq[m]; if (q < qm) { qm = q; ms = m; } }
                S.scal[1] = (float)ms; S.scal[2] = qm;
            }
            BF_SYNC();                 // I (+J)
            }
            BF_SYNC();                 // K
            if (mode == 0) { float *sw = Pcur; Pcur = Pnext; Pnext = sw; }
        }
    } else {
    // ================= geometry waves (0-3)
    for (int it = 0; it < n_iters; ++it) {
#ifdef BF_STAMP
        int sidx = 0;
        const long long t_iter = 0;
        BF_T(it == n_iters - 1 ? 23 : 22);        // top of the last two iterations: their difference is one iteration
#else
        const long long t_iter = 0;
#endif
        // ================= phase A: forward chain (waves 0-2) | shaped selector verts (wave 3) | GMM (waves 4-7)
        if (wave < 3) {
            float Ri[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rc[4], rel0 = 0.f, rel1 = 0.f, rel2 = 0.f;
            float4 row = {0.f, 0.f, 0.f, 0.f};
            float jj0 = 0.f, jj1 = 0.f, jj2 = 0.f;
            const int wjq = bf_launder(wj);               // (this phase's LDS addresses are formed here, not kept across the loop)
            if (cw_on) {
                float th0 = w_pm0, th1 = w_pm1, th2 = w_pm2;
                // (SMPL: the joint's parameter offset is arithmetic; kept in a register across the loop it gets spilled and the
                //  reload delays the first read of the chain)
                const int woff = NJ == 24 ? (wjq > 0 ? T.off_pose + 3 * (wjq - 1) : T.off_orient) : w_off;
                if (ROT_AHEAD) { }
                else if (NJ == 24 || w_kind == 0) { th0 += Pcur[woff]; th1 += Pcur[woff + 1]; th2 += Pcur[woff + 2]; }
                else if (NJ != 24 && w_kind >= 2) {
                    float t3[3];
                    bf_theta3(Pcur, wj, t3, S.thk, S.tho, S.pmean, S.hcomp, T.n_pca, T.off_lh, T.off_rh);
                    th0 = t3[0]; th1 = t3[1]; th2 = t3[2];
                }
                // rel_j = J_j - J_parent (rel_0 = J_0) was formed from the betas by wave 3 at the end of the previous
                // iteration (or in the prologue); read it before the Rodrigues arithmetic so the latency hides under it
                float a0, a1, a2;
                if (ROT_AHEAD) {
                    const float4 l0 = *(const float4 *)(S.L + wjq * 12), l1 = *(const float4 *)(S.L + wjq * 12 + 4), l2 = *(const float4 *)(S.L + wjq * 12 + 8);
                    Ri[0] = l0.x; Ri[1] = l0.y; Ri[2] = l0.z; Ri[3] = l1.x; Ri[4] = l1.y; Ri[5] = l1.z; Ri[6] = l2.x; Ri[7] = l2.y; Ri[8] = l2.z;
                    a0 = l0.w; a1 = l1.w; a2 = l2.w;
                } else { a0 = S.rel[wjq * 3]; a1 = S.rel[wjq * 3 + 1]; a2 = S.rel[wjq * 3 + 2]; }
                jj0 = S.J[wjq * 3]; jj1 = S.J[wjq * 3 + 1]; jj2 = S.J[wjq * 3 + 2];
                __builtin_amdgcn_sched_barrier(0);
                if (!ROT_AHEAD) {
                rodrigues_fwd(th0, th1, th2, Ri, rc);
                if (wave == 2) { S.theta[wjq * 3] = th0; S.theta[wjq * 3 + 1] = th1; S.theta[wjq * 3 + 2] = th2; }
                }
                rel0 = a0; rel1 = a1; rel2 = a2;
                // bookkeeping stores spread over the three (otherwise identical) chain waves
                if (!ROT_AHEAD && wave == 0 && mode == 1) {                 // (debug dump only)
#pragma unroll
                    for (int e = 0; e < 9; ++e) S.R[wjq * 9 + e] = Ri[e];
                }
                if (!ROT_AHEAD && wave == 2) *(float4 *)(S.rc + wjq * 4) = make_float4(rc[0], rc[1], rc[2], rc[3]);
                if (!ROT_AHEAD && wave == 1 && wj > 0) {
                    float *f = S.feat + (wjq > 0 ? wjq - 1 : 0) * 9;
                    f[0] = Ri[0] - 1.f; f[1] = Ri[1]; f[2] = Ri[2]; f[3] = Ri[3]; f[4] = Ri[4] - 1.f;
                    f[5] = Ri[5]; f[6] = Ri[6]; f[7] = Ri[7]; f[8] = Ri[8] - 1.f;
                }
                if (wj == 0) {
                    row.x = wave == 0 ? Ri[0] : (wave == 1 ? Ri[3] : Ri[6]);
                    row.y = wave == 0 ? Ri[1] : (wave == 1 ? Ri[4] : Ri[7]);
                    row.z = wave == 0 ? Ri[2] : (wave == 1 ? Ri[5] : Ri[8]);
                    row.w = wave == 0 ? rel0 : (wave == 1 ? rel1 : rel2);
                    *(float4 *)(S.G + (wjq * 3 + wave) * 4) = row;
                }
            }
            BF_MARK(40, 0, it, t_iter);
            // level sweep, two tree levels per round.  Every lane first composes its local transform with its parent's,
            // M_j = L_parent L_j (the parent's 3x4 comes out of the parent lane's registers with ds_bpermute: every chain
            // wave holds the full local transforms); a joint of depth d then needs the finished row of its ancestor at
            // depth d - 2 (d - 1 for the first level): ceil(depth / 2) dependent rounds instead of depth.
            const int wp4 = wp * 4;
            float Mr[9], Mt[3];
            {
                float pr[9], pt[3];
#pragma unroll
                for (int e = 0; e < 9; ++e) pr[e] = __int_as_float(__builtin_amdgcn_ds_bpermute(wp4, __float_as_int(Ri[e])));
                pt[0] = __int_as_float(__builtin_amdgcn_ds_bpermute(wp4, __float_as_int(rel0)));
                pt[1] = __int_as_float(__builtin_amdgcn_ds_bpermute(wp4, __float_as_int(rel1)));
                pt[2] = __int_as_float(__builtin_amdgcn_ds_bpermute(wp4, __float_as_int(rel2)));
                const bool two = wd >= 2;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = pr[r * 3] * Ri[c] + pr[r * 3 + 1] * Ri[3 + c] + pr[r * 3 + 2] * Ri[6 + c];
                        Mr[r * 3 + c] = two ? v : Ri[r * 3 + c];
                    }
                    const float v = pr[r * 3] * rel0 + pr[r * 3 + 1] * rel1 + pr[r * 3 + 2] * rel2 + pt[r];
                    Mt[r] = two ? v : (r == 0 ? rel0 : (r == 1 ? rel1 : rel2));
                }
            }
            // (the composition itself stays scalar: packed, with the pairs assembled for it, it measured slower - 10.67 k against 10.50 k cycles)
            typedef float c2f __attribute__((ext_vector_type(2)));
            const c2f Mxy[3] = {{Mr[0], Mr[1]}, {Mr[3], Mr[4]}, {Mr[6], Mr[7]}};
            const c2f Mzw[3] = {{Mr[2], Mt[0]}, {Mr[5], Mt[1]}, {Mr[8], Mt[2]}};
            const int wa4 = (wd >= 2 ? w_gp : wp) * 4;            // ancestor lane: grandparent (parent on the first level)
            const int wround = (wd + 1) >> 1;                      // the round this joint is finished in
            // (the four outputs of a round as two packed pairs, (x, y) and (z, w): six v_pk_* instead of twelve multiply-adds on the
            //  critical wave; every component sees the same mul, fma, fma [, add] sequence as the scalar form)
            for (int rd = 1; 2 * rd - 1 < T.n_levels; ++rd) {
                float gx = __int_as_float(__builtin_amdgcn_ds_bpermute(wa4, __float_as_int(row.x)));
                float gy = __int_as_float(__builtin_amdgcn_ds_bpermute(wa4, __float_as_int(row.y)));
                float gz = __int_as_float(__builtin_amdgcn_ds_bpermute(wa4, __float_as_int(row.z)));
                float gw_ = __int_as_float(__builtin_amdgcn_ds_bpermute(wa4, __float_as_int(row.w)));
                const bool mine = wround == rd;
                const c2f nxy = gx * Mxy[0] + gy * Mxy[1] + gz * Mxy[2];
                c2f nzw = gx * Mzw[0] + gy * Mzw[1] + gz * Mzw[2];
                nzw.y = nzw.y + gw_;
                row.x = mine ? nxy.x : row.x; row.y = mine ? nxy.y : row.y; row.z = mine ? nzw.x : row.z; row.w = mine ? nzw.y : row.w;
            }
            if (cw_on && wj > 0) *(float4 *)(S.G + (wjq * 3 + wave) * 4) = row;
            // A_j translation row: Gt_j - GR_j J_j (J of this pass was formed with the betas, in the Adam phase)
            if (cw_on) S.At[wjq * 3 + wave] = row.w - (row.x * jj0 + row.y * jj1 + row.z * jj2);
            BF_MARK(41, 0, it, t_iter);
        } else if (EXT && door) {         // wave 3 has nothing of its own in this phase
            door_token = it + 1;
            door_state(Pcur);
            BF_MARK(59, 192, it, t_iter);
        }
        if (EXT && door && wave < 3) {
            // row `wave` of the published state's chain (the pose state's own arithmetic, from wave 3's rotations)
            volatile int *cue = (volatile int *)((int *)S.part + BF_POSE_STATE_FLAG);
            while (*cue != it + 1) __builtin_amdgcn_s_sleep(1);
            BF_WAVE_FENCE();
            bf_pose_chain_row(nj, T.n_levels, wave, lane, S.part, S.par, S.lvl + nj + 67);
        }
        BF_SYNC();
        if (EXT && door) door_mid(it, Pcur, std::true_type());

        if (merge_bc) {
            // ================= phase B (+C): pose blend and skinning of the selector vertices in one phase
            skin_only();
        } else {
        {
        // ================= phase B: pose blend of the selector vertices, partial sums over row slices
        pose_blend(std::integral_constant<int, 1>());
        }
                if (!NO_VERT) BF_SYNC();
        {
        // ================= phase C: finish the pose blend; skin the selector vertices.  Lane b of a quad owns
        // column b of row k of T_s = sum_j w_sj A_j
        const int tq = bf_launder(tid);          // (fresh per phase: keeps this phase's address arithmetic out of the loop-invariant set)
        for (int base = 0; base < ns3 * 4; base += nt) {
            int o = (base + tq) >> 2, b = tq & 3;
            bool ok = o < ns3;
            int sv = ok ? o / 3 : 0, k = ok ? o - sv * 3 : 0;
            float t = 0.f, vpb = 1.f;
            constexpr int CSL = (NJ && NS) ? BF_FIT_THREADS / (NS * 3) : 0;      // compile-time slice count (SMPL instance)
            if (ok) {
                const float *A = b < 3 ? S.G + k * 4 + b : S.At + k;
                const int stride = b < 3 ? 12 : 3;
                const int bb = b < 3 ? b : 0;
                if (CSL > 0 && sel_nnz > 0) {
                    // all LDS reads of the lane in two batches: (pose-blend slices, weights, bone indices), then the bones
                    float pv[CSL > 0 ? CSL : 1], wq[BF_SEL_NNZ], aq[BF_SEL_NNZ];
                    int jq[BF_SEL_NNZ];
#pragma unroll
                    for (int sl = 0; sl < CSL; ++sl) pv[sl] = S.vpp[sl * ns3 + sv * 3 + bb];
                    const float vs0 = S.vs[sv * 3 + bb];
#pragma unroll
                    for (int q = 0; q < BF_SEL_NNZ; ++q) { wq[q] = S.nzw[sv * BF_SEL_NNZ + q]; jq[q] = S.nzj[sv * BF_SEL_NNZ + q]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < BF_SEL_NNZ; ++q) aq[q] = A[jq[q] * stride];
                    float acc = 0.f;
#pragma unroll
                    for (int sl = 0; sl < CSL; ++sl) acc += pv[sl];
                    if (b < 3) { vpb = vs0 + acc; if (k == 0) S.vp[sv * 3 + b] = vpb; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < BF_SEL_NNZ; ++q) t += wq[q] * aq[q];
                } else {
                    if (b < 3) {
                        float acc = 0.f;
                        for (int sl = 0; sl < NSL; ++sl) acc += S.vpp[sl * ns3 + sv * 3 + b];
                        vpb = S.vs[sv * 3 + b] + acc;
                        if (k == 0) S.vp[sv * 3 + b] = vpb;
                    }
                    if (sel_nnz > 0) {                 // the non-zero skinning weights only (exact: the rest add 0)
#pragma unroll
                        for (int q = 0; q < BF_SEL_NNZ; ++q) t += S.nzw[sv * BF_SEL_NNZ + q] * A[S.nzj[sv * BF_SEL_NNZ + q] * stride];
                    } else {
                        const float *w = S.sel_w + sv * nj;
#pragma unroll
                        for (int j = 0; j < nj; ++j) t += w[j] * A[j * stride];
                    }
                }
                if (b < 3) S.TR[sv * 9 + k * 3 + b] = t;
            }
            float contrib = ok ? t * vpb : 0.f;
            contrib = quad_sum(contrib);
            if (ok && b == 0) S.vsel[o] = contrib;
        }
        }
                }
        if (MERGE_BD) BF_WAVE_FENCE();       // (the wave projects the selector vertices it has just skinned: no workgroup barrier)
        else if (!NO_VERT) BF_SYNC();

        // ================= phase D: similarity, multi-view projection, GMoF and its gradient
        if (nl > 0) project(it == n_iters - 1 || mode == 1);      // (no loss joints in this launch - the dense keypoint path: nothing to project)
        if (!NO_VERT) BF_SYNC();

        // (this step's Adam constants: a global read, issued ahead of its use - as a VECTOR load, the address made lane-dependent on
        //  purpose: a scalar load returns through the counter the LDS reads use, so every wave's first LDS wait of this phase also
        //  waited for the scalar cache - a miss to L2 every fifth iteration, the table being 12 bytes per iteration)
        const float *at = adam_tab + (size_t)(adam_t0 + it) * 3 + bf_launder(0);
        const float at0 = at[0], at1 = at[1], at2 = at[2];
        constexpr bool MERGE_FG = NJ > 0 && NJ <= 32 && NS > 0 && NS * 3 <= 36;
        if (MERGE_FG) {
        // ================= phase F (+G, H): reverse skinning AND the subtree sums in one phase.  Wave r (0-2) owns row r of
        // every joint, lane = depth-first position: a subtree is a contiguous run of positions, so with X_k = (row r of
        // M_k, dL/dGt_k[r]) the two subtree sums are P[last(k)] - P[k] of ONE inclusive prefix sum over the lanes (four DPP
        // row_shr stages + a cross-row fix-up) - no LDS round trip, no barrier between the reverse skinning that produces
        // X_k and the sums that consume it.  Wave 3 meanwhile forms d(pose feature) = sel_pd . dvp.
        const int lq = bf_launder(lane);
        float *strip = S.vpp + wave * 64;
        if (wave == 3 && lq < ns3) {                 // dvp = T_s^T dvsel
            const int sv = lq / 3, b = lq - sv * 3;
            const float dvp = S.TR[sv * 9 + b] * S.dvsel[sv * 3] + S.TR[sv * 9 + 3 + b] * S.dvsel[sv * 3 + 1] +
                              S.TR[sv * 9 + 6 + b] * S.dvsel[sv * 3 + 2];
            strip[lq] = dvp;
            S.dvp[lq] = dvp;                         // (for the betas' gradient in the next phase)
        }
        if (wave == 3 && lq >= 40 && lq < 45) {      // the projection phase's 16 pair slots -> d/dt, d/ds, loss value (slot order)
            const int q = lq - 40;
            float pq[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) pq[w] = S.part[w * 8 + q];
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) acc += pq[w];
            S.scal[3 + q] = nl > 0 ? acc : 0.f;               // (the slots are only written by the projection phase)
        }
        if (wave < 3) {
            constexpr int NSC = NS > 0 ? NS : 1;
            const int r = wave, k = fg_k;
            const bool on = lq < nj;
            // LDS reads first.  dL/dvsel and the posed vertices (33 floats each, the same for every lane) are read ONCE per
            // wave, one float per lane, and reach the arithmetic as scalars (v_readlane): 66 broadcast b128 reads per wave
            // cost ~700 cycles, 66 readlanes ~300.
            const float4 gi0 = *(const float4 *)(S.G + k * 12), gi1 = *(const float4 *)(S.G + k * 12 + 4),
                         gi2 = *(const float4 *)(S.G + k * 12 + 8);
            const float j0 = S.J[k * 3], j1 = S.J[k * 3 + 1], j2 = S.J[k * 3 + 2];
            const float c0 = GT_(0, 0), c1 = GT_(0, 1), c2 = GT_(0, 2);
            const float routed = S.dGt[k * 3 + r];
            const int lcl = min(lq, NSC * 3 - 1);
            const float my_dv = S.dvsel[lcl], my_vp = S.vp[lcl];
            float wv[NSC];
#pragma unroll
            for (int sv = 0; sv < NSC; ++sv) wv[sv] = S.sel_w[sv * nj + k];
            __builtin_amdgcn_sched_barrier(0);
            float da0 = 0.f, da1 = 0.f, da2 = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int sv = 0; sv < NSC; ++sv) {
                const float g0s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_dv), sv * 3));
                const float g1s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_dv), sv * 3 + 1));
                const float g2s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_dv), sv * 3 + 2));
                const float v0s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_vp), sv * 3));
                const float v1s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_vp), sv * 3 + 1));
                const float v2s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_vp), sv * 3 + 2));
                da0 += wv[sv] * g0s; da1 += wv[sv] * g1s; da2 += wv[sv] * g2s;
                const float wd_ = wv[sv] * (r == 0 ? g0s : (r == 1 ? g1s : g2s));
                r0 += wd_ * v0s; r1 += wd_ * v1s; r2 += wd_ * v2s;
            }
            float dat = r == 0 ? da0 : (r == 1 ? da1 : da2);
            if (ext) {                      // dense vertex losses: sum_v w_vj dv (x) [vp | 1] from bf_mesh_bwd_kernel (all rows of dAt)
                const float *ea = ext + EXT_A + k * 12;
                r0 += ea[r * 4]; r1 += ea[r * 4 + 1]; r2 += ea[r * 4 + 2];
                da0 += ea[3]; da1 += ea[7]; da2 += ea[11];
                dat = r == 0 ? da0 : (r == 1 ? da1 : da2);
            }
            const float gt_fin = routed + (ext ? dat + ext[EXT_G + k * 3 + r] : dat);
            const float d0 = r0 - dat * j0, d1 = r1 - dat * j1, d2 = r2 - dat * j2;       // row r of D_k
            const float dg0 = d0 * gi0.x + d1 * gi0.y + d2 * gi0.z;                      // row r of Dg_k = D_k GR_k^T
            const float dg1 = d0 * gi1.x + d1 * gi1.y + d2 * gi1.z;
            const float dg2 = d0 * gi2.x + d1 * gi2.y + d2 * gi2.z;
            const float u0 = gi0.w - c0, u1 = gi1.w - c1, u2 = gi2.w - c2;              // Gt_k - Gt_0
            // X_k = (row r of M_k = Dg_k + dGt_k (Gt_k - Gt_0)^T, dGt_k[r]); inclusive prefix sum over the DFS positions
            float x[4] = {on ? dg0 + gt_fin * u0 : 0.f, on ? dg1 + gt_fin * u1 : 0.f, on ? dg2 + gt_fin * u2 : 0.f, on ? gt_fin : 0.f};
            float p[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = dpp_add<0x111>(p[c]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = dpp_add<0x112>(p[c]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = dpp_add<0x114>(p[c]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = dpp_add<0x118>(p[c]);
            float sres[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float t15 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[c]), 15));
                p[c] += lq >= 16 ? t15 : 0.f;                       // (positions 16..31 sit in the second DPP row)
                const float pl = __int_as_float(__builtin_amdgcn_ds_bpermute(fg_last4, __float_as_int(p[c])));
                sres[c] = pl - p[c];                                // sum over the strict subtree
            }
            if (on) {
                S.tt[k * 3 + r] = sres[3] + x[3];                   // t_k: dL/dGt over the whole subtree
                const float s0 = sres[0] - sres[3] * u0 + dg0, s1 = sres[1] - sres[3] * u1 + dg1, s2 = sres[2] - sres[3] * u2 + dg2;
                float4 tot = {s0 * gi0.x + s1 * gi1.x + s2 * gi2.x, s0 * gi0.y + s1 * gi1.y + s2 * gi2.y,
                              s0 * gi0.z + s1 * gi1.z + s2 * gi2.z, 0.f};
                *(float4 *)(S.dGR + (k * 3 + r) * 4) = tot;
                const float k0 = r == 0 ? gi0.x : (r == 1 ? gi0.y : gi0.z), k1 = r == 0 ? gi1.x : (r == 1 ? gi1.y : gi1.z),
                            k2 = r == 0 ? gi2.x : (r == 1 ? gi2.y : gi2.z);
                S.dJ[k * 3 + r] = -(k0 * da0 + k1 * da1 + k2 * da2);
            }
        }
        BF_WAVE_FENCE();
        // d(pose feature) = sel_pd . dvp: all of it on wave 3 (dvp stays in its registers across the passes), in the shadow of
        // the row waves
        BF_MARK(46, 0, it, t_iter);
        if (wave == 3 && !merge_bc) dfeat_rows(strip, lq, npf, 64);
        if (wave == 3 && GBLEND) gmm_prior_grad();       // (all eight q values are in LDS since the barrier behind phase D)
        BF_MARK(47, 192, it, t_iter);
        BF_SYNC();
        } else {
        {
        // ================= phase F: reverse skinning of the selector vertices
        const int tq = bf_launder(tid);          // (fresh per phase: keeps this phase's address arithmetic out of the loop-invariant set)
        const int ci = tq / 3, cr = tq - ci * 3;
        if (c_on) {
            float dat = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
            if (NS > 0) {
                // every LDS read first (weights, dL/dvsel, the posed vertices as b128), then the arithmetic
                constexpr int NSC = NS > 0 ? NS : 1;
                float wv[NSC], dv[NSC];
                float4 vq[(NSC * 3 + 3) / 4];
#pragma unroll
                for (int sv = 0; sv < NSC; ++sv) { wv[sv] = S.sel_w[sv * nj + ci]; dv[sv] = S.dvsel[sv * 3 + cr]; }
#pragma unroll
                for (int q = 0; q < (NSC * 3 + 3) / 4; ++q) vq[q] = ((const float4 *)__builtin_assume_aligned(S.vp, 16))[q];
                __builtin_amdgcn_sched_barrier(0);
                const float *vpr = (const float *)vq;
#pragma unroll
                for (int sv = 0; sv < NSC; ++sv) {
                    float wd_ = wv[sv] * dv[sv];
                    dat += wd_;
                    r0 += wd_ * vpr[sv * 3]; r1 += wd_ * vpr[sv * 3 + 1]; r2 += wd_ * vpr[sv * 3 + 2];
                }
            } else {
                for (int sv = 0; sv < ns; ++sv) {
                    float wd_ = S.sel_w[sv * nj + ci] * S.dvsel[sv * 3 + cr];
                    dat += wd_;
                    r0 += wd_ * S.vp[sv * 3]; r1 += wd_ * S.vp[sv * 3 + 1]; r2 += wd_ * S.vp[sv * 3 + 2];
                }
            }
            if (ext) {                      // dense vertex losses: sum_v w_vj dv (x) [vp | 1] from bf_mesh_bwd_kernel
                const float *ea = ext + EXT_A + ci * 12 + cr * 4;
                r0 += ea[0]; r1 += ea[1]; r2 += ea[2]; dat += ea[3];
            }
            S.dAt[tq] = dat;
            const float gt_fin = S.dGt[tq] + (ext ? dat + ext[EXT_G + tq] : dat);     // + dL/d(chain joint) of the dense keypoint loss
            const float d0 = r0 - dat * S.J[ci * 3], d1 = r1 - dat * S.J[ci * 3 + 1], d2 = r2 - dat * S.J[ci * 3 + 2];   // row cr of D_i
            // Row cr of Dg_i = D_i GR_i^T and of M_i = Dg_i + dGt_i (Gt_i - Gt_0)^T.  With these, the sum over the strict
            // subtree of p of N_i = Dg_i + t_i (Gt_i - Gt_parent(i))^T (t_i itself a subtree sum) telescopes to
            //   sum_k M_k - (sum_k dGt_k) (Gt_p - Gt_0)^T,
            // two INDEPENDENT masked sums, so the next phase does both at once (one phase less than summing t_i first).
            const float4 gi0 = *(const float4 *)(S.G + ci * 12), gi1 = *(const float4 *)(S.G + ci * 12 + 4),
                         gi2 = *(const float4 *)(S.G + ci * 12 + 8);
            const float c0 = GT_(0, 0), c1 = GT_(0, 1), c2 = GT_(0, 2);
            const float dg0 = d0 * gi0.x + d1 * gi0.y + d2 * gi0.z;
            const float dg1 = d0 * gi1.x + d1 * gi1.y + d2 * gi1.z;
            const float dg2 = d0 * gi2.x + d1 * gi2.y + d2 * gi2.z;
            float4 drow = {dg0, dg1, dg2, 0.f};
            *(float4 *)(S.Dg + tq * 4) = drow;
            float4 mrow = {dg0 + gt_fin * (gi0.w - c0), dg1 + gt_fin * (gi1.w - c1), dg2 + gt_fin * (gi2.w - c2), gt_fin};
            *(float4 *)(S.N + tq * 4) = mrow;
        }
        if (tq >= 128 && tq < 133) {            // the projection phase's 16 pair slots -> d/dt, d/ds, loss value (slot order)
            const int q = tq - 128;
            float pq[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) pq[w] = S.part[w * 8 + q];
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) acc += pq[w];
            S.scal[3 + q] = nl > 0 ? acc : 0.f;               // (the slots are only written by the projection phase)
        }
        for (int idx = NG - 1 - tq; idx < ns3; idx += NG) {   // taken from the far end of the geometry waves
            int sv = idx / 3, b = idx - sv * 3;
            S.dvp[idx] = S.TR[sv * 9 + b] * S.dvsel[sv * 3] + S.TR[sv * 9 + 3 + b] * S.dvsel[sv * 3 + 1] +
                         S.TR[sv * 9 + 6 + b] * S.dvsel[sv * 3 + 2];
        }
        }
        BF_SYNC();

        {
        // ================= phase G (+H): per (joint p, row r) the two masked subtree sums (M rows and dL/dGt, one b128
        // read each), t_p, dL/dGR_p (total) = (Dg_p + sum_k M_k - st (Gt_p - Gt_0)^T) GR_p, direct dJ | d(pose feature)
        const int tq = bf_launder(tid);          // (fresh per phase: keeps this phase's address arithmetic out of the loop-invariant set)
        // (who takes what: with selector vertices d(pose feature) is real work and gets waves 2-3, the (joint, row) items waves 0-1 -
        //  two passes for 55 joints; without them - SMPL-X in the dense schedule - it is a copy of the outside block, wave 3 takes it
        //  alone and the 165 items fit waves 0-2 in ONE pass: this phase was 7.5 k of the resident launch's 18 k cycles per iteration)
        constexpr bool G_WIDE = NS == 0 && NJ > 0 && NJ * 3 <= 192;
        constexpr int G_ITEM_WAVES = G_WIDE ? 3 : 2, G_ITEM_THREADS = G_ITEM_WAVES * 64;
        if (wave >= G_ITEM_WAVES) dfeat_rows(S.dvp, tq - G_ITEM_THREADS, npf, NG - G_ITEM_THREADS);
        for (int q = tq; q < nj3 && wave < G_ITEM_WAVES; q += G_ITEM_THREADS) {
            const int p = q / 3, r = q - p * 3;
            // (one pass: q is this thread's own (joint, row), whose descendant mask has been in registers since the prologue - T.desc is
            //  in global memory, and a load of it at the top of this phase was ~2 k cycles of every iteration)
            const unsigned long long mk = G_WIDE ? cmask : T.desc[p];
            const unsigned mk_lo = (unsigned)mk, mk_hi = (unsigned)(mk >> 32);
            const float4 own = *(const float4 *)(S.N + q * 4), dgr = *(const float4 *)(S.Dg + q * 4);
            const float4 gp0 = *(const float4 *)(S.G + p * 12), gp1 = *(const float4 *)(S.G + p * 12 + 4),
                         gp2 = *(const float4 *)(S.G + p * 12 + 8);
            const float c0 = GT_(0, 0), c1 = GT_(0, 1), c2 = GT_(0, 2);
            const float da0 = S.dAt[p * 3], da1 = S.dAt[p * 3 + 1], da2 = S.dAt[p * 3 + 2];
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, st = 0.f;
            if (NJ > 0) {
                constexpr int NJC = NJ > 0 ? NJ : 1, HB = (NJC + 2) / 3;       // three batches of b128 reads (registers)
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    float4 nq[HB];
#pragma unroll
                    for (int i = 0; i < HB; ++i) nq[i] = *(const float4 *)(S.N + (min(h * HB + i, nj - 1) * 3 + r) * 4);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < HB; ++i) {
                        const int k = h * HB + i;
                        const bool in = (k < nj) && (((k < 32 ? mk_lo >> (k & 31) : mk_hi >> (k & 31)) & 1u) != 0u);     // (k is a constant here: one bit test)
                        s0 += in ? nq[i].x : 0.f; s1 += in ? nq[i].y : 0.f; s2 += in ? nq[i].z : 0.f; st += in ? nq[i].w : 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll 4
                for (int k = 0; k < nj; ++k) {
                    const float4 n = *(const float4 *)(S.N + (k * 3 + r) * 4);
                    const bool in = (mk >> k) & 1ull;
                    s0 += in ? n.x : 0.f; s1 += in ? n.y : 0.f; s2 += in ? n.z : 0.f; st += in ? n.w : 0.f;
                }
            }
            S.tt[q] = own.w + st;                                   // t_p: dL/dGt summed over the whole subtree of p
            s0 += dgr.x - st * (gp0.w - c0); s1 += dgr.y - st * (gp1.w - c1); s2 += dgr.z - st * (gp2.w - c2);
            float4 tot = {s0 * gp0.x + s1 * gp1.x + s2 * gp2.x, s0 * gp0.y + s1 * gp1.y + s2 * gp2.y,
                          s0 * gp0.z + s1 * gp1.z + s2 * gp2.z, 0.f};
            *(float4 *)(S.dGR + q * 4) = tot;
            const float k0 = r == 0 ? gp0.x : (r == 1 ? gp0.y : gp0.z), k1 = r == 0 ? gp1.x : (r == 1 ? gp1.y : gp1.z),
                        k2 = r == 0 ? gp2.x : (r == 1 ? gp2.y : gp2.z);
            S.dJ[q] = -(k0 * da0 + k1 * da1 + k2 * da2);
        }
        }
        BF_SYNC();

        }
        constexpr bool MERGE_IK = NJ == 24 && NB > 0 && NB <= 10 && NS > 0 && NS * 3 <= 36;     // plain axis-angle body (SMPL)
        if (MERGE_IK) {
        // ================= phase I (+K): the reverse sweep's last step and the Adam step in ONE phase, each parameter
        // stepped by the lane that finishes its gradient:
        //   wave 0, lane = joint: dL/dR_i, Rodrigues reverse, + GMM / angle prior, Adam for the joint's three pose dofs
        //   wave 1, lanes 0-3:    transl / scale from the pair-slot sums of the projection phase
        //   wave 3:               geometric dL/dbeta (6 lanes per beta, summed through a wave-private LDS strip), + shape
        //                         prior, Adam, then everything the next forward pass derives from the betas
        // Every consumer takes the arg-min GMM component from the eight q values itself (no hand-off between waves).
        const int tq = bf_launder(tid);
        auto adam = [&](int pidx, float pval, float am, float av, float grad) {
            // torch.optim.Adam, single-tensor path; v_sqrt_f32 / v_rcp_f32 (1 ulp) on the critical path
            if (mode == 1) { if (io.grads) io.grads[(size_t)frame * np + pidx] = grad; return; }
            am = am + (grad - am) * (1.0f - hp.beta1);
            av = av * hp.beta2 + (1.0f - hp.beta2) * grad * grad;
            const float denom = __builtin_amdgcn_sqrtf(av) * __builtin_amdgcn_rcpf(at2) + hp.eps;
            const float step = pidx < 4 ? at0 : at1;
            Pnext[pidx] = pval - step * (am * __builtin_amdgcn_rcpf(denom));
            S.am[pidx] = am; S.av[pidx] = av;
        };
        if (tq < nj) {
            // all LDS reads of the lane first
            const float4 g0 = *(const float4 *)(S.G + wp * 12), g1 = *(const float4 *)(S.G + wp * 12 + 4),
                         g2 = *(const float4 *)(S.G + wp * 12 + 8);
            const float4 c0 = *(const float4 *)(S.dGR + tq * 12), c1 = *(const float4 *)(S.dGR + tq * 12 + 4),
                         c2 = *(const float4 *)(S.dGR + tq * 12 + 8);
            const float t0 = S.tt[tq * 3], t1 = S.tt[tq * 3 + 1], t2 = S.tt[tq * 3 + 2];
            const float *dfp = S.dfeat + (tq > 0 ? tq - 1 : 0) * 9;
            float df[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) df[e] = dfp[e];
            const float th0 = S.theta[tq * 3], th1 = S.theta[tq * 3 + 1], th2 = S.theta[tq * 3 + 2];
            const float4 rc4 = *(const float4 *)(S.rc + tq * 4);
            float rcl[4] = {rc4.x, rc4.y, rc4.z, rc4.w};
            const int pb0 = tq > 0 ? 3 * (tq - 1) : 0;                       // body-pose dof of the joint's first component
            const int pi0 = tq > 0 ? T.off_pose + pb0 : T.off_orient;        // its parameter index
            float pv[3], pm[3], pw[3], pgm[3], pan[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pv[c] = Pcur[pi0 + c]; pm[c] = S.am[pi0 + c]; pw[c] = S.av[pi0 + c];
                pgm[c] = S.gtail[pb0 + c]; pan[c] = S.gtail[BF_GMM_LD + pb0 + c];          // the priors' terms (gmm_prior_grad, phase F)
            }
            __builtin_amdgcn_sched_barrier(0);
            float dRl[9], drl[3];
            if (tq == 0) {
                dRl[0] = c0.x; dRl[1] = c0.y; dRl[2] = c0.z; dRl[3] = c1.x; dRl[4] = c1.y; dRl[5] = c1.z;
                dRl[6] = c2.x; dRl[7] = c2.y; dRl[8] = c2.z;
                drl[0] = t0; drl[1] = t1; drl[2] = t2;
            } else {
                dRl[0] = g0.x * c0.x + g1.x * c1.x + g2.x * c2.x + df[0];
                dRl[1] = g0.x * c0.y + g1.x * c1.y + g2.x * c2.y + df[1];
                dRl[2] = g0.x * c0.z + g1.x * c1.z + g2.x * c2.z + df[2];
                dRl[3] = g0.y * c0.x + g1.y * c1.x + g2.y * c2.x + df[3];
                dRl[4] = g0.y * c0.y + g1.y * c1.y + g2.y * c2.y + df[4];
                dRl[5] = g0.y * c0.z + g1.y * c1.z + g2.y * c2.z + df[5];
                dRl[6] = g0.z * c0.x + g1.z * c1.x + g2.z * c2.x + df[6];
                dRl[7] = g0.z * c0.y + g1.z * c1.y + g2.z * c2.y + df[7];
                dRl[8] = g0.z * c0.z + g1.z * c1.z + g2.z * c2.z + df[8];
                drl[0] = g0.x * t0 + g1.x * t1 + g2.x * t2;
                drl[1] = g0.y * t0 + g1.y * t1 + g2.y * t2;
                drl[2] = g0.z * t0 + g1.z * t1 + g2.z * t2;
            }
            float gl[3];
            rodrigues_bwd(th0, th1, th2, rcl, dRl, gl);
            BF_MARK(50, 0, it, t_iter);
            if (mode == 1) {                         // (kept for the debug dump only)
#pragma unroll
                for (int e = 0; e < 9; ++e) S.dR[tq * 9 + e] = dRl[e];
                S.drel[tq * 3] = drl[0]; S.drel[tq * 3 + 1] = drl[1]; S.drel[tq * 3 + 2] = drl[2];
                S.gth[tq * 3] = gl[0]; S.gth[tq * 3 + 1] = gl[1]; S.gth[tq * 3 + 2] = gl[2];
            }
            // body-pose dofs: + GMM and angle priors (loss.py:54-61: dofs 52, 55, 9, 12), branch-free: the sign is 0 for every
            // other dof (and for the root joint), which zeroes the exponential term exactly.  The joint's three Adam steps
            // are written stage by stage so that their dependent chains (sqrt, two rcp) interleave.
            {
                const float body = tq > 0 ? 1.f : 0.f;          // (the root joint has no prior: its slots above are another joint's)
                float grad[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) grad[c] = gl[c] + body * pgm[c] + body * pan[c];
                if (mode == 1) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) if (io.grads) io.grads[(size_t)frame * np + pi0 + c] = grad[c];
                } else {
                    float sq[3], rd[3];
                    const float ir2 = __builtin_amdgcn_rcpf(at2);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { pm[c] = pm[c] + (grad[c] - pm[c]) * (1.0f - hp.beta1); pw[c] = pw[c] * hp.beta2 + (1.0f - hp.beta2) * grad[c] * grad[c]; }
#pragma unroll
                    for (int c = 0; c < 3; ++c) sq[c] = __builtin_amdgcn_sqrtf(pw[c]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) rd[c] = __builtin_amdgcn_rcpf(sq[c] * ir2 + hp.eps);
                    float pn[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { pn[c] = pv[c] - at1 * (pm[c] * rd[c]); Pnext[pi0 + c] = pn[c]; S.am[pi0 + c] = pm[c]; S.av[pi0 + c] = pw[c]; }
                    if (ROT_AHEAD && it + 1 < n_iters) rotations_ahead(tq, 0.f + pn[0], 0.f + pn[1], 0.f + pn[2]);
                }
            }
            BF_MARK(51, 0, it, t_iter);
        }
        if (tq >= 64 && tq < 68) {                                   // transl / scale
            const int pidx = tq - 64;
            const float pval = Pcur[pidx], am = S.am[pidx], av = S.av[pidx], psum = S.scal[3 + pidx], sc3 = Pcur[3];
            const float grad = psum * (pidx < 3 ? sc3 * cscale : cscale) + (ext ? ext[EXT_T + pidx] + ext[EXT_K + pidx] : 0.f);
            S.g[pidx] = grad;                                        // (kept for the debug dump)
            adam(pidx, pval, am, av, grad);
        }
        if (wave == 3) {
            // geometric part of dL/dbeta: sum Jd.dJ + Jdrel.drel + sel_sd.dvp with dL/drel_i = GR_p^T t_i formed inline;
            // lane = (beta l = lane / 6, slice sl = lane % 6): joints sl, sl + 6, sl + 12, sl + 18 and outputs sl + 6 m
            const int l = min(lane / 6, nb - 1), sl = lane - (lane / 6) * 6;
            const bool on = lane < 6 * nb;
            float acc = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                  // two joints at a time (registers)
                int pj[2];
                float tv[2][3], dj[2][3], jd[2][3], jr[2][3];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int i = sl + 6 * (2 * h + m);
                    pj[m] = S.par[i];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        tv[m][k] = S.tt[i * 3 + k]; dj[m][k] = S.dJ[i * 3 + k];
                        jd[m][k] = S.Jd[(i * 3 + k) * nbp + l]; jr[m][k] = S.Jdrel[(i * 3 + k) * nbp + l];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                float4 ga[2][3];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ga[m][0] = *(const float4 *)(S.G + pj[m] * 12); ga[m][1] = *(const float4 *)(S.G + pj[m] * 12 + 4);
                    ga[m][2] = *(const float4 *)(S.G + pj[m] * 12 + 8);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int i = sl + 6 * (2 * h + m);
                    float e0 = tv[m][0], e1 = tv[m][1], e2 = tv[m][2];
                    if (i > 0) {
                        e0 = ga[m][0].x * tv[m][0] + ga[m][1].x * tv[m][1] + ga[m][2].x * tv[m][2];
                        e1 = ga[m][0].y * tv[m][0] + ga[m][1].y * tv[m][1] + ga[m][2].y * tv[m][2];
                        e2 = ga[m][0].z * tv[m][0] + ga[m][1].z * tv[m][1] + ga[m][2].z * tv[m][2];
                    }
                    acc += jd[m][0] * dj[m][0] + jr[m][0] * e0;
                    acc += jd[m][1] * dj[m][1] + jr[m][1] * e1;
                    acc += jd[m][2] * dj[m][2] + jr[m][2] * e2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                float sw[6], sd[6];
#pragma unroll
                for (int m = 0; m < 6; ++m) { const int o = min(sl + 6 * m, ns3 - 1); sw[m] = S.sel_sd[o * nbp + l]; sd[m] = S.dvp[o]; }
#pragma unroll
                for (int m = 0; m < 6; ++m) acc += (sl + 6 * m < ns3) ? sw[m] * sd[m] : 0.f;
            }
            BF_MARK(52, 192, it, t_iter);
            float *strip = S.vpp + 4 * 64;                 // (the pose-blend strips of waves 0-3 are dead by now; this is a fifth)
            strip[lane] = on ? acc : 0.f;
            // this lane's beta (lanes 0..nb-1): value, moments
            const int pidx = T.off_beta + min(lane, nb - 1);
            const float pval = Pcur[pidx], am = S.am[pidx], av = S.av[pidx];
            BF_WAVE_FENCE();
            if (lane < nb) {
                float pr[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) pr[i] = strip[lane * 6 + i];
                float g = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i) g += pr[i];
                if (ext) g += ext[EXT_B + lane];
                S.g[pidx] = g;                                       // (kept for the debug dump)
                adam(pidx, pval, am, av, g + 2.f * hp.w_shape * pval);
            }
            BF_MARK(53, 192, it, t_iter);
            BF_WAVE_FENCE();
            beta_dependent(mode == 0 ? Pnext : Pcur);
            BF_MARK(54, 192, it, t_iter);
        }
        } else {
        {
        // ================= phase I: per joint (wave 0, lane = joint) dL/dR_i = GR_p^T dGR_i + d(pose feature), then the
        // Rodrigues reverse on the same lane | geometric part of dL/dbeta (waves 1-3), with dL/drel_i = GR_p^T t_i
        // formed inline so that nothing here waits for another wave
        const int tq = bf_launder(tid);          // (fresh per phase: keeps this phase's address arithmetic out of the loop-invariant set)
        if (tq < nj) {
            // all LDS reads of the lane first
            const float4 g0 = *(const float4 *)(S.G + wp * 12), g1 = *(const float4 *)(S.G + wp * 12 + 4),
                         g2 = *(const float4 *)(S.G + wp * 12 + 8);
            const float4 c0 = *(const float4 *)(S.dGR + tq * 12), c1 = *(const float4 *)(S.dGR + tq * 12 + 4),
                         c2 = *(const float4 *)(S.dGR + tq * 12 + 8);
            const float t0 = S.tt[tq * 3], t1 = S.tt[tq * 3 + 1], t2 = S.tt[tq * 3 + 2];
            const float *dfp = S.dfeat + (tq > 0 ? tq - 1 : 0) * 9;
            float df[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) df[e] = dfp[e];
            const float th0 = S.theta[tq * 3], th1 = S.theta[tq * 3 + 1], th2 = S.theta[tq * 3 + 2];
            const float4 rc4 = *(const float4 *)(S.rc + tq * 4);
            float rcl[4] = {rc4.x, rc4.y, rc4.z, rc4.w};
            __builtin_amdgcn_sched_barrier(0);
            float dRl[9], drl[3];
            if (tq == 0) {
                dRl[0] = c0.x; dRl[1] = c0.y; dRl[2] = c0.z; dRl[3] = c1.x; dRl[4] = c1.y; dRl[5] = c1.z;
                dRl[6] = c2.x; dRl[7] = c2.y; dRl[8] = c2.z;
                drl[0] = t0; drl[1] = t1; drl[2] = t2;
            } else {
                dRl[0] = g0.x * c0.x + g1.x * c1.x + g2.x * c2.x + df[0];
                dRl[1] = g0.x * c0.y + g1.x * c1.y + g2.x * c2.y + df[1];
                dRl[2] = g0.x * c0.z + g1.x * c1.z + g2.x * c2.z + df[2];
                dRl[3] = g0.y * c0.x + g1.y * c1.x + g2.y * c2.x + df[3];
                dRl[4] = g0.y * c0.y + g1.y * c1.y + g2.y * c2.y + df[4];
                dRl[5] = g0.y * c0.z + g1.y * c1.z + g2.y * c2.z + df[5];
                dRl[6] = g0.z * c0.x + g1.z * c1.x + g2.z * c2.x + df[6];
                dRl[7] = g0.z * c0.y + g1.z * c1.y + g2.z * c2.y + df[7];
                dRl[8] = g0.z * c0.z + g1.z * c1.z + g2.z * c2.z + df[8];
                drl[0] = g0.x * t0 + g1.x * t1 + g2.x * t2;
                drl[1] = g0.y * t0 + g1.y * t1 + g2.y * t2;
                drl[2] = g0.z * t0 + g1.z * t1 + g2.z * t2;
            }
            if (mode == 1) {                         // (kept for the debug dump only)
#pragma unroll
                for (int e = 0; e < 9; ++e) S.dR[tq * 9 + e] = dRl[e];
                S.drel[tq * 3] = drl[0]; S.drel[tq * 3 + 1] = drl[1]; S.drel[tq * 3 + 2] = drl[2];
            }
            float gl[3];
            rodrigues_bwd(th0, th1, th2, rcl, dRl, gl);
            S.gth[tq * 3] = gl[0]; S.gth[tq * 3 + 1] = gl[1]; S.gth[tq * 3 + 2] = gl[2];
        }
        if (wave >= 1 && wave < 4) {
            // 16 lanes per beta component: sum Jd.dJ + Jdrel.drel + sel_sd.dvp
            int q = tq - 64, l = q >> 4, sl = q & 15;
            float acc = 0.f;
            if (l < nb) {
                if (NJ > 0 && NJ <= 32 && NS > 0 && NS * 3 <= 48) {
                    // this lane's (at most) two joints and three selector outputs, reads in two batches
                    const int i0 = sl, i1 = sl + 16 < nj ? sl + 16 : 0;
                    const bool h1 = sl + 16 < nj;
                    const int p0 = S.par[i0], p1 = S.par[i1];
                    float tA[3], tB[3], jA[3], jB[3], rA[3], rB[3], dA[3], dB[3], sw[3], sd[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        tA[k] = S.tt[i0 * 3 + k]; tB[k] = S.tt[i1 * 3 + k];
                        dA[k] = S.dJ[i0 * 3 + k]; dB[k] = S.dJ[i1 * 3 + k];
                        jA[k] = S.Jd[(i0 * 3 + k) * nbp + l]; jB[k] = S.Jd[(i1 * 3 + k) * nbp + l];
                        rA[k] = S.Jdrel[(i0 * 3 + k) * nbp + l]; rB[k] = S.Jdrel[(i1 * 3 + k) * nbp + l];
                        const int o = min(sl + 16 * k, ns3 - 1);
                        sw[k] = S.sel_sd[o * nbp + l]; sd[k] = S.dvp[o];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float4 a0 = *(const float4 *)(S.G + p0 * 12), a1 = *(const float4 *)(S.G + p0 * 12 + 4),
                                 a2 = *(const float4 *)(S.G + p0 * 12 + 8);
                    const float4 b0 = *(const float4 *)(S.G + p1 * 12), b1 = *(const float4 *)(S.G + p1 * 12 + 4),
                                 b2 = *(const float4 *)(S.G + p1 * 12 + 8);
                    __builtin_amdgcn_sched_barrier(0);
                    float eA0 = tA[0], eA1 = tA[1], eA2 = tA[2];
                    if (i0 > 0) {
                        eA0 = a0.x * tA[0] + a1.x * tA[1] + a2.x * tA[2];
                        eA1 = a0.y * tA[0] + a1.y * tA[1] + a2.y * tA[2];
                        eA2 = a0.z * tA[0] + a1.z * tA[1] + a2.z * tA[2];
                    }
                    const float eB0 = b0.x * tB[0] + b1.x * tB[1] + b2.x * tB[2];
                    const float eB1 = b0.y * tB[0] + b1.y * tB[1] + b2.y * tB[2];
                    const float eB2 = b0.z * tB[0] + b1.z * tB[1] + b2.z * tB[2];
                    acc += jA[0] * dA[0] + rA[0] * eA0;
                    acc += jA[1] * dA[1] + rA[1] * eA1;
                    acc += jA[2] * dA[2] + rA[2] * eA2;
                    if (h1) {
                        acc += jB[0] * dB[0] + rB[0] * eB0;
                        acc += jB[1] * dB[1] + rB[1] * eB1;
                        acc += jB[2] * dB[2] + rB[2] * eB2;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc += (sl + 16 * k < ns3) ? sw[k] * sd[k] : 0.f;
                } else {
                    for (int i = sl; i < nj; i += 16) {
                        const int p = i > 0 ? S.par[i] : 0;
                        const float t0 = S.tt[i * 3], t1 = S.tt[i * 3 + 1], t2 = S.tt[i * 3 + 2];
                        float e0 = t0, e1 = t1, e2 = t2;
                        if (i > 0) {
                            const float4 g0 = *(const float4 *)(S.G + p * 12), g1 = *(const float4 *)(S.G + p * 12 + 4),
                                         g2 = *(const float4 *)(S.G + p * 12 + 8);
                            e0 = g0.x * t0 + g1.x * t1 + g2.x * t2;
                            e1 = g0.y * t0 + g1.y * t1 + g2.y * t2;
                            e2 = g0.z * t0 + g1.z * t1 + g2.z * t2;
                        }
                        const float *jd = S.Jd + i * 3 * nbp + l, *jr = S.Jdrel + i * 3 * nbp + l;
                        acc += jd[0] * S.dJ[i * 3] + jr[0] * e0;
                        acc += jd[nbp] * S.dJ[i * 3 + 1] + jr[nbp] * e1;
                        acc += jd[2 * nbp] * S.dJ[i * 3 + 2] + jr[2 * nbp] * e2;
                    }
                    for (int o = sl; o < ns3; o += 16) acc += S.sel_sd[o * nbp + l] * S.dvp[o];
                }
            }
            acc = row16_sum(acc);
            if (l < nb && sl == 0) S.g[T.off_beta + l] = ext ? acc + ext[EXT_B + l] : acc;
        }
        }
        BF_SYNC();

        {
        // ================= phase K: priors, gradient assembly, Adam (one parameter per thread).  Two batches of LDS
        // reads: (value, moments, descriptor), then every candidate gradient source at a clamped index; the parameter
        // kind selects afterwards.
        const int tq = bf_launder(tid);          // (fresh per phase: keeps this phase's address arithmetic out of the loop-invariant set)
        // The betas belong to wave 3 (lanes 0..nb-1) instead of the threads numbered like them: once they are stepped the
        // same wave rebuilds everything the next forward pass derives from them, in the shadow of the other updates.
        const bool isbeta = tq >= T.off_beta && tq < T.off_beta + nb;
        const int pidx = (tq < np && !isbeta) ? tq : ((wave == 3 && lane < nb) ? T.off_beta + lane : -1);
        float grad = 0.f, pval = 0.f, am = 0.f, av = 0.f;
        if (pidx >= 0) {
            pval = Pcur[pidx];
            am = S.am[pidx]; av = S.av[pidx];
            const int pk = S.pk[pidx], pa = S.pa_[pidx], pb = S.pb_[pidx];
            const int mstar = (int)S.scal[1];
            const float sc3 = Pcur[3];
            __builtin_amdgcn_sched_barrier(0);
            const int t8 = pidx < 8 ? pidx : 0;
            const float psum = S.scal[3 + (t8 < 5 ? t8 : 0)];
            const float gth_v = S.gth[pk == 1 ? pa : 0];
            const float gy_v = S.gy[mstar * BF_GMM_LD + (pb >= 0 && pb < BF_GMM_LD ? pb : 0)];
            const float g_v = S.g[pidx];
            __builtin_amdgcn_sched_barrier(0);
            // angle prior sign (loss.py:54-61: body dofs 52, 55, 9, 12)
            const float ang_sg = (pk == 1 && pb >= 0) ? (pb == 52 ? 1.f : ((pb == 55 || pb == 9 || pb == 12) ? -1.f : 0.f)) : 0.f;
            if (pk == 0) {                                           // transl / scale: the geometry waves' shares in wave order
                const float acc = psum;
                grad = acc * (pidx < 3 ? sc3 * cscale : cscale) + (ext ? ext[EXT_T + pidx] + ext[EXT_K + pidx] : 0.f);
                S.g[pidx] = grad;                                    // (kept for the debug dump)
            }
            else if (pk == 1) {
                grad = gth_v;
                if (pb >= 0) {                                       // body-pose dof pb: GMM + angle priors
                    grad += hp.w_pose * gy_v;
                    if (ang_sg != 0.f) { float e = __expf(pval * ang_sg); grad += hp.w_angle * 2.f * e * e * ang_sg; }
                }
            } else if (pk == 2) grad = g_v + 2.f * hp.w_shape * pval;
            else {                                                   // hand PCA coefficient pb of hand pa
                // (45 products, e ascending; the operands of fifteen requested before the first multiply-add: rolled, the loop was 45
                //  dependent pairs of LDS reads on the wave that holds the hand coefficients, in every iteration's Adam phase)
                const float *comp = S.hcomp + (pa * T.n_pca + pb) * 45;
                const float *gh = S.gth + (pa == 0 ? hand_j0_l : hand_j0_r) * 3;
                float acc = 0.f;
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    float cv[15], gv[15];
#pragma unroll
                    for (int e = 0; e < 15; ++e) { cv[e] = comp[h * 15 + e]; gv[e] = gh[h * 15 + e]; }
#pragma unroll
                    for (int e = 0; e < 15; ++e) acc += cv[e] * gv[e];
                }
                grad = acc;
            }
        }
        grad_last = grad;
        pidx_last = pidx;
        if (mode == 0 && pidx >= 0) {
            // torch.optim.Adam, single-tensor path
            am = am + (grad - am) * (1.0f - hp.beta1);
            av = av * hp.beta2 + (1.0f - hp.beta2) * grad * grad;
            // (v_sqrt_f32 / v_rcp_f32, 1 ulp each, on the critical path of every iteration)
            float denom = __builtin_amdgcn_sqrtf(av) * __builtin_amdgcn_rcpf(at2) + hp.eps;
            float step = pidx < 4 ? at0 : at1;
            Pnext[pidx] = pval - step * (am * __builtin_amdgcn_rcpf(denom));
            S.am[pidx] = am; S.av[pidx] = av;
        }
        if (wave == 3) {
            BF_WAVE_FENCE();
            beta_dependent(mode == 0 ? Pnext : Pcur);
        }
        }
        }
        BF_SYNC();
        if (mode == 0) { float *sw = Pcur; Pcur = Pnext; Pnext = sw; }
    }
    {   // ---- after the last iteration: loss terms (loss.py:219-224) and the pose state of the LAST forward pass, i.e. of
        // the parameters before the final Adam step (they sit in Pnext after the swap)
        const float *Pold = mode == 0 ? Pnext : Pcur;
        const float grad = grad_last;
        {
            if (tid == 0) {
                float *tm = io.terms + (size_t)frame * 4;
                if (!T.kp_dense) {                                   // (the dense keypoint kernel owns it otherwise)
                    float acc = 0.f;
                    acc = S.scal[3 + 4];
                    tm[0] = acc / ndiv_f;
                }
                tm[1] = hp.w_pose * S.scal[2];
            }
            if (tid == 64) {
                float acc = 0.f;
                const int ai[4] = {52, 55, 9, 12};
                const float as[4] = {1.f, -1.f, -1.f, -1.f};
                for (int k = 0; k < 4; ++k) {
                    float th = ai[k] < T.nbp ? Pold[T.off_pose + ai[k]] : 0.f;
                    float e = expf(th * as[k]);
                    acc += e * e;
                }
                io.terms[(size_t)frame * 4 + 2] = hp.w_angle * acc;
            }
            if (tid == 128) {
                float acc = 0.f;
                for (int l = 0; l < nb; ++l) acc += Pold[T.off_beta + l] * Pold[T.off_beta + l];
                io.terms[(size_t)frame * 4 + 3] = hp.w_shape * acc;
            }
            StateView st = bf_state_view(io.state + (size_t)frame * bf_state_stride(nj, npf, nb), nj, npf, nb);
            for (int i = tid; i < nj * 9; i += NG) { int j = i / 9, e = i - j * 9; st.GR[i] = GR_(j, e / 3, e % 3); }
            for (int i = tid; i < nj3; i += NG) { st.At[i] = S.At[i]; st.Gt[i] = GT_(i / 3, i % 3); st.theta[i] = S.theta[i]; }
            for (int p = tid; p < npf; p += NG) st.feat[p] = S.feat[p];
            if (tid < nb) st.beta[tid] = Pold[T.off_beta + tid];
            if (tid < 3) st.t[tid] = Pold[tid];
            if (tid == 3) { st.sc[0] = Pold[3]; st.sc[1] = cscale; }
            if (io.grads && pidx_last >= 0 && !(NJ == 24 && NB > 0 && NB <= 10 && NS > 0 && NS * 3 <= 36)) io.grads[(size_t)frame * np + pidx_last] = grad;
        }
        if (io.debug && frame == 0 && mode == 1) {
            float *d = io.debug;
            int o = 0;
            auto dump = [&](const float *src, int n) { for (int i = tid; i < n; i += NG) d[o + i] = src[i]; o += n; };
            dump(S.R, nj * 9); dump(S.J, nj3); dump(S.G, nj * 12); dump(S.vp, ns3);
            dump(S.vsel, ns3); dump(S.g, 4);
            for (int i = tid; i < nj * 9; i += NG) d[o + i] = S.dGR[(i / 3) * 4 + i % 3];
            o += nj * 9;
            dump(S.tt, nj3); dump(S.dR, nj * 9);
            dump(S.gth, nj3); dump(S.gq, BF_GMM_M); dump(S.dfeat, npf); dump(S.dJ, nj3); dump(S.drel, nj3);
        }
    }

    }

#ifdef BF_STAMP
    __syncthreads();
    if (tid < 64 && io.debug && frame == 0) io.debug[4096 + tid] = S.stamp[tid];
    if (tid < 192 && io.debug && frame == 0) io.debug[4352 + tid] = S.stamp[64 + tid];
#endif
    BF_KMARK(4, 0); BF_KMARK(5, 256);
    if (mode == 0 && tid < np) {
        io.params[(size_t)frame * np + tid] = Pcur[tid];
        io.adam_m[(size_t)frame * np + tid] = S.am[tid];
        io.adam_v[(size_t)frame * np + tid] = S.av[tid];
    }
}

// This file is compiled TWICE (Makefile): BF_FIT_PART 0 - the instances with compile-time sizes and everything on the host side, with
// the scheduler's max-ILP strategy - and BF_FIT_PART 1 (fit_kernels_table.o) - the two table-driven instances behind
// bf_fit_launch_table, with the default scheduler: same box, max-ILP is 2 % faster on the sized SMPL instance and 8 % SLOWER on the
// table-driven one (837 against 909 frames/s).
#ifndef BF_FIT_PART
#define BF_FIT_PART 0
#endif

// launches one of two instantiations; `slot`: the per-device cache entry of the dynamic-LDS attribute already set for it
template <class K>
static hipError_t fit_launch_one(K kern, size_t *have, const FitTab *T, const FrameIO *io, const HyperDev *hp, int n_iters, int mode,
                                 const float *adam_tab, int adam_t0, size_t smem, hipStream_t stream, hipEvent_t done) {
    if (smem > 64 * 1024 && smem > *have) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *have = smem;
    }
    // (`done`: an event that completes with THIS dispatch - its own completion signal, no marker packet behind it on the queue)
    if (done) hipExtLaunchKernelGGL(kern, dim3(io->n_frames), dim3(BF_FIT_THREADS), smem, stream, nullptr, done, 0, *T, *io, *hp, n_iters, mode, adam_tab, adam_t0);
    else hipLaunchKernelGGL(kern, dim3(io->n_frames), dim3(BF_FIT_THREADS), smem, stream, *T, *io, *hp, n_iters, mode, adam_tab, adam_t0);
    return hipGetLastError();
}

#if BF_FIT_PART == 1
// the table-driven instances (any model the sized ones do not cover)
extern "C" hipError_t bf_fit_launch_table(const FitTab *T, const FrameIO *io, const HyperDev *hp, int n_iters, int mode,
                                          const float *adam_tab, int adam_t0, size_t smem, hipStream_t stream, hipEvent_t done) {
    // hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: the cache of what was set is keyed by device
    static size_t attr[16][2] = {};
    size_t none = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    const bool ext = io->ext != nullptr;
    size_t *have = (dev >= 0 && dev < 16) ? &attr[dev][ext ? 1 : 0] : &none;
    if (ext) return fit_launch_one(fit_kernel<0, 0, 0, 0, true>, have, T, io, hp, n_iters, mode, adam_tab, adam_t0, smem, stream, done);
    return fit_launch_one(fit_kernel<0, 0, 0, 0, false>, have, T, io, hp, n_iters, mode, adam_tab, adam_t0, smem, stream, done);
}
#else
extern "C" hipError_t bf_fit_launch_table(const FitTab *, const FrameIO *, const HyperDev *, int, int, const float *, int, size_t, hipStream_t, hipEvent_t);

// The three contiguous runs of model-constant arrays in the carve, as (first float4, float4 count): Jtrel .. nzj | sel_pd2 .. par |
// pk .. pb_ (each run may contain a scratch array or two; copying them is cheaper than splitting the run)
extern "C" void bf_fit_image_segments(int nj, int nb, int npf, int ns, int nl, int np, int seg[6]) {
    FitSmem s;
    float *base = (float *)(uintptr_t)4096;
    fit_smem_carve(s, base, nj, nb, npf, ns, nl, np, 0);
    auto off4 = [&](const void *p) { return (int)(((const float *)p - base) / 4); };
    seg[0] = off4(s.Jtrel); seg[1] = off4(s.kp) - seg[0];
    seg[2] = off4(s.sel_pd2); seg[3] = off4(s.pa) - seg[2];
    seg[4] = off4(s.pk); seg[5] = off4(s.am) - seg[4];
}

extern "C" size_t bf_fit_smem_bytes(int nj, int nb, int npf, int ns, int nl, int np, int nviews) {
    FitSmem s;
    return fit_smem_carve(s, nullptr, nj, nb, npf, ns, nl, np, nviews, false);
}

// (the compile-time-sized SMPL instance also assumes at most 4 bones per selector vertex - true of SMPL's skinning weights)
extern "C" bool bf_fit_is_sized_smpl(const FitTab *T) {
    return T->nj == 24 && T->nb == 10 && T->ns == 11 && T->nl == 25 && T->sel_nnz > 0 && T->sel_nnz <= 4 && T->bd_ok;
}

// Host-side launcher: picks the compile-time-sized instantiation for SMPL, the table-driven one otherwise.
extern "C" hipError_t bf_fit_launch(const FitTab *T, const FrameIO *io, const HyperDev *hp, int n_iters, int mode,
                                    const float *adam_tab, int adam_t0, size_t smem, hipStream_t stream, hipEvent_t done) {
    const bool smpl = bf_fit_is_sized_smpl(T);
    const bool ext = io->ext != nullptr;
    // SMPL-X in the dense schedule (keypoints through bf_kp_loss_kernel: no selector vertices, no loss joints here): sizes fixed
    // at compile time like SMPL's, the phases stay the table-driven ones
    const bool smplx_dense = ext && T->nj == 55 && T->nb == 10 && T->ns == 0 && T->nl == 0;
    if (!smpl && !smplx_dense) return bf_fit_launch_table(T, io, hp, n_iters, mode, adam_tab, adam_t0, smem, stream, done);
    // hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: the cache of what was set is keyed by device
    static size_t attr[16][3] = {};
    size_t none = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    size_t *have = (dev >= 0 && dev < 16) ? &attr[dev][smplx_dense ? 2 : (ext ? 1 : 0)] : &none;
    if (smplx_dense) return fit_launch_one(fit_kernel<55, 10, 0, 0, true>, have, T, io, hp, n_iters, mode, adam_tab, adam_t0, smem, stream, done);
    if (ext) return fit_launch_one(fit_kernel<24, 10, 11, 25, true>, have, T, io, hp, n_iters, mode, adam_tab, adam_t0, smem, stream, done);
    return fit_launch_one(fit_kernel<24, 10, 11, 25, false>, have, T, io, hp, n_iters, mode, adam_tab, adam_t0, smem, stream, done);
}
#endif
