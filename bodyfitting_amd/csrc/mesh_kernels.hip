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
#include <hip/hip_ext.h>
#include "pose_state_body.h"
#include "joints_body.h"
#include "loss_bodies.h"
#include <cstring>

namespace {
__device__ inline float m_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
}  // namespace

// One 128-thread workgroup per parameter set (body: pose_state_body.h).
// A pose state published by the resident fit launch, read by a kernel that waited for its doorbell.  `coherent` (bit 30 of the
// door target): device-scope loads that are served past this XCD's non-coherent caches - correctness then no longer rests on
// no line of `state` having entered this XCD's L2 since the kernel started (measured cost in DESIGN.md 4.3).
#define BF_DOOR_COHERENT_BIT 0x40000000
__device__ __forceinline__ float bf_ld_state(const float *p, bool coherent) {
    return coherent ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

extern "C" __global__ void __launch_bounds__(128)
bf_pose_state_kernel(FitTab T, const float *__restrict__ betas, const float *__restrict__ orient,
                     const float *__restrict__ body_pose, const float *__restrict__ sim, float *state,
                     const float *__restrict__ packed, const float *__restrict__ cscale, float cscale_all) {
    __shared__ float lds[BF_POSE_STATE_LDS];
    if (packed) bf_pose_state_body<true>(T, betas, orient, body_pose, sim, state, packed, cscale, cscale_all, blockIdx.x, threadIdx.x, 128, lds, bf_pose_tabs(T));
    else bf_pose_state_body<false>(T, betas, orient, body_pose, sim, state, packed, cscale, cscale_all, blockIdx.x, threadIdx.x, 128, lds, bf_pose_tabs(T));
}

// grid (ceil(NV/32), F), block 96 x 8.  vraw = model-space vertices (lbs output), vout = (v + t) s c.
// Every operand a thread will need is requested before the first barrier (posedirs slice by all
// threads; lbs_weights row by row-group 0, shapedirs + template by row-group 1, the extra-regressor
// slice by row-group 2), so the kernel pays one memory round trip instead of one per phase.
#define BF_MESH_PF 26      // posedirs rows prefetched per thread (= ceil(207 / 8) for SMPL); more are streamed
#define BF_MESH_WPF 24     // lbs weights prefetched per vertex thread
// (SPAN: the instance bf_batch_mesh_span launches - every workgroup leaves its start and end on the 100 MHz wall clock in span[0]
//  (minimum) / span[1] (maximum): the kernel's own duration, with no event record or launch gap in it)
template <bool SPAN>
__device__ __forceinline__ void bf_mesh_body(const MeshTab &M, const float *__restrict__ state, float *__restrict__ vraw, float *__restrict__ vout,
               float *__restrict__ xpart, float *__restrict__ vposed, const float *__restrict__ pose_off, int *door, int door_target,
               unsigned long long *span) {
    const unsigned long long t_begin = SPAN ? (unsigned long long)wall_clock64() : 0ull;
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
    const int frame = blockIdx.y, tile = blockIdx.x;
    StateView st = bf_state_view(const_cast<float *>(state) + (size_t)frame * bf_state_stride(nj, npf, nb), nj, npf, nb);
    const int ncols = 3 * nv;
    const int gcol = tile * COLS + col;
    const bool ok = gcol < ncols;
    const int vl = col / 3, k = col - vl * 3, v = tile * BF_MESH_TILE + vl;

    // ---- request everything ------------------------------------------------------------------
    const int rows = (npf + BF_MESH_RG - 1) / BF_MESH_RG;
    const int p0 = rg * rows, p1 = min(npf, p0 + rows);
    const int pdp = M.pd_pitch;                 // (floats between posedirs rows: 3 nv rounded up to 128 bytes, so that a tile's 384-byte slice of a row is three whole lines)
    const float *pd = M.posedirs + (size_t)p0 * pdp + (ok ? gcol : 0);
    float pv[BF_MESH_PF];
#pragma unroll
    for (int i = 0; i < BF_MESH_PF; ++i) pv[i] = (!pose_off && p0 + i < p1) ? pd[(size_t)i * pdp] : 0.f;
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
    const bool sparse4 = M.v_nnz == 4;             // four bones per vertex (SMPL): exact, the dropped weights are zeros
    int zj[4] = {0, 0, 0, 0};
    if (rg == 0 && ok) {
        if (sparse4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { zj[q] = M.v_nzj[(size_t)v * 4 + q] * 12; wreg[q] = M.v_nzw[(size_t)v * 4 + q]; }
        } else {
            const float *w = M.lbs_weights + (size_t)v * nj;
#pragma unroll
            for (int j = 0; j < BF_MESH_WPF; ++j) if (j < nj) wreg[j] = w[j];
        }
    } else if (rg == 1 && ok) {
        const float *sd = M.shapedirs + (size_t)gcol * nb;
#pragma unroll
        for (int l = 0; l < 12; ++l) if (l < nb) sdreg[l] = sd[l];
        vt = M.v_template[gcol];
    } else if (rg == 2 && col < ne3 && xpart) {
        const int e = col / 3, v0 = tile * BF_MESH_TILE;
        const float *row = M.j_extra + (size_t)e * nv + v0;
#pragma unroll
        for (int i = 0; i < BF_MESH_TILE; ++i) if (v0 + i < nv) jx[i] = row[i];
    }
    const bool coh = door && (door_target & BF_DOOR_COHERENT_BIT);
    if (door) {        // the pose state comes from the persistent fit launch (BfDoor): wait for it under the requests above
        if (tid == 0) bf_door_wait(door, BF_DOOR_STATE + (int)((blockIdx.x + 7 * blockIdx.y) % BF_DOOR_COPIES) * BF_DOOR_COPY_STRIDE, door_target & ~BF_DOOR_COHERENT_BIT);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // (cold caches at kernel start, first read of the state below)
    }
    for (int i = tid; i < npf; i += nt) s_feat[i] = bf_ld_state(st.feat + i, coh);
    for (int i = tid; i < nj * 12; i += nt) {
        int j = i / 12, e = i % 12, a = e / 4, b = e % 4;
        s_A[i] = bf_ld_state(b < 3 ? st.GR + j * 9 + a * 3 + b : st.At + j * 3 + a, coh);
    }
    if (tid < nb + 5) s_beta[tid] = bf_ld_state(st.beta + tid, coh);   // beta, t, sc are contiguous in the state record
    __syncthreads();

    // ---- pose blend: this row group's share, reduced through LDS -----------------------------------
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < BF_MESH_PF; ++i) acc += (p0 + i < p1 ? s_feat[p0 + i] : 0.f) * pv[i];
    if (!pose_off) for (int p = p0 + BF_MESH_PF; p < p1; ++p) acc += s_feat[p] * pd[(size_t)(p - p0) * pdp];
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
            if (sparse4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = *(const float4 *)(s_A + zj[q] + k * 4);
                    t0 += wreg[q] * a.x; t1 += wreg[q] * a.y; t2 += wreg[q] * a.z; tt += wreg[q] * a.w;
                }
            } else {
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
            xpart[((size_t)frame * gridDim.x + tile) * ne3 + col] = a3;
        }
    }
    if (SPAN) {
        __syncthreads();                                   // (the workgroup's stores are issued; the clock is read after them)
        if (tid == 0) { atomicMin(span, t_begin); atomicMax(span + 1, (unsigned long long)wall_clock64()); }
    }
}
extern "C" __global__ void __launch_bounds__(BF_MESH_TILE * 3 * BF_MESH_RG)
bf_mesh_kernel(MeshTab M, const float *__restrict__ state, float *__restrict__ vraw, float *__restrict__ vout,
               float *__restrict__ xpart, float *__restrict__ vposed, const float *__restrict__ pose_off, int *door, int door_target) {
    bf_mesh_body<false>(M, state, vraw, vout, xpart, vposed, pose_off, door, door_target, nullptr);
}
extern "C" __global__ void __launch_bounds__(BF_MESH_TILE * 3 * BF_MESH_RG)
bf_mesh_span_kernel(MeshTab M, const float *__restrict__ state, float *__restrict__ vraw, float *__restrict__ vout,
                    float *__restrict__ xpart, unsigned long long *span) {
    bf_mesh_body<true>(M, state, vraw, vout, xpart, nullptr, nullptr, nullptr, 0, span);
}

// Small batches (2..15 frames) and models whose pose feature is longer than the 8 x BF_MESH_PF rows the kernel above
// keeps in flight (SMPL-X: 486): ONE workgroup streams a tile's posedirs slice once for up to FPW frames.
// grid (n_tiles, ceil(F / FPW)), block 96 x 8 as above.  The rows of a row group are requested in chunks of BF_MM_CH,
// every chunk's loads in flight together, and each loaded value feeds one fma per frame (pose features of the frames sit
// frame-minor in LDS: one b128 read serves four frames).  After the split-K reduce the eight row groups become eight
// FRAMES: row group f shape-blends, skins and stores frame f, so the epilogue keeps all 768 threads busy.
// Arithmetic and its order per frame are those of bf_mesh_kernel (rows ascending inside a row group, row groups
// ascending, template + shape offset first, pose offset added to it), so a frame's result does not depend on the
// batch it was in.
// Round 6: the eight-frame instance asks for its rows in chunks of 8 (two chunks in flight) instead of 64 and for six waves per SIMD:
// 66 registers instead of 154, so TWO workgroups fit a CU and config 5's 328 tiles are all resident at once - at one workgroup per CU
// the last 72 ran as a second round with nothing to hide their stream behind (forward pass 37.6 -> 31 us per dense iteration, config 5's
// fit 54.9 -> 53.5 ms, same box; chunks of 12: the same; chunks of 16 need 90 registers - held to 80 they spill and the pass takes 44 us).
// The bytes in flight per CU are what they were (24 waves x 16 rows instead of 12 x 64); the one-, two- and four-frame instances keep 64.
#ifndef BF_MM_CH8
#define BF_MM_CH8 8
#endif
#ifndef BF_MM_OCC8
#define BF_MM_OCC8 6
#endif
#define BF_MM_CH (FPW == 8 ? BF_MM_CH8 : 64)
template <int FPW>
__global__ void __launch_bounds__(BF_MESH_TILE * 3 * BF_MESH_RG, FPW == 8 ? BF_MM_OCC8 : 1)
bf_mesh_multi_kernel(MeshTab M, const float *__restrict__ state, int n_frames, float *__restrict__ vraw, float *__restrict__ vout,
                     float *__restrict__ xpart, float *__restrict__ vposed, float *__restrict__ dvzero, MaskProj mp, int *door, int door_target) {
    static_assert(FPW == 1 || FPW == 2 || FPW == 4 || FPW == 8, "frames per workgroup");
    constexpr int COLS = BF_MESH_TILE * 3;
    extern __shared__ __align__(16) float sm[];
    const int nj = M.nj, nb = M.nb, npf = M.npf, nv = M.nv;
    const int rows = (npf + BF_MESH_RG - 1) / BF_MESH_RG, npad = rows * BF_MESH_RG;
    float *s_feat = sm;                                  // [npad][FPW]  (frame-minor; rows beyond npf are zero)
    float *s_A = s_feat + npad * FPW;                    // [FPW][nj][12]
    float *s_red = s_A + FPW * nj * 12;                  // [FPW][RG][COLS]
    float *s_vp = s_red + FPW * BF_MESH_RG * COLS;       // [FPW][COLS]
    float *s_beta = s_vp + FPW * COLS;                   // [FPW][32]: beta[nb], t[3], sc[2]
    const int tid = threadIdx.x, nt = COLS * BF_MESH_RG;
    const int col = tid % COLS, rg = tid / COLS;
    const int tile = blockIdx.x, fbase = blockIdx.y * FPW, nf = min(FPW, n_frames - fbase);
    const int ncols = 3 * nv, gcol = tile * COLS + col;
    const bool ok = gcol < ncols;
    const int vl = col / 3, k = col - vl * 3, v = tile * BF_MESH_TILE + vl;
    const size_t sstride = bf_state_stride(nj, npf, nb);

    // ---- first chunk of the stream, then the per-frame state into LDS ----------------------------------------------
    const int p0 = rg * rows, p1 = min(npf, p0 + rows);
    const int pdp = M.pd_pitch;
    const float *pd = M.posedirs + (size_t)p0 * pdp + (ok ? gcol : 0);
    constexpr int CH = BF_MM_CH;
    float pv[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) pv[i] = p0 + i < p1 ? pd[(size_t)i * pdp] : 0.f;
    const bool coh = door && (door_target & BF_DOOR_COHERENT_BIT);
    // ... and so are the model rows the epilogue needs for this thread's vertex coordinate (shape directions, template, the sparse
    // skinning row): they depend on nothing that is waited for below, and requested here their two memory round trips are over
    // before the pose blend is (they used to start behind it, on the 96 threads that do the epilogue of a frame)
    // (one frame per workgroup only: with eight, the 28 registers of the rows take the kernel from 3 to 2 waves per SIMD - config 5's
    //  iteration went from 0.202 to 0.214 ms)
    const bool pre = FPW == 1 && rg < nf && ok && nb <= 10 && (M.v_nnz == 4 || M.v_nnz == 8);
    float pre_sd[10], pre_vt = 0.f, pre_w[8];
    int pre_j[8];
    if (pre) {
        const float *sd = M.shapedirs + (size_t)gcol * nb;
#pragma unroll
        for (int l = 0; l < 10; ++l) pre_sd[l] = l < nb ? sd[l] : 0.f;
        pre_vt = M.v_template[gcol];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            pre_w[q] = q < M.v_nnz ? M.v_nzw[(size_t)v * M.v_nnz + q] : 0.f;
            pre_j[q] = q < M.v_nnz ? M.v_nzj[(size_t)v * M.v_nnz + q] : 0;
        }
    }
    const int door_copy = (int)((blockIdx.x + 7 * blockIdx.y) % BF_DOOR_COPIES) * BF_DOOR_COPY_STRIDE;
    if (door) {        // the pose FEATURES come first from the persistent fit launch (BF_DOOR_FEAT): wait for all of them, under the first chunk's loads
        if (tid == 0) bf_door_wait(door, BF_DOOR_FEAT + door_copy, door_target & ~BF_DOOR_COHERENT_BIT);
        __syncthreads();
        // (this kernel started with invalidated caches and reads the states for the first time below: nothing stale to drop)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int i = tid; i < npad * FPW; i += nt) {
        const int p = i / FPW, f = i - p * FPW;
        s_feat[i] = (p < npf && f < nf) ? bf_ld_state(bf_state_view(const_cast<float *>(state) + (fbase + f) * sstride, nj, npf, nb).feat + p, coh) : 0.f;
    }
    // (without doorbells the whole state is there: everything is loaded now, as before; with them the chain matrices and the
    //  betas are fetched behind the pose blend, when BF_DOOR_STATE has been rung - the fit launch is still forming them)
    if (!door)
    for (int i = tid; i < FPW * nj * 12; i += nt) {
        const int f = i % FPW, r = i / FPW, j = r / 12, e = r - j * 12, a = e >> 2, b = e & 3;       // (FPW, 12: compile-time divisors)
        float x = 0.f;
        if (f < nf) {
            StateView st = bf_state_view(const_cast<float *>(state) + (fbase + f) * sstride, nj, npf, nb);
            x = bf_ld_state(b < 3 ? st.GR + j * 9 + a * 3 + b : st.At + j * 3 + a, coh);
        }
        s_A[f * nj * 12 + r] = x;
    }
    if (!door)
    for (int i = tid; i < FPW * 32; i += nt) {
        const int f = i >> 5, l = i & 31;
        s_beta[i] = (f < nf && l < nb + 5) ? bf_ld_state(bf_state_view(const_cast<float *>(state) + (fbase + f) * sstride, nj, npf, nb).beta + l, coh) : 0.f;
    }
    __syncthreads();

    // ---- pose blend: the row group's rows, chunk by chunk, one fma per frame and loaded value ------------------------
    float acc[FPW];
#pragma unroll
    for (int f = 0; f < FPW; ++f) acc[f] = 0.f;
    for (int c0 = 0; c0 < rows; c0 += CH) {
        float nx[CH];
        const bool more = c0 + CH < rows;
        if (more) {
#pragma unroll
            for (int i = 0; i < CH; ++i) nx[i] = p0 + c0 + CH + i < p1 ? pd[(size_t)(c0 + CH + i) * pdp] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (c0 + i < rows) {                             // (s_feat rows up to npad exist and are zero beyond npf)
                const float *fr = s_feat + (size_t)(p0 + c0 + i) * FPW;
                if constexpr (FPW >= 4) {
#pragma unroll
                    for (int f4 = 0; f4 < FPW; f4 += 4) {
                        const float4 x = *(const float4 *)(fr + f4);
                        acc[f4] += x.x * pv[i]; acc[f4 + 1] += x.y * pv[i]; acc[f4 + 2] += x.z * pv[i]; acc[f4 + 3] += x.w * pv[i];
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < FPW; ++f) acc[f] += fr[f] * pv[i];
                }
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < CH; ++i) pv[i] = nx[i];
        }
    }
#pragma unroll
    for (int f = 0; f < FPW; ++f) s_red[(f * BF_MESH_RG + rg) * COLS + col] = acc[f];
    if (door) {        // the rest of the states: chain matrices A_j, betas, similarity
        if (tid == 0) bf_door_wait(door, BF_DOOR_STATE + door_copy, door_target & ~BF_DOOR_COHERENT_BIT);
        __syncthreads();
        for (int i = tid; i < FPW * nj * 12; i += nt) {
            const int f = i % FPW, r = i / FPW, j = r / 12, e = r - j * 12, a = e >> 2, b = e & 3;
            float x = 0.f;
            if (f < nf) {
                StateView st = bf_state_view(const_cast<float *>(state) + (fbase + f) * sstride, nj, npf, nb);
                x = bf_ld_state(b < 3 ? st.GR + j * 9 + a * 3 + b : st.At + j * 3 + a, coh);
            }
            s_A[f * nj * 12 + r] = x;
        }
        for (int i = tid; i < FPW * 32; i += nt) {
            const int f = i >> 5, l = i & 31;
            s_beta[i] = (f < nf && l < nb + 5) ? bf_ld_state(bf_state_view(const_cast<float *>(state) + (fbase + f) * sstride, nj, npf, nb).beta + l, coh) : 0.f;
        }
        __syncthreads();
    }
    // ---- from here on row group rg works for FRAME rg -----------------------------------------------------------------
    const int f = rg;
    const bool mine = f < nf;
    float a2 = 0.f, vt = 0.f;
    if (mine && ok) {
        if (pre) {
            vt = pre_vt;
#pragma unroll
            for (int l = 0; l < 10; ++l) if (l < nb) a2 += pre_sd[l] * s_beta[f * 32 + l];
        } else {
            const float *sd = M.shapedirs + (size_t)gcol * nb;
            vt = M.v_template[gcol];
            for (int l = 0; l < nb; ++l) a2 += sd[l] * s_beta[f * 32 + l];
        }
    }
    __syncthreads();
    if (mine) {
        float off = 0.f;
#pragma unroll
        for (int q = 0; q < BF_MESH_RG; ++q) off += s_red[(f * BF_MESH_RG + q) * COLS + col];
        float vp = ok ? vt + a2 : 0.f;
        vp += off;
        s_vp[f * COLS + col] = vp;
    }
    __syncthreads();
    // ---- skinning ---------------------------------------------------------------------------------------------------
    float r = 0.f;
    if (mine && ok) {
        const float *A = s_A + f * nj * 12;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, tt = 0.f;
        if (pre) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < M.v_nnz) {
                    const float w = pre_w[q];
                    const float4 a = *(const float4 *)(A + pre_j[q] * 12 + k * 4);
                    t0 += w * a.x; t1 += w * a.y; t2 += w * a.z; tt += w * a.w;
                }
        } else if (M.v_nnz) {
            const int nnz = M.v_nnz;
            for (int q = 0; q < nnz; ++q) {
                const float w = M.v_nzw[(size_t)v * nnz + q];
                const float4 a = *(const float4 *)(A + M.v_nzj[(size_t)v * nnz + q] * 12 + k * 4);
                t0 += w * a.x; t1 += w * a.y; t2 += w * a.z; tt += w * a.w;
            }
        } else {
            for (int j = 0; j < nj; ++j) {
                const float w = M.lbs_weights[(size_t)v * nj + j];
                const float4 a = *(const float4 *)(A + j * 12 + k * 4);
                t0 += w * a.x; t1 += w * a.y; t2 += w * a.z; tt += w * a.w;
            }
        }
        const float *vp = s_vp + f * COLS + vl * 3;
        r = t0 * vp[0] + t1 * vp[1] + t2 * vp[2] + tt;
        const size_t o = (size_t)(fbase + f) * ncols + gcol;
        const float *sb = s_beta + f * 32 + nb;
        if (vraw) vraw[o] = r;
        if (vout) vout[o] = (r + sb[k]) * sb[3] * sb[4];
        if (vposed) vposed[o] = vp[k];
        if (dvzero) dvzero[o] = 0.f;                          // dL/dvertices starts the iteration at zero: saves a memset launch
    }
    if (mp.on) {
        // silhouette loss: the tile's sampled vertices (every 4th, loss.py:99) projected into every mask view while they are at hand -
        // bf_mask_project_kernel's arithmetic on the value just stored to vout (no launch of its own)
        __syncthreads();                                     // (s_red is free: every sum above has been taken)
        if (mine) {
            const float *sb = s_beta + f * 32 + nb;
            s_red[f * COLS + col] = ok ? (r + sb[k]) * sb[3] * sb[4] : 0.f;
        }
        __syncthreads();
        {
            // (every thread of the workgroup takes (frame, sample, view) items - with one frame per workgroup only the 96 threads of row
            //  group 0 used to: three passes of projection + four dependent mask reads each instead of one)
            const int sst = mp.K.sstride, spt = BF_MESH_TILE / sst;        // samples of a tile: every 4th vertex, or (sub-model) all that are samples
            const int per = spt * mp.K.n_masks;
            for (int idx = tid; idx < per * nf; idx += nt) {
                const int ff = idx / per, q = idx - ff * per;
                const int sv = q % spt, m = q / spt, vv = tile * BF_MESH_TILE + sv * sst, sidx = sst == 4 ? vv >> 2 : vv;
                if (vv < nv && sidx < mp.K.ns) {
                    const float *X = s_red + ff * COLS + sv * sst * 3;
                    (void)bf_mask_project_one(mp.K, X[0], X[1], X[2], mp.proj, fbase + ff, m, sidx, mp.uvi, mp.duvb);
                }
            }
        }
    }
    if (xpart) {
        // this tile's share of J_regressor_extra . vertices (models/smpl.py:72), per frame
        const int ne3 = M.n_extra * 3;
        __syncthreads();                                     // (s_red is free: every sum above has been taken)
        if (mine) s_red[f * COLS + col] = r;
        __syncthreads();
        if (mine && col < ne3) {
            const int e = col / 3, v0 = tile * BF_MESH_TILE;
            const float *row = M.j_extra + (size_t)e * nv + v0;
            float jx[BF_MESH_TILE];
#pragma unroll
            for (int i = 0; i < BF_MESH_TILE; ++i) jx[i] = v0 + i < nv ? row[i] : 0.f;
            float a3 = 0.f;
#pragma unroll
            for (int i = 0; i < BF_MESH_TILE; ++i) a3 += jx[i] * s_red[f * COLS + i * 3 + k];
            xpart[((size_t)(fbase + f) * gridDim.x + tile) * ne3 + col] = a3;
        }
    }
}

extern "C" int bf_mesh_multi_launch(const MeshTab *M, const float *state, int n, float *vraw, float *vout, float *xpart, float *vposed,
                                    float *dvzero, hipStream_t stream, const MaskProj *mproj, int *door, int door_target, hipEvent_t done) {
    // (`done`: an event that completes with this dispatch - its own completion signal instead of a marker packet behind it)
    MaskProj mp;
    if (mproj) mp = *mproj; else { std::memset(&mp, 0, sizeof mp); }
    constexpr int COLS = BF_MESH_TILE * 3;
    const int fpw = n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : 8));
    const int rows = (M->npf + BF_MESH_RG - 1) / BF_MESH_RG;
    const size_t smem = sizeof(float) * ((size_t)rows * BF_MESH_RG * fpw + (size_t)fpw * M->nj * 12 + (size_t)fpw * BF_MESH_RG * COLS +
                                         (size_t)fpw * COLS + (size_t)fpw * 32);
    const dim3 grid(M->n_tiles, (n + fpw - 1) / fpw), block(COLS * BF_MESH_RG);
    if (smem > 64 * 1024) {
        hipError_t e = hipSuccess;
        if (fpw == 8) e = hipFuncSetAttribute((const void *)bf_mesh_multi_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        else if (fpw == 4) e = hipFuncSetAttribute((const void *)bf_mesh_multi_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    switch (fpw) {
    case 1: if (done) hipExtLaunchKernelGGL(bf_mesh_multi_kernel<1>, grid, block, smem, stream, nullptr, done, 0, *M, state, n, vraw, vout, xpart, vposed, dvzero, mp, door, door_target);
             else hipLaunchKernelGGL(bf_mesh_multi_kernel<1>, grid, block, smem, stream, *M, state, n, vraw, vout, xpart, vposed, dvzero, mp, door, door_target);
             break;
    case 2: if (done) hipExtLaunchKernelGGL(bf_mesh_multi_kernel<2>, grid, block, smem, stream, nullptr, done, 0, *M, state, n, vraw, vout, xpart, vposed, dvzero, mp, door, door_target);
             else hipLaunchKernelGGL(bf_mesh_multi_kernel<2>, grid, block, smem, stream, *M, state, n, vraw, vout, xpart, vposed, dvzero, mp, door, door_target);
             break;
    case 4: if (done) hipExtLaunchKernelGGL(bf_mesh_multi_kernel<4>, grid, block, smem, stream, nullptr, done, 0, *M, state, n, vraw, vout, xpart, vposed, dvzero, mp, door, door_target);
             else hipLaunchKernelGGL(bf_mesh_multi_kernel<4>, grid, block, smem, stream, *M, state, n, vraw, vout, xpart, vposed, dvzero, mp, door, door_target);
             break;
    default: if (done) hipExtLaunchKernelGGL(bf_mesh_multi_kernel<8>, grid, block, smem, stream, nullptr, done, 0, *M, state, n, vraw, vout, xpart, vposed, dvzero, mp, door, door_target);
             else hipLaunchKernelGGL(bf_mesh_multi_kernel<8>, grid, block, smem, stream, *M, state, n, vraw, vout, xpart, vposed, dvzero, mp, door, door_target);
             break;
    }
    return (int)hipGetLastError();
}

// which forward the single-launch path takes: the multi-frame kernel for 2..15 frames, and for one frame when the pose
// feature has more rows than bf_mesh_kernel keeps in flight
extern "C" int bf_mesh_use_multi(int npf, int n) { return n >= 2 || npf > BF_MESH_PF * BF_MESH_RG; }

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
    __shared__ float lds[BF_JOINTS_LDS];
    bf_joints_body<256>(M, state, vraw, xpart, joints, joints_ori, jraw, lmk_vid, lmk_w, blockIdx.x, lds);
}


// MFMA fragments (v_mfma_f32_32x32x2_f32): A: lane l holds A[i = l & 31][k = l >> 5]; B: lane l holds
// B[k = l >> 5][j = l & 31]; accumulator: column = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A operand of the pose-blend GEMM: featT[p][f] = pose feature p of frame f, frame-minor so that the GEMM's lanes read
// it coalesced; rows past npf and frames past n are zero.
extern "C" __global__ void __launch_bounds__(256)
bf_pack_feat_kernel(MeshTab M, const float *__restrict__ state, int n_frames, int kpad, int fpad, float *__restrict__ featT) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kpad * fpad) return;
    const int p = idx / fpad, f = idx - p * fpad;
    const int stride = bf_state_stride(M.nj, M.npf, M.nb), feat_off = M.nj * 15;      // GR 9 + At 3 + Gt 3 per joint, then feat
    featT[idx] = (p < M.npf && f < n_frames) ? state[(size_t)f * stride + feat_off + p] : 0.f;
}

// Pose blend of a whole batch on the matrix cores: pose_off[f][c] = sum_p feat[f][p] posedirs[p][c] (SURVEY.md 8a,
// lbs.py pose_offsets), fp32 MFMA 32x32x2.  A workgroup owns 64 columns x 128 frames: each of its four waves takes 32
// frames (two 32x32 accumulator tiles).  posedirs streams through LDS in blocks of 2 * BF_GEMM_KB rows, double
// buffered: the global loads of block k + 1 (float2, coalesced 256-byte rows, 13 per thread) are in flight while the
// MFMAs of block k run, and every posedirs element is fetched once per 128 frames.  The A operand (the frames' pose
// features, frame-minor featT) sits in VGPRs, one block at a time with the next prefetched.
extern "C" __global__ void __launch_bounds__(256, 3)
bf_poseblend_gemm_kernel(MeshTab M, const float *__restrict__ featT, int kpad, int fpad, int n_frames, float *__restrict__ pose_off) {
    constexpr int KB = BF_GEMM_KB, ROWS = 2 * KB, NLD = (ROWS * 32 + 255) / 256;      // float2 loads per thread and block
    __shared__ __align__(16) float s_b[2][ROWS][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npf = M.npf, ncols = 3 * M.nv;
    const int mi = lane & 31, kh = lane >> 5;
    const int f0 = (blockIdx.y * 4 + wave) * 32;
    const bool live = f0 < n_frames;                      // (idle waves still help with the staging)
    const int cbase = blockIdx.x * 64;
    const float *ap = featT + min(f0, fpad - 32) + mi;
    // staging role: thread -> (row r0 + 8 q, column pair c2)
    const int c2 = (tid & 31) * 2, r0 = tid >> 5;
    const bool cpair = cbase + c2 + 1 < ncols, csingle = cbase + c2 < ncols;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a_cur[KB];
    float2 st[NLD];
    auto fetch = [&](int kb) {                           // rows 2 kb .. 2 kb + ROWS of posedirs -> registers
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int r = r0 + 8 * q, p = 2 * kb + r;
            float2 v = {0.f, 0.f};
            if (r < ROWS && p < npf) {
                const float *src = M.posedirs + (size_t)p * M.pd_pitch + cbase + c2;
                if (cpair) v = *(const float2 *)src; else if (csingle) v.x = src[0];
            }
            st[q] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) { const int r = r0 + 8 * q; if (r < ROWS) *(float2 *)&s_b[buf][r][c2] = st[q]; }
    };
    fetch(0);
#pragma unroll
    for (int i = 0; i < KB; ++i) a_cur[i] = ap[(size_t)(2 * i + kh) * fpad];
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int kb = 0; kb < kpad / 2; kb += KB) {
        const bool more = kb + KB < kpad / 2;
        if (more) fetch(kb + KB);
        if (live) {
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const float b0 = s_b[buf][2 * i + kh][mi], b1 = s_b[buf][2 * i + kh][32 + mi];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b1, acc1, 0, 0, 0);
            }
        }
        if (more) {
            stash(buf ^ 1);
#pragma unroll
            for (int i = 0; i < KB; ++i) a_cur[i] = ap[(size_t)(2 * (kb + KB + i) + kh) * fpad];
        }
        __syncthreads();
        buf ^= 1;
    }
    if (!live) return;
    const int c0 = cbase + mi, c1 = c0 + 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = f0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (f < n_frames) {
            if (c0 < ncols) pose_off[(size_t)f * ncols + c0] = acc0[r];
            if (c1 < ncols) pose_off[(size_t)f * ncols + c1] = acc1[r];
        }
    }
}

// frames [0, n) of the batch
extern "C" hipError_t bf_poseblend_launch(const MeshTab *M, const float *state, int n, float *featT, int kpad, int fpad,
                                          float *pose_off, hipStream_t stream) {
    const int ncols = 3 * M->nv;
    hipLaunchKernelGGL(bf_pack_feat_kernel, dim3((kpad * fpad + 255) / 256), dim3(256), 0, stream, *M, state, n, kpad, fpad, featT);
    hipLaunchKernelGGL(bf_poseblend_gemm_kernel, dim3((ncols + 63) / 64, fpad / 128), dim3(256), 0, stream, *M, (const float *)featT,
                       kpad, fpad, n, pose_off);
    return hipGetLastError();
}

// Batched epilogue (4-sparse skinning rows, nb <= 10): one THREAD per vertex, 128 vertices per workgroup, walking
// BF_EPI_FRAMES frames.  The vertex's tables stay in registers (three shapedirs rows, template, four bones + weights);
// the frames' bone transforms and (beta | t | s) records are staged in LDS once.  Per vertex and frame: shaped vertex +
// pose offset (from the GEMM) -> T = sum_4 w A_j -> skinning -> similarity: ~100 VALU instructions, 16 b128 LDS reads,
// nothing of the model re-read.  The extra-joint partial sums keep the 32-vertex tiling the joints kernel expects.
extern "C" __global__ void __launch_bounds__(128)
bf_mesh_epilogue_batch_kernel(MeshTab M, const float *__restrict__ state, const float *__restrict__ pose_off, int n_frames,
                              float *__restrict__ vraw, float *__restrict__ vout, float *__restrict__ xpart) {
    constexpr int FE = BF_EPI_FRAMES, VT = 128;
    extern __shared__ __align__(16) float s_dyn[];          // [FE][nj * 12] bone transforms | [FE][VT * 3] raw vertices
    __shared__ __align__(16) float s_beta[FE][16];            // beta (12, zero padded) | t (3) | s * cscale
    const int nj = M.nj, nb = M.nb, npf = M.npf, nv = M.nv, ncols = 3 * nv;
    const int tid = threadIdx.x, nj12 = nj * 12;
    float *s_raw = s_dyn + FE * nj12;
    const int fbase = blockIdx.y * FE, nf = min(FE, n_frames - fbase);
    const size_t stride = bf_state_stride(nj, npf, nb);
    for (int e0 = tid; e0 < nj12; e0 += 128) {
        const int j = e0 / 12, e = e0 - j * 12, a = e >> 2, b = e & 3;
        const size_t src = b < 3 ? (size_t)(j * 9 + a * 3 + b) : (size_t)(nj * 9 + j * 3 + a);
        float tmp[FE];
#pragma unroll
        for (int f = 0; f < FE; ++f) tmp[f] = f < nf ? state[(size_t)(fbase + f) * stride + src] : 0.f;
#pragma unroll
        for (int f = 0; f < FE; ++f) s_dyn[f * nj12 + e0] = tmp[f];
    }
    {
        const size_t boff = (size_t)nj * 15 + npf + (size_t)nj * 3;           // beta, t, sc are contiguous in the state record
        for (int i = tid; i < FE * 16; i += 128) {
            const int f = i >> 4, e = i & 15;
            float v = 0.f;
            if (f < nf) {
                const float *rec = state + (size_t)(fbase + f) * stride + boff;
                if (e < 12) v = e < nb ? rec[e] : 0.f;
                else if (e < 15) v = rec[nb + (e - 12)];
                else v = rec[nb + 3] * rec[nb + 4];
            }
            s_beta[f][e] = v;
        }
    }
    const int v = blockIdx.x * VT + tid;
    const bool ok = v < nv;
    const int vc = ok ? v : nv - 1;
    float sd[3][10], vt[3], w4[4];
    int j4[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *sp = M.shapedirs + (size_t)(vc * 3 + c) * nb;
#pragma unroll
        for (int l = 0; l < 10; ++l) sd[c][l] = l < nb ? sp[l] : 0.f;
        vt[c] = M.v_template[vc * 3 + c];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { j4[q] = M.v_nzj[(size_t)vc * 4 + q] * 12; w4[q] = M.v_nzw[(size_t)vc * 4 + q]; }
    // the pose offsets of all the frames are requested up front (one memory latency, not one per frame)
    float ofs[FE][3];
#pragma unroll
    for (int f = 0; f < FE; ++f) {
        const float *po = pose_off + (size_t)(fbase + (f < nf ? f : 0)) * ncols + (size_t)vc * 3;
        ofs[f][0] = po[0]; ofs[f][1] = po[1]; ofs[f][2] = po[2];
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < FE; ++f) {
        if (f < nf) {
            const float4 bq0 = *(const float4 *)&s_beta[f][0], bq1 = *(const float4 *)&s_beta[f][4], bq2 = *(const float4 *)&s_beta[f][8];
            const float be[10] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w, bq2.x, bq2.y};
            float vp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a2 = 0.f;
#pragma unroll
                for (int l = 0; l < 10; ++l) a2 += sd[c][l] * be[l];
                vp[c] = vt[c] + a2 + ofs[f][c];
            }
            float T[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 *A = (const float4 *)(s_dyn + f * nj12 + j4[q]);
                const float4 r0 = A[0], r1 = A[1], r2 = A[2];
                T[0] += w4[q] * r0.x; T[1] += w4[q] * r0.y; T[2] += w4[q] * r0.z; T[3] += w4[q] * r0.w;
                T[4] += w4[q] * r1.x; T[5] += w4[q] * r1.y; T[6] += w4[q] * r1.z; T[7] += w4[q] * r1.w;
                T[8] += w4[q] * r2.x; T[9] += w4[q] * r2.y; T[10] += w4[q] * r2.z; T[11] += w4[q] * r2.w;
            }
            const float x0 = T[0] * vp[0] + T[1] * vp[1] + T[2] * vp[2] + T[3];
            const float x1 = T[4] * vp[0] + T[5] * vp[1] + T[6] * vp[2] + T[7];
            const float x2 = T[8] * vp[0] + T[9] * vp[1] + T[10] * vp[2] + T[11];
            s_raw[(f * VT + tid) * 3] = ok ? x0 : 0.f; s_raw[(f * VT + tid) * 3 + 1] = ok ? x1 : 0.f; s_raw[(f * VT + tid) * 3 + 2] = ok ? x2 : 0.f;
        }
    }
    __syncthreads();
    // coalesced stores out of the LDS copy: 384 consecutive floats per frame
    {
        const int c0 = blockIdx.x * VT * 3;
#pragma unroll
        for (int f = 0; f < FE; ++f) {
            if (f < nf) {
                const float4 bq3 = *(const float4 *)&s_beta[f][12];
                const size_t o = (size_t)(fbase + f) * ncols + c0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int c = tid + 128 * r, kk = c % 3;
                    if (c0 + c < ncols) {
                        const float x = s_raw[f * VT * 3 + c];
                        if (vraw) vraw[o + c] = x;
                        if (vout) vout[o + c] = (x + (kk == 0 ? bq3.x : (kk == 1 ? bq3.y : bq3.z))) * bq3.w;
                    }
                }
            }
        }
    }
    if (xpart) {
        // extra-joint partial sums: a thread keeps one (32-vertex sub-tile, extra joint, coordinate) regressor row in
        // registers and walks the frames
        const int ne3 = M.n_extra * 3, nsub = VT / BF_MESH_TILE;
        for (int i = tid; i < nsub * ne3; i += 128) {
            const int sub = i / ne3, q = i - sub * ne3, e = q / 3, kk = q - e * 3;
            const int tile = blockIdx.x * nsub + sub, v0 = tile * BF_MESH_TILE;
            if (tile >= M.n_tiles) continue;
            const float *row = M.j_extra + (size_t)e * nv + v0;
            float rv[BF_MESH_TILE];
#pragma unroll
            for (int t = 0; t < BF_MESH_TILE; ++t) rv[t] = v0 + t < nv ? row[t] : 0.f;
            for (int f = 0; f < nf; ++f) {
                float a3 = 0.f;
#pragma unroll
                for (int t = 0; t < BF_MESH_TILE; ++t) a3 += rv[t] * s_raw[(f * VT + sub * BF_MESH_TILE + t) * 3 + kk];
                xpart[((size_t)(fbase + f) * M.n_tiles + tile) * ne3 + q] = a3;
            }
        }
    }
}

// Final mesh of a 32-frame block in ONE launch (config 4's per-GPU shard is exactly one block): pose blend on the
// matrix cores with the shape blend, skinning and similarity BEHIND THE ACCUMULATORS - no featT, no pose_off round trip.
// grid (n_tiles, ceil(F / 32)), 512 threads; SMPL-sized models (bf_mesh_batch32_fits).
// A workgroup owns 32 vertices x 32 frames.  The MFMA column j of block cb is coordinate cb of vertex j, so a lane's B
// operand for a row is its vertex's 12 contiguous bytes (one dwordx3, 384 contiguous bytes per half wave) and the
// accumulators come out vertex-major.  K is dealt over the eight waves (wave w takes the row pairs 8 s + w).  Every global
// read of the kernel - the block's pose features, bone transforms and records (coalesced float2 along each frame's state
// record), the wave's 13 B operands, the vertex's tables, the tile's extra-joint rows - is requested before the first one
// is used, in the order of use: a wave has at most 64 loads in flight, so that is two memory latencies for the kernel.
// Then 13 x 3 v_mfma_f32_32x32x2_f32 per wave (A operand: pose features frame-minor in LDS, conflict free), the eight
// partial tiles meet in LDS (over the pose features) and are added in wave order; thread (vertex, frame pair) does what
// bf_mesh_epilogue_batch_kernel does per vertex and frame.  Stores leave through LDS, 384 consecutive bytes per frame.
typedef float bf_f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float bf_f2u __attribute__((ext_vector_type(2), aligned(8)));
extern "C" __global__ void __launch_bounds__(512)
bf_mesh_batch32_kernel(MeshTab M, const float *__restrict__ state, int n_frames, float *__restrict__ vraw, float *__restrict__ vout,
                       float *__restrict__ xpart) {
    constexpr int FB = 32, COLS = BF_MESH_TILE * 3, NW = 8, KS = 13, KROWS = 2 * NW * KS, FLD = FB + 1, ALD = COLS + 1, XLD = BF_MESH_TILE + 1;
    constexpr int NF2 = FB / 4;                             // 8 float2 of pose features per thread (threads 0..415: four frames a step)
    constexpr int NA2 = (FB + 2) / 3;                       // 11 float2 of bone transforms per thread (threads 0..3 nj6 - 1: three frames a step)
    constexpr int NX = (24 * BF_MESH_TILE + 511) / 512;     // 2 extra-joint regressor values per thread (n_extra <= 24)
    extern __shared__ __align__(16) float s_dyn[];
    float *s_feat = s_dyn;                                  // [KROWS][FLD]   pose features, frame-minor (dead after the K loop, under s_acc)
    float *s_acc = s_dyn;                                   // [NW][FB][ALD]  the waves' partial tiles; [0] becomes the raw vertices
    float *s_beta = s_acc + NW * FB * ALD;                  // [FB][16]       beta (12, zero padded) | t (3) | s * cscale
    float *s_A = s_beta + FB * 16;                          // [FB][nj * 12]  bone transforms
    float *s_x = s_A + FB * 24 * 12;                        // [n_extra][XLD] extra-joint regressor rows over this tile
    const int nj = M.nj, nb = M.nb, npf = M.npf, nv = M.nv, ncols = 3 * nv, nj12 = nj * 12, nj6 = nj * 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mi = lane & 31, kh = lane >> 5;
    const int tile = blockIdx.x, cbase = tile * COLS, v0 = tile * BF_MESH_TILE, fbase = blockIdx.y * FB, nf = min(FB, n_frames - fbase);
    const size_t stride = bf_state_stride(nj, npf, nb);
    const int feat_off = nj * 15;
    // ---- the loads, in the order of use
    bf_f2u tfe[NF2], tA[NA2];
    float tbe, tbs, tx[NX];
    const float *sblk = state + (size_t)fbase * stride;      // (the block's records: 32-bit offsets from here on)
    const int istride = (int)stride;
    constexpr int HP = KROWS / 2;                            // row pairs of a frame
    const int fe_hi = (tid >= HP ? 1 : 0) + (tid >= 2 * HP ? 1 : 0) + (tid >= 3 * HP ? 1 : 0), fe_p = 2 * (tid - fe_hi * HP);   // frame mod 4, row pair
    if (tid < 4 * HP) {
        // (unconditional: frames past the block repeat its last one, rows past npf read on into the record - zeroed when stored)
#pragma unroll
        for (int r = 0; r < NF2; ++r) tfe[r] = *(const bf_f2u *)(sblk + min(4 * r + fe_hi, nf - 1) * istride + feat_off + fe_p);
    }
    const int fa = (tid >= nj6 ? 1 : 0) + (tid >= 2 * nj6 ? 1 : 0) + (tid >= 3 * nj6 ? 1 : 0), ea = tid - fa * nj6;      // frame mod 3, pair
    if (fa < 3) {
#pragma unroll
        for (int r = 0; r < NA2; ++r) tA[r] = *(const bf_f2u *)(sblk + min(3 * r + fa, nf - 1) * istride + 2 * ea);
    }
    {
        // (beta | t | s cscale) records: plain loads, no branches - a wait inside this stream would serialise everything behind it
        const size_t boff = (size_t)nj * 15 + npf + (size_t)nj * 3;           // beta, t, sc are contiguous in the state record
        const int f = min(tid >> 4, nf - 1), e = tid & 15;
        const float *rec = sblk + f * istride + (int)boff;
        tbe = rec[e < 12 ? min(e, nb - 1) : nb + (e - 12)];                  // e == 15: body scale
        tbs = rec[nb + 4];                                                    //          x constant scale
    }
    // B operands of this wave: rows 2 (4 s + wave) + kh, the three coordinates of vertex v0 + mi
    // (unconditional, addresses clamped: a row past npf meets a zero A operand, a vertex past nv is never stored)
    bf_f3u bq[KS];
    {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int pdp = M.pd_pitch;
        const float *bcol = M.posedirs + (size_t)(2 * wv) * pdp + min(v0 + mi, nv - 1) * 3 + kh * pdp;
#pragma unroll
        for (int s2 = 0; s2 < KS - 1; ++s2) bq[s2] = *(const bf_f3u *)(bcol + (size_t)(2 * NW * s2) * pdp);
        bq[KS - 1] = *(const bf_f3u *)(bcol + (size_t)(min(2 * NW * (KS - 1) + 2 * wv + kh, npf - 1) - 2 * wv - kh) * pdp);
    }
    // the vertex's tables (thread = vertex tid & 31, frames 4 (tid >> 5) ..)
    const int vl = tid & 31, fo = tid >> 5, v = v0 + vl;
    const bool ok = v < nv;
    const int vc = ok ? v : nv - 1;
    float sd[30], w4[4];
    int j4[4];
    {
        const bf_f2u *sp = (const bf_f2u *)(M.shapedirs + (size_t)vc * 30);       // (nb == 10: three rows of ten, 8-byte aligned)
#pragma unroll
        for (int l = 0; l < 15; ++l) { const bf_f2u x = sp[l]; sd[2 * l] = x.x; sd[2 * l + 1] = x.y; }
    }
    const bf_f3u vt = *(const bf_f3u *)(M.v_template + (size_t)vc * 3);
    {
        const int4 jj = *(const int4 *)(M.v_nzj + (size_t)vc * 4);
        const float4 ww = *(const float4 *)(M.v_nzw + (size_t)vc * 4);
        j4[0] = jj.x * 12; j4[1] = jj.y * 12; j4[2] = jj.z * 12; j4[3] = jj.w * 12;
        w4[0] = ww.x; w4[1] = ww.y; w4[2] = ww.z; w4[3] = ww.w;
    }
    const int nxr = xpart ? M.n_extra * BF_MESH_TILE : 0;
#pragma unroll
    for (int r = 0; r < NX; ++r) tx[r] = 0.f;
    if (nxr) {
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const int i = tid + 512 * r, e = min(i >> 5, M.n_extra - 1), t = i & 31;
            tx[r] = M.j_extra[(size_t)e * nv + min(v0 + t, nv - 1)];
        }
    }
    // ---- LDS: pose features frame-minor, bone transforms as [joint][3][4], records, regressor rows
    if (tid < 4 * HP) {
        const bool px = fe_p < npf, py = fe_p + 1 < npf;
#pragma unroll
        for (int r = 0; r < NF2; ++r) {
            s_feat[fe_p * FLD + 4 * r + fe_hi] = px ? tfe[r].x : 0.f; s_feat[(fe_p + 1) * FLD + 4 * r + fe_hi] = py ? tfe[r].y : 0.f;
        }
    }
    if (fa < 3) {
#pragma unroll
        for (int r = 0; r < NA2; ++r) {
            const int f = 3 * r + fa;
            if (f < FB) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // GR[nj][3][3] | At[nj][3] -> [nj][3][4]: 9 j + 3 a + b -> 12 j + 4 a + b = src + src / 3;  9 nj + 3 j + a -> 12 j + 4 a + 3
                    const int src = 2 * ea + h;
                    const int dst = src < nj * 9 ? src + src / 3 : 4 * (src - nj * 9) + 3;
                    s_A[f * nj12 + dst] = h ? tA[r].y : tA[r].x;
                }
            }
        }
    }
    {
        const int f = tid >> 4, e = tid & 15;
        s_beta[tid] = f >= nf || (e < 12 && e >= nb) ? 0.f : (e == 15 ? tbe * tbs : tbe);
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) { const int i = tid + 512 * r; if (i < nxr) s_x[(i >> 5) * XLD + (i & 31)] = v0 + (i & 31) < nv ? tx[r] : 0.f; }
    __syncthreads();
    // ---- this wave's eighth of K
    f32x16 acc[3];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
        const float a = s_feat[(2 * (NW * s2 + wave) + kh) * FLD + mi];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[s2].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[s2].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[s2].z, acc[2], 0, 0, 0);
    }
    __syncthreads();                                         // (the partial tiles go where the pose features were)
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = (r & 3) + 8 * (r >> 2) + 4 * kh;
            s_acc[(wave * FB + f) * ALD + mi * 3 + cb] = acc[cb][r];
        }
    __syncthreads();
    // ---- behind the accumulators: shape blend + pose offset -> skinning, 2 frames per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = fo * 2 + i;
        const float *bp = s_beta + f * 16;
        const float4 bq0 = *(const float4 *)bp, bq1 = *(const float4 *)(bp + 4), bq2 = *(const float4 *)(bp + 8);
        const float be[10] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w, bq2.x, bq2.y};
        float vp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int o = f * ALD + vl * 3 + c;
            float po = s_acc[o];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) po += s_acc[w2 * FB * ALD + o];       // (wave order)
            float a2 = 0.f;
#pragma unroll
            for (int l = 0; l < 10; ++l) a2 += sd[c * 10 + l] * be[l];
            vp[c] = (c == 0 ? vt.x : (c == 1 ? vt.y : vt.z)) + a2 + po;
        }
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 *A = (const float4 *)(s_A + f * nj12 + j4[q]);
            const float4 r0 = A[0], r1 = A[1], r2 = A[2];
            T[0] += w4[q] * r0.x; T[1] += w4[q] * r0.y; T[2] += w4[q] * r0.z; T[3] += w4[q] * r0.w;
            T[4] += w4[q] * r1.x; T[5] += w4[q] * r1.y; T[6] += w4[q] * r1.z; T[7] += w4[q] * r1.w;
            T[8] += w4[q] * r2.x; T[9] += w4[q] * r2.y; T[10] += w4[q] * r2.z; T[11] += w4[q] * r2.w;
        }
        const float x0 = T[0] * vp[0] + T[1] * vp[1] + T[2] * vp[2] + T[3];
        const float x1 = T[4] * vp[0] + T[5] * vp[1] + T[6] * vp[2] + T[7];
        const float x2 = T[8] * vp[0] + T[9] * vp[1] + T[10] * vp[2] + T[11];
        // (the slots this thread just read in partial tile 0: nobody else touches them)
        float *raw = s_acc + f * ALD + vl * 3;
        raw[0] = ok ? x0 : 0.f; raw[1] = ok ? x1 : 0.f; raw[2] = ok ? x2 : 0.f;
    }
    __syncthreads();
    if (tid < 4 * COLS) {
        // threads 0..383: column tid % 96 of frames 4 r + tid / 96
        const int hi = (tid >= COLS ? 1 : 0) + (tid >= 2 * COLS ? 1 : 0) + (tid >= 3 * COLS ? 1 : 0), c = tid - hi * COLS, kk = c % 3;
        if (cbase + c < ncols) {
            size_t o = (size_t)(fbase + hi) * ncols + cbase + c;
#pragma unroll
            for (int r = 0; r < FB / 4; ++r, o += 4 * (size_t)ncols) {
                const int f = 4 * r + hi;
                if (f < nf) {
                    const float x = s_acc[f * ALD + c], tk = s_beta[f * 16 + 12 + kk], sc = s_beta[f * 16 + 15];
                    if (vraw) vraw[o] = x;
                    if (vout) vout[o] = (x + tk) * sc;
                }
            }
        }
    }
    if (xpart) {
        // extra-joint partial sums of this tile, xpart[f][tile][e][kk] = sum_t j_extra[e][v0 + t] raw[f][t][kk]: one 32x32 MFMA tile
        // (extra joint x frame, K = the tile's 32 vertices) per coordinate, waves 0..2
        if (wave < 3) {
            const int kk = wave, ne3 = M.n_extra * 3;
            f32x16 xa;
#pragma unroll
            for (int r = 0; r < 16; ++r) xa[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < BF_MESH_TILE / 2; ++ks) {
                const int t = 2 * ks + kh;
                xa = __builtin_amdgcn_mfma_f32_32x32x2f32(s_x[min(mi, 23) * XLD + t], s_acc[mi * ALD + t * 3 + kk], xa, 0, 0, 0);
            }
            if (mi < nf) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (e < M.n_extra) xpart[((size_t)(fbase + mi) * M.n_tiles + tile) * ne3 + e * 3 + kk] = xa[r];
                }
            }
        }
    }
}

extern "C" bool bf_mesh_batch32_fits(const MeshTab *M) {
    // (float2 reads along a frame's record: the record and its pose-feature block have to start on 8-byte boundaries)
    if (bf_state_stride(M->nj, M->npf, M->nb) % 2 != 0 || (M->nj * 15) % 2 != 0) return false;
    // npf >= 2 NW (KS - 1) + 1 = 193: only the LAST of a wave's 13 row pairs is clamped to the table's end (bq[KS - 1]); with fewer
    // rows an earlier pair would read past posedirs (nj = 22: npf = 189 - the operand it meets is zero, but 0 x garbage may be NaN)
    return M->npf <= 208 && M->npf >= 193 && M->nj <= 24 && M->nj >= 22 /* three frames' bone transforms per 512-thread step */ && M->v_nnz == 4 &&
           M->nb == 10 && M->n_extra <= 24;
}
extern "C" hipError_t bf_mesh_batch32_launch(const MeshTab *M, const float *state, int n, float *vraw, float *vout, float *xpart,
                                             hipStream_t stream) {
    constexpr int FB = 32, COLS = BF_MESH_TILE * 3;
    const size_t smem = sizeof(float) * (8 * (size_t)FB * (COLS + 1) + (size_t)FB * 16 + (size_t)FB * 24 * 12 + 24 * (size_t)(BF_MESH_TILE + 1));
    {   // (every launch: the attribute belongs to the current device, and a group drives several from one process)
        hipError_t e = hipFuncSetAttribute((const void *)bf_mesh_batch32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(bf_mesh_batch32_kernel, dim3(M->n_tiles, (n + FB - 1) / FB), dim3(512), smem, stream, *M, state, n, vraw, vout, xpart);
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

// out[c][r] = in[r][c]: posedirs [npf][3NV] -> posedirsT [3NV][npf] for the reverse pass, once per model.  32 x 32 tiles through
// LDS (padded row: no bank conflicts), both sides coalesced.  grid (ceil(cols / 32), ceil(rows / 32)), 256 threads.
extern "C" __global__ void __launch_bounds__(256) bf_transpose_kernel(const float *__restrict__ in, int rows, int cols, float *__restrict__ out, int in_pitch) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + k * 8, c = c0 + tx;
        tile[ty + k * 8][tx] = (r < rows && c < cols) ? in[(size_t)r * in_pitch + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + k * 8, r = r0 + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][ty + k * 8];
    }
}
