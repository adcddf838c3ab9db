// Internal layouts shared by the HIP kernels and the host side of libbodyfit (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BF_GMM_M 8          // reference hard-codes num_gaussians=8 (smplify/smplify.py:47)
#define BF_GMM_D 69         // and a 69-dof body pose (smplify/prior.py:154)
#define BF_GMM_LD 72        // padded row length of the d / y vectors in LDS
#define BF_FIT_THREADS 512  // one workgroup (8 wave64, two per SIMD) per frame
#define BF_VSUB 16          // view lanes per loss-joint pair in the projection phase (one DPP row)
#ifndef BF_RED_COLS
#define BF_RED_COLS 64       // outputs per workgroup of bf_ext_reduce_kernel (x 8 chunks of partial rows = its threads)
#endif
#define BF_KP_ROUNDS 3      // keypoint records staged in LDS for V <= 16*3 = 48 views
#define BF_SEL_NNZ 8         // compacted skinning weights per selector vertex (real SMPL has <= 4)
#define BF_SKIN_PER_WAVE 5   // selector vertices a geometry wave can skin in one pass (12 lanes each) in the merged skin + projection phase
#define BF_MFMA_MIN_FRAMES 16
#define BF_BATCH32_MAX_FRAMES 64   // up to here the final mesh of a batch is bf_mesh_batch32_kernel (32-frame blocks, fused epilogue)
#define BF_EPI_FRAMES 8      // frames one workgroup of the batched mesh epilogue walks over (its tile's tables stay in registers)
#define BF_GEMM_KB 26        // K pairs per register block of the pose-blend GEMM (A operand resident in VGPRs)  // from this batch size on the pose blend runs as one fp32-MFMA GEMM for all frames
#define BF_MESH_TILE 32     // vertices per workgroup of the full-mesh forward
#define BF_MESH_RG 8        // pose-feature row groups per workgroup (split-K inside the workgroup)

// Model-level tables of the fit, all resident in HBM for the life of the model.
struct FitTab {
    int nj, nb, npf, ns, nl, np;     // joints, betas, pose features 9(nj-1), selector verts, loss joints, params
    int n_levels;                    // depth of the kinematic tree
    int nbp;                         // optimised body-pose dofs (69 SMPL); the GMM sees them zero-padded to 69
    int off_pose, off_beta, off_orient;
    // full-pose assembly theta_j = pose_mean_j + (params | hand PCA | 0), SURVEY.md 10B
    int n_pca, off_lh, off_rh;       // hand PCA coefficient blocks inside the parameter vector (n_pca = 0 for SMPL)
    const int *th_kind;              // [nj] 0 = 3 parameters at th_off, 1 = constant (jaw), 2 / 3 = left / right hand joint th_off
    const int *th_off;               // [nj]
    const float *pose_mean;          // [nj*3] or null
    const float *hand_comp;          // [2][n_pca][45] or null
    const int *p_kind;               // [np] 0 = g[i] (transl, scale), 1 = d/dtheta[p_a] (+ body priors if p_b >= 0 = body dof),
    const int *p_a;                  //      2 = beta (+ shape prior), 3 = hand PCA coefficient p_b of hand p_a
    const int *p_b;
    int kp_dense;                    // 1 = the keypoint loss arrives through `ext` (more than 32 loss joints)
    const int *parents;              // [nj]
    const int *depth;                // [nj]
    const unsigned long long *desc;  // [nj] bit k set = joint k is a strict descendant
    const int *dfs_order;            // [nj] joint at depth-first position i (children in index order)
    const int *dfs_last;             // [nj] last depth-first position of the subtree rooted at position i (inclusive)
    const int *level_start;          // [n_levels+1]  joints sorted by depth
    const int *level_joints;         // [nj]
    const int *child_start;          // [nj+1]        CSR children lists
    const int *child_list;           // [nj-1]
    const int *lj_kind;              // [nl] 0 = chain joint, 1 = selector vertex slot
    const int *lj_index;             // [nl]
    // The deal of the merged skinning + projection phase (sized SMPL instance): loss-joint PAIR (2 p, 2 p + 1) pair_slot[4 w + q] is
    // projected by the q-th sixteen lanes of geometry wave w (-1: nobody), and wave w skins selector vertices skin_vert[5 w ..]
    // (-1: none) - every selector vertex a wave's pairs read is skinned by that wave, so no workgroup barrier separates the two.
    // bd_ok = 0: no such deal exists for this model (it takes the table-driven instance).
    int bd_ok;
    int pair_slot[16];
    int skin_vert[4 * BF_SKIN_PER_WAVE];
    const float *Jt;                 // [nj*3]        J_regressor v_template
    const float *Jd;                 // [nj*3][nb]    J_regressor shapedirs
    const float *Jdrel;              // [nj*3][nb]    Jd[j] - Jd[parent(j)]   (Jd[0] for the root)
    const float *Jtrel;              // [nj*3]        Jt[j] - Jt[parent(j)]   (Jt[0] for the root)
    const float *sel_vt;             // [ns*3]
    const float *sel_sd;             // [ns*3][nb]
    const float *sel_pd;             // [npf][ns*3]   posedirs columns of the selector vertices
    const float *sel_w;              // [ns][nj]
    int sel_nnz;                     // max non-zeros per row if <= BF_SEL_NNZ, else 0
    const float *sel_nzw;            // [ns][BF_SEL_NNZ] non-zero weights in joint order, zero padded
    const int *sel_nzj;              // [ns][BF_SEL_NNZ] their joints (padding points at joint 0)
    const float *g_means;            // [M][D]
    const float *g_psym;             // [M][D][D]     0.5 (P + P^T)
    const float *g_logw;             // [M]           -log(nll_weights)
    const float *g_plane;            // [M][72][64]   Psym[m][row=lane][col=j] lane-major (rows 0..63), zero padded
    const float *g_ptail;            // [4][12][64]   rows 64..68 of components (2w, 2w+1) cut into 60 twelve-column pieces
    // the fit kernel's LDS segment as its prologue leaves it (everything but the per-frame arrays), for the dense schedule's
    // one-iteration launches: one flat copy instead of two dozen dependent load -> store loops.  Null until built.
    const float *lds_image;
    int lds_image_n4;                // float4s of the dump (the LDS segment up to the projection matrices)
    int img_seg[3][2];               // the three runs of model-constant arrays in it: (first float4, count)
};

// Full model tensors for the dense mesh kernels.
struct MeshTab {
    int nv, nj, nb, npf;
    int n_selector, n_extra, n_joint_map;
    const float *v_template;         // [nv*3]
    const float *shapedirs;          // [nv*3][nb]
    const float *posedirs;           // [npf][pd_pitch]: rows of 3 nv floats, padded to a multiple of 128 bytes
    int pd_pitch;
    const float *lbs_weights;        // [nv][nj]
    const float *j_extra;            // [n_extra][nv]
    const int *selector_ids;         // [n_selector]
    const int *joint_map;            // [n_joint_map]
    int n_tiles;                     // ceil(nv / BF_MESH_TILE)
    // the non-zero skinning weights of every vertex, padded to v_nnz (4 or 8) entries; v_nnz = 0: some vertex has
    // more than 8 bones, use the dense rows
    int v_nnz;
    const int *v_nzj;                // [nv][v_nnz]
    const float *v_nzw;              // [nv][v_nnz]
    // face landmarks (SMPL-X): 51 static + 17 contour landmarks picked by the neck's yaw out of a 79-row table
    int n_lmk_static, n_lmk_dyn, n_dyn_rows, neck_joint;
    const int *faces;                // [nf][3]
    const int *lmk_faces;            // [n_lmk_static]
    const float *lmk_bary;           // [n_lmk_static][3]
    const int *dyn_faces;            // [rows][n_lmk_dyn]
    const float *dyn_bary;           // [rows][n_lmk_dyn][3]
    const int *lmk_fv, *dyn_fv;      // [n_lmk_static][3] / [rows][n_lmk_dyn][3]: faces[lmk_faces[l]] / faces[dyn_faces[..]] looked up once, on the host
                                     // (the joints prologue of the dense keypoint workgroup had three dependent global loads per landmark: face ->
                                     //  its corners -> their coordinates; now two)
};

// theta_j[k] of the full pose: shared by the fit kernel (LDS copies of the tables) and the pose-state kernel
__device__ inline float bf_theta(const float *params, int j, int k, const int *th_kind, const int *th_off, const float *pose_mean,
                                 const float *hand_comp, int n_pca, int off_lh, int off_rh) {
    float th = pose_mean ? pose_mean[j * 3 + k] : 0.f;
    const int kind = th_kind[j], off = th_off[j];
    if (kind == 0) th += params[off + k];
    else if (kind >= 2) {
        const float *comp = hand_comp + (kind - 2) * n_pca * 45 + off * 3 + k;
        const float *c = params + (kind == 2 ? off_lh : off_rh);
        for (int q = 0; q < n_pca; ++q) th += c[q] * comp[q * 45];
    }
    return th;
}

// The three components of theta_j at once, bit for bit bf_theta's values: the PCA hands' operands (n_pca coefficients and 3 n_pca
// component entries) are requested in batches of six BEFORE the first multiply-add, and the three sums (independent chains, q
// ascending as above) run side by side - bf_theta called three times is 3 x n_pca dependent pairs of LDS reads on one wave.
__device__ inline void bf_theta3(const float *params, int j, float th[3], const int *th_kind, const int *th_off, const float *pose_mean,
                                 const float *hand_comp, int n_pca, int off_lh, int off_rh) {
    const int kind = th_kind[j], off = th_off[j];
    th[0] = pose_mean ? pose_mean[j * 3] : 0.f; th[1] = pose_mean ? pose_mean[j * 3 + 1] : 0.f; th[2] = pose_mean ? pose_mean[j * 3 + 2] : 0.f;
    if (kind == 0) { th[0] += params[off]; th[1] += params[off + 1]; th[2] += params[off + 2]; }
    else if (kind >= 2) {
        const float *comp = hand_comp + (kind - 2) * n_pca * 45 + off * 3;
        const float *c = params + (kind == 2 ? off_lh : off_rh);
        for (int q0 = 0; q0 < n_pca; q0 += 6) {
            float cq[6], m0[6], m1[6], m2[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int q = q0 + i < n_pca ? q0 + i : n_pca - 1;
                cq[i] = c[q]; m0[i] = comp[q * 45]; m1[i] = comp[q * 45 + 1]; m2[i] = comp[q * 45 + 2];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (q0 + i < n_pca) { th[0] += cq[i] * m0[i]; th[1] += cq[i] * m1[i]; th[2] += cq[i] * m2[i]; }
        }
    }
}

// Per-frame pose state handed from the fit / pose-prep kernel to the mesh kernel.
// layout per frame (floats): GR[nj*9] At[nj*3] Gt[nj*3] feat[npf] theta[nj*3] beta[nb] t[3] s c
__host__ __device__ inline int bf_state_stride(int nj, int npf, int nb) {
    return nj * 9 + nj * 3 + nj * 3 + npf + nj * 3 + nb + 3 + 2;
}
struct StateView {
    float *GR, *At, *Gt, *feat, *theta, *beta, *t, *sc;   // sc[0] = body_scale, sc[1] = constant_scale
};
__host__ __device__ inline StateView bf_state_view(float *base, int nj, int npf, int nb) {
    StateView s;
    s.GR = base;
    s.At = s.GR + nj * 9;
    s.Gt = s.At + nj * 3;
    s.feat = s.Gt + nj * 3;
    s.theta = s.feat + npf;
    s.beta = s.theta + nj * 3;
    s.t = s.beta + nb;
    s.sc = s.t + 3;
    return s;
}



// One scan mesh with its uniform search grid (device pointers).
struct ScanDev {
    int nv, nf, nx, ny, nz;
    float ox, oy, oz, step, height;
    const float *verts;       // [nv][3]
    const int *faces;         // [nf][3]
    const int *cell_start;    // [nx*ny*nz + 1], cell = (x*ny + y)*nz + z
    const int *cell_tris;     // triangles per cell, ascending face id
    const float4 *cell_pack;  // per cell-list entry: the triangle's corners and id, 3 x float4 = (p0.xyz p1.x)(p1.yz p2.xy)(p2.z id 0 0):
                              // one contiguous 48-byte record instead of the list -> faces -> vertices pointer chase
    const float4 *cell_box;   // per cell-list entry, for the search's screen: 2 x float4 = (box lo.xyz, m)(box hi.xyz, 0) of the triangle, m =
                              // 2.1e-5 x the box's squared diagonal (what the rule's value may lie below the true distance, scan_kernels.hip)
};

// Silhouette-loss inputs of one batch (device pointers).
struct MaskIO {
    int nv, ns, n_views, n_masks, H, W, cmax, part_stride, proj_blocks;
    int cdist;                            // 1 = distances in torch.cdist's expanded fp32 form (loss.py:108), 0 = exact
    int sstride;                          // the sampled vertices are vertices s * sstride: 4 on the model (loss.py:99), 1 on the sampled-first sub-model
    float imsize, eps, weight;            // weight = 5 (smplify.py:210)
    const int *view_index;                // [M] index of each mask view among the V views
    const unsigned char *masks;           // [F][M][H][W], 1 = foreground (already > 128, smplify.py:139)
    const int *contour_start;             // [F*M] offsets into contour_xy
    const int *contour_count;             // [F*M]
    const float *contour_xy;              // [sum][2] (x, y) contour points (loss.py:73-83)
    // Round 5: the contour term's gradient per (view, sampled vertex) summed as 64-bit FIXED-POINT numbers by atomic adds of the contour
    // scan itself (exact sums: the order of the additions does not matter, so the result is reproducible without the ordered walk of
    // bf_mask_gather_kernel - a 13 us launch of the dense iteration).  [F][M][ns][2] (du, dv) scaled by 2^BF_ACC_SHIFT; zeroed by the
    // projection of the same iteration; null = off (the contour scan then only writes choice / cgrad for the gather kernel).
    unsigned long long *acc;
};
#define BF_ACC_SHIFT 40       // |du|, |dv| <= weight x eps x contour points < 2^18: 2^58 at most, steps of 2^-40
__device__ __forceinline__ unsigned long long bf_acc_fixed(float x) { return (unsigned long long)__double2ll_rn((double)x * (double)(1ull << BF_ACC_SHIFT)); }
__device__ __forceinline__ float bf_acc_float(unsigned long long a) { return (float)((double)(long long)a * (1.0 / (double)(1ull << BF_ACC_SHIFT))); }

// What the reverse mesh pass needs to turn those sums into dL/dvertex itself (bf_mask_gather_kernel's closing step, per view, views in order)
struct MaskFold {
    const unsigned long long *acc;        // null = off
    const float *uvi, *duvb, *proj;       // bf_mask_project_one's records; [F][V][12]
    const int *view_index;                // [M]
    int n_views;                          // V
};

// What the forward mesh pass needs to project its sampled vertices into the mask views (on = 0: nothing to do).
struct MaskProj {
    int on;
    MaskIO K;
    const float *proj;        // [F][V][12]
    float *uvi, *duvb;        // bf_mask_project_kernel's outputs
};


// Doorbells between the dense kernels of an iteration (batch stream) and the persistent fit launch (second stream) - the fit
// kernel stays resident for all dense iterations of a call instead of being launched once per iteration (a launch starts with
// cold instruction and data caches: ~25 us for this kernel, against ~6 us for a warm iteration).
//   door[BF_DOOR_EXT]    = k: the outside gradient blocks `ext` of dense iteration k (1-based) are complete (rung by the last
//                          workgroup of bf_ext_reduce_kernel)
//   door[BF_DOOR_STATE + c * BF_DOOR_COPY_STRIDE], c < BF_DOOR_COPIES
//                        = number of (frame, iteration) pose states the fit launch has published so far (each of its workgroups adds
//                          1 per iteration, to every copy): iteration k's forward mesh pass waits for F * k.  Hundreds of workgroups
//                          poll it at once, so it is replicated over cache lines 4 KB + 128 B apart (different memory channels) and a
//                          workgroup polls the copy its index selects - one line would queue every poll behind the others.
//   door[BF_DOOR_TICKET] = running count of finished bf_ext_reduce_kernel workgroups
//   door[BF_DOOR_ERR]    = sticky: a wait ran into its time limit (the call fails with BF_ERR_HIP; nobody waits any more)
#define BF_DOOR_EXT 0
#define BF_DOOR_TICKET 1
#define BF_DOOR_ERR 2
#define BF_DOOR_KP 3            // running count of finished keypoint workgroups that ran beside the closest-point search (F per dense iteration): bf_pc_grad_kernel waits for F x k
#define BF_DOOR_STATE 64
#define BF_DOOR_FEAT 96          // (+ c * BF_DOOR_COPY_STRIDE) like BF_DOOR_STATE, one cache line further: the pose FEATURES of that many (frame,
                                 // iteration) states are published - rung early in the fit launch's phase A, as soon as the rotations exist, so a
                                 // forward mesh pass runs its pose blend (most of its work) under the rest of that phase and only then waits for
                                 // the chain matrices (BF_DOOR_STATE)
#define BF_DOOR_COPIES 32
#define BF_DOOR_COPY_STRIDE 1056
#define BF_DOOR_INTS (BF_DOOR_COPIES * BF_DOOR_COPY_STRIDE)
#define BF_DOOR_LIMIT_TICKS 100000000LL        // 1 s of the 100 MHz wall clock

#ifdef __HIPCC__
// one thread: wait until *flag >= target (acquire at device scope); false = gave up (error flag set by somebody, or the time limit)
__device__ inline bool bf_door_wait(int *door, int which, int target) {
    // (polled with RELAXED device-scope loads: they are served past the non-coherent caches)
    bool ok = true;
    if (__hip_atomic_load(door + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        const long long t0 = wall_clock64();
        for (;;) {
            __builtin_amdgcn_s_sleep(8);
            if (__hip_atomic_load(door + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            if (__hip_atomic_load(door + BF_DOOR_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
            if (wall_clock64() - t0 > BF_DOOR_LIMIT_TICKS) {
                __hip_atomic_store(door + BF_DOOR_ERR, which == BF_DOOR_EXT ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (1: the fit launch gave up, 2: a mesh pass)
                ok = false;
                break;
            }
        }
    }
    // (NO acquire fence here: on this multi-XCD part a device-scope acquire invalidates the XCD's whole L2, and hundreds of waiting
    //  workgroups doing that under a streaming kernel doubled its run time.  Callers either have nothing stale to see - a kernel
    //  starts with invalidated caches and has not touched the data before its wait - or read the data past the caches.)
    return ok;
}
#endif

// Dense keypoint loss inputs (device pointers / sizes).
struct KpIO {
    int nl, n_views, nj, npf, nb, nv, n_all, n_selector, n_extra, n_lmk, n_cj_list;
    float sigma2, coeff;
    const int *joint_map;        // [nl] index into the all-joints array
    const int *selector_ids;     // [n_selector]
    const int *cj_start;         // [nj+1]  loss joints mapping to each chain joint (CSR)
    const int *cj_list;
    const float *j_extra;        // [n_extra][nv]  J_regressor_extra rows over THIS mesh's vertices (models/smpl.py:62-64, 72)
};

struct FrameIO {
    int n_frames, n_views;
    const float *proj;        // [F][V][12]   K [R|t], world -> pixel
    const float *keypoints;   // [F][V][nl][3]
    const int *ndiv;          // [F]
    float *params;            // [F][np]
    const float *params0;     // [F][np] or null: start from these parameters with zero Adam moments (in-kernel re-arm)
    float *adam_m;            // [F][np]
    float *adam_v;            // [F][np]
    float *grads;             // [F][np]       (grad-only mode)
    float *terms;             // [F][4]
    float *state;             // [F][state_stride]
    float *debug;             // optional dump of the first iteration's intermediates
    const float *cscale;      // [F] per-frame constant scale (scan_height / 1.7, smplify.py:156) or null
    float *image_out;         // mode 2 only: receives the LDS image (see FitTab::lds_image)
    int *door;                // dense schedule, persistent fit launch: the doorbells it shares with the dense kernels (BfDoor), or null
    int *door_resident;       // host-visible counter: every workgroup of the persistent launch adds 1 when it starts running
    const float *ext;         // [F][npf + nj*12 + nb + 4 + nj*3] gradients arriving from the dense losses (dfeat | per joint
                              //  rows of sum w dv (x) [vp|1] | dbeta | dt ds | dL/d(chain joint positions)), or null
};

struct HyperDev {
    float sigma2, w_pose, w_angle, w_shape, cscale, coeff;
    float beta1, beta2, eps;
};
