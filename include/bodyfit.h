/*
 * libbodyfit - MI355X-native multi-view SMPLify inner loop, C ABI.
 *
 * The reference (generalizable-neural-performer/bodyfitting) has no C/plugin API for this path;
 * its boundary is Python (`smplify.smplify.SMPLify`, `smplify.body_fitting.BodyFitting`,
 * `models.smpl.SMPL`).  This header is the native surface the build's Python mirror of those
 * classes (bodyfitting_amd/smplify.py, smpl.py) binds through ctypes.  Each entry point names the
 * reference code it replaces.
 *
 * Conventions: every function returns 0 on success and a negative bf_status otherwise;
 * bf_last_error() gives the message for the calling thread.  All pointers are HOST pointers to
 * caller-owned, C-contiguous buffers unless the name ends in `_dev`.  Device memory is owned by the
 * library behind the opaque handles.  One handle is used by one host thread at a time.  Work is
 * queued on the batch's own HIP stream; nothing blocks except bf_batch_sync and the getters.
 * All floating point is fp32 (the reference's dtype); indices are int32.
 */
#ifndef BODYFIT_H
#define BODYFIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bf_status {
    BF_OK = 0,
    BF_ERR_INVALID = -1,      /* bad argument / inconsistent sizes */
    BF_ERR_HIP = -2,          /* a HIP runtime call failed (message has the hipError string) */
    BF_ERR_UNSUPPORTED = -3,  /* size outside what the kernels were built for */
    BF_ERR_NO_DEVICE = -4     /* no usable gfx950 device */
} bf_status;

typedef struct bf_model bf_model;   /* body model + GMM prior resident on one device */
typedef struct bf_batch bf_batch;   /* F independent frames being fitted against one model */
typedef struct bf_scan bf_scan;     /* one scan mesh + its uniform closest-point grid on one device */

/*
 * Body model tensors exactly as smplx==0.1.13 stores them and reference models/smpl.py:56-66 extends
 * them, plus the buffers of MaxMixturePrior (reference smplify/prior.py:143-160).
 */
typedef struct bf_model_desc {
    int32_t n_verts;               /* NV: 6890 (SMPL) */
    int32_t n_joints;              /* NJ: 24 */
    int32_t n_betas;               /* NB: 10 */
    const float *v_template;       /* [NV,3] */
    const float *shapedirs;        /* [NV,3,NB] */
    const float *posedirs;         /* [9(NJ-1), 3NV]  row p = pose-feature element, col = 3v+k */
    const float *j_regressor;      /* [NJ,NV] dense */
    const float *lbs_weights;      /* [NV,NJ] dense */
    const int32_t *parents;        /* [NJ], parents[0] = -1, parents[i] < i */
    int32_t n_selector;            /* 21: VertexJointSelector vertex ids appended after the chain joints */
    const int32_t *selector_ids;   /* [n_selector] */
    int32_t n_extra;               /* 9: rows of J_regressor_extra (models/smpl.py:62-64,72) */
    const float *j_regressor_extra;/* [n_extra,NV] dense */
    int32_t n_joint_map;           /* 49: output joints = cat(chain, selector, extra)[joint_map] */
    const int32_t *joint_map;      /* [n_joint_map] (models/smpl.py:61,75) */
    int32_t n_loss_joints;         /* 25: SKELETON_LENGTH, the leading joints that enter the loss (loss.py:17,163) */
    int32_t gmm_components;        /* 8 */
    int32_t gmm_dim;               /* 69 */
    const float *gmm_means;        /* [M,D] */
    const float *gmm_precisions;   /* [M,D,D] = inv(covars) */
    const float *gmm_nll_weights;  /* [M] (prior.py:153-160) */
    int32_t n_faces;               /* body-model topology, only needed by the SMPL+D stage; may be 0 */
    const int32_t *faces;          /* [n_faces,3] */
    /* ---- SMPL-X (smplx.create(model_type='smplx', use_face_contour=True, use_pca with 6 comps), smplify.py:59-80).
     * model_kind 0 = SMPL (everything below ignored), 1 = SMPL-X: joints 0 root, 1..21 body, 22 jaw, 23/24 eyes,
     * 25..39 / 40..54 left / right hand; n_betas = 10 (expression stays 0, smplify.py:167-173); the optimised
     * vector is transl3 scale1 body_pose63 betas10 global_orient3 leye3 reye3 left_hand_pca6 right_hand_pca6 (98). */
    int32_t model_kind;
    const float *pose_mean;                 /* [3 NJ], added to the assembled full pose */
    int32_t n_hand_pca;                     /* 6 */
    const float *left_hand_components;      /* [n_hand_pca,45] */
    const float *right_hand_components;     /* [n_hand_pca,45] */
    int32_t n_lmk_static;                   /* 51 */
    const int32_t *lmk_faces_idx;           /* [n_lmk_static] face ids */
    const float *lmk_bary_coords;           /* [n_lmk_static,3] */
    int32_t n_lmk_dynamic;                  /* 17 */
    int32_t n_dyn_rows;                     /* 79 */
    const int32_t *dynamic_lmk_faces_idx;   /* [n_dyn_rows,n_lmk_dynamic] */
    const float *dynamic_lmk_bary_coords;   /* [n_dyn_rows,n_lmk_dynamic,3] */
    int32_t neck_joint;                     /* 12: its global rotation's yaw picks the contour row */
} bf_model_desc;

/* Loss weights and optimiser constants; bf_hyper_default() fills the reference's literals. */
typedef struct bf_hyper {
    float sigma;               /* 100   loss.py:139 */
    float pose_prior_weight;   /* 4.78  loss.py:141 */
    float angle_prior_weight;  /* 15.2  loss.py:140 */
    float shape_prior_weight;  /* 5     loss.py:140 */
    float constant_scale;      /* 0.3   smplify.py:160 */
    float imsize;              /* 512   scale_coeff = imsize/1024, loss.py:155 */
    float lr;                  /* 1e-2  smplify.py:174 */
    float lr_transl_scale;     /* 0.1   smplify.py:167-168 */
    float adam_beta1;          /* 0.9 */
    float adam_beta2;          /* 0.999 */
    float adam_eps;            /* 1e-8 */
    float lr_displacement;     /* 5e-2  smplify.py:233 */
    float mask_cdist_form;     /* 1     silhouette loss: 1 = contour-to-vertex distances in the fp32 form torch.cdist evaluates for
                                        these sizes (loss.py:108: |a|^2 + |b|^2 - 2ab as one 4-term fma chain, ~1e-2 px of
                                        round-off at 512 px - the reference's own numbers, so the nearest-vertex choice matches
                                        it); 0 = exact (a - b)^2 sums */
    float dense_after;         /* -1    the silhouette / scan losses are active for iterations i > dense_after, i counted from the
                                        last reset; -1 = num_iters // 3 of the bf_fit call (smplify.py:197,205).  Lets a caller
                                        cut one reference loop into several bf_fit calls (snapshots) */
} bf_hyper;

/* bf_fit flags */
#define BF_FIT_DEFAULT      0u
#define BF_FIT_DENSE        1u   /* evaluate the full mesh every iteration, like the reference does
                                    (smplify.py:179-190), instead of only the vertices that carry
                                    gradient; same results, used for measurement */
#define BF_FIT_NO_VERTICES  2u   /* skip the final full-mesh evaluation (parameters only) */
#define BF_FIT_RESET        8u   /* bf_batch_reset() first, inside the same call */
#define BF_FIT_GRAPH       16u   /* with BF_FIT_RESET on the keypoint-only path: capture the call's whole command sequence
                                    into a hipGraph once and replay it - one host command per fit (per-kernel device times are
                                    then not split: bf_batch_last_timing charges everything to ms[0]) */
#define BF_FIT_NOTIME      32u   /* do not bracket the parts of this call with HIP events (bf_batch_last_timing / timing_sum skip it):
                                     four event records per call are a measurable share of a 0.65 ms step.  With BF_FIT_RESET |
                                     BF_FIT_FETCH on the keypoint-only path the mesh / joints / result hand-over of the call then runs
                                     on the batch's second stream, under the fit kernel of the NEXT call (frame after frame) */
#define BF_FIT_FETCH        4u   /* queue the device->host copies of the result into the batch's pinned
                                    staging buffers behind the kernels (bf_batch_get_result then only
                                    waits for them) */

const char *bf_last_error(void);
const char *bf_version(void);
int bf_device_count(void);
void bf_hyper_default(bf_hyper *h);

/* Replaces the per-frame model + prior construction of SMPLify.__init__ (smplify.py:46-56): the
 * tensors are uploaded once per process, and the model-level tables of the fit are derived. */
int bf_model_create(const bf_model_desc *desc, int device, bf_model **out);
void bf_model_destroy(bf_model *m);
/* number of optimised scalars per frame: 86 for SMPL, laid out in the reference's optimiser order
 * (smplify.py:167-171): global_transl[3] body_scale[1] body_pose[69] betas[10] global_orient[3] */
int bf_model_n_params(const bf_model *m);
/* which instance of the persistent keypoint fit this model takes (no reference counterpart: the reference has one code path):
 * 1 = sizes fixed at compile time (24 joints, 10 betas, 11 loss selector vertices with at most 4 bones each, 25 loss joints: SMPL
 * as models/smpl.py:56-66 builds it), 0 = table-driven (any other model; ~2.5x more cycles per iteration) */
int bf_model_fit_instance(const bf_model *m);

/* models.smpl.SMPL.forward (models/smpl.py:69-83) for `n` parameter sets:
 * betas[n,NB], global_orient[n,3], body_pose[n,3(NJ-1)] ->
 * vertices[n,NV,3], joints[n,n_joint_map,3], joints_ori[n,NJ+n_selector,3] (either output may be NULL) */
int bf_smpl_forward(bf_model *m, int n, const float *betas, const float *global_orient,
                    const float *body_pose, float *vertices, float *joints, float *joints_ori);

/* The model's forward for `n` packed parameter vectors params[n,n_params] (any model kind): vertices[n,NV,3] in
 * model space and joints[n,n_joint_map,3], both before the similarity (either may be NULL). */
int bf_model_forward(bf_model *m, int n, const float *params, float *vertices, float *joints);

/* One batch = F frames that SMPLify.__call__ (smplify.py:84-250) would process one after another
 * (apps/genebody_fitting.py:183-192), each with V calibrated views. */
int bf_batch_create(bf_model *m, int n_frames, int n_views, bf_batch **out);
void bf_batch_destroy(bf_batch *b);

/* c2w[F,V,4,4], K[F,V,3,3]: what the caller passes as `c2ws`, `Ks` (smplify.py:84); inverted to
 * world-to-camera here (smplify.py:131-135). */
int bf_batch_set_cameras(bf_batch *b, const float *c2w, const float *K);
/* keypoints[F,V,n_loss_joints,3] = (x, y, confidence): keypoints[i]['pose'] of loss.py:160.  A view
 * without a detection (None, loss.py:157) is passed with all confidences 0.  n_use_frames[F] is the
 * divisor len(use_frames) of loss.py:197 (NULL -> V). */
int bf_batch_set_keypoints(bf_batch *b, const float *keypoints, const int32_t *n_use_frames);
/* init_betas[F,NB], init_pose[F,72] = net_output of smplify.py:103 (SMPL-X takes [:, 3:66] as body pose,
 * :110-112; eyes / hand PCA start at 0, :118-122); transl=0, scale=1 (:126-128) */
int bf_batch_set_init(bf_batch *b, const float *init_betas, const float *init_pose);
/* The NEXT frame's keypoints[F,V,n_loss_joints,3], n_use_frames[F] (NULL -> V), init_betas[F,NB], init_pose[F,72] (layouts of
 * bf_batch_set_keypoints / bf_batch_set_init) without waiting for the work in flight: the frame loop of
 * apps/genebody_fitting.py:183-192 hands SMPLify.__call__ new detections and a new HMR estimate every frame (and loss.py:160
 * uploads the keypoints again every iteration).  The arrays are copied into pinned staging before the call returns; their
 * transfer into the device arena the running fit does not read is queued behind that fit on the batch's stream - or, in the
 * frame-after-frame loop (fits with BF_FIT_RESET | BF_FIT_FETCH | BF_FIT_NOTIME), on the batch's second stream, where it runs
 * UNDER the fit in flight (round 5; the fit that reads it waits for it, on the host).  The next bf_fit must carry
 * BF_FIT_RESET (anything else fails with BF_ERR_INVALID).  Cameras, masks and scans are not staged: they stay as set. */
int bf_batch_stage_inputs(bf_batch *b, const float *keypoints, const int32_t *n_use_frames, const float *init_betas, const float *init_pose);
/* Re-arm the batch for another fit of the same inputs without touching the host: restores the
 * parameters of the last bf_batch_set_init / bf_batch_set_params and clears the Adam state, as
 * stream-ordered device copies.  (The reference rebuilds everything per frame, body_fitting.py:82.) */
int bf_batch_reset(bf_batch *b);
/* direct access to the packed optimised scalars [F,n_params] (for stage-level tests / warm starts) */
int bf_batch_set_params(bf_batch *b, const float *params);
int bf_batch_get_params(bf_batch *b, float *params);

/* The optimisation loop smplify.py:177-213: n_iters Adam steps on every frame.  Asynchronous. */
int bf_fit(bf_batch *b, int n_iters, const bf_hyper *hyper, uint32_t flags);
/* One evaluation of multiview_keypoint_loss (loss.py:139-230) and its gradient at the current
 * parameters, no update: terms[F,4] = reprojection, pose_prior, angle_prior, shape_prior
 * (loss.py:219-224); grads[F,n_params]. */
int bf_loss_grad(bf_batch *b, const bf_hyper *hyper, float *terms, float *grads);
int bf_batch_sync(bf_batch *b);

/* rtn_dict of smplify.py:216-226 after bf_fit (any pointer may be NULL):
 * vertices[F,NV,3], joints[F,n_joint_map,3], full_pose[F,3NJ] come from the LAST forward pass
 * (parameters before the final step, as in the reference); the stepped parameters come from
 * bf_batch_get_params.  loss_terms[F,4] are those of the last evaluated iteration. */
int bf_batch_get_result(bf_batch *b, float *vertices, float *joints, float *full_pose, float *loss_terms);
/* The result of the fit issued BEFORE the last one, without waiting for the last one: with bf_batch_stage_inputs this makes the
 * frame loop a two-deep pipeline (frame i's rtn_dict, smplify.py:216-226, is read while frame i+1 is being fitted).  Both fits
 * must have been issued with BF_FIT_RESET | BF_FIT_FETCH | BF_FIT_NOTIME on the keypoint-only path - their results then sit in
 * the batch's two result arenas.  params[F,n_params] + the outputs of bf_batch_get_result; any pointer may be NULL. */
int bf_batch_get_previous(bf_batch *b, float *params, float *vertices, float *joints, float *full_pose, float *loss_terms);
/* packed [F,n_params] stepped parameters copied into a DEVICE buffer (e.g. the send buffer of the
 * final RCCL all-gather when frames are sharded over GPUs) */
int bf_batch_export_params_dev(bf_batch *b, void *dst_dev);

/* ---- scan closest-point path (use_mesh, BASELINE config 5) ------------------------------------------
 * MeshGridSearcher(verts, faces) (utils/mesh_grid_searcher.py:51-79 -> insert_grid_surface,
 * thirdparty/mesh_grid/mesh_grid.cpp:31-52): verts[n_verts,3], faces[n_faces,3] int32. */
/* ORDER OF DESTRUCTION: a scan may be destroyed while batches still hold it (bf_batch_set_scans): bf_scan_destroy then waits for
 * the device and DETACHES every scan from those batches, which are marked: their next bf_fit / bf_fit_displacement returns
 * BF_ERR_INVALID ("a scan this batch held was destroyed") until bf_batch_set_scans is called again - with NULL to go on without
 * scans (rounds 4-5 let the fit run silently without the closest-point loss).  A batch may be destroyed before its scans.
 * Not thread-safe against a bf_fit / bf_batch_set_scans of a holding batch running at the same moment on another thread.
 * bf_scan_create builds the grid on the NULL stream and waits for that stream only (the library's streams are non-blocking):
 * a fit in flight on a batch's stream keeps running. */
int bf_scan_create(int device, int n_verts, const float *verts, int n_faces, const int32_t *faces, bf_scan **out);
void bf_scan_destroy(bf_scan *s);
float bf_scan_height(const bf_scan *s);                 /* (max - min)[1], smplify.py:150-151 */
/* The library caches the device blocks of destroyed scans per device (at most 2 GB; a failed hipMalloc empties it and retries):
 * give them back to the runtime now.  -> bytes released, < 0 on a bad device index.  (Waits for the device, as hipFree does.) */
int64_t bf_device_cache_trim(int device);
int bf_scan_grid_info(const bf_scan *s, int32_t dims[3], float origin_step[4]);
/* The tensors insert_grid_surface leaves with its caller (mesh_grid.cpp:129-136, mesh_grid_kernel.cu:178-236; built on
 * the device by bf_scan_create): tri_num[nx*ny*nz] = inclusive cumulative triangle count per cell (cell = (x*ny+y)*nz+z),
 * tri_idx[*n_entries] = face id + 1 per list entry, ascending inside a cell (the reference's order inside a cell is
 * whatever its atomicCAS race produced).  Any pointer may be NULL; call once with tri_idx NULL to learn n_entries. */
int bf_scan_grid_lists(const bf_scan *s, int32_t *tri_num, int32_t *tri_idx, int32_t *n_entries);
/* MeshGridSearcher.nearest_points / search_nearest_point (mesh_grid.cpp:54-72): points[n,3] ->
 * face_ids[n] int32, nearest[n,3], bary[n,3] (any output may be NULL) */
int bf_scan_nearest(bf_scan *s, int n, const float *points, int32_t *face_ids, float *nearest, float *bary);
/* The same search with a guess per query, hint[n,3] (NULL: none) = where the nearest point is believed to be - the fit loop hands the
 * search its previous iteration's answer this way.  The guess bounds the search and is CHECKED against what was found (searched again
 * without it when it was wrong): the results are bf_scan_nearest's for any hint.  reps > 0 with kernel_us != NULL: the launch is repeated
 * and its mean device time returned (microseconds). */
int bf_scan_nearest_hinted(bf_scan *s, int n, const float *points, const float *hint, int32_t *face_ids, float *nearest, float *bary,
                           int reps, float *kernel_us);
/* The per-triangle arithmetic of every closest-point search of the process (bf_scan_nearest, the scan loss of bf_fit, SMPL+D).
 * BF_NEAREST_REFERENCE (default): search_nearest_proj as the reference's source evaluates it in float32 - Gram matrix of the corner
 * vectors, the bordered 4 x 4 system through solve4 / solve3 with their pivot order and absolute 1e-9 rank tests, IEEE divisions,
 * no fused multiply-adds (mesh_grid_kernel.cu:12-109, matrix.h:13-316): face ids, coefficients and points are those of the
 * reference's arithmetic wherever no two faces return the same distance bit for bit (such ties go to the lowest face id; the
 * reference's own order inside a cell is an atomicCAS race).  BF_NEAREST_FAST: the same rule through the 2 x 2 normal equations
 * and v_rcp_f32 - about half the instructions, other last bits (DESIGN.md 2.3 has the measured difference).
 * Also read once from the environment: BF_NEAREST_RULE=reference|fast. */
#define BF_NEAREST_REFERENCE 0
#define BF_NEAREST_FAST      1
int bf_nearest_rule_set(int rule);                     /* 0 on success, -1 for an unknown rule */
int bf_nearest_rule_get(void);
/* How the silhouette loss's contour gradients (loss.py:110-119: every contour point pulls its nearest projected vertex) reach
 * dL/dvertices inside bf_fit.  BF_MASK_FOLD_SUMS (default, round 5): the contour scan adds each point's pull onto its vertex as a
 * 64-bit fixed-point number (steps of 2^-40, exact sums: the order of the atomic additions does not matter, so a fit is reproducible
 * bit for bit) and the reverse mesh pass maps the sums back through the projection.  BF_MASK_FOLD_GATHER: the ordered walk of
 * rounds 2-4 (bf_mask_gather_kernel: a vertex adds its contour points up in contour order, in float32) - one more launch per
 * iteration, last bits of the sum differ.  bf_batch_mask_loss always takes the ordered walk.
 * Also read once from the environment: BF_MASK_FOLD=sums|gather. */
#define BF_MASK_FOLD_SUMS   0
#define BF_MASK_FOLD_GATHER 1
int bf_mask_fold_set(int mode);                        /* 0 on success, -1 for an unknown mode */
int bf_mask_fold_get(void);
/* Self-tests of the reference-arithmetic rule: out[i] = num[i] / den[i] through the kernel's division helper (exact IEEE division
 * for 1e-8 < |den| < 4, |num / den| < 2^90); the per-triangle rule itself on patches[n][9] = the corners relative to the query
 * (mesh_grid_kernel.cu:305-311) -> dist[n], coeff[n][3]; general = 0 evaluates the straight-line paths alone and returns -1 where
 * they decline. */
int bf_nearest_selftest_quot(int device, int n, const float *num, const float *den, float *out);
int bf_nearest_selftest_rule(int device, int n, const float *patches, int general, float *dist, float *coeff);
/* SurfaceNearest.backward with respect to the query points (utils/mesh_grid_searcher.py:17-49; search_nearest_point_backward,
 * mesh_grid.cpp:120-128, mesh_grid_kernel.cu:354-382 - left unfinished in the reference: its kernel never inverts the KKT matrix).
 * face_ids[n], bary[n,3] as bf_scan_nearest returned them, dnearest[n,3] = dL/d(nearest point) -> dpoints[n,3] = dL/d(query):
 * the projection onto the plane (I - n n^T), the edge (d d^T / |d|^2) or the corner (0) the closest point lies on. */
int bf_scan_nearest_backward(bf_scan *s, int n, const int32_t *face_ids, const float *bary, const float *dnearest, float *dpoints);
/* MeshGridSearcher.inside_mesh / search_inside_mesh (utils/mesh_grid_searcher.py:86-91, mesh_grid.cpp:74-90,
 * mesh_grid_kernel.cu:569-641): signs[n] = +1 inside (odd number of triangles crossed by the axis ray towards the
 * nearest grid wall), -1 outside or off the grid. */
int bf_scan_inside(bf_scan *s, int n, const float *points, float *signs);
/* MeshGridSearcher.intersects_any / search_intersect (utils/mesh_grid_searcher.py:93-99, mesh_grid.cpp:92-110,
 * mesh_grid_kernel.cu:742-1026,1029-1231): hit[n] = 1 when the ray origins[i] + t directions[i], t >= 0, meets a triangle by the
 * reference's per-triangle test intersect_tri2, its branches for rays inside a triangle's plane and for degenerate triangles included. */
int bf_scan_intersects(bf_scan *s, int n, const float *origins, const float *directions, uint8_t *hit);
/* use_mesh=True: scans[F], one per frame (NULL detaches).  Sets each frame's constant scale to
 * scan_height / 1.7 (smplify.py:156); bf_fit then adds 5 * point_cloud_loss / scan_height * imsize for
 * iterations i > n_iters // 3 (smplify.py:205-210). */
int bf_batch_set_scans(bf_batch *b, bf_scan *const *scans);
/* ---- silhouette loss (use_mask, BASELINE config 3) ----------------------------------------------------
 * masks[F,M,H,W] uint8 as loaded (> 128 = foreground, smplify.py:139); view_index[M] = position of each mask
 * view among the V views (smplify.py:141-142); per (frame, mask view) contour_count[F*M] contour points,
 * concatenated as (x, y) pairs in contour_xy (what extract_countours returns, loss.py:73-83).  bf_fit then
 * adds 5 * multview_mask_loss for iterations i > n_iters // 3 (smplify.py:197-199,210).  n_masks = 0 detaches. */
/* contour_count == NULL (and contour_xy == NULL): the contours are extracted from the masks on the device - Suzuki-Abe
 * border following = cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) - and ONE external border per mask is kept, chosen by
 * contour_select (ignored when the caller passes contours).  The reference keeps
 * `contour[np.argmax([a.shape[1] for a in contour])]` (loss.py:80); every OpenCV contour has shape [C,1,2], so the argmax
 * runs over ones and returns OpenCV's FIRST listed contour.  OpenCV lists external borders in the reverse of the order its
 * raster scan meets them (each new contour is inserted as the first child of the frame), so that is the border whose start
 * pixel comes LAST in raster order: BF_CONTOUR_OPENCV_FIRST, the default.  For a one-component silhouette all three coincide.
 * In this form the call returns once the mask upload and the border following are QUEUED (on the batch's second stream); the
 * next bf_fit / bf_batch_mask_loss collects the contours right before the first kernel that reads them - in a fit, under the
 * iterations that need no silhouette yet.  The caller's `masks` buffer is copied before the call returns. */
#define BF_CONTOUR_OPENCV_FIRST 0   /* the last external border the raster scan meets = contours[0] of OpenCV = what loss.py:80 keeps */
#define BF_CONTOUR_RASTER_FIRST 1   /* the first external border the raster scan meets */
#define BF_CONTOUR_LONGEST      2   /* the longest external border (first on ties): the evident intent of loss.py:80 */
int bf_batch_set_masks(bf_batch *b, int n_masks, const int32_t *view_index, int H, int W, const uint8_t *masks,
                       const int32_t *contour_count, const float *contour_xy, int contour_select);
/* The NEXT frame's silhouettes without draining the work in flight (a capture hands SMPLify new masks with every frame,
 * apps/genebody_fitting.py:183-192; smplify.py:138-144): same views and image shape as the masks attached with bf_batch_set_masks
 * (contour_count = NULL there: contours on the device).  Binarisation, upload and border following (loss.py:73-83) go into a second
 * arena, on the batch's second stream, under the fit in flight; the next bf_fit uses them.  Two-deep: the call waits for the fit that
 * last read the arena it overwrites. */
int bf_batch_stage_masks(bf_batch *b, int n_masks, const int32_t *view_index, int H, int W, const uint8_t *masks, int contour_select);
/* extract_countours (smplify/loss.py:73-83) on its own: masks[n,H,W] uint8 (non-zero = foreground) -> counts[n] and,
 * when xy != NULL, the (x, y) points of the n contours concatenated (sum(counts) pairs; call with xy == NULL first). */
int bf_extract_contours(int device, int n, int H, int W, const uint8_t *masks, int32_t *counts, float *xy, int select);
/* one evaluation of multview_mask_loss (loss.py:85-130) at the current parameters: loss[F] and its gradient
 * w.r.t. body_vertices, dverts[F,NV,3] (either may be NULL) */
int bf_batch_mask_loss(bf_batch *b, const bf_hyper *hyper, float *loss, float *dverts);

/* SMPL+D stage (displacement=True, smplify.py:228-247): n_iters Adam steps (lr 5e-2) on a per-vertex
 * displacement of the vertices returned by the last bf_fit, against each frame's scan:
 * loss = icp + (normal_loss + laplacian) * constant_scale * 0.1.  Needs faces in the model and scans. */
int bf_fit_displacement(bf_batch *b, int n_iters, const bf_hyper *hyper);
int bf_batch_get_displacement(bf_batch *b, float *displacement /*[F,NV,3]*/);


/* ---- frames sharded over the GPUs of one node (BASELINE config 4 / 5) -----------------------------------
 * Replaces the serial `for frame` loop of apps/genebody_fitting.py:183-192: frames are independent, so frame f goes to
 * shard f // ceil(F / n) in contiguous blocks, model and cameras are replicated, nothing is exchanged during the fit, and
 * the fitted parameters are all-gathered ONCE over RCCL (xGMI) at the end.  No PyTorch on this path; librccl is opened
 * lazily the first time a communicator is needed. */
/* host arithmetic, no device: the partition (blocks differ by at most one frame), the per-shard capacity the all-gather is
 * padded to, and the unpacking of a gathered [n_shards][capacity][width] block into [n_frames][width] */
int bf_shard_range(int n_frames, int n_shards, int shard, int32_t *first, int32_t *count);
int bf_shard_capacity(int n_frames, int n_shards);
int bf_shard_unpack(const float *gathered, int n_frames, int n_shards, int width, float *out);
/* block starts of the silhouette contours (bf_batch_set_masks' contour_count[n_frames * n_masks] / contour_xy) per shard:
 * xy_first[n_shards + 1] = (x, y) pairs in front of every shard's first contour, then their total */
int bf_shard_contour_offsets(int n_frames, int n_shards, int n_masks, const int32_t *contour_count, int64_t *xy_first);

/* ONE process driving n devices: a bf_model + bf_batch + stream per device (devices == NULL -> 0..n-1),
 * ncclCommInitAll on first use.  The setters take the arrays of the whole job [F, ...] (layouts of the bf_batch_set_*
 * counterparts) and hand every device its block; bf_group_fit queues bf_fit on every device and returns. */
typedef struct bf_group bf_group;
int bf_group_create(const bf_model_desc *desc, int n_devices, const int32_t *devices, int n_frames, int n_views, bf_group **out);
void bf_group_destroy(bf_group *g);
int bf_group_n_devices(const bf_group *g);
int bf_group_n_params(const bf_group *g);
int bf_group_shard(const bf_group *g, int i, int32_t *device, int32_t *first, int32_t *count);
bf_batch *bf_group_batch(bf_group *g, int i);      /* device i's block as an ordinary batch (masks, scans, results, timing) */
bf_model *bf_group_model(bf_group *g, int i);
int bf_group_set_cameras(bf_group *g, const float *c2w, const float *K);
int bf_group_set_keypoints(bf_group *g, const float *keypoints, const int32_t *n_use_frames);
int bf_group_set_init(bf_group *g, const float *init_betas, const float *init_pose);
/* the whole job's next frames without draining the devices (bf_batch_stage_inputs per device) */
int bf_group_stage_inputs(bf_group *g, const float *keypoints, const int32_t *n_use_frames, const float *init_betas, const float *init_pose);
/* masks[F,M,H,W] + optional contours of the whole job (bf_batch_set_masks per device; the devices extract their contours side by side) */
int bf_group_set_masks(bf_group *g, int n_masks, const int32_t *view_index, int H, int W, const uint8_t *masks,
                       const int32_t *contour_count, const float *contour_xy, int contour_select);
int bf_group_stage_masks(bf_group *g, int n_masks, const int32_t *view_index, int H, int W, const uint8_t *masks, int contour_select);   /* bf_batch_stage_masks, every device its block */
/* scans[F]: frame f's scan must have been created on the device of f's block (bf_group_shard); NULL detaches */
int bf_group_set_scans(bf_group *g, bf_scan *const *scans);
/* Every device's bf_fit is issued from a host thread of its own (created with the group), so the devices run side by side also
 * when a call enqueues hundreds of launches (silhouette / scan losses); returns when all calls have returned. */
int bf_group_fit(bf_group *g, int n_iters, const bf_hyper *hyper, uint32_t flags);
int bf_group_fit_displacement(bf_group *g, int n_iters, const bf_hyper *hyper);
int bf_group_sync(bf_group *g);
/* ranks of the group's RCCL communicator (created on first use) */
int bf_group_comm_size(bf_group *g);
/* the path's one collective: grouped ncclAllGather of the packed parameters, stream-ordered behind the fits;
 * params[F][n_params] (host) = the copy that arrived on device `from_peer` */
int bf_group_gather_params(bf_group *g, float *params, int from_peer);

/* One process PER device (launched like `torch.distributed.run`: RANK / LOCAL_RANK / WORLD_SIZE in the environment):
 * rank 0 calls bf_comm_unique_id and hands the 128 bytes to the other ranks through the host (bodyfitting_amd/shard.py
 * uses the file system), every rank calls bf_comm_create (ncclCommInitRank). */
typedef struct bf_comm bf_comm;
int bf_comm_unique_id(uint8_t id[128]);
int bf_comm_create(const uint8_t id[128], int rank, int world, int device, bf_comm **out);
void bf_comm_destroy(bf_comm *c);
int bf_comm_size(const bf_comm *c);
int bf_comm_barrier(bf_comm *c);                              /* device idle + every rank arrived */
int bf_comm_allreduce(bf_comm *c, double *value, int op);    /* in place over the ranks; op 0 = sum, 1 = max */
/* the batch holds block `rank` of bf_shard_range(n_frames, world, .); params[n_frames][n_params] (host) on every rank */
int bf_comm_gather_params(bf_comm *c, bf_batch *b, int n_frames, float *params);

/* ---- texture fitting (smplify/texture_fitting.py:220-301) -------------------------------------------------------
 * The loop of TextureFitting.__call__ (:240-275): per iteration both meshes are rendered from one view with neural_renderer
 * (Renderer.render_rgb, camera_mode='projection', ambient light only, anti-aliasing by 2 x 2 super-sampling), the loss is
 * sum |scan_img - smpl_img| and Adam steps the per-face texture cubes of the SMPL+D mesh; only the textures are differentiated.
 * The rasteriser, texture sampling and backward_textures of thirdparty/neural_renderer (cuda/rasterize_cuda_kernel.cu:24-252,
 * 498-540) are restated as HIP kernels; the UV-space texture image of :298 is bf_texfit_render_ndc;
 * file formats (OBJ / MTL / texture images), the inpainting CNN and the cv2 morphology of render_texture_map's `morph` branch are
 * out of scope. */
typedef struct bf_texfit bf_texfit;
int bf_texfit_create(int device, int image_size, int texture_size, float near, float far, const float *background /*[3] or NULL = white*/,
                     int anti_aliasing, bf_texfit **out);
void bf_texfit_destroy(bf_texfit *x);
/* which: 0 = target (textured scan), 1 = the mesh whose textures are fitted; textures[n_faces][ts][ts][ts][3] */
int bf_texfit_set_mesh(bf_texfit *x, int which, int n_verts, const float *verts, int n_faces, const int32_t *faces, const float *textures);
/* Renderer.render_rgb: R[9], t[3] (world to camera), K[9], orig_size -> rgb[3][image_size][image_size] */
int bf_texfit_render(bf_texfit *x, int which, const float *R, const float *t, const float *K, float orig_size, float *rgb);
/* Renderer.render_texture (thirdparty/neural_renderer/neural_renderer/renderer.py:294-346), the rasteriser behind render_texture_map
 * (smplify/texture_fitting.py:149-151) and the UV-space texture image smpl.png (:298): a mesh whose vertices ndc[n_verts][3] are
 * ALREADY normalised device coordinates (the OBJ's vt lines mapped to [-1, 1], z = 1; the caller appends the reversed faces with
 * their cube axes swapped, renderer.py:338-340) is rasterised without projection -> rgb[3][image_size][image_size],
 * depth[image_size][image_size] (far where nothing was drawn); either may be NULL. */
int bf_texfit_render_ndc(bf_texfit *x, int n_verts, const float *ndc, int n_faces, const int32_t *faces, const float *textures, float *rgb, float *depth);
/* one iteration (:262-270) from this view; *loss (may be NULL) = the loss before the step */
int bf_texfit_step(bf_texfit *x, const float *R, const float *t, const float *K, float orig_size, float lr, double *loss);
/* loss and d loss / d textures [n_faces][ts][ts][ts][3] of the fitted mesh from this view, without a step */
int bf_texfit_loss_grad(bf_texfit *x, const float *R, const float *t, const float *K, float orig_size, double *loss, float *grad);
int bf_texfit_get_textures(bf_texfit *x, float *textures);

/* Device time of the kernels of the last bf_fit on this batch, from HIP events on the batch's
 * stream: ms[0] = fit loop kernel(s), ms[1] = final full-mesh forward kernel, ms[2] = joints kernel +
 * result fetch, ms[3] = whole call.  (With BF_FIT_DENSE every iteration's mesh pass is inside ms[0].) */
int bf_batch_last_timing(bf_batch *b, float ms[4]);
/* The same, summed over every bf_fit since bf_batch_timing_reset (at most 1024 calls are kept);
 * *n_calls receives how many were summed. */
int bf_batch_timing_reset(bf_batch *b);
int bf_batch_timing_sum(bf_batch *b, float ms[4], int32_t *n_calls);
/* Duration of the single-frame full-mesh forward (bf_mesh_kernel, the HBM-bound kernel of the keypoint-only path) measured INSIDE the
 * kernel: first workgroup's start to last workgroup's end on the device's 100 MHz clock, `reps` launches on frame 0's current pose state;
 * us[3] = mean, min, max in microseconds.  What bench.py's roofline_mesh divides the kernel's algorithmic bytes by (an event bracket
 * around a 6 us kernel is half record overhead).  SMPL-sized models. */
int bf_batch_mesh_span(bf_batch *b, int reps, float us[3]);
/* Device time of the kernel classes of the LAST dense iteration (smplify.py:177-213 with use_mask / use_mesh / SMPL-X keypoints) of the
 * last bf_fit, from HIP events on the batch's stream: ms[0] pose state + forward mesh pass, ms[1] keypoint loss and / or silhouette kernels,
 * ms[2] closest-point search, ms[3] point-cloud loss + gradient, ms[4] reverse mesh pass, ms[5] reduction of the partial blocks.  `enable`
 * switches the recording for the following fits (the events cost a few microseconds of that one iteration); ms may be NULL. */
int bf_batch_dense_timing(bf_batch *b, int enable, float ms[6]);
/* How the dense iterations of this batch's fits have run: 1 = with the fit kernel resident (one launch per fit on its own stream, paced by
 * doorbells), 0 = one fit launch per iteration (the self-test found the two streams on one hardware queue - libbodyfit says so once on
 * stderr - or BF_DENSE_PERSISTENT=0, or 16+ frames), -1 = no dense fit has run yet.  Same results either way; ~3x the time without. */
int bf_batch_dense_resident(const bf_batch *b);

/* ---- test hooks (bring-up / parity tests only; not part of the drop-in surface) ----------------------------------
 * first-iteration intermediates of frame 0 written by the last bf_loss_grad launch (layout: tests/gpu_debug.py) */
int bf_batch_debug_dump(bf_batch *b, float *dst, int n);
/* first Adam moment of the SMPL+D displacement (after one step = 0.1 x the gradient) */
int bf_batch_debug_disp_moment(bf_batch *b, float *m_out);

#ifdef __cplusplus
}
#endif
#endif /* BODYFIT_H */
