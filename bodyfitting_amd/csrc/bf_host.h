// Host-side private definitions shared by api.hip and scan_api.hip.
#pragma once
#include "../../include/bodyfit.h"
#include "bf_internal.h"

#include <atomic>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

std::string &bf_err_slot();
int bf_fail(int code, const std::string &msg);
#define fail bf_fail

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(BF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

// hipMemset of device memory returns before the fill has run, and the fill is on the NULL stream: work enqueued afterwards on a
// non-blocking stream (every stream of this library) is not ordered behind it.  Fills that are not stream-ordered wait here.
inline hipError_t bf_memset_sync(void *p, int value, size_t bytes) {
    hipError_t e = hipMemset(p, value, bytes);
    return e == hipSuccess ? hipStreamSynchronize(nullptr) : e;
}
// A cache of freed device blocks per device (bf_pool_alloc / bf_pool_free), for objects that come and go with every frame of a
// capture: a scan is created, attached, fitted against and destroyed once per frame (apps/genebody_fitting.py:183-192), and every
// hipFree waits for the whole device - i.e. for the fit of the PREVIOUS frame that is still running - while a hipMalloc of a fresh
// block costs tens of microseconds.  A block is handed out again for a request of its size up to 25 % smaller (and comes back under its true size); the
// cache holds at most 2 GB per device (beyond that a block is really freed), gives everything back when a hipMalloc fails, and
// bf_pool_trim() empties it.  The caller guarantees what hipFree used to: nothing on the device
// still uses a block it gives back (bf_scan_destroy waits for the device itself when the scan is still attached to a batch, and detaches it).
struct BfPool {
    std::mutex mu;
    std::multimap<size_t, void *> blocks[16];
    size_t held[16] = {0};
};
inline BfPool &bf_pool() { static BfPool P; return P; }
// every cached block of a device goes back to the runtime (device idle or not: hipFree waits); -> bytes released
inline size_t bf_pool_trim(int dev) {
    std::vector<void *> drop;
    size_t bytes = 0;
    if (dev >= 0 && dev < 16) {
        BfPool &P = bf_pool();
        std::lock_guard<std::mutex> lk(P.mu);
        for (auto &kv : P.blocks[dev]) drop.push_back(kv.second);
        bytes = P.held[dev];
        P.blocks[dev].clear();
        P.held[dev] = 0;
    }
    for (void *q : drop) (void)hipFree(q);
    return bytes;
}
// `*got` = the size of the block handed out (>= bytes): what bf_pool_free must be told, so that the cache's accounting holds
inline hipError_t bf_pool_alloc(void **p, size_t bytes, size_t *got) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bytes = (bytes + 255) & ~(size_t)255;
    *got = bytes;
    if (dev >= 0 && dev < 16) {
        BfPool &P = bf_pool();
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.blocks[dev].lower_bound(bytes);
        if (it != P.blocks[dev].end() && it->first <= bytes + bytes / 4 + 4096) {
            *p = it->second;
            *got = it->first;
            P.held[dev] -= it->first;
            P.blocks[dev].erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && bf_pool_trim(dev) > 0) {          // the cache itself may be what exhausted the device: give it back, once
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    return e;
}
// `bytes`: the block's size as bf_pool_alloc reported it
inline void bf_pool_free(void *p, size_t bytes, int dev) {
    if (!p) return;
    bytes = (bytes + 255) & ~(size_t)255;
    if (dev >= 0 && dev < 16) {
        BfPool &P = bf_pool();
        std::lock_guard<std::mutex> lk(P.mu);
        if (P.held[dev] + bytes <= ((size_t)2 << 30)) { P.blocks[dev].emplace(bytes, p); P.held[dev] += bytes; return; }
    }
    (void)hipFree(p);
}
inline int bf_alloc_index() { static std::atomic<int> counter{0}; return counter++; }      // (of this translation unit's allocations, all types; bf_group's workers allocate side by side)
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool view = false;          // a slice of another allocation: not freed here
    int pool_dev = -1;          // >= 0: the block came from (and goes back to) bf_pool of that device
    size_t pool_bytes = 0;
    void slice(T *base, size_t count) { p = base; n = count; view = true; }
    hipError_t alloc_pooled(size_t count) {       // for buffers of per-frame objects (see BfPool); contents undefined
        n = count;
        (void)hipGetDevice(&pool_dev);
        return bf_pool_alloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T), &pool_bytes);
    }
    hipError_t upload_pooled(const T *h, size_t count) {
        hipError_t e = alloc_pooled(count);
        if (e != hipSuccess) return e;
        return count == 0 ? hipSuccess : hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice);
    }
    hipError_t alloc(size_t count) {
        n = count;
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
        // BF_POISON=<byte>: fill every fresh allocation with that byte (255: NaNs) - a read of memory nobody wrote shows up in the
        // results instead of depending on what the allocator hands out (bring-up switch)
        // (BF_POISON_ONLY=<k>: only the k-th allocation of the process; BF_POISON_LOG=1 lists them on stderr)
        static const int poison = [] { const char *v = std::getenv("BF_POISON"); return v ? std::atoi(v) : -1; }();
        static const int only = [] { const char *v = std::getenv("BF_POISON_ONLY"); return v ? std::atoi(v) : -1; }();
        static const bool log = std::getenv("BF_POISON_LOG") != nullptr;
        const int k = bf_alloc_index();
        if (log) std::fprintf(stderr, "alloc %d: %zu x %zu bytes\n", k, count, sizeof(T));
        if (e == hipSuccess && poison >= 0 && (only < 0 || only == k)) {
            e = bf_memset_sync(p, poison, std::max<size_t>(count, 1) * sizeof(T));
        }
        return e;
    }
    hipError_t upload(const std::vector<T> &h) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess) return e;
        return h.empty() ? hipSuccess : hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void drop() { if (p && !view) { if (pool_dev >= 0) bf_pool_free(p, pool_bytes, pool_dev); else (void)hipFree(p); } }
    void release() { drop(); p = nullptr; n = 0; view = false; pool_dev = -1; }
    ~DevBuf() { drop(); }
};

// Scratch of the MFMA batch path of the full-mesh forward (>= BF_MFMA_MIN_FRAMES frames).  Owned by whoever owns the stream
// the forward runs on (a bf_batch, or a one-off forward call): two batches of one model never share it.
struct MeshScratch {
    DevBuf<float> pose_off;       // [F][3NV] batched pose-blend result, grown on demand
    DevBuf<float> featT;          // [K padded][F padded] pose features of a batch, frame-minor (the GEMM's A operand)
};

struct bf_model {
    int device = 0;
    int nv = 0, nj = 0, nb = 0, npf = 0, ns = 0, nl = 0, np = 0, n_levels = 0;
    int n_selector = 0, n_extra = 0, n_joint_map = 0;
    FitTab fit{};
    MeshTab mesh{};
    size_t mesh_smem = 0;
    DevBuf<float> v_template, shapedirs, posedirs, lbs_weights, j_extra;
    DevBuf<int> selector_ids, joint_map;
    DevBuf<int> depth_d, sel_nzj;
    DevBuf<float> sel_nzw;
    DevBuf<unsigned long long> desc_d;
    DevBuf<int> dfs_order, dfs_last;
    DevBuf<float> g_plane, g_ptail;
    DevBuf<int> parents, level_start, level_joints, child_start, child_list, lj_kind, lj_index;
    DevBuf<float> Jtrel;
    DevBuf<float> Jt, Jd, Jdrel, sel_vt, sel_sd, sel_pd, sel_w, g_means, g_psym, g_logw;
    // SMPL-X pose assembly / parameter routing / landmarks / dense keypoint loss
    int kind = 0, kp_dense = 0, n_lmk = 0, n_all = 0, nl_loss = 0;
    DevBuf<int> th_kind, th_off, p_kind, p_a, p_b, faces_lm, lmk_faces, dyn_faces, kp_jm, cj_start, cj_list, lmk_fv, dyn_fv;
    DevBuf<float> pose_mean, hand_comp, lmk_bary, dyn_bary;
    KpIO kp{};
    DevBuf<int> v_nzj;            // sparse skinning rows (MeshTab::v_nnz)
    DevBuf<float> v_nzw;
    // The sampled-first sub-model: the vertices the dense losses of a fit WITHOUT scans can touch - every 4th vertex (the
    // silhouette loss, loss.py:99) first, then the selector and landmark vertices of the dense keypoint loss - with the model's
    // tables gathered for them.  The loop's forward / reverse mesh passes then stream ~30 % of posedirs; the result mesh after
    // the loop is the full model's.
    struct Sub {
        bool on = false;
        int ns = 0;                   // sampled vertices = the first ns of the sub-model
        MeshTab mesh{};
        KpIO kp{};
        DevBuf<float> v_template, shapedirs, posedirs, lbs_weights, j_extra, v_nzw, posedirsT;
        DevBuf<int> v_nzj, selector_ids, faces, lmk_fv, dyn_fv;
    } sub, sub_kp;                // (sub_kp: the keypoint-only sub-model of the iterations before the dense losses switch on - no sampled vertices)
    DevBuf<float> posedirsT;      // [3NV][npf], built on first use of the dense reverse pass (under `lazy`, device-synchronised)
    DevBuf<float> fit_image;      // FitTab::lds_image of the dense-schedule fit instance, built on first use (under `lazy`)
    std::mutex lazy;              // guards the build-on-first-use tables (posedirsT, faces_d / adj)
    std::vector<int> faces_host;  // body-model topology (for the SMPL+D stage), optional
    DevBuf<int> faces_d, adj_start, adj;   // faces and the vertex -> (face, corner) lists, built on first use
};

struct bf_graph_key { int n_iters; uint32_t flags; int arena; bf_hyper h; };     // (arena: the result arena its nodes point at)

struct bf_batch {
    bf_model *m = nullptr;
    int F = 0, V = 0;
    size_t fit_smem = 0;            // dynamic LDS of the fit kernel for THIS batch's view count (the carve depends on V)
    MeshScratch scratch;            // pose_off / featT of the MFMA batch path, used on this batch's stream only
    hipStream_t stream = nullptr;
    static constexpr int kRing = 1024;
    std::vector<hipEvent_t> ring;   // kRing x 4 events: | fit | mesh | joints + fetch |
    int ring_n = 0;                 // calls recorded since the last timing reset
    hipEvent_t *ev = nullptr;       // the triple of the last call
    bool timed = false;
    DevBuf<float> params0;          // parameters of the last set_init / set_params / stage_inputs (a view into the current input arena)
    // Per-frame inputs [keypoints | params0 | ndiv] live in TWO device arenas, each fed from its own pinned staging buffer:
    // bf_batch_stage_inputs packs the next frame's inputs into the staging buffer the fit in flight does not read and queues
    // their transfer on the batch stream - the setter never drains the stream (the reference pays 48 keypoint host-to-device
    // copies per ITERATION, loss.py:160).  `keypoints`, `ndiv`, `params0` are views into arena in_cur.
    DevBuf<float> in_dev[2];
    float *h_in[2] = {nullptr, nullptr};
    size_t in_off[3] = {0, 0, 0}, in_total = 0;          // float offsets of keypoints, params0, ndiv inside an arena
    hipEvent_t ev_in[2] = {nullptr, nullptr};            // the transfer out of staging buffer k has finished
    bool in_pending[2] = {false, false};
    int in_cur = 0;
    // Staging ASIDE (round 5): the transfer of the next frame's inputs rides on the second stream, AHEAD of the mesh / hand-over tail of
    // the fit in flight - which is why that tail is enqueued late (`tail_k`: at the next entry point, bf_flush_tail) - so that the batch
    // stream holds fit kernel after fit kernel with nothing in between.  in_aside[k]: arena k's transfer is on the second stream and
    // the fit that reads it must see ev_in[k] first; in_reader[k]: sequence number of the last fit that read arena k (-1: none);
    // tail_seq: the last fit whose tail - it starts by waiting for that fit - is already on the second stream.
    bool in_aside[2] = {false, false};
    long long in_reader[2] = {-1, -1};
    long long tail_seq = -1;
    int tail_k = -1;                // result arena whose tail is still to be enqueued (-1: none)
    bool tail_big = false;
    bool in_host = false;           // the views point at the pinned staging buffer itself (BF_STAGE_MODE=zerocopy)
    int stage_mode = 0;             // 0 = copy kernel reading pinned memory, 1 = hipMemcpyAsync, 2 = zero-copy
    bool staged = false;            // inputs were staged since the last fit: the next bf_fit must carry BF_FIT_RESET
    long long fit_seq = 0;          // fits issued so far; arena_seq[k] = the fit whose result result-arena k holds
    long long arena_seq[2] = {-1, -1};
    bool arena_fetched[2] = {false, false}, arena_has_v[2] = {false, false};
    // results live in ONE device arena [params | terms | state | joints | vout] mirrored by ONE pinned host arena, so a
    // fetch is a single device-to-host copy of the prefix that is wanted
    // Two such pairs: a fresh fit (BF_FIT_RESET + GRAPH + FETCH) writes the arena the previous fit did not use and its
    // fetch runs on a second stream, under the next fit's kernels.
    DevBuf<float> res;              // (device arena 0; arena 1 is res_b)
    DevBuf<float> res_b;
    float *h_res = nullptr, *h_res_b = nullptr;
    size_t res_small = 0;           // floats up to the end of `joints` (everything but the vertices)
    size_t res_off[5] = {0, 0, 0, 0, 0}, res_cnt[5] = {0, 0, 0, 0, 0};   // params, terms, state, joints, vout
    int cur = 0;                    // arena the DevBuf views / h_* pointers are on
    hipStream_t copy_stream = nullptr;
    // dense schedule with the fit kernel resident for the whole call (BfDoor, bf_internal.h)
    hipStream_t fit_stream = nullptr;
    hipEvent_t ev_aux[2] = {nullptr, nullptr};   // fork / join of the dense keypoint loss on the second stream
    // bf_batch_dense_timing: events between the kernel classes of the LAST dense iteration of a fit (recorded only when asked for)
    bool dense_timing = false, dense_timed = false;
    hipEvent_t ev_dense[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool door_usable = false;           // the self-test at first use found the fit stream running beside the batch stream
    bool scans_lost = false;            // a scan this batch held was destroyed under it: fits fail until bf_batch_set_scans is called again
    bool kp_door_ok = false;            // ... and the second stream beside the batch stream (the doorbell join of config 5's keypoint workgroups)
    int dense_resident = -1;            // bf_batch_dense_resident: how the last dense fit ran
    hipEvent_t ev_door[2] = {nullptr, nullptr};
    DevBuf<int> door;
    int *h_resident = nullptr;          // pinned, device-visible: workgroups of the persistent launch that have started (this call)
    int kp_tickets = 0;                 // keypoint workgroups launched beside the search since the doors were last zeroed (BF_DOOR_KP's target)
    int *h_door_err = nullptr;          // pinned; copied from door[BF_DOOR_ERR] at the end of a call, read by bf_sync_all
    hipEvent_t ev_done[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    bool copy_pending[2] = {false, false};
    hipGraphExec_t graph_pipe[2] = {nullptr, nullptr};   // kernels-only graphs of the pipelined path, one per arena
    bf_graph_key graph_pipe_key[2]{};
    float *h_params = nullptr, *h_vout = nullptr, *h_joints = nullptr, *h_terms = nullptr, *h_state = nullptr;
    bool fetched = false;
    int steps_done = 0;
    bf_hyper adam_hyper{};
    int adam_cap = 0;
    DevBuf<float> proj, keypoints, params, adam_m, adam_v, grads, terms, state, vraw, vout, joints, adam_tab, debug, xpart;
    DevBuf<int> ndiv;
    bool have_result = false;
    hipGraphExec_t graph_exec = nullptr;     // BF_FIT_GRAPH: the captured [re-arm, fit, mesh, joints, fetch] sequence
    bf_graph_key graph_key{};
    // dense vertex losses (use_mesh, smplify.py:146-156,205-206)
    std::vector<struct bf_scan *> scans;
    DevBuf<ScanDev> scan_dev;
    float *h_pc_weight = nullptr;   // pinned staging of pc_weight
    DevBuf<float> cscale, pc_weight, pc_partial, pc_loss, dvout, vposed, cpts, ext_part, ext;
    DevBuf<int> cface, lmk_vid;
    bool cface_valid = false;       // cface holds the faces of an earlier closest-point call for these scans (warm start)
    DevBuf<float> jraw, lmk_w;
    // silhouette loss (use_mask, smplify.py:138-144,197-199)
    bool has_masks = false;
    MaskIO mask{};
    // Contours extracted on the device are DEFERRED: bf_batch_set_masks queues the upload of the binarised masks and the border
    // following on the second stream and returns; the bookkeeping that needs their lengths (cmax-sized buffers, offsets) is finished
    // by bf_masks_finalize right before the first kernel that reads them - in a fit that is the first iteration past dense_after, by
    // which time the keypoint-only iterations queued in front have long covered the extraction.
    bool masks_pending = false;
    unsigned char *h_masks = nullptr;   // pinned staging of the binarised masks
    size_t h_masks_n = 0, h_ccount_n = 0;
    int *h_ccount = nullptr;            // pinned: [2 * F * M] contour lengths | which half of the slab holds them
    hipEvent_t ev_masks = nullptr;
    DevBuf<float> mk_slab;              // [F*M][2][cap][2] the contour kernel's two-slot slabs
    DevBuf<int> mk_cnt2;
    DevBuf<unsigned> mk_planes;         // bit planes of images too large for LDS
    int mk_cap = 0, mk_select = 0;
    bool mk_on_device = false;          // the attached masks' contours were followed on the device (contour_count = NULL)
    std::vector<float *> mk_retired;    // outgrown buffers a finalize inside a fit could not free
    // bf_batch_stage_masks: the NEXT frame's silhouettes in an arena of their own (pinned staging, device masks, the contour kernel's
    // slab and counts), filled on the second stream under the fit in flight; the next bf_fit swaps the two arenas' pointers.
    struct MaskStage {
        unsigned char *h_masks = nullptr;
        size_t h_masks_n = 0, h_ccount_n = 0;
        int *h_ccount = nullptr;
        hipEvent_t ev = nullptr;        // the arena's upload + border following have finished
        hipEvent_t ev_used = nullptr;   // the last fit that read the arena has finished
        DevBuf<unsigned char> masks;
        DevBuf<float> slab;
        DevBuf<int> cnt2;
        DevBuf<unsigned> planes;
        int select = 0;
        bool staged = false;
    } mk_stage;
    hipEvent_t ev_masks_used = nullptr; // (the active arena's ev_used)
    std::vector<int> mk_view_host;      // the view indices the active masks were set with
    DevBuf<int> mk_view, mk_cstart, mk_ccount, mk_choice;
    DevBuf<unsigned char> mk_masks;
    DevBuf<float> mk_cxy, mk_uvi, mk_duvb, mk_cgrad, mk_part, mk_loss, mk_gpart;
    DevBuf<unsigned long long> mk_acc;       // [F][M][ns][2] fixed-point contour-gradient sums (MaskIO::acc)
    // SMPL+D stage (smplify.py:228-247)
    DevBuf<float> disp, disp_m, disp_v, disp_base, disp_P, disp_fn, disp_vn, disp_dv, disp_dPf;
    DevBuf<const float *> scan_fn;
    int disp_steps = 0;
    bool have_disp = false;
};

struct bf_scan {
    int device = 0, nv = 0, nf = 0, n_entries = 0;
    std::vector<struct bf_batch *> holders;   // batches that hold this scan (bf_batch_set_scans), one entry per frame slot, under bf_scan_links():
                                              // destroying the scan waits for the device and detaches those batches' scans first
    ScanDev dev{};
    DevBuf<float> verts, face_norms;
    DevBuf<int> faces, cell_start, cell_tris;
    DevBuf<float> cell_pack, cell_box;
};


inline std::mutex &bf_scan_links() { static std::mutex mu; return mu; }      // guards bf_scan::holders and bf_batch::scans of every object
extern "C" void bf_batch_unlink_scans(struct bf_batch *b);                                // (caller holds bf_scan_links(); device idle) batch forgets its scans, scans forget the batch

// shared between api.hip and scan_api.hip
extern "C" {
int bf_ensure_fit_image(struct bf_batch *b, FrameIO io, const HyperDev &hd);
int bf_launch_mesh(bf_model *m, MeshScratch *scr, int n, const float *state_dev, float *vraw, float *vout, float *xpart, float *joints,
                   float *joints_ori, hipStream_t stream, hipEvent_t after_mesh, float *vposed, float *jraw = nullptr,
                   int *lmk_vid = nullptr, float *lmk_w = nullptr, float *dvzero = nullptr, bool *zeroed = nullptr, bool want_xpart = false,
                   const MaskProj *mproj = nullptr, bool *projected = nullptr, int *door = nullptr, int door_target = 0,
                   const MeshTab *tab = nullptr, hipEvent_t mesh_done = nullptr, bool *mesh_done_set = nullptr);
// (dvzero: a [n][NV][3] buffer the forward pass should zero while it is at it - only the 1..15-frame kernel does, *zeroed says so;
//  want_xpart: fill xpart although no joints are asked for here - the caller forms them itself;
//  mproj: project the sampled vertices into the mask views as well - only the 1..15-frame kernel does, *projected says so)
int bf_fit_with_scans(bf_batch *b, int n_iters, const bf_hyper &h, const HyperDev &hd, FrameIO io);
int bf_dense_loss_grad(bf_batch *b, const bf_hyper &h, const HyperDev &hd, FrameIO io);
int bf_ensure_dense_buffers(bf_batch *b);
int bf_masks_finalize(bf_batch *b);      // no-op unless a deferred bf_batch_set_masks is pending
void bf_masks_commit(bf_batch *b);       // no-op unless bf_batch_stage_masks has staged the next frame's silhouettes
HyperDev bf_to_dev(const bf_hyper &h);
int bf_flush_tail(bf_batch *b);          // enqueue the deferred mesh / hand-over tail of the last frame-after-frame fit (api.hip)
int bf_sync_all(bf_batch *b);            // copy stream, then compute stream
int bf_guard_arena(bf_batch *b);         // the compute stream waits for a fetch still reading the current arena
void bf_use_arena(bf_batch *b, int k);
void bf_use_inputs(bf_batch *b, int k, bool host);
FrameIO bf_frame_io(bf_batch *b, bool want_grads);
}
extern "C" void bf_fit_image_segments(int nj, int nb, int npf, int ns, int nl, int np, int seg[6]);
extern "C" size_t bf_fit_smem_bytes(int nj, int nb, int npf, int ns, int nl, int np, int nviews);
extern "C" hipError_t bf_fit_launch(const FitTab *, const FrameIO *, const HyperDev *, int, int, const float *, int, size_t, hipStream_t, hipEvent_t);
extern "C" bool bf_fit_is_sized_smpl(const FitTab *);
