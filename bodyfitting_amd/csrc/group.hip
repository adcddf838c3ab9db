// Frame sharding over the GPUs of one node inside libbodyfit: no PyTorch anywhere on this path.
//
// Frames are independent (the reference fits them in a serial loop, apps/genebody_fitting.py:183-192), so a job of F frames
// is cut into contiguous blocks, one per GPU; the body model and the cameras are replicated; nothing is exchanged during the
// fit.  The ONE collective of the path is the final gather of the packed parameters [frames_per_gpu, n_params]: an
// ncclAllGather over RCCL (xGMI inside the node).
//
//   bf_group  - ONE process drives n devices: one bf_model + bf_batch + stream per device, ncclCommInitAll, grouped all-gather.
//   bf_comm   - one process PER device (torchrun-style launch): ncclCommInitRank from a 128-byte id the ranks exchange through
//               the host (bodyfitting_amd/shard.py does that through the file system), the same all-gather, plus the barrier
//               and the max-over-ranks reduction the benchmark contract needs.
//
// librccl (570 MB of code objects) is opened lazily with dlopen the first time a communicator is needed: the single-GPU
// product path never pays for it.
#include "bf_host.h"

#include <condition_variable>
#include <dlfcn.h>
#include <functional>
#include <rccl/rccl.h>
#include <thread>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl *rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        // BF_RCCL_PATH first (a full path), then the loader's search path (a process that imported torch finds torch's copy there),
        // then ROCm's own install
        std::string tried;
        const char *env = std::getenv("BF_RCCL_PATH");
        for (const char *name : {env ? env : "", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
            if (!name[0]) continue;
            R.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.handle) break;
            const char *why = dlerror();
            tried += std::string(tried.empty() ? "" : "; ") + name + " (" + (why ? why : "?") + ")";
        }
        if (!R.handle) {
            R.error = "RCCL could not be loaded - tried: " + tried + ".  Multi-GPU jobs (bf_group_* with more than one device, bf_comm_*) need "
                      "librccl.so.1 on the loader's path or BF_RCCL_PATH=/full/path/librccl.so.1; single-GPU fitting does not use it";
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(R.handle, n);
            if (!p && R.error.empty()) R.error = std::string("librccl has no symbol ") + n;
            return p;
        };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.CommCount = (decltype(R.CommCount))sym("ncclCommCount");
        R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
        R.AllReduce = (decltype(R.AllReduce))sym("ncclAllReduce");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    });
    return &R;
}

#define RCCL_TRY(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(BF_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(r_)); \
    } while (0)

int need_rccl() {
    Rccl *R = rccl();
    if (!R->error.empty()) return fail(BF_ERR_UNSUPPORTED, "RCCL is not usable: " + R->error);
    return BF_OK;
}

}  // namespace

// One host thread per device of a group.  A dense bf_fit (silhouette / scan losses) enqueues a few launches per iteration for
// hundreds of iterations and waits for its resident fit launch to be running before it returns (scan_api.hip): issued from ONE
// thread, device k+1 would start after device k's whole sequence is queued.  Every device's calls therefore come from its own
// thread; the caller's thread posts the job to all of them and waits until every one has returned.
struct bf_worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, quit = false;
    int rc = 0;
    std::string err;

    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            std::function<int()> j = std::move(job);
            lk.unlock();
            bf_err_slot().clear();
            const int r = j();
            std::string e = r ? bf_err_slot() : std::string();      // (the error message lives in THIS thread's slot: carry it over)
            lk.lock();
            rc = r; err = std::move(e);
            has_job = false;
            cv.notify_all();
        }
    }
    void post(std::function<int()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j); has_job = true; rc = 0;
        cv.notify_all();
    }
    int wait(std::string *msg) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !has_job; });
        if (rc && msg) *msg = err;
        return rc;
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};

struct bf_group_peer {
    int device = 0, first = 0, count = 0;
    bf_model *model = nullptr;
    bf_batch *batch = nullptr;
    ncclComm_t comm = nullptr;
    DevBuf<float> send, recv;      // [cap][np], [n][cap][np]
    bf_worker *worker = nullptr;   // null in a one-device group (the caller's thread does the work)
};

struct bf_group {
    int n = 0, F = 0, V = 0, np = 0, nl = 0, nb = 0, cap = 0;
    std::vector<bf_group_peer> peers;
    bool comm_ready = false;
    std::vector<float> host;       // [n][cap][np] staging of the gathered block
};

// fn(peer) on every device of the group, each from that device's own host thread, all running side by side; returns when every
// one has returned, with the first failure (in device order) as this thread's error
static int run_all(bf_group *g, const std::function<int(bf_group_peer &)> &fn) {
    if (!g->peers[0].worker) {
        for (auto &p : g->peers) { int rc = fn(p); if (rc) return rc; }
        return BF_OK;
    }
    for (auto &p : g->peers) { bf_group_peer *pp = &p; p.worker->post([pp, &fn] { return fn(*pp); }); }
    int first_rc = BF_OK;
    std::string first_msg;
    for (auto &p : g->peers) {
        std::string msg;
        const int rc = p.worker->wait(&msg);
        if (rc && !first_rc) { first_rc = rc; first_msg = "device " + std::to_string(p.device) + ": " + msg; }
    }
    return first_rc ? fail(first_rc, first_msg) : BF_OK;
}

struct bf_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<float> send, recv;
    DevBuf<double> scalar;
    std::vector<float> host;
};

extern "C" {

// Contiguous block of frames owned by shard `shard` of `n_shards`: blocks differ by at most one frame, earlier shards take
// the remainder.  Host arithmetic only (no device needed) - the one place the partition is defined.
int bf_shard_range(int n_frames, int n_shards, int shard, int32_t *first, int32_t *count) {
    if (n_frames < 0 || n_shards <= 0 || shard < 0 || shard >= n_shards || !first || !count)
        return fail(BF_ERR_INVALID, "bf_shard_range: bad argument");
    const int base = n_frames / n_shards, rem = n_frames % n_shards;
    *first = shard * base + std::min(shard, rem);
    *count = base + (shard < rem ? 1 : 0);
    return BF_OK;
}

// frames per shard the all-gather is padded to (the largest block)
int bf_shard_capacity(int n_frames, int n_shards) {
    if (n_frames < 0 || n_shards <= 0) return 0;
    return (n_frames + n_shards - 1) / n_shards;
}

// What an all-gather of per-shard blocks padded to bf_shard_capacity() leaves behind, gathered[n_shards][cap][width], to the
// dense [n_frames][width] array in frame order.  Host arithmetic only.
int bf_shard_unpack(const float *gathered, int n_frames, int n_shards, int width, float *out) {
    if (!gathered || !out || n_frames < 0 || n_shards <= 0 || width <= 0) return fail(BF_ERR_INVALID, "bf_shard_unpack: bad argument");
    const int cap = bf_shard_capacity(n_frames, n_shards);
    for (int s = 0; s < n_shards; ++s) {
        int32_t first, count;
        bf_shard_range(n_frames, n_shards, s, &first, &count);
        std::memcpy(out + (size_t)first * width, gathered + (size_t)s * cap * width, (size_t)count * width * sizeof(float));
    }
    return BF_OK;
}

// Where every shard's block starts inside the per-job arrays the dense setters take.  contour_count[n_frames * n_masks] = points of
// every (frame, mask view) contour, concatenated in contour_xy: xy_first[s] = (x, y) pairs in front of shard s's first contour,
// xy_first[n_shards] = all of them.  Host arithmetic only.
int bf_shard_contour_offsets(int n_frames, int n_shards, int n_masks, const int32_t *contour_count, int64_t *xy_first) {
    if (n_frames < 0 || n_shards <= 0 || n_masks < 0 || !xy_first || (n_masks > 0 && n_frames > 0 && !contour_count))
        return fail(BF_ERR_INVALID, "bf_shard_contour_offsets: bad argument");
    int64_t acc = 0;
    for (int s = 0; s < n_shards; ++s) {
        int32_t first, count;
        bf_shard_range(n_frames, n_shards, s, &first, &count);
        xy_first[s] = acc;
        for (int64_t q = (int64_t)first * n_masks; q < (int64_t)(first + count) * n_masks; ++q) {
            if (contour_count[q] < 0) return fail(BF_ERR_INVALID, "bf_shard_contour_offsets: negative contour length");
            acc += contour_count[q];
        }
    }
    xy_first[n_shards] = acc;
    return BF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// one process, n devices
// ---------------------------------------------------------------------------------------------------------------------------
void bf_group_destroy(bf_group *g) {
    if (!g) return;
    for (auto &p : g->peers)
        if (p.worker) { p.worker->stop(); delete p.worker; p.worker = nullptr; }
    for (auto &p : g->peers) {
        (void)hipSetDevice(p.device);
        if (p.batch) (void)bf_batch_sync(p.batch);
        if (p.comm) (void)rccl()->CommDestroy(p.comm);
        if (p.send.p) { (void)hipFree(p.send.p); p.send.p = nullptr; }
        if (p.recv.p) { (void)hipFree(p.recv.p); p.recv.p = nullptr; }
        if (p.batch) bf_batch_destroy(p.batch);
        if (p.model) bf_model_destroy(p.model);
    }
    delete g;
}

int bf_group_create(const bf_model_desc *desc, int n_devices, const int32_t *devices, int n_frames, int n_views, bf_group **out) {
    if (!desc || !out || n_devices <= 0 || n_frames < n_devices || n_views <= 0)
        return fail(BF_ERR_INVALID, "bf_group_create: need a model, >= 1 device and at least one frame per device");
    *out = nullptr;
    const int have = bf_device_count();
    for (int i = 0; i < n_devices; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= have)
            return fail(BF_ERR_NO_DEVICE, "bf_group_create: device " + std::to_string(d) + " requested, " + std::to_string(have) + " visible");
        for (int k = 0; k < i; ++k)
            if ((devices ? devices[k] : k) == d) return fail(BF_ERR_INVALID, "bf_group_create: a device is listed twice");
    }
    auto *g = new bf_group();
    g->n = n_devices; g->F = n_frames; g->V = n_views; g->cap = bf_shard_capacity(n_frames, n_devices);
    g->peers.resize(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        bf_group_peer &p = g->peers[i];
        p.device = devices ? devices[i] : i;
        int32_t first, count;
        bf_shard_range(n_frames, n_devices, i, &first, &count);
        p.first = first; p.count = count;
        int rc = bf_model_create(desc, p.device, &p.model);
        if (!rc) rc = bf_batch_create(p.model, p.count, n_views, &p.batch);
        if (rc) { std::string keep = bf_err_slot(); bf_group_destroy(g); return fail(rc, keep); }
    }
    g->np = g->peers[0].model->np; g->nl = g->peers[0].model->nl_loss; g->nb = g->peers[0].model->nb;
    // (BF_GROUP_THREADS=1: a worker also for a one-device group - lets a 1-GPU box exercise the threaded issue path of N devices)
    const char *force = std::getenv("BF_GROUP_THREADS");
    if (n_devices > 1 || (force && force[0] == '1'))
        for (auto &p : g->peers) {
            p.worker = new bf_worker();
            bf_worker *w = p.worker;
            w->th = std::thread([w] { w->loop(); });
        }
    *out = g;
    return BF_OK;
}

int bf_group_n_devices(const bf_group *g) { return g ? g->n : 0; }
int bf_group_n_params(const bf_group *g) { return g ? g->np : 0; }

int bf_group_shard(const bf_group *g, int i, int32_t *device, int32_t *first, int32_t *count) {
    if (!g || i < 0 || i >= g->n) return fail(BF_ERR_INVALID, "bf_group_shard: bad argument");
    if (device) *device = g->peers[i].device;
    if (first) *first = g->peers[i].first;
    if (count) *count = g->peers[i].count;
    return BF_OK;
}

bf_batch *bf_group_batch(bf_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->peers[i].batch : nullptr; }
bf_model *bf_group_model(bf_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->peers[i].model : nullptr; }

// The setters take the arrays of the WHOLE job, [F, ...] in frame order, and hand every device its block.
int bf_group_set_cameras(bf_group *g, const float *c2w, const float *K) {
    if (!g || !c2w || !K) return fail(BF_ERR_INVALID, "bf_group_set_cameras: null argument");
    for (auto &p : g->peers) {
        int rc = bf_batch_set_cameras(p.batch, c2w + (size_t)p.first * g->V * 16, K + (size_t)p.first * g->V * 9);
        if (rc) return rc;
    }
    return BF_OK;
}

int bf_group_set_keypoints(bf_group *g, const float *keypoints, const int32_t *n_use_frames) {
    if (!g || !keypoints) return fail(BF_ERR_INVALID, "bf_group_set_keypoints: null argument");
    for (auto &p : g->peers) {
        int rc = bf_batch_set_keypoints(p.batch, keypoints + (size_t)p.first * g->V * g->nl * 3, n_use_frames ? n_use_frames + p.first : nullptr);
        if (rc) return rc;
    }
    return BF_OK;
}

int bf_group_set_init(bf_group *g, const float *init_betas, const float *init_pose) {
    if (!g || !init_betas || !init_pose) return fail(BF_ERR_INVALID, "bf_group_set_init: null argument");
    for (auto &p : g->peers) {
        int rc = bf_batch_set_init(p.batch, init_betas + (size_t)p.first * g->nb, init_pose + (size_t)p.first * 72);
        if (rc) return rc;
    }
    return BF_OK;
}

/* bf_batch_stage_inputs for the whole job: keypoints[F,V,nl,3], n_use_frames[F] or NULL, init_betas[F,NB], init_pose[F,72] */
int bf_group_stage_inputs(bf_group *g, const float *keypoints, const int32_t *n_use_frames, const float *init_betas, const float *init_pose) {
    if (!g || !keypoints || !init_betas || !init_pose) return fail(BF_ERR_INVALID, "bf_group_stage_inputs: null argument");
    for (auto &p : g->peers) {
        int rc = bf_batch_stage_inputs(p.batch, keypoints + (size_t)p.first * g->V * g->nl * 3, n_use_frames ? n_use_frames + p.first : nullptr,
                                       init_betas + (size_t)p.first * g->nb, init_pose + (size_t)p.first * 72);
        if (rc) return rc;
    }
    return BF_OK;
}

/* bf_batch_set_masks for the whole job: masks[F,M,H,W]; contour_count[F*M] / contour_xy as bf_batch_set_masks takes them, or
 * both NULL = extract the contours on the devices (every device its own frames, side by side) */
int bf_group_set_masks(bf_group *g, int n_masks, const int32_t *view_index, int H, int W, const uint8_t *masks,
                       const int32_t *contour_count, const float *contour_xy, int contour_select) {
    if (!g) return fail(BF_ERR_INVALID, "bf_group_set_masks: null group");
    if (n_masks > 0 && !masks) return fail(BF_ERR_INVALID, "bf_group_set_masks: null masks");
    std::vector<int64_t> xy_first(g->n + 1, 0);     // (float pairs in front of every device's block of contour points)
    if (contour_count && n_masks > 0) {
        int rc = bf_shard_contour_offsets(g->F, g->n, n_masks, contour_count, xy_first.data());
        if (rc) return rc;
    }
    return run_all(g, [&](bf_group_peer &p) {
        const size_t i = &p - g->peers.data();
        return bf_batch_set_masks(p.batch, n_masks, view_index, H, W, n_masks > 0 ? masks + (size_t)p.first * n_masks * H * W : nullptr,
                                  contour_count ? contour_count + (size_t)p.first * n_masks : nullptr,
                                  contour_xy ? contour_xy + 2 * xy_first[i] : nullptr, contour_select);
    });
}

/* bf_batch_stage_masks for the whole job: the next step's masks[F,M,H,W], every device its own block, under the fits in flight */
int bf_group_stage_masks(bf_group *g, int n_masks, const int32_t *view_index, int H, int W, const uint8_t *masks, int contour_select) {
    if (!g || !masks || n_masks <= 0) return fail(BF_ERR_INVALID, "bf_group_stage_masks: bad argument");
    return run_all(g, [&](bf_group_peer &p) {
        return bf_batch_stage_masks(p.batch, n_masks, view_index, H, W, masks + (size_t)p.first * n_masks * H * W, contour_select);
    });
}

/* bf_batch_set_scans for the whole job: scans[F], scan f created (bf_scan_create) on the device bf_group_shard reports for f's
 * block; NULL detaches */
int bf_group_set_scans(bf_group *g, bf_scan *const *scans) {
    if (!g) return fail(BF_ERR_INVALID, "bf_group_set_scans: null group");
    if (scans)
        for (auto &p : g->peers)
            for (int f = 0; f < p.count; ++f) {
                const bf_scan *s = scans[p.first + f];
                if (!s) return fail(BF_ERR_INVALID, "bf_group_set_scans: null scan");
                if (s->device != p.device)
                    return fail(BF_ERR_INVALID, "bf_group_set_scans: scan of frame " + std::to_string(p.first + f) + " lives on device " +
                                                    std::to_string(s->device) + ", its block on device " + std::to_string(p.device));
            }
    return run_all(g, [&](bf_group_peer &p) { return bf_batch_set_scans(p.batch, scans ? scans + p.first : nullptr); });
}

// bf_fit on every device's block, each issued from the device's own host thread: the n devices run side by side also when a
// call enqueues hundreds of launches (the dense losses).  Returns when every call has returned (= its work is queued).  No
// communication.
int bf_group_fit(bf_group *g, int n_iters, const bf_hyper *hyper, uint32_t flags) {
    if (!g) return fail(BF_ERR_INVALID, "bf_group_fit: null group");
    return run_all(g, [&](bf_group_peer &p) { return bf_fit(p.batch, n_iters, hyper, flags); });
}

/* bf_fit_displacement (the SMPL+D stage, smplify.py:228-247) on every device's block, side by side */
int bf_group_fit_displacement(bf_group *g, int n_iters, const bf_hyper *hyper) {
    if (!g) return fail(BF_ERR_INVALID, "bf_group_fit_displacement: null group");
    return run_all(g, [&](bf_group_peer &p) { return bf_fit_displacement(p.batch, n_iters, hyper); });
}

int bf_group_sync(bf_group *g) {
    if (!g) return fail(BF_ERR_INVALID, "bf_group_sync: null group");
    for (auto &p : g->peers) {
        int rc = bf_batch_sync(p.batch);
        if (rc) return rc;
    }
    return BF_OK;
}

static int group_comm(bf_group *g) {
    if (g->comm_ready) return BF_OK;
    int rc = need_rccl();
    if (rc) return rc;
    std::vector<int> devs(g->n);
    std::vector<ncclComm_t> comms(g->n, nullptr);
    for (int i = 0; i < g->n; ++i) devs[i] = g->peers[i].device;
    RCCL_TRY(rccl()->CommInitAll(comms.data(), g->n, devs.data()));
    const size_t blk = (size_t)g->cap * g->np;
    bool ok = true;
    for (int i = 0; i < g->n && ok; ++i) {
        bf_group_peer &p = g->peers[i];
        ok = hipSetDevice(p.device) == hipSuccess && p.send.alloc(blk) == hipSuccess && p.recv.alloc(blk * g->n) == hipSuccess &&
             bf_memset_sync(p.send.p, 0, blk * sizeof(float)) == hipSuccess;
    }
    if (!ok) {                                    // (a retry must not find half a communicator)
        for (int i = 0; i < g->n; ++i) {
            (void)rccl()->CommDestroy(comms[i]);
            g->peers[i].send.release(); g->peers[i].recv.release();
        }
        return fail(BF_ERR_HIP, "bf_group: allocating the all-gather buffers failed");
    }
    for (int i = 0; i < g->n; ++i) g->peers[i].comm = comms[i];
    g->host.resize(blk * g->n);
    g->comm_ready = true;
    return BF_OK;
}

// number of ranks of the RCCL communicator (creates it on first use): what the benchmark prints as evidence
int bf_group_comm_size(bf_group *g) {
    if (!g) return fail(BF_ERR_INVALID, "bf_group_comm_size: null group");
    int rc = group_comm(g);
    if (rc) return rc;
    int n = 0;
    RCCL_TRY(rccl()->CommCount(g->peers[0].comm, &n));
    return n;
}

// The path's one collective.  Every device copies its block of fitted parameters into its send buffer and the n devices
// all-gather over RCCL, each on its batch's own stream (stream-ordered behind the fit: no host synchronisation before the
// collective).  params[F][n_params] (host) receives the copy that landed on device `from_peer`.
int bf_group_gather_params(bf_group *g, float *params, int from_peer) {
    if (!g || !params || from_peer < 0 || from_peer >= g->n) return fail(BF_ERR_INVALID, "bf_group_gather_params: bad argument");
    int rc = group_comm(g);
    if (rc) return rc;
    const size_t blk = (size_t)g->cap * g->np;
    for (auto &p : g->peers) {
        HIP_TRY(hipSetDevice(p.device));
        rc = bf_guard_arena(p.batch);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(p.send.p, p.batch->params.p, (size_t)p.count * g->np * sizeof(float), hipMemcpyDeviceToDevice, p.batch->stream));
    }
    RCCL_TRY(rccl()->GroupStart());
    for (auto &p : g->peers) {
        ncclResult_t r = rccl()->AllGather(p.send.p, p.recv.p, blk, ncclFloat, p.comm, p.batch->stream);
        if (r != ncclSuccess) { (void)rccl()->GroupEnd(); return fail(BF_ERR_HIP, std::string("ncclAllGather: ") + rccl()->GetErrorString(r)); }
    }
    RCCL_TRY(rccl()->GroupEnd());
    bf_group_peer &src = g->peers[from_peer];
    HIP_TRY(hipSetDevice(src.device));
    HIP_TRY(hipMemcpyAsync(g->host.data(), src.recv.p, blk * g->n * sizeof(float), hipMemcpyDeviceToHost, src.batch->stream));
    for (auto &p : g->peers) {                     // the collective is finished when every participant's stream has drained
        HIP_TRY(hipSetDevice(p.device));
        HIP_TRY(hipStreamSynchronize(p.batch->stream));
    }
    return bf_shard_unpack(g->host.data(), g->F, g->n, g->np, params);
}

// ---------------------------------------------------------------------------------------------------------------------------
// one process per device
// ---------------------------------------------------------------------------------------------------------------------------
int bf_comm_unique_id(uint8_t id[128]) {
    if (!id) return fail(BF_ERR_INVALID, "bf_comm_unique_id: null argument");
    int rc = need_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId u;
    RCCL_TRY(rccl()->GetUniqueId(&u));
    std::memcpy(id, &u, 128);
    return BF_OK;
}

void bf_comm_destroy(bf_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bf_comm_create(const uint8_t id[128], int rank, int world, int device, bf_comm **out) {
    if (!id || !out || world <= 0 || rank < 0 || rank >= world) return fail(BF_ERR_INVALID, "bf_comm_create: bad argument");
    *out = nullptr;
    if (device < 0 || device >= bf_device_count()) return fail(BF_ERR_NO_DEVICE, "bf_comm_create: no such HIP device");
    int rc = need_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    auto *c = new bf_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    ncclResult_t r = rccl()->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return fail(BF_ERR_HIP, std::string("ncclCommInitRank: ") + rccl()->GetErrorString(r)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || c->scalar.alloc(2) != hipSuccess) {
        bf_comm_destroy(c);
        return fail(BF_ERR_HIP, "bf_comm_create: stream / buffer allocation failed");
    }
    *out = c;
    return BF_OK;
}

int bf_comm_size(const bf_comm *c) {
    if (!c) return 0;
    int n = 0;
    if (rccl()->CommCount(c->comm, &n) != ncclSuccess) return 0;
    return n;
}

// in-place max / sum over the ranks of one double (op: 0 = sum, 1 = max)
int bf_comm_allreduce(bf_comm *c, double *value, int op) {
    if (!c || !value || (op != 0 && op != 1)) return fail(BF_ERR_INVALID, "bf_comm_allreduce: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->scalar.p, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(rccl()->AllReduce(c->scalar.p, c->scalar.p + 1, 1, ncclDouble, op ? ncclMax : ncclSum, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(value, c->scalar.p + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BF_OK;
}

// every rank has reached this point and this rank's device is idle
int bf_comm_barrier(bf_comm *c) {
    if (!c) return fail(BF_ERR_INVALID, "bf_comm_barrier: null communicator");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    double one = 1.0;
    int rc = bf_comm_allreduce(c, &one, 0);
    if (rc) return rc;
    if ((int)(one + 0.5) != c->world) return fail(BF_ERR_HIP, "bf_comm_barrier: the all-reduce did not see every rank");
    return BF_OK;
}

// The final gather when every rank owns one device: this rank's batch holds block `rank` of the bf_shard_range partition of
// n_frames; params[n_frames][n_params] (host) receives everybody's.  Runs on the batch's stream, behind the fit.
int bf_comm_gather_params(bf_comm *c, bf_batch *b, int n_frames, float *params) {
    if (!c || !b || !params || n_frames < c->world) return fail(BF_ERR_INVALID, "bf_comm_gather_params: bad argument");
    int32_t first, count;
    bf_shard_range(n_frames, c->world, c->rank, &first, &count);
    if (count != b->F) return fail(BF_ERR_INVALID, "bf_comm_gather_params: the batch does not hold this rank's block of the partition");
    if (b->m->device != c->device) return fail(BF_ERR_INVALID, "bf_comm_gather_params: batch and communicator live on different devices");
    HIP_TRY(hipSetDevice(c->device));
    const int np = b->m->np;
    const size_t blk = (size_t)bf_shard_capacity(n_frames, c->world) * np;
    if (c->send.n != blk) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->send.p) { (void)hipFree(c->send.p); c->send.p = nullptr; }
        if (c->recv.p) { (void)hipFree(c->recv.p); c->recv.p = nullptr; }
        HIP_TRY(c->send.alloc(blk));
        HIP_TRY(c->recv.alloc(blk * c->world));
        HIP_TRY(bf_memset_sync(c->send.p, 0, blk * sizeof(float)));
        c->host.resize(blk * c->world);
    }
    int rc = bf_guard_arena(b);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->send.p, b->params.p, (size_t)count * np * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
    RCCL_TRY(rccl()->AllGather(c->send.p, c->recv.p, blk, ncclFloat, c->comm, b->stream));
    HIP_TRY(hipMemcpyAsync(c->host.data(), c->recv.p, blk * c->world * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return bf_shard_unpack(c->host.data(), n_frames, c->world, np, params);
}

}  // extern "C"
