// Host side of libbodyfit: the C ABI of include/bodyfit.h over the gfx950 kernels.
#include "bf_host.h"
#include <chrono>

extern "C" hipError_t bf_fit_launch(const FitTab *, const FrameIO *, const HyperDev *, int, int, const float *, int, size_t, hipStream_t, hipEvent_t);
extern "C" __global__ void bf_pose_state_kernel(FitTab, const float *, const float *, const float *, const float *, float *, const float *, const float *, float);
extern "C" __global__ void bf_mesh_kernel(MeshTab, const float *, float *, float *, float *, float *, const float *, int *, int);
extern "C" __global__ void bf_mesh_span_kernel(MeshTab, const float *, float *, float *, float *, unsigned long long *);
extern "C" int bf_mesh_use_multi(int npf, int n);
extern "C" int bf_mesh_multi_launch(const MeshTab *, const float *, int, float *, float *, float *, float *, float *, hipStream_t, const MaskProj *, int *, int, hipEvent_t);
extern "C" __global__ void bf_mesh_epilogue_kernel(MeshTab, const float *, const float *, float *, float *, float *, float *);
extern "C" hipError_t bf_poseblend_launch(const MeshTab *M, const float *state, int n, float *featT, int kpad, int fpad, float *pose_off, hipStream_t stream);
extern "C" bool bf_mesh_batch32_fits(const MeshTab *M);
extern "C" hipError_t bf_mesh_batch32_launch(const MeshTab *M, const float *state, int n, float *vraw, float *vout, float *xpart, hipStream_t stream);
extern "C" __global__ void bf_mesh_epilogue_batch_kernel(MeshTab M, const float *state, const float *pose_off, int n_frames, float *vraw, float *vout, float *xpart);
extern "C" __global__ void bf_joints_kernel(MeshTab, const float *, const float *, const float *, float *, float *, float *, int *, float *);
extern "C" size_t bf_fit_smem_bytes(int, int, int, int, int, int, int);
extern "C" size_t bf_mesh_smem_bytes(int, int, int);

// The HIP runtime multiplexes all streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
// queue run in order.  A batch uses up to three streams that must run side by side (batch stream; second stream for a call's mesh tail
// / the side kernels of a dense iteration; the resident fit launch's stream), and an RCCL communicator in the same process brings
// streams of its own: with four queues the resident launch then lands on the batch stream's queue, its self-test fails and every
// dense iteration pays a fit launch (measured: config 3 48.6 ms per fit instead of 16.4 with a communicator created first).  Eight
// queues restore it.  Set when the library is loaded - before the first HIP call of a host that has not initialised HIP itself -
// and never over a value the user chose.
__attribute__((constructor)) static void bf_more_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

std::string &bf_err_slot() { thread_local std::string e; return e; }
int bf_fail(int code, const std::string &msg) { bf_err_slot() = msg; return code; }

extern "C" {

const char *bf_last_error(void) { return bf_err_slot().c_str(); }
const char *bf_version(void) { return "bodyfit-mi355x 0.1 (gfx950)"; }

int bf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void bf_hyper_default(bf_hyper *h) {
    if (!h) return;
    h->sigma = 100.f;
    h->pose_prior_weight = 4.78f;
    h->angle_prior_weight = 15.2f;
    h->shape_prior_weight = 5.f;
    h->constant_scale = 0.3f;
    h->imsize = 512.f;
    h->lr = 1e-2f;
    h->lr_transl_scale = 0.1f;
    h->adam_beta1 = 0.9f;
    h->adam_beta2 = 0.999f;
    h->adam_eps = 1e-8f;
    h->lr_displacement = 5e-2f;
    h->mask_cdist_form = 1.f;
    h->dense_after = -1.f;
}

int bf_model_create(const bf_model_desc *d, int device, bf_model **out) {
    if (!d || !out) return fail(BF_ERR_INVALID, "bf_model_create: null argument");
    *out = nullptr;
    if (bf_device_count() <= device || device < 0) return fail(BF_ERR_NO_DEVICE, "bf_model_create: no such HIP device");
    if (d->n_verts <= 0 || d->n_joints < 2 || d->n_joints > 64 || d->n_betas <= 0 || d->n_betas > 12)
        return fail(BF_ERR_UNSUPPORTED, "bf_model_create: need 2..64 joints and 1..12 betas");
    if (d->gmm_components != BF_GMM_M || d->gmm_dim != BF_GMM_D)
        return fail(BF_ERR_UNSUPPORTED, "bf_model_create: the GMM prior must be 8 components x 69 dims");
    if (d->n_loss_joints <= 0 || d->n_loss_joints > 192 || d->n_loss_joints > d->n_joint_map)
        return fail(BF_ERR_UNSUPPORTED, "bf_model_create: 1..192 loss joints supported");
    const bool smplx = d->model_kind == 1;
    const int n_lmk = smplx ? d->n_lmk_static + d->n_lmk_dynamic : 0;
    if (d->n_extra > 32 || d->n_joints + d->n_selector + d->n_extra + n_lmk > 256)
        return fail(BF_ERR_UNSUPPORTED, "bf_model_create: too many auxiliary joints");
    if (smplx) {
        if (d->n_joints != 55 || d->n_hand_pca <= 0 || d->n_hand_pca > 6 || !d->pose_mean || !d->left_hand_components ||
            !d->right_hand_components || !d->faces || d->n_faces <= 0 || !d->lmk_faces_idx || !d->lmk_bary_coords ||
            (d->n_lmk_dynamic > 0 && (!d->dynamic_lmk_faces_idx || !d->dynamic_lmk_bary_coords || d->n_dyn_rows < 79)) ||
            d->neck_joint < 0 || d->neck_joint >= 55)
            return fail(BF_ERR_INVALID, "bf_model_create: incomplete SMPL-X description");
    }
    const int nv = d->n_verts, nj = d->n_joints, nb = d->n_betas, npf = 9 * (nj - 1);
    if (d->parents[0] != -1) return fail(BF_ERR_INVALID, "bf_model_create: parents[0] must be -1");
    for (int j = 1; j < nj; ++j)
        if (d->parents[j] < 0 || d->parents[j] >= j) return fail(BF_ERR_INVALID, "bf_model_create: parents[i] must be in [0,i)");
    for (int i = 0; i < d->n_selector; ++i)
        if (d->selector_ids[i] < 0 || d->selector_ids[i] >= nv) return fail(BF_ERR_INVALID, "bf_model_create: selector id out of range");
    const int n_all = nj + d->n_selector + d->n_extra + ((d->model_kind == 1) ? d->n_lmk_static + d->n_lmk_dynamic : 0);
    for (int i = 0; i < d->n_joint_map; ++i)
        if (d->joint_map[i] < 0 || d->joint_map[i] >= n_all) return fail(BF_ERR_INVALID, "bf_model_create: joint_map entry out of range");
    HIP_TRY(hipSetDevice(device));

    auto *m = new bf_model();
    m->device = device;
    m->nv = nv; m->nj = nj; m->nb = nb; m->npf = npf;
    m->n_selector = d->n_selector; m->n_extra = d->n_extra; m->n_joint_map = d->n_joint_map;
    m->nl = d->n_loss_joints;
    m->nl_loss = d->n_loss_joints;
    m->kind = d->model_kind; m->n_lmk = n_lmk; m->n_all = n_all;
    m->kp_dense = d->n_loss_joints > 32;
    const int n_body = smplx ? 21 : nj - 1, n_pca = smplx ? d->n_hand_pca : 0;
    m->np = smplx ? 3 + 1 + 63 + nb + 3 + 3 + 3 + 2 * n_pca : 3 + 1 + 3 * (nj - 1) + nb + 3;

    // ---- kinematic tree: depth levels and children lists ----------------------------------
    std::vector<int> parents(d->parents, d->parents + nj), depth(nj, 0);
    int n_levels = 1;
    for (int j = 1; j < nj; ++j) { depth[j] = depth[parents[j]] + 1; n_levels = std::max(n_levels, depth[j] + 1); }
    std::vector<int> level_start(n_levels + 1, 0), level_joints;
    for (int l = 0; l < n_levels; ++l) {
        level_start[l] = (int)level_joints.size();
        for (int j = 0; j < nj; ++j) if (depth[j] == l) level_joints.push_back(j);
    }
    level_start[n_levels] = (int)level_joints.size();
    std::vector<int> child_start(nj + 1, 0), child_list;
    for (int p = 0; p < nj; ++p) {
        child_start[p] = (int)child_list.size();
        for (int j = 1; j < nj; ++j) if (parents[j] == p) child_list.push_back(j);
    }
    child_start[nj] = (int)child_list.size();
    m->n_levels = n_levels;

    // ---- loss joints -> chain joint or selector-vertex slot (loss.py:163, models/smpl.py:75) ----
    if (m->kp_dense) m->nl = 0;                   // the keypoint loss goes through the dense path (bf_kp_loss_kernel)
    std::vector<int> lj_kind(m->nl), lj_index(m->nl), sel;
    for (int k = 0; k < m->nl; ++k) {
        int s = d->joint_map[k];
        if (s < nj) { lj_kind[k] = 0; lj_index[k] = s; }
        else if (s < nj + d->n_selector) {
            int vid = d->selector_ids[s - nj];
            auto it = std::find(sel.begin(), sel.end(), vid);
            if (it == sel.end()) { sel.push_back(vid); it = sel.end() - 1; }
            lj_kind[k] = 1; lj_index[k] = (int)(it - sel.begin());
        } else {
            delete m;
            return fail(BF_ERR_UNSUPPORTED, "bf_model_create: a loss joint maps to an extra-regressor joint");
        }
    }
    const int ns = (int)sel.size();
    m->ns = ns;

    // ---- pre-contracted joint regressor (float64 accumulate, rounded once) --------------------
    std::vector<float> Jt(nj * 3), Jd((size_t)nj * 3 * nb), Jdrel((size_t)nj * 3 * nb), Jtrel(nj * 3);
    {
        std::vector<double> acc((size_t)3 + 3 * nb);
        std::vector<double> Jd64((size_t)nj * 3 * nb), Jt64((size_t)nj * 3);
        for (int j = 0; j < nj; ++j) {
            std::fill(acc.begin(), acc.end(), 0.0);
            const float *row = d->j_regressor + (size_t)j * nv;
            for (int v = 0; v < nv; ++v) {
                double w = row[v];
                if (w == 0.0) continue;
                for (int k = 0; k < 3; ++k) {
                    acc[k] += w * d->v_template[(size_t)v * 3 + k];
                    const float *sd = d->shapedirs + ((size_t)v * 3 + k) * nb;
                    for (int l = 0; l < nb; ++l) acc[3 + k * nb + l] += w * sd[l];
                }
            }
            for (int k = 0; k < 3; ++k) {
                Jt[j * 3 + k] = (float)acc[k];
                Jt64[j * 3 + k] = acc[k];
                for (int l = 0; l < nb; ++l) {
                    Jd64[((size_t)j * 3 + k) * nb + l] = acc[3 + k * nb + l];
                    Jd[((size_t)j * 3 + k) * nb + l] = (float)acc[3 + k * nb + l];
                }
            }
        }
        for (int j = 0; j < nj; ++j)
            for (int k = 0; k < 3; ++k) Jtrel[j * 3 + k] = (float)(Jt64[j * 3 + k] - (j > 0 ? Jt64[parents[j] * 3 + k] : 0.0));
        for (int j = 0; j < nj; ++j)
            for (int e = 0; e < 3 * nb; ++e) {
                double v = Jd64[(size_t)j * 3 * nb + e];
                if (j > 0) v -= Jd64[(size_t)parents[j] * 3 * nb + e];
                Jdrel[(size_t)j * 3 * nb + e] = (float)v;
            }
    }
    // ---- selector-vertex slices of the model ----------------------------------------------------
    std::vector<float> sel_vt(ns * 3), sel_sd((size_t)ns * 3 * nb), sel_pd((size_t)npf * ns * 3), sel_w((size_t)ns * nj);
    for (int s = 0; s < ns; ++s) {
        int v = sel[s];
        for (int k = 0; k < 3; ++k) {
            sel_vt[s * 3 + k] = d->v_template[(size_t)v * 3 + k];
            for (int l = 0; l < nb; ++l) sel_sd[((size_t)s * 3 + k) * nb + l] = d->shapedirs[((size_t)v * 3 + k) * nb + l];
            for (int p = 0; p < npf; ++p) sel_pd[(size_t)p * ns * 3 + s * 3 + k] = d->posedirs[(size_t)p * 3 * nv + 3 * v + k];
        }
        for (int j = 0; j < nj; ++j) sel_w[(size_t)s * nj + j] = d->lbs_weights[(size_t)v * nj + j];
    }
    std::vector<float> nzw((size_t)ns * BF_SEL_NNZ, 0.f);
    std::vector<int> nzj((size_t)ns * BF_SEL_NNZ, 0);
    int sel_nnz = 0;
    for (int s = 0; s < ns; ++s) {
        int c = 0;
        for (int j = 0; j < nj; ++j) {
            float w = sel_w[(size_t)s * nj + j];
            if (w == 0.f) continue;
            if (c < BF_SEL_NNZ) { nzw[(size_t)s * BF_SEL_NNZ + c] = w; nzj[(size_t)s * BF_SEL_NNZ + c] = j; }
            ++c;
        }
        sel_nnz = std::max(sel_nnz, c);
    }
    if (sel_nnz > BF_SEL_NNZ) sel_nnz = 0;
    // ---- GMM: symmetrised precisions and -log of the merged weights (prior.py:188-189) ---------
    const int M = BF_GMM_M, D = BF_GMM_D;
    std::vector<float> psym((size_t)M * D * D), logw(M), means(d->gmm_means, d->gmm_means + (size_t)M * D);
    for (int c = 0; c < M; ++c) {
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j)
                psym[((size_t)c * D + i) * D + j] = (float)(0.5 * ((double)d->gmm_precisions[((size_t)c * D + i) * D + j] +
                                                                   (double)d->gmm_precisions[((size_t)c * D + j) * D + i]));
        logw[c] = (float)(-std::log((double)d->gmm_nll_weights[c]));
    }

    // lane-major register images of Psym for the fit kernel (coalesced one-off load)
    std::vector<float> plane((size_t)M * BF_GMM_LD * 64, 0.f), ptail((size_t)4 * 12 * 64, 0.f);
    for (int c = 0; c < M; ++c)
        for (int j = 0; j < D; ++j)
            for (int l = 0; l < 64; ++l) plane[((size_t)c * BF_GMM_LD + j) * 64 + l] = psym[((size_t)c * D + l) * D + j];
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 60; ++l) {
            int comp = l < 30 ? 2 * w : 2 * w + 1, row = 64 + (l % 30) / 6, col = 12 * (l % 6);
            for (int e = 0; e < 12; ++e)
                if (col + e < D) ptail[((size_t)w * 12 + e) * 64 + l] = psym[((size_t)comp * D + row) * D + col + e];
        }
    int max_children = 0;
    for (int p = 0; p < nj; ++p) max_children = std::max(max_children, child_start[p + 1] - child_start[p]);
    if (max_children > 6) { delete m; return fail(BF_ERR_UNSUPPORTED, "bf_model_create: a joint has more than 6 children"); }

    bool okay = true;
    auto up_f = [&](DevBuf<float> &b, const float *src, size_t n) {
        std::vector<float> h(src, src + n);
        okay = okay && b.upload(h) == hipSuccess;
    };
    auto up_vf = [&](DevBuf<float> &b, const std::vector<float> &h) { okay = okay && b.upload(h) == hipSuccess; };
    auto up_vi = [&](DevBuf<int> &b, const std::vector<int> &h) { okay = okay && b.upload(h) == hipSuccess; };
    up_f(m->v_template, d->v_template, (size_t)nv * 3);
    up_f(m->shapedirs, d->shapedirs, (size_t)nv * 3 * nb);
    // posedirs rows are padded to a multiple of 128 bytes: a tile's 96 columns are 384 bytes, and with the natural pitch (SMPL: 82,680 B)
    // every slice straddled a fourth line that the neighbouring tile's workgroup - usually on another XCD - fetched again (counter
    // traffic 1.30 x the algorithmic bytes in rounds 2-4)
    auto pd_pitch_of = [](int cols) { return (cols + 31) & ~31; };
    auto up_rows = [&](DevBuf<float> &b, const float *src, int rows, int cols, int pitch) {
        std::vector<float> h((size_t)rows * pitch, 0.f);
        for (int r = 0; r < rows; ++r) memcpy(h.data() + (size_t)r * pitch, src + (size_t)r * cols, (size_t)cols * sizeof(float));
        okay = okay && b.upload(h) == hipSuccess;
    };
    up_rows(m->posedirs, d->posedirs, npf, 3 * nv, pd_pitch_of(3 * nv));
    if (d->n_faces > 0 && d->faces) {
        for (int i = 0; i < d->n_faces * 3; ++i)
            if (d->faces[i] < 0 || d->faces[i] >= nv) { delete m; return fail(BF_ERR_INVALID, "bf_model_create: face index out of range"); }
        m->faces_host.assign(d->faces, d->faces + (size_t)d->n_faces * 3);
    }
    up_f(m->lbs_weights, d->lbs_weights, (size_t)nv * nj);
    int v_nnz = 0;
    {   // sparse skinning rows: exact (the dropped entries are zeros)
        int mx = 0;
        for (int v = 0; v < nv; ++v) { int c = 0; for (int j = 0; j < nj; ++j) c += d->lbs_weights[(size_t)v * nj + j] != 0.f; mx = std::max(mx, c); }
        v_nnz = mx <= 4 ? 4 : (mx <= 8 ? 8 : 0);
        std::vector<int> zj((size_t)nv * std::max(v_nnz, 1), 0);
        std::vector<float> zw((size_t)nv * std::max(v_nnz, 1), 0.f);
        for (int v = 0; v < nv && v_nnz; ++v) {
            int c = 0;
            for (int j = 0; j < nj; ++j) {
                float w = d->lbs_weights[(size_t)v * nj + j];
                if (w != 0.f) { zj[(size_t)v * v_nnz + c] = j; zw[(size_t)v * v_nnz + c] = w; ++c; }
            }
        }
        up_vi(m->v_nzj, zj); up_vf(m->v_nzw, zw);
    }
    up_f(m->j_extra, d->j_regressor_extra, (size_t)d->n_extra * nv);
    up_vi(m->selector_ids, std::vector<int>(d->selector_ids, d->selector_ids + d->n_selector));
    up_vi(m->joint_map, std::vector<int>(d->joint_map, d->joint_map + d->n_joint_map));
    {
        std::vector<unsigned long long> desc(nj, 0ull);
        for (int j = nj - 1; j >= 1; --j) desc[parents[j]] |= desc[j] | (1ull << j);
        okay = okay && m->desc_d.upload(desc) == hipSuccess;
        // depth-first order (children in index order): a subtree is a contiguous run of positions, so subtree sums are
        // differences of a prefix sum
        std::vector<int> order, last(nj, 0), pos(nj, 0);
        std::vector<int> stack{0};
        while (!stack.empty()) {
            int j = stack.back(); stack.pop_back();
            pos[j] = (int)order.size(); order.push_back(j);
            for (int c = nj - 1; c >= 1; --c) if (parents[c] == j) stack.push_back(c);
        }
        for (int i = 0; i < nj; ++i) {
            int j = order[i], cnt = 1;
            for (int k = 0; k < nj; ++k) if ((desc[j] >> k) & 1ull) ++cnt;
            last[i] = i + cnt - 1;
        }
        okay = okay && m->dfs_order.upload(order) == hipSuccess && m->dfs_last.upload(last) == hipSuccess;
    }
    up_vi(m->depth_d, depth); up_vf(m->sel_nzw, nzw); up_vi(m->sel_nzj, nzj); up_vf(m->g_plane, plane); up_vf(m->g_ptail, ptail);
    up_vi(m->parents, parents); up_vi(m->level_start, level_start); up_vi(m->level_joints, level_joints);
    up_vi(m->child_start, child_start); up_vi(m->child_list, child_list);
    up_vi(m->lj_kind, lj_kind); up_vi(m->lj_index, lj_index);
    up_vf(m->Jt, Jt); up_vf(m->Jd, Jd); up_vf(m->Jdrel, Jdrel); up_vf(m->Jtrel, Jtrel);
    up_vf(m->sel_vt, sel_vt); up_vf(m->sel_sd, sel_sd); up_vf(m->sel_pd, sel_pd); up_vf(m->sel_w, sel_w);
    up_vf(m->g_means, means); up_vf(m->g_psym, psym); up_vf(m->g_logw, logw);
    if (!okay) { delete m; return fail(BF_ERR_HIP, "bf_model_create: device allocation / upload failed"); }

    FitTab &T = m->fit;
    T.nj = nj; T.nb = nb; T.npf = npf; T.ns = ns; T.nl = m->nl; T.np = m->np; T.n_levels = n_levels;
    T.nbp = 3 * n_body;
    T.off_pose = 4; T.off_beta = 4 + 3 * n_body; T.off_orient = T.off_beta + nb;
    T.n_pca = n_pca; T.off_lh = T.off_orient + 9; T.off_rh = T.off_lh + n_pca;
    T.kp_dense = m->kp_dense;
    {
        const int off_leye = T.off_orient + 3, off_reye = T.off_orient + 6;
        std::vector<int> thk(nj, 0), tho(nj, 0), pk(m->np, 0), pa(m->np, 0), pb(m->np, -1);
        for (int j = 0; j < nj; ++j) {
            if (j == 0) { thk[j] = 0; tho[j] = T.off_orient; }
            else if (j <= n_body) { thk[j] = 0; tho[j] = T.off_pose + 3 * (j - 1); }
            else if (j == 22) { thk[j] = 1; }
            else if (j == 23) { thk[j] = 0; tho[j] = off_leye; }
            else if (j == 24) { thk[j] = 0; tho[j] = off_reye; }
            else if (j < 40) { thk[j] = 2; tho[j] = j - 25; }
            else { thk[j] = 3; tho[j] = j - 40; }
        }
        for (int i = 0; i < m->np; ++i) {
            if (i < 4) pk[i] = 0;
            else if (i < T.off_beta) { int ip = i - T.off_pose; pk[i] = 1; pa[i] = 3 + ip; pb[i] = ip; }
            else if (i < T.off_orient) pk[i] = 2;
            else if (i < T.off_orient + 3) { pk[i] = 1; pa[i] = i - T.off_orient; }
            else if (i < off_reye) { pk[i] = 1; pa[i] = 23 * 3 + (i - off_leye); }
            else if (i < T.off_lh) { pk[i] = 1; pa[i] = 24 * 3 + (i - off_reye); }
            else if (i < T.off_rh) { pk[i] = 3; pa[i] = 0; pb[i] = i - T.off_lh; }
            else { pk[i] = 3; pa[i] = 1; pb[i] = i - T.off_rh; }
        }
        bool up = m->th_kind.upload(thk) == hipSuccess && m->th_off.upload(tho) == hipSuccess && m->p_kind.upload(pk) == hipSuccess &&
                  m->p_a.upload(pa) == hipSuccess && m->p_b.upload(pb) == hipSuccess;
        if (smplx) {
            std::vector<float> hc((size_t)2 * n_pca * 45);
            std::memcpy(hc.data(), d->left_hand_components, sizeof(float) * n_pca * 45);
            std::memcpy(hc.data() + (size_t)n_pca * 45, d->right_hand_components, sizeof(float) * n_pca * 45);
            up = up && m->hand_comp.upload(hc) == hipSuccess &&
                 m->pose_mean.upload(std::vector<float>(d->pose_mean, d->pose_mean + (size_t)nj * 3)) == hipSuccess;
        }
        if (!up) { delete m; return fail(BF_ERR_HIP, "bf_model_create: device allocation failed (pose tables)"); }
        T.th_kind = m->th_kind.p; T.th_off = m->th_off.p; T.p_kind = m->p_kind.p; T.p_a = m->p_a.p; T.p_b = m->p_b.p;
        T.pose_mean = smplx ? m->pose_mean.p : nullptr; T.hand_comp = smplx ? m->hand_comp.p : nullptr;
    }
    T.sel_nnz = sel_nnz; T.sel_nzw = m->sel_nzw.p; T.sel_nzj = m->sel_nzj.p;
    T.depth = m->depth_d.p; T.desc = m->desc_d.p; T.dfs_order = m->dfs_order.p; T.dfs_last = m->dfs_last.p; T.g_plane = m->g_plane.p; T.g_ptail = m->g_ptail.p;
    T.parents = m->parents.p; T.level_start = m->level_start.p; T.level_joints = m->level_joints.p;
    T.child_start = m->child_start.p; T.child_list = m->child_list.p;
    T.lj_kind = m->lj_kind.p; T.lj_index = m->lj_index.p;
    {   // deal pairs and selector vertices to the four geometry waves (FitTab::pair_slot): pairs with the most selector vertices
        // first, each to the wave that already owns its vertices, else to the wave with the fewest vertices, then the fewest pairs
        const int npairs = (m->nl + 1) / 2;
        for (int &x : T.pair_slot) x = -1;
        for (int &x : T.skin_vert) x = -1;
        T.bd_ok = (npairs >= 1 && npairs <= 16 && ns <= 4 * BF_SKIN_PER_WAVE) ? 1 : 0;
        std::vector<int> owner(ns, -1), order(std::max(npairs, 0)), nverts(4, 0), npw(4, 0);
        auto verts_of = [&](int p) {
            std::vector<int> v;
            for (int l = 2 * p; l < std::min(2 * p + 2, m->nl); ++l)
                if (lj_kind[l] == 1 && std::find(v.begin(), v.end(), lj_index[l]) == v.end()) v.push_back(lj_index[l]);
            return v;
        };
        for (int p = 0; p < npairs; ++p) order[p] = p;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return verts_of(a).size() > verts_of(b).size(); });
        for (int p : order) {
            if (!T.bd_ok) break;
            const std::vector<int> v = verts_of(p);
            int w = -1;
            for (int x : v) if (owner[x] >= 0) { if (w >= 0 && w != owner[x]) T.bd_ok = 0; w = owner[x]; }
            int fresh = 0;
            for (int x : v) if (owner[x] < 0) ++fresh;
            if (w < 0) {
                for (int c = 0; c < 4; ++c) {
                    if (npw[c] >= 4 || nverts[c] + fresh > BF_SKIN_PER_WAVE) continue;
                    if (w < 0 || nverts[c] < nverts[w] || (nverts[c] == nverts[w] && npw[c] < npw[w])) w = c;
                }
            }
            if (w < 0 || npw[w] >= 4 || nverts[w] + fresh > BF_SKIN_PER_WAVE) { T.bd_ok = 0; break; }
            T.pair_slot[4 * w + npw[w]++] = p;
            for (int x : v) if (owner[x] < 0) { owner[x] = w; T.skin_vert[BF_SKIN_PER_WAVE * w + nverts[w]++] = x; }
        }
        for (int x = 0; x < ns && T.bd_ok; ++x) if (owner[x] < 0) T.bd_ok = 0;       // (a selector vertex no pair reads: cannot happen, sel is built from the loss joints)
    }
    T.Jt = m->Jt.p; T.Jd = m->Jd.p; T.Jdrel = m->Jdrel.p; T.Jtrel = m->Jtrel.p;
    T.sel_vt = m->sel_vt.p; T.sel_sd = m->sel_sd.p; T.sel_pd = m->sel_pd.p; T.sel_w = m->sel_w.p;
    T.g_means = m->g_means.p; T.g_psym = m->g_psym.p; T.g_logw = m->g_logw.p;
    MeshTab &Q = m->mesh;
    Q.nv = nv; Q.nj = nj; Q.nb = nb; Q.npf = npf;
    Q.n_selector = d->n_selector; Q.n_extra = d->n_extra; Q.n_joint_map = d->n_joint_map;
    Q.v_template = m->v_template.p; Q.shapedirs = m->shapedirs.p; Q.posedirs = m->posedirs.p; Q.pd_pitch = pd_pitch_of(3 * nv);
    Q.lbs_weights = m->lbs_weights.p; Q.j_extra = m->j_extra.p;
    Q.selector_ids = m->selector_ids.p; Q.joint_map = m->joint_map.p;
    Q.n_tiles = (nv + BF_MESH_TILE - 1) / BF_MESH_TILE;
    Q.v_nnz = v_nnz; Q.v_nzj = m->v_nzj.p; Q.v_nzw = m->v_nzw.p;
    Q.n_lmk_static = smplx ? d->n_lmk_static : 0; Q.n_lmk_dyn = smplx ? d->n_lmk_dynamic : 0;
    Q.n_dyn_rows = smplx ? d->n_dyn_rows : 0; Q.neck_joint = smplx ? d->neck_joint : 0;
    if (smplx) {
        bool up = m->faces_lm.upload(std::vector<int>(d->faces, d->faces + (size_t)d->n_faces * 3)) == hipSuccess &&
                  m->lmk_faces.upload(std::vector<int>(d->lmk_faces_idx, d->lmk_faces_idx + d->n_lmk_static)) == hipSuccess &&
                  m->lmk_bary.upload(std::vector<float>(d->lmk_bary_coords, d->lmk_bary_coords + (size_t)d->n_lmk_static * 3)) == hipSuccess &&
                  m->dyn_faces.upload(std::vector<int>(d->dynamic_lmk_faces_idx, d->dynamic_lmk_faces_idx + (size_t)d->n_dyn_rows * d->n_lmk_dynamic)) == hipSuccess &&
                  m->dyn_bary.upload(std::vector<float>(d->dynamic_lmk_bary_coords, d->dynamic_lmk_bary_coords + (size_t)d->n_dyn_rows * d->n_lmk_dynamic * 3)) == hipSuccess;
        if (!up) { delete m; return fail(BF_ERR_HIP, "bf_model_create: device allocation failed (landmarks)"); }
        for (int i = 0; i < d->n_lmk_static; ++i)
            if (d->lmk_faces_idx[i] < 0 || d->lmk_faces_idx[i] >= d->n_faces) { delete m; return fail(BF_ERR_INVALID, "bf_model_create: landmark face out of range"); }
        for (size_t i = 0; i < (size_t)d->n_dyn_rows * d->n_lmk_dynamic; ++i)
            if (d->dynamic_lmk_faces_idx[i] < 0 || d->dynamic_lmk_faces_idx[i] >= d->n_faces) { delete m; return fail(BF_ERR_INVALID, "bf_model_create: dynamic landmark face out of range"); }
        Q.faces = m->faces_lm.p; Q.lmk_faces = m->lmk_faces.p; Q.lmk_bary = m->lmk_bary.p; Q.dyn_faces = m->dyn_faces.p; Q.dyn_bary = m->dyn_bary.p;
        // the landmarks' corner vertices, looked up once (MeshTab::lmk_fv / dyn_fv)
        std::vector<int> sfv((size_t)d->n_lmk_static * 3), dfv((size_t)d->n_dyn_rows * d->n_lmk_dynamic * 3);
        for (int i = 0; i < d->n_lmk_static; ++i)
            for (int c = 0; c < 3; ++c) sfv[(size_t)i * 3 + c] = d->faces[(size_t)d->lmk_faces_idx[i] * 3 + c];
        for (size_t i = 0; i < (size_t)d->n_dyn_rows * d->n_lmk_dynamic; ++i)
            for (int c = 0; c < 3; ++c) dfv[i * 3 + c] = d->faces[(size_t)d->dynamic_lmk_faces_idx[i] * 3 + c];
        if (dfv.empty()) dfv.push_back(0);
        if (m->lmk_fv.upload(sfv) != hipSuccess || m->dyn_fv.upload(dfv) != hipSuccess) { delete m; return fail(BF_ERR_HIP, "bf_model_create: device allocation failed (landmark corners)"); }
        Q.lmk_fv = m->lmk_fv.p; Q.dyn_fv = m->dyn_fv.p;
    }
    if (m->kp_dense) {
        // dense keypoint loss tables: loss joints -> all-joints index; per chain joint the loss joints that use it
        std::vector<int> jm(d->joint_map, d->joint_map + m->nl_loss), cs(nj + 1, 0), cl;
        for (int j = 0; j < nj; ++j) {
            cs[j] = (int)cl.size();
            for (int q = 0; q < m->nl_loss; ++q) if (jm[q] == j) cl.push_back(q);
        }
        cs[nj] = (int)cl.size();
        if (cl.empty()) cl.push_back(0);
        bool up = m->kp_jm.upload(jm) == hipSuccess && m->cj_start.upload(cs) == hipSuccess && m->cj_list.upload(cl) == hipSuccess;
        if (!up) { delete m; return fail(BF_ERR_HIP, "bf_model_create: device allocation failed (keypoint tables)"); }
        KpIO &K = m->kp;
        K.nl = m->nl_loss; K.nj = nj; K.npf = npf; K.nb = nb; K.nv = nv; K.n_all = n_all; K.n_selector = d->n_selector;
        K.n_extra = d->n_extra; K.n_lmk = n_lmk;
        K.joint_map = m->kp_jm.p; K.selector_ids = m->selector_ids.p; K.cj_start = m->cj_start.p; K.cj_list = m->cj_list.p;
        K.n_cj_list = (int)cl.size();
        K.j_extra = m->j_extra.p;
    }
    m->mesh_smem = bf_mesh_smem_bytes(nj, npf, nb);
    {
        // ---- sub-models (bf_model::Sub): the model's tables gathered for a subset of its vertices --------------------------
        //   sub     "sampled first": every 4th vertex (the silhouette loss, loss.py:99) first, then what the dense keypoint loss reads;
        //   sub_kp  (round 5) only what the dense keypoint loss reads - selector vertices, the landmark faces' corners, the support of
        //           the extra regressor: the iterations BEFORE the silhouette / scan losses switch on (i <= num_iters // 3,
        //           smplify.py:197,205) touch nothing else, with or without a scan attached.
        std::vector<char> extra(nv, 0);
        for (int i = 0; i < d->n_selector; ++i) extra[d->selector_ids[i]] = 1;
        if (smplx) {
            auto face = [&](int fidx) { if (fidx >= 0 && fidx < d->n_faces) for (int c = 0; c < 3; ++c) extra[d->faces[(size_t)fidx * 3 + c]] = 1; };
            for (int i = 0; i < d->n_lmk_static; ++i) face(d->lmk_faces_idx[i]);
            for (size_t i = 0; i < (size_t)d->n_dyn_rows * d->n_lmk_dynamic; ++i) face(d->dynamic_lmk_faces_idx[i]);
        }
        // (the extra-joint regressor rows are gathered for the sub-model's vertices: every vertex that carries regressor weight
        //  must be one of them, or the extra joints of the dense loop would be partial sums)
        for (int e = 0; e < d->n_extra; ++e)
            for (int v = 0; v < nv; ++v) if (d->j_regressor_extra[(size_t)e * nv + v] != 0.f) extra[v] = 1;
        auto build_sub = [&](bf_model::Sub &U, bool sampled_first, int max_tenths) -> int {
            std::vector<int> pos(nv, -1), S;
            if (sampled_first) for (int v = 0; v < nv; v += 4) { pos[v] = (int)S.size(); S.push_back(v); }
            const int n_samp = (int)S.size();
            for (int v = 0; v < nv; ++v) if (extra[v] && pos[v] < 0) { pos[v] = (int)S.size(); S.push_back(v); }
            const int sv = (int)S.size();
            if (sv == 0 || sv * 10 > nv * max_tenths) return BF_OK;          // (not worth it: the full model serves)
            std::vector<float> vt((size_t)sv * 3), sd((size_t)sv * 3 * nb), pd((size_t)npf * pd_pitch_of(3 * sv), 0.f), lw((size_t)sv * nj), jx((size_t)std::max(d->n_extra, 0) * sv);
            const int nnz = m->mesh.v_nnz;
            std::vector<int> zj((size_t)sv * std::max(nnz, 1), 0), sel(d->n_selector), fc;
            std::vector<float> zw((size_t)sv * std::max(nnz, 1), 0.f);
            for (int i = 0; i < sv; ++i) {
                const int v = S[i];
                for (int k = 0; k < 3; ++k) {
                    vt[(size_t)i * 3 + k] = d->v_template[(size_t)v * 3 + k];
                    for (int l = 0; l < nb; ++l) sd[((size_t)i * 3 + k) * nb + l] = d->shapedirs[((size_t)v * 3 + k) * nb + l];
                    for (int p = 0; p < npf; ++p) pd[(size_t)p * pd_pitch_of(3 * sv) + (size_t)i * 3 + k] = d->posedirs[(size_t)p * 3 * nv + (size_t)v * 3 + k];
                }
                int c = 0;
                for (int j = 0; j < nj; ++j) {
                    const float w = d->lbs_weights[(size_t)v * nj + j];
                    lw[(size_t)i * nj + j] = w;
                    if (nnz && w != 0.f) { zj[(size_t)i * nnz + c] = j; zw[(size_t)i * nnz + c] = w; ++c; }
                }
                for (int e = 0; e < d->n_extra; ++e) jx[(size_t)e * sv + i] = d->j_regressor_extra[(size_t)e * nv + v];
            }
            for (int i = 0; i < d->n_selector; ++i) sel[i] = pos[d->selector_ids[i]];
            if (smplx) {            // faces re-indexed; a corner outside the sub-model belongs to a face no landmark uses
                fc.resize((size_t)d->n_faces * 3);
                for (size_t i = 0; i < fc.size(); ++i) fc[i] = std::max(pos[d->faces[i]], 0);
            }
            bool up = U.v_template.upload(vt) == hipSuccess && U.shapedirs.upload(sd) == hipSuccess && U.posedirs.upload(pd) == hipSuccess &&
                      U.lbs_weights.upload(lw) == hipSuccess && U.j_extra.upload(jx) == hipSuccess && U.v_nzj.upload(zj) == hipSuccess &&
                      U.v_nzw.upload(zw) == hipSuccess && U.selector_ids.upload(sel) == hipSuccess && U.faces.upload(fc) == hipSuccess;
            if (!up) return fail(BF_ERR_HIP, "bf_model_create: device allocation failed (sub-model)");
            U.mesh = m->mesh;
            U.mesh.nv = sv; U.mesh.n_tiles = (sv + BF_MESH_TILE - 1) / BF_MESH_TILE;
            U.mesh.v_template = U.v_template.p; U.mesh.shapedirs = U.shapedirs.p; U.mesh.posedirs = U.posedirs.p; U.mesh.pd_pitch = pd_pitch_of(3 * sv);
            U.mesh.lbs_weights = U.lbs_weights.p; U.mesh.j_extra = U.j_extra.p; U.mesh.selector_ids = U.selector_ids.p;
            U.mesh.v_nzj = U.v_nzj.p; U.mesh.v_nzw = U.v_nzw.p;
            if (smplx) {
                U.mesh.faces = U.faces.p;
                std::vector<int> sfv((size_t)d->n_lmk_static * 3), dfv((size_t)d->n_dyn_rows * d->n_lmk_dynamic * 3);
                for (int i = 0; i < d->n_lmk_static; ++i)
                    for (int c = 0; c < 3; ++c) sfv[(size_t)i * 3 + c] = fc[(size_t)d->lmk_faces_idx[i] * 3 + c];
                for (size_t i = 0; i < (size_t)d->n_dyn_rows * d->n_lmk_dynamic; ++i)
                    for (int c = 0; c < 3; ++c) dfv[i * 3 + c] = fc[(size_t)d->dynamic_lmk_faces_idx[i] * 3 + c];
                if (dfv.empty()) dfv.push_back(0);
                if (U.lmk_fv.upload(sfv) != hipSuccess || U.dyn_fv.upload(dfv) != hipSuccess) return fail(BF_ERR_HIP, "bf_model_create: device allocation failed (sub-model landmark corners)");
                U.mesh.lmk_fv = U.lmk_fv.p; U.mesh.dyn_fv = U.dyn_fv.p;
            }
            U.kp = m->kp; U.kp.nv = sv; U.kp.selector_ids = U.selector_ids.p; U.kp.j_extra = U.j_extra.p;
            U.ns = n_samp;
            U.on = true;
            return BF_OK;
        };
        int rs = build_sub(m->sub, true, 6);
        if (rs == BF_OK && m->kp_dense) rs = build_sub(m->sub_kp, false, 5);      // (only models whose keypoint loss is dense have keypoint-only dense iterations)
        if (rs != BF_OK) { delete m; return rs; }
    }
    *out = m;
    return BF_OK;
}

void bf_model_destroy(bf_model *m) { delete m; }
int bf_model_n_params(const bf_model *m) { return m ? m->np : 0; }
int bf_model_fit_instance(const bf_model *m) { return (m && bf_fit_is_sized_smpl(&m->fit)) ? 1 : 0; }

int bf_launch_mesh(bf_model *m, MeshScratch *scr, int n, const float *state_dev, float *vraw, float *vout, float *xpart, float *joints,
                   float *joints_ori, hipStream_t stream, hipEvent_t after_mesh, float *vposed, float *jraw, int *lmk_vid,
                   float *lmk_w, float *dvzero, bool *zeroed, bool want_xpart, const MaskProj *mproj, bool *projected, int *door, int door_target,
                   const MeshTab *tab, hipEvent_t mesh_done, bool *mesh_done_set) {
    if (mesh_done_set) *mesh_done_set = false;
    if (zeroed) *zeroed = false;
    if (projected) *projected = false;
    const bool need_x = joints || joints_ori || jraw || want_xpart;
    const MeshTab &Q = tab ? *tab : m->mesh;          // (the sampled-first sub-model inside a dense loop without scans)
    dim3 grid(Q.n_tiles, n);
    const float *pose_off = nullptr;
    if (n >= BF_MFMA_MIN_FRAMES && n <= BF_BATCH32_MAX_FRAMES && !tab && !vposed && bf_mesh_batch32_fits(&m->mesh)) {
        // one or two 32-frame blocks: pose blend on the matrix cores with the epilogue behind the accumulators, ONE launch
        // (13.4 us instead of 4.6 + 15.5 + 14.5 at 32 frames; from 128 frames on the 128-frame GEMM tile below wins)
        HIP_TRY(bf_mesh_batch32_launch(&m->mesh, state_dev, n, vraw, vout, need_x ? xpart : (float *)nullptr, stream));
    } else if (n >= BF_MFMA_MIN_FRAMES && !tab) {
        // batched pose blend on the matrix cores (posedirs streamed once for up to 256 frames), then the per-frame
        // shape / skinning part only
        const size_t ncols = (size_t)m->nv * 3;
        if (!scr) return fail(BF_ERR_INVALID, "bf_launch_mesh: the batched path needs a scratch owner");
        // (the scratch belongs to this stream's owner, so draining this stream is enough before a buffer is replaced)
        if (scr->pose_off.n < (size_t)n * ncols) {
            if (scr->pose_off.p) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(scr->pose_off.p); scr->pose_off.p = nullptr; }
            HIP_TRY(scr->pose_off.alloc((size_t)n * ncols));
        }
        // A operand of the GEMM: the pose features of the batch, frame-minor and zero padded
        const int kpad = ((m->npf + 2 * BF_GEMM_KB - 1) / (2 * BF_GEMM_KB)) * 2 * BF_GEMM_KB, fpad = ((n + 127) / 128) * 128;
        if (scr->featT.n < (size_t)kpad * fpad) {
            if (scr->featT.p) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(scr->featT.p); scr->featT.p = nullptr; }
            HIP_TRY(scr->featT.alloc((size_t)kpad * fpad));
        }
        HIP_TRY(bf_poseblend_launch(&m->mesh, state_dev, n, scr->featT.p, kpad, fpad, scr->pose_off.p, stream));
        pose_off = scr->pose_off.p;
        if (m->mesh.v_nnz == 4 && m->nb <= 10 && !vposed) {
            hipLaunchKernelGGL(bf_mesh_epilogue_batch_kernel, dim3((m->nv + 127) / 128, (n + BF_EPI_FRAMES - 1) / BF_EPI_FRAMES), dim3(128),
                               (size_t)BF_EPI_FRAMES * (m->nj * 12 + 128 * 3) * sizeof(float), stream, m->mesh, state_dev, pose_off, n, vraw, vout,
                               need_x ? xpart : (float *)nullptr);
        } else
        hipLaunchKernelGGL(bf_mesh_epilogue_kernel, grid, dim3(128), 0, stream, m->mesh, state_dev, pose_off, vraw, vout,
                           need_x ? xpart : (float *)nullptr, vposed);
    } else if (bf_mesh_use_multi(m->npf, n)) {
        const int e = bf_mesh_multi_launch(&Q, state_dev, n, vraw, vout, need_x ? xpart : (float *)nullptr, vposed,
                                           dvzero, stream, mproj, door, door_target, mesh_done);
        if (e) return fail(BF_ERR_HIP, std::string("bf_mesh_multi_kernel: ") + hipGetErrorString((hipError_t)e));
        if (mesh_done && mesh_done_set) *mesh_done_set = true;          // (the event completes with the mesh dispatch: the caller records nothing)
        if (projected && mproj) *projected = true;
        if (zeroed && dvzero) *zeroed = true;
    } else
    hipLaunchKernelGGL(bf_mesh_kernel, grid, dim3(BF_MESH_TILE * 3 * BF_MESH_RG), m->mesh_smem, stream, Q,
                       state_dev, vraw, vout, need_x ? xpart : (float *)nullptr, vposed, pose_off, door, door_target);
    HIP_TRY(hipGetLastError());
    if (after_mesh) HIP_TRY(hipEventRecord(after_mesh, stream));
    if (joints || joints_ori || jraw) {
        hipLaunchKernelGGL(bf_joints_kernel, dim3(n), dim3(256), 0, stream, Q, state_dev, (const float *)vraw,
                           (const float *)xpart, joints, joints_ori, jraw, lmk_vid, lmk_w);
        HIP_TRY(hipGetLastError());
    }
    return BF_OK;
}

int bf_smpl_forward(bf_model *m, int n, const float *betas, const float *global_orient, const float *body_pose,
                    float *vertices, float *joints, float *joints_ori) {
    if (!m || n <= 0 || !betas || !global_orient || !body_pose) return fail(BF_ERR_INVALID, "bf_smpl_forward: bad argument");
    HIP_TRY(hipSetDevice(m->device));
    const int nj = m->nj, nb = m->nb, nv = m->nv;
    const size_t stride = bf_state_stride(nj, m->npf, nb);
    DevBuf<float> d_beta, d_or, d_bp, d_state, d_vraw, d_j, d_jo, d_xp;
    MeshScratch scratch;
    HIP_TRY(d_beta.upload(std::vector<float>(betas, betas + (size_t)n * nb)));
    HIP_TRY(d_or.upload(std::vector<float>(global_orient, global_orient + (size_t)n * 3)));
    HIP_TRY(d_bp.upload(std::vector<float>(body_pose, body_pose + (size_t)n * 3 * (nj - 1))));
    HIP_TRY(d_state.alloc((size_t)n * stride));
    HIP_TRY(d_vraw.alloc((size_t)n * nv * 3));
    HIP_TRY(d_xp.alloc((size_t)n * m->mesh.n_tiles * m->n_extra * 3));
    HIP_TRY(d_j.alloc((size_t)n * m->n_joint_map * 3));
    HIP_TRY(d_jo.alloc((size_t)n * (nj + m->n_selector) * 3));
    hipLaunchKernelGGL(bf_pose_state_kernel, dim3(n), dim3(128), 0, 0, m->fit, (const float *)d_beta.p,
                       (const float *)d_or.p, (const float *)d_bp.p, (const float *)nullptr, d_state.p,
                       (const float *)nullptr, (const float *)nullptr, 1.0f);
    HIP_TRY(hipGetLastError());
    int rc = bf_launch_mesh(m, &scratch, n, d_state.p, d_vraw.p, nullptr, d_xp.p, d_j.p, d_jo.p, 0, nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (vertices) HIP_TRY(hipMemcpy(vertices, d_vraw.p, (size_t)n * nv * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (joints) HIP_TRY(hipMemcpy(joints, d_j.p, d_j.n * sizeof(float), hipMemcpyDeviceToHost));
    if (joints_ori) HIP_TRY(hipMemcpy(joints_ori, d_jo.p, d_jo.n * sizeof(float), hipMemcpyDeviceToHost));
    return BF_OK;
}

// A small fetch as a kernel: device arena -> pinned host mirror with plain stores over PCIe (posted writes, visible to the
// host when the stream drains); a copy node costs ~20 us of fixed overhead for the same 90 KB.
__global__ void __launch_bounds__(256) bf_publish_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// result arena k becomes the one the DevBuf views and the pinned mirrors point at
void bf_use_arena(bf_batch *b, int k) {
    float *d = k ? b->res_b.p : b->res.p, *h = k ? b->h_res_b : b->h_res;
    b->params.slice(d + b->res_off[0], b->res_cnt[0]); b->terms.slice(d + b->res_off[1], b->res_cnt[1]);
    b->state.slice(d + b->res_off[2], b->res_cnt[2]); b->joints.slice(d + b->res_off[3], b->res_cnt[3]);
    b->vout.slice(d + b->res_off[4], b->res_cnt[4]);
    b->h_params = h + b->res_off[0]; b->h_terms = h + b->res_off[1]; b->h_state = h + b->res_off[2];
    b->h_joints = h + b->res_off[3]; b->h_vout = h + b->res_off[4];
    b->cur = k;
}

// input arena k becomes the one `keypoints`, `params0`, `ndiv` point at (host = true: at its pinned staging buffer itself)
void bf_use_inputs(bf_batch *b, int k, bool host) {
    float *base = host ? b->h_in[k] : b->in_dev[k].p;
    const bf_model *m = b->m;
    b->keypoints.slice(base + b->in_off[0], (size_t)b->F * b->V * m->nl_loss * 3);
    b->params0.slice(base + b->in_off[1], (size_t)b->F * m->np);
    b->ndiv.slice((int *)(base + b->in_off[2]), (size_t)b->F);
    b->in_cur = k;
    b->in_host = host;
}

// a synchronous setter's write into the current input arena (the stream is idle)
static hipError_t write_input(bf_batch *b, void *dst, const void *src, size_t bytes) {
    if (b->in_host) { std::memcpy(dst, src, bytes); return hipSuccess; }
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}

// the deferred part of a frame-after-frame call (fit_impl, `tail_aside`): mesh + joints + hand-over of result arena tail_k on the
// second stream, behind the fit that filled it.  Every entry point that looks at results, events of the second stream or starts
// another fit comes through here first (bf_sync_all, bf_fit, bf_batch_get_previous, bf_batch_stage_inputs, bf_batch_destroy).
static int flush_tail_body(bf_batch *b, int k) {
    bf_model *m = b->m;
    HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->ev_done[k], 0));
    float *d = k ? b->res_b.p : b->res.p;            // (arena k's slices by address: the views may already have moved on)
    int rc = bf_launch_mesh(m, &b->scratch, b->F, d + b->res_off[2], b->vraw.p, d + b->res_off[4], b->xpart.p, d + b->res_off[3], nullptr,
                            b->copy_stream, nullptr, nullptr);
    if (rc) return rc;
    if (b->tail_big) {
        HIP_TRY(hipMemcpyAsync(k ? b->h_res_b : b->h_res, d, b->res.n * sizeof(float), hipMemcpyDeviceToHost, b->copy_stream));
    } else {
        const size_t n4 = b->res.n / 4;
        hipLaunchKernelGGL(bf_publish_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 64)), dim3(256), 0, b->copy_stream,
                           (const float4 *)d, (float4 *)(k ? b->h_res_b : b->h_res), n4);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(b->ev_copied[k], b->copy_stream));
    return BF_OK;
}
int bf_flush_tail(bf_batch *b) {
    const int k = b->tail_k;
    if (k < 0) return BF_OK;
    b->tail_k = -1;
    const int rc = flush_tail_body(b, k);
    if (rc) {
        // the mesh / copy / event of arena k did not all go out: ev_copied[k] still carries its previous (completed) record, so a
        // wait on it would pass and hand out the pinned buffer's OLD contents.  Nothing of this arena may be read any more:
        // bf_batch_get_previous (arena_fetched) and bf_batch_get_result (have_result) then fail instead.
        b->arena_fetched[k] = false;
        b->arena_has_v[k] = false;
        b->copy_pending[k] = false;
        if (k == b->cur) b->have_result = false;
        return rc;
    }
    b->tail_seq = b->arena_seq[k];
    return BF_OK;
}

int bf_sync_all(bf_batch *b) {
    { int rt_ = bf_flush_tail(b); if (rt_) return rt_; }
    if (b->copy_stream) HIP_TRY(hipStreamSynchronize(b->copy_stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->copy_pending[0] = b->copy_pending[1] = false;
    if (b->h_door_err && *b->h_door_err) {
        const int who = *b->h_door_err;
        *b->h_door_err = 0;
        b->door_usable = false;          // this batch keeps one fit launch per iteration from now on (same results): a second call does not run into the same wait
        int d[BF_DOOR_STATE + 1] = {};
        (void)hipMemcpy(d, b->door.p, sizeof d, hipMemcpyDeviceToHost);
        return fail(BF_ERR_HIP, "dense schedule (bells: ext " + std::to_string(d[BF_DOOR_EXT]) + ", states " + std::to_string(d[BF_DOOR_STATE]) +
                                    ", tickets " + std::to_string(d[BF_DOOR_TICKET]) + ", waiter " + std::to_string(who) + "): a doorbell wait between the persistent fit launch and the dense kernels ran into its time limit");
    }
    return BF_OK;
}

int bf_guard_arena(bf_batch *b) {
    { int rt_ = bf_flush_tail(b); if (rt_) return rt_; }
    if (b->copy_pending[b->cur]) {
        HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_copied[b->cur], 0));
        b->copy_pending[b->cur] = false;
    }
    return BF_OK;
}

int bf_batch_create(bf_model *m, int n_frames, int n_views, bf_batch **out) {
    if (!m || !out || n_frames <= 0 || n_views <= 0) return fail(BF_ERR_INVALID, "bf_batch_create: bad argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(m->device));
    size_t smem = bf_fit_smem_bytes(m->nj, m->nb, m->npf, m->ns, m->nl, m->np, n_views);
    if (smem > 160 * 1024) return fail(BF_ERR_UNSUPPORTED, "bf_batch_create: too many views for one workgroup's LDS");
    auto *b = new bf_batch();
    b->m = m; b->F = n_frames; b->V = n_views;
    const size_t F = n_frames, np = m->np;
    bool ok = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess;
    b->ring.assign((size_t)bf_batch::kRing * 4, nullptr);
    b->ev = b->ring.data();          // (the events of a slot are created when the slot is first used: 4096 creations cost 4 ms)
    {
        auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
        const size_t n_kp = F * n_views * m->nl_loss * 3;
        b->in_off[0] = 0; b->in_off[1] = up(n_kp); b->in_off[2] = b->in_off[1] + up(F * np);
        b->in_total = b->in_off[2] + up(F);
        if (const char *e = getenv("BF_STAGE_MODE")) b->stage_mode = !strcmp(e, "memcpy") ? 1 : (!strcmp(e, "zerocopy") ? 2 : 0);
        for (int k = 0; k < 2; ++k) {
            ok = ok && b->in_dev[k].alloc(b->in_total) == hipSuccess && bf_memset_sync(b->in_dev[k].p, 0, b->in_total * sizeof(float)) == hipSuccess;
            ok = ok && hipHostMalloc((void **)&b->h_in[k], b->in_total * sizeof(float)) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&b->ev_in[k], hipEventDisableTiming) == hipSuccess;
            if (ok) std::memset(b->h_in[k], 0, b->in_total * sizeof(float));
        }
        if (ok) bf_use_inputs(b, 0, false);
    }
    {
        auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };          // 256-byte slices
        const size_t n_par = F * np, n_terms = F * 4, n_state = F * bf_state_stride(m->nj, m->npf, m->nb),
                     n_joints = F * m->n_joint_map * 3, n_v = F * m->nv * 3;
        const size_t o_terms = up(n_par), o_state = o_terms + up(n_terms), o_joints = o_state + up(n_state),
                     o_v = o_joints + up(n_joints), total = o_v + up(n_v);
        ok = ok && b->res.alloc(total) == hipSuccess && b->res_b.alloc(total) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&b->h_res, total * sizeof(float)) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&b->h_res_b, total * sizeof(float)) == hipSuccess;
        {
            // the second stream on the LOWEST priority: the runtime keeps a pool of hardware queues per priority, so it never shares a
            // queue with a batch's main stream (streams that share one run in order, and the hand-over of a call would no longer run
            // under the next call's fit kernel - observed with two live batches)
            int least = 0, greatest = 0;
            ok = ok && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
                 hipStreamCreateWithPriority(&b->copy_stream, hipStreamNonBlocking, least) == hipSuccess;
        }
        for (int k = 0; k < 2; ++k)
            ok = ok && hipEventCreateWithFlags(&b->ev_done[k], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&b->ev_copied[k], hipEventDisableTiming) == hipSuccess;
        if (ok) {
            ok = bf_memset_sync(b->res.p, 0, total * sizeof(float)) == hipSuccess && bf_memset_sync(b->res_b.p, 0, total * sizeof(float)) == hipSuccess;
            const size_t offs[5] = {0, o_terms, o_state, o_joints, o_v}, cnts[5] = {n_par, n_terms, n_state, n_joints, n_v};
            for (int i = 0; i < 5; ++i) { b->res_off[i] = offs[i]; b->res_cnt[i] = cnts[i]; }
            b->res_small = o_v;
            bf_use_arena(b, 0);
        }
    }
    ok = ok && b->proj.alloc(F * n_views * 12) == hipSuccess;
    if (ok) {
        const std::vector<int> nd(F, n_views);
        for (int k = 0; k < 2; ++k)
            ok = ok && hipMemcpy(b->in_dev[k].p + b->in_off[2], nd.data(), F * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    }
    ok = ok && b->adam_m.alloc(F * np) == hipSuccess;
    ok = ok && b->adam_v.alloc(F * np) == hipSuccess && b->grads.alloc(F * np) == hipSuccess;
    ok = ok && b->vraw.alloc(F * m->nv * 3) == hipSuccess;
    ok = ok && b->xpart.alloc(F * m->mesh.n_tiles * std::max(m->n_extra, 1) * 3) == hipSuccess;
    ok = ok && b->debug.alloc(8192) == hipSuccess;
    if (ok) {
        ok = bf_memset_sync(b->adam_m.p, 0, F * np * sizeof(float)) == hipSuccess &&
             bf_memset_sync(b->adam_v.p, 0, F * np * sizeof(float)) == hipSuccess &&
             bf_memset_sync(b->params.p, 0, F * np * sizeof(float)) == hipSuccess &&
             bf_memset_sync(b->debug.p, 0, 8192 * sizeof(float)) == hipSuccess;
    }
    if (!ok) { bf_batch_destroy(b); return fail(BF_ERR_HIP, "bf_batch_create: device allocation failed"); }
    b->fit_smem = smem;
    *out = b;
    return BF_OK;
}

void bf_batch_destroy(bf_batch *b) {
    if (!b) return;
    b->tail_k = -1;                       // (a tail never enqueued: nobody will read that result)
    if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    { std::lock_guard<std::mutex> lk(bf_scan_links()); bf_batch_unlink_scans(b); }      // (its scans outlive it: they forget this batch)
    if (b->graph_exec) (void)hipGraphExecDestroy(b->graph_exec);
    for (auto &e : b->ring) if (e) (void)hipEventDestroy(e);
    if (b->h_res) (void)hipHostFree(b->h_res);
    if (b->h_res_b) (void)hipHostFree(b->h_res_b);
    if (b->h_pc_weight) (void)hipHostFree(b->h_pc_weight);
    if (b->h_masks) (void)hipHostFree(b->h_masks);
    for (float *q : b->mk_retired) (void)hipFree(q);
    if (b->h_ccount) (void)hipHostFree(b->h_ccount);
    if (b->ev_masks) (void)hipEventDestroy(b->ev_masks);
    if (b->ev_masks_used) (void)hipEventDestroy(b->ev_masks_used);
    if (b->mk_stage.ev) (void)hipEventDestroy(b->mk_stage.ev);
    if (b->mk_stage.ev_used) (void)hipEventDestroy(b->mk_stage.ev_used);
    if (b->mk_stage.h_masks) (void)hipHostFree(b->mk_stage.h_masks);
    if (b->mk_stage.h_ccount) (void)hipHostFree(b->mk_stage.h_ccount);
    for (auto &e : b->ev_dense) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) {
        if (b->h_in[k]) (void)hipHostFree(b->h_in[k]);
        if (b->ev_in[k]) (void)hipEventDestroy(b->ev_in[k]);
        if (b->graph_pipe[k]) (void)hipGraphExecDestroy(b->graph_pipe[k]);
        if (b->ev_done[k]) (void)hipEventDestroy(b->ev_done[k]);
        if (b->ev_copied[k]) (void)hipEventDestroy(b->ev_copied[k]);
    }
    if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
    if (b->fit_stream) { (void)hipStreamSynchronize(b->fit_stream); (void)hipStreamDestroy(b->fit_stream); }
    for (auto &e : b->ev_door) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ev_aux) if (e) (void)hipEventDestroy(e);
    if (b->h_door_err) (void)hipHostFree(b->h_door_err);
    if (b->h_resident) (void)hipHostFree(b->h_resident);
    delete b;
}

// general 4x4 inverse, Gauss-Jordan with partial pivoting, in double
static bool invert4(const float *src, double *inv) {
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { a[r][c] = src[r * 4 + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c) for (int k = 0; k < 8; ++k) std::swap(a[piv][k], a[c][k]);
        double d = a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            double f = a[r][c];
            if (f != 0.0) for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) inv[r * 4 + c] = a[r][4 + c];
    return true;
}

int bf_batch_set_cameras(bf_batch *b, const float *c2w, const float *K) {
    if (!b || !c2w || !K) return fail(BF_ERR_INVALID, "bf_batch_set_cameras: null argument");
    HIP_TRY(hipSetDevice(b->m->device));
    const size_t n = (size_t)b->F * b->V;
    std::vector<float> proj(n * 12);
    for (size_t i = 0; i < n; ++i) {
        double w2c[16];
        if (!invert4(c2w + i * 16, w2c)) return fail(BF_ERR_INVALID, "bf_batch_set_cameras: singular c2w");
        const float *k = K + i * 9;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c)
                proj[i * 12 + r * 4 + c] =
                    (float)((double)k[r * 3] * w2c[c] + (double)k[r * 3 + 1] * w2c[4 + c] + (double)k[r * 3 + 2] * w2c[8 + c]);
    }
    // bf_fit is asynchronous on the batch's own (non-blocking) stream: a fit still in flight reads these inputs
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    HIP_TRY(hipMemcpy(b->proj.p, proj.data(), proj.size() * sizeof(float), hipMemcpyHostToDevice));
    return BF_OK;
}

int bf_batch_set_keypoints(bf_batch *b, const float *keypoints, const int32_t *n_use_frames) {
    if (!b || !keypoints) return fail(BF_ERR_INVALID, "bf_batch_set_keypoints: null argument");
    HIP_TRY(hipSetDevice(b->m->device));
    std::vector<int> nd(b->F, b->V);
    if (n_use_frames)
        for (int f = 0; f < b->F; ++f) {
            if (n_use_frames[f] <= 0) return fail(BF_ERR_INVALID, "bf_batch_set_keypoints: n_use_frames must be positive");
            nd[f] = n_use_frames[f];
        }
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }      // (a fit still in flight reads the old keypoints)
    HIP_TRY(write_input(b, b->keypoints.p, keypoints, b->keypoints.n * sizeof(float)));
    HIP_TRY(write_input(b, b->ndiv.p, nd.data(), nd.size() * sizeof(int)));
    return BF_OK;
}

static int reset_adam(bf_batch *b, const float *params_host) {
    HIP_TRY(write_input(b, b->params0.p, params_host, b->params.n * sizeof(float)));
    HIP_TRY(bf_memset_sync(b->adam_m.p, 0, b->adam_m.n * sizeof(float)));
    HIP_TRY(bf_memset_sync(b->adam_v.p, 0, b->adam_v.n * sizeof(float)));
    b->steps_done = 0;
    b->have_result = false;
    b->fetched = false;
    b->staged = false;             // (a fresh start from these parameters: what BF_FIT_RESET would do for staged inputs)
    return BF_OK;
}

int bf_batch_reset(bf_batch *b) {
    if (!b) return fail(BF_ERR_INVALID, "bf_batch_reset: null batch");
    HIP_TRY(hipSetDevice(b->m->device));
    { int rg_ = bf_guard_arena(b); if (rg_) return rg_; }
    // (hipMemcpyDefault: with BF_STAGE_MODE=zerocopy params0 is a view into the pinned staging buffer, not device memory)
    HIP_TRY(hipMemcpyAsync(b->params.p, b->params0.p, b->params.n * sizeof(float), hipMemcpyDefault, b->stream));
    HIP_TRY(hipMemsetAsync(b->adam_m.p, 0, b->adam_m.n * sizeof(float), b->stream));
    HIP_TRY(hipMemsetAsync(b->adam_v.p, 0, b->adam_v.n * sizeof(float), b->stream));
    b->steps_done = 0;
    b->have_result = false;
    b->fetched = false;
    b->staged = false;             // (re-armed from the staged parameters: a following bf_fit needs no BF_FIT_RESET)
    return BF_OK;
}

// net_output of smplify.py:103 -> the packed optimiser vector: transl = 0, scale = 1 (:126-128), body pose, betas, root orientation
static void pack_init(const bf_batch *b, const float *init_betas, const float *init_pose, float *dst) {
    const bf_model *m = b->m;
    const int np = m->np, nb = m->nb;
    const int pose_stride = 72;                  // net_output poses are [F,72] for both model kinds (smplify.py:108-112)
    std::memset(dst, 0, (size_t)b->F * np * sizeof(float));
    for (int f = 0; f < b->F; ++f) {
        float *q = dst + (size_t)f * np;
        q[3] = 1.0f;                                                             // body_scale = 1, transl = 0
        std::memcpy(q + m->fit.off_pose, init_pose + (size_t)f * pose_stride + 3, sizeof(float) * m->fit.nbp);
        std::memcpy(q + m->fit.off_beta, init_betas + (size_t)f * nb, sizeof(float) * nb);
        std::memcpy(q + m->fit.off_orient, init_pose + (size_t)f * pose_stride, sizeof(float) * 3);
    }
}

/* The next frame's keypoints and initial estimate, WITHOUT draining the work in flight (apps/genebody_fitting.py:183-192 hands
 * SMPLify a new frame's detections and HMR estimate every call; loss.py:160 re-uploads the keypoints every iteration).  The
 * inputs are packed into the pinned staging buffer the fit in flight does not use and their transfer into the other device
 * arena is queued on the batch stream, behind that fit - or, frame after frame, on the second stream under it (below); the next
 * bf_fit - which must carry BF_FIT_RESET - reads them. */
int bf_batch_stage_inputs(bf_batch *b, const float *keypoints, const int32_t *n_use_frames, const float *init_betas, const float *init_pose) {
    if (!b || !keypoints || !init_betas || !init_pose) return fail(BF_ERR_INVALID, "bf_batch_stage_inputs: null argument");
    if (n_use_frames)
        for (int f = 0; f < b->F; ++f)
            if (n_use_frames[f] <= 0) return fail(BF_ERR_INVALID, "bf_batch_stage_inputs: n_use_frames must be positive");
    HIP_TRY(hipSetDevice(b->m->device));
    const int k = b->in_cur ^ 1;
    if (b->in_pending[k]) {                 // (the transfer of two stagings ago: behind a fit that has long finished)
        HIP_TRY(hipEventSynchronize(b->ev_in[k]));
        b->in_pending[k] = false;
    }
    float *h = b->h_in[k];
    std::memcpy(h + b->in_off[0], keypoints, (size_t)b->F * b->V * b->m->nl_loss * 3 * sizeof(float));
    pack_init(b, init_betas, init_pose, h + b->in_off[1]);
    int *nd = (int *)(h + b->in_off[2]);
    for (int f = 0; f < b->F; ++f) nd[f] = n_use_frames ? n_use_frames[f] : b->V;
    if (b->stage_mode == 2) {
        // zero-copy: the fit kernel's prologue reads the pinned buffer itself; bf_fit records ev_in[k] behind the fit that read it
        bf_use_inputs(b, k, true);
    } else {
        const size_t n4 = b->in_total / 4;
        b->in_aside[k] = false;
        if (b->stage_mode == 0 && b->tail_k >= 0 && b->copy_stream && b->in_reader[k] <= b->tail_seq) {
            // Frame after frame (the fit in flight has its tail still to be enqueued): the transfer goes on the second stream, AHEAD of
            // that tail, and runs under the fit in flight; the batch stream then holds that fit and the next one back to back (the
            // transfer in between cost its 4.6 us and a second dispatch gap: ~13 us of a 430 us step).  Arena k was last read by fit
            // in_reader[k], whose tail - it starts by waiting for that fit - is already on the second stream: the transfer is ordered
            // behind it.  The fit that reads arena k waits for ev_in[k] (bf_fit: on the host while the fit in flight runs).
            hipLaunchKernelGGL(bf_publish_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 64)), dim3(256), 0, b->copy_stream,
                               (const float4 *)h, (float4 *)b->in_dev[k].p, n4);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(b->ev_in[k], b->copy_stream));
            b->in_pending[k] = true;
            b->in_aside[k] = true;
            bf_use_inputs(b, k, false);
            b->staged = true;
            return bf_flush_tail(b);
        }
        { int rt_ = bf_flush_tail(b); if (rt_) return rt_; }
        if (b->stage_mode == 1) {
            HIP_TRY(hipMemcpyAsync(b->in_dev[k].p, h, b->in_total * sizeof(float), hipMemcpyHostToDevice, b->stream));
        } else {
            // (a kernel that reads the pinned buffer over PCIe with coalesced 16-byte loads: one more dispatch on the queue the fit
            //  kernel is on, no engine hand-over)
            hipLaunchKernelGGL(bf_publish_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 64)), dim3(256), 0, b->stream,
                               (const float4 *)h, (float4 *)b->in_dev[k].p, n4);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(b->ev_in[k], b->stream));
        b->in_pending[k] = true;
        bf_use_inputs(b, k, false);
    }
    b->staged = true;
    return BF_OK;
}

int bf_batch_set_init(bf_batch *b, const float *init_betas, const float *init_pose) {
    if (!b || !init_betas || !init_pose) return fail(BF_ERR_INVALID, "bf_batch_set_init: null argument");
    HIP_TRY(hipSetDevice(b->m->device));
    const bf_model *m = b->m;
    std::vector<float> p((size_t)b->F * m->np, 0.f);
    pack_init(b, init_betas, init_pose, p.data());
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    HIP_TRY(hipMemcpy(b->params.p, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
    return reset_adam(b, p.data());
}

int bf_batch_set_params(bf_batch *b, const float *params) {
    if (!b || !params) return fail(BF_ERR_INVALID, "bf_batch_set_params: null argument");
    HIP_TRY(hipSetDevice(b->m->device));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    HIP_TRY(hipMemcpy(b->params.p, params, b->params.n * sizeof(float), hipMemcpyHostToDevice));
    return reset_adam(b, params);
}

int bf_batch_get_params(bf_batch *b, float *params) {
    if (!b || !params) return fail(BF_ERR_INVALID, "bf_batch_get_params: null argument");
    HIP_TRY(hipSetDevice(b->m->device));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    if (b->fetched) std::memcpy(params, b->h_params, b->params.n * sizeof(float));
    else HIP_TRY(hipMemcpy(params, b->params.p, b->params.n * sizeof(float), hipMemcpyDeviceToHost));
    return BF_OK;
}

HyperDev bf_to_dev(const bf_hyper &h) {
    HyperDev d;
    d.sigma2 = h.sigma * h.sigma;
    d.w_pose = h.pose_prior_weight * h.pose_prior_weight;
    d.w_angle = h.angle_prior_weight * h.angle_prior_weight;
    d.w_shape = h.shape_prior_weight * h.shape_prior_weight;
    d.cscale = h.constant_scale;
    d.coeff = h.imsize / 1024.0f;
    d.beta1 = h.adam_beta1; d.beta2 = h.adam_beta2; d.eps = h.adam_eps;
    return d;
}

// torch.optim.Adam evaluates the bias corrections in python floats (double): SURVEY.md 10C
static int ensure_adam_tab(bf_batch *b, const bf_hyper &h, int upto) {
    bool same = b->adam_cap >= upto && b->adam_hyper.lr == h.lr && b->adam_hyper.lr_transl_scale == h.lr_transl_scale &&
                b->adam_hyper.adam_beta1 == h.adam_beta1 && b->adam_hyper.adam_beta2 == h.adam_beta2;
    if (same) return BF_OK;
    int cap = std::max(upto, 1024);
    std::vector<float> tab((size_t)(cap + 1) * 3);            // (+1: a kernel may look one step ahead)
    const double b1 = (double)h.adam_beta1, b2 = (double)h.adam_beta2;
    for (int t = 1; t <= cap + 1; ++t) {
        double bc1 = 1.0 - std::pow(b1, t), bc2 = 1.0 - std::pow(b2, t);
        tab[(size_t)(t - 1) * 3 + 0] = (float)((double)h.lr_transl_scale / bc1);
        tab[(size_t)(t - 1) * 3 + 1] = (float)((double)h.lr / bc1);
        tab[(size_t)(t - 1) * 3 + 2] = (float)std::sqrt(bc2);
    }
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    if (b->adam_tab.p) { (void)hipFree(b->adam_tab.p); b->adam_tab.p = nullptr; }
    HIP_TRY(b->adam_tab.upload(tab));
    b->adam_cap = cap;
    b->adam_hyper = h;
    return BF_OK;
}

FrameIO bf_frame_io(bf_batch *b, bool want_grads) {
    FrameIO io;
    io.n_frames = b->F; io.n_views = b->V;
    io.proj = b->proj.p; io.keypoints = b->keypoints.p; io.ndiv = b->ndiv.p;
    io.params = b->params.p; io.params0 = nullptr; io.adam_m = b->adam_m.p; io.adam_v = b->adam_v.p;
    io.grads = want_grads ? b->grads.p : nullptr;
    io.terms = b->terms.p; io.state = b->state.p;
    io.debug = b->debug.p;
    io.cscale = b->cscale.p;          // null unless scans are attached
    io.ext = nullptr;
    io.image_out = nullptr;
    io.door = nullptr;
    io.door_resident = nullptr;
    return io;
}

// stream work of one sparse-schedule call: [re-arm] -> persistent fit -> [mesh + joints] -> [fetch]; `ev` = event
// records between the parts (null inside a graph capture)
static int enqueue_plain(bf_batch *b, int n_iters, const HyperDev &hd, const FrameIO &io, bool reset, bool want_v, bool fetch,
                         int adam_t0, hipEvent_t *ev) {
    bf_model *m = b->m;
    const size_t fb = sizeof(float);
    FrameIO io2 = io;
    if (reset) io2.params0 = b->params0.p;      // re-arm inside the fit kernel: no copy / memset commands
    HIP_TRY(bf_fit_launch(&m->fit, &io2, &hd, n_iters, 0, b->adam_tab.p, adam_t0, b->fit_smem, b->stream, nullptr));
    if (ev) HIP_TRY(hipEventRecord(ev[1], b->stream));
    if (want_v) {
        int rc = bf_launch_mesh(m, &b->scratch, b->F, b->state.p, b->vraw.p, b->vout.p, b->xpart.p, b->joints.p, nullptr, b->stream,
                                ev ? ev[2] : nullptr, nullptr);
        if (rc) return rc;
    } else if (ev) HIP_TRY(hipEventRecord(ev[2], b->stream));
    if (fetch) {
        // one copy of the result arena: [params | terms | state | joints] and, when they were built, the vertices
        const size_t nfl = want_v ? b->res.n : b->res_small;          // (slices are 256-byte multiples: float4 clean)
        if (nfl * fb < (size_t)512 * 1024) {
            const size_t n4 = nfl / 4;
            hipLaunchKernelGGL(bf_publish_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 64)), dim3(256), 0, b->stream,
                               (const float4 *)(b->cur ? b->res_b.p : b->res.p), (float4 *)(b->cur ? b->h_res_b : b->h_res), n4);
            HIP_TRY(hipGetLastError());
        } else
        HIP_TRY(hipMemcpyAsync(b->cur ? b->h_res_b : b->h_res, b->cur ? b->res_b.p : b->res.p, nfl * fb, hipMemcpyDeviceToHost, b->stream));
    }
    return BF_OK;
}

static int fit_impl(bf_batch *b, int n_iters, const bf_hyper *hyper, uint32_t flags);

int bf_fit(bf_batch *b, int n_iters, const bf_hyper *hyper, uint32_t flags) {
    if (!b || n_iters <= 0) return fail(BF_ERR_INVALID, "bf_fit: bad argument");
    if (b->scans_lost)
        return fail(BF_ERR_INVALID, "bf_fit: a scan this batch held was destroyed (bf_scan_destroy) - call bf_batch_set_scans again (NULL: go on without scans)");
    if (b->staged && !(flags & BF_FIT_RESET))
        return fail(BF_ERR_INVALID, "bf_fit: inputs were staged with bf_batch_stage_inputs - the fit of a new frame starts from its initial estimate (BF_FIT_RESET)");
    bf_masks_commit(b);                       // (silhouettes staged with bf_batch_stage_masks become this fit's)
    int rc = bf_flush_tail(b);
    if (rc) return rc;
    if (b->in_aside[b->in_cur]) {
        // inputs staged aside: their transfer was queued on the second stream a few microseconds ago and runs under the fit in flight.
        // Waiting for it HERE, on the host, keeps a wait packet out of the batch stream (the fit in flight has hundreds of
        // microseconds to go); only if it does not show up in time does the stream wait for it.
        hipError_t q = hipErrorNotReady;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(500);
        while ((q = hipEventQuery(b->ev_in[b->in_cur])) == hipErrorNotReady && std::chrono::steady_clock::now() < t_end) { }
        if (q == hipErrorNotReady) HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_in[b->in_cur], 0));
        else HIP_TRY(q);
        b->in_aside[b->in_cur] = false;
    }
    rc = fit_impl(b, n_iters, hyper, flags);
    if (rc) return rc;
    b->in_reader[b->in_cur] = b->fit_seq;       // (this fit's number, given to its arena below)
    if (b->has_masks) {                       // (the arena these masks live in may be overwritten once this fit is done)
        if (!b->ev_masks_used) HIP_TRY(hipEventCreateWithFlags(&b->ev_masks_used, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(b->ev_masks_used, b->stream));
    }
    b->staged = false;
    if (b->in_host) { HIP_TRY(hipEventRecord(b->ev_in[b->in_cur], b->stream)); b->in_pending[b->in_cur] = true; }   // (zero-copy: this fit read the pinned buffer)
    b->arena_seq[b->cur] = b->fit_seq++;
    b->arena_fetched[b->cur] = b->fetched;
    b->arena_has_v[b->cur] = b->have_result;
    return BF_OK;
}

static int fit_impl(bf_batch *b, int n_iters, const bf_hyper *hyper, uint32_t flags) {
    bf_model *m = b->m;
    HIP_TRY(hipSetDevice(m->device));
    bf_hyper h;
    if (hyper) h = *hyper; else bf_hyper_default(&h);
    const bool reset = flags & BF_FIT_RESET;
    if (reset) { b->steps_done = 0; b->have_result = false; }
    int rc = ensure_adam_tab(b, h, b->steps_done + n_iters);
    if (rc) return rc;
    HyperDev hd = bf_to_dev(h);
    FrameIO io = bf_frame_io(b, false);
    rc = bf_ensure_fit_image(b, io, hd);          // (once per model: the fit kernel's batched prologue, before any graph captures a launch)
    if (rc) return rc;
    const bool dense_losses = !b->scans.empty() || b->has_masks || m->kp_dense;
    if (dense_losses) flags &= ~BF_FIT_DENSE;
    const bool dense = flags & BF_FIT_DENSE, want_v = !(flags & BF_FIT_NO_VERTICES), fetch = flags & BF_FIT_FETCH;
    b->ev = b->ring.data() + (size_t)(b->ring_n % bf_batch::kRing) * 4;
    for (int k = 0; k < 4; ++k)
        if (!b->ev[k]) HIP_TRY(hipEventCreate(&b->ev[k]));
    const size_t fb = sizeof(float);
    // (a small fetch is cheaper as a copy node inside the graph than as a second stream with two event hand-offs)
    const bool big_fetch = (want_v ? b->res.n : b->res_small) * sizeof(float) >= (size_t)512 * 1024;
    const bool pipelined = (flags & BF_FIT_GRAPH) && reset && !dense_losses && !dense && fetch && big_fetch;
    // Calls issued back to back without timing records (frame after frame, as the reference's loop does): the fit kernel is a
    // latency chain of one workgroup per frame, so the mesh / joints / result hand-over of a call runs on the second stream UNDER the
    // next call's fit kernel; the two result arenas alternate as in the pipelined fetch.
    const bool tail_aside = (flags & BF_FIT_NOTIME) && !(flags & BF_FIT_GRAPH) && reset && !dense_losses && !dense && fetch && want_v;
    if (!pipelined && !tail_aside) {                // (a pipelined fetch may still be reading the arena this call writes)
        rc = bf_guard_arena(b);
        if (rc) return rc;
    }
    if ((flags & BF_FIT_GRAPH) && reset && !dense_losses && !dense) {
        // the whole call as one hipGraph launch: the host issues a single command per fit
        // the MFMA batch path may grow its scratch buffer: make sure that happened before capturing
        if (want_v && b->F >= BF_MFMA_MIN_FRAMES && b->scratch.pose_off.n < (size_t)b->F * m->nv * 3) {
            { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
            rc = bf_launch_mesh(m, &b->scratch, b->F, b->state.p, b->vraw.p, b->vout.p, b->xpart.p, b->joints.p, nullptr, b->stream, nullptr, nullptr);
            if (rc) return rc;
            { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
        }
        bf_graph_key key{n_iters, flags, (pipelined ? (b->cur ^ 1) : b->cur) | (b->in_cur << 4) | ((int)b->in_host << 5), h};   // (the captured nodes hold the arenas' addresses)
        if (pipelined) {
            // pipelined fetch: this fit writes the result arena the previous one did not use; its device-to-host copy
            // runs on the copy stream, under the kernels of whatever is enqueued next
            const int k = b->cur ^ 1;
            if (b->copy_pending[k]) { HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_copied[k], 0)); b->copy_pending[k] = false; }
            bf_use_arena(b, k);
            io = bf_frame_io(b, false);
            if (!b->graph_pipe[k] || std::memcmp(&key, &b->graph_pipe_key[k], sizeof key) != 0) {
                if (b->graph_pipe[k]) { (void)hipGraphExecDestroy(b->graph_pipe[k]); b->graph_pipe[k] = nullptr; }
                hipGraph_t graph = nullptr;
                HIP_TRY(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
                rc = enqueue_plain(b, n_iters, hd, io, true, want_v, false, 0, nullptr);
                hipError_t e = hipStreamEndCapture(b->stream, &graph);
                if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                HIP_TRY(e);
                HIP_TRY(hipGraphInstantiate(&b->graph_pipe[k], graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                b->graph_pipe_key[k] = key;
            }
            HIP_TRY(hipEventRecord(b->ev[0], b->stream));
            HIP_TRY(hipGraphLaunch(b->graph_pipe[k], b->stream));
            HIP_TRY(hipEventRecord(b->ev[1], b->stream));       // (no events inside a graph: the whole call is charged to ms[0])
            HIP_TRY(hipEventRecord(b->ev[2], b->stream));
            HIP_TRY(hipEventRecord(b->ev[3], b->stream));
            HIP_TRY(hipEventRecord(b->ev_done[k], b->stream));
            HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->ev_done[k], 0));
            HIP_TRY(hipMemcpyAsync(k ? b->h_res_b : b->h_res, k ? b->res_b.p : b->res.p, (want_v ? b->res.n : b->res_small) * fb,
                                   hipMemcpyDeviceToHost, b->copy_stream));
            HIP_TRY(hipEventRecord(b->ev_copied[k], b->copy_stream));
            b->copy_pending[k] = true;
        } else {
        if (!b->graph_exec || std::memcmp(&key, &b->graph_key, sizeof key) != 0) {
            if (b->graph_exec) { (void)hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            HIP_TRY(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue_plain(b, n_iters, hd, io, true, want_v, fetch, 0, nullptr);
            hipError_t e = hipStreamEndCapture(b->stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            HIP_TRY(e);
            HIP_TRY(hipGraphInstantiate(&b->graph_exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            b->graph_key = key;
        }
        HIP_TRY(hipEventRecord(b->ev[0], b->stream));
        HIP_TRY(hipGraphLaunch(b->graph_exec, b->stream));
        HIP_TRY(hipEventRecord(b->ev[1], b->stream));       // (no events inside a graph: the whole call is charged to ms[0])
        HIP_TRY(hipEventRecord(b->ev[2], b->stream));
        HIP_TRY(hipEventRecord(b->ev[3], b->stream));
        }
        b->fetched = fetch;
        b->ring_n += 1;
        b->steps_done = n_iters;
        b->timed = true;
        b->have_result = want_v;
        return BF_OK;
    }
    const bool notime = (flags & BF_FIT_NOTIME) && !dense_losses && !dense;
    if (tail_aside) {
        const int k = b->cur ^ 1;
        if (b->copy_pending[k]) {      // (arena k's hand-over of two calls ago: normally long done - then no wait packet goes into the stream)
            if (hipEventQuery(b->ev_copied[k]) != hipSuccess) HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_copied[k], 0));
            b->copy_pending[k] = false;
        }
        bf_use_arena(b, k);
        FrameIO io2 = bf_frame_io(b, false);
        io2.params0 = b->params0.p;                  // re-arm inside the fit kernel
        static const bool own_signal = [] { const char *e = getenv("BF_FIT_DONE_EVENT"); return !(e && e[0] == '0'); }();
        static const int n_cus = [] { int v = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
        const bool crowded = b->F >= n_cus;
        const bool signal_here = own_signal && !crowded;       // (ev_done[k] completes with the fit's own dispatch: no marker packet between this fit and the next)
        HIP_TRY(bf_fit_launch(&m->fit, &io2, &hd, n_iters, 0, b->adam_tab.p, b->steps_done, b->fit_smem, b->stream, signal_here ? b->ev_done[k] : nullptr));
        // A batch that fills the machine (a frame's workgroup per CU, one workgroup fits per CU): under the NEXT fit the mesh kernels
        // would only get the CUs that fit's workgroups leave as they finish - measured 379 us for a 33 us GEMM at 256 frames.  The mesh
        // then goes on the fit's own stream, ahead of the next fit (0.462 + 0.074 ms instead of 0.601); only the copy stays aside.
        if (crowded) {
            rc = bf_launch_mesh(m, &b->scratch, b->F, b->state.p, b->vraw.p, b->vout.p, b->xpart.p, b->joints.p, nullptr, b->stream, nullptr, nullptr);
            if (rc) return rc;
        }
        if (!signal_here) HIP_TRY(hipEventRecord(b->ev_done[k], b->stream));
        if (!crowded) {
            // the rest - wait for this fit, mesh, joints, hand-over, on the second stream - is enqueued at the next entry point
            // (bf_flush_tail): a bf_batch_stage_inputs that follows puts the next frame's inputs ahead of it
            b->tail_k = k; b->tail_big = big_fetch;
            b->copy_pending[k] = true;
            b->fetched = true;
            b->steps_done += n_iters;
            b->have_result = true;
            return BF_OK;
        }
        HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->ev_done[k], 0));
        if (big_fetch) {
            HIP_TRY(hipMemcpyAsync(k ? b->h_res_b : b->h_res, k ? b->res_b.p : b->res.p, b->res.n * fb, hipMemcpyDeviceToHost, b->copy_stream));
        } else {
            const size_t n4 = b->res.n / 4;
            hipLaunchKernelGGL(bf_publish_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 64)), dim3(256), 0, b->copy_stream,
                               (const float4 *)(k ? b->res_b.p : b->res.p), (float4 *)(k ? b->h_res_b : b->h_res), n4);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(b->ev_copied[k], b->copy_stream));
        b->copy_pending[k] = true;
        b->fetched = true;
        b->steps_done += n_iters;
        b->have_result = true;
        return BF_OK;
    }
    if (notime) {
        rc = enqueue_plain(b, n_iters, hd, io, reset, want_v, fetch, b->steps_done, nullptr);
        if (rc) return rc;
        b->fetched = fetch;
        b->steps_done += n_iters;
        b->have_result = want_v;
        return BF_OK;
    }
    HIP_TRY(hipEventRecord(b->ev[0], b->stream));
    if (!dense_losses && !dense) {
        rc = enqueue_plain(b, n_iters, hd, io, reset, want_v, fetch, b->steps_done, b->ev);
        if (rc) return rc;
    } else {
        if (reset) {
            HIP_TRY(hipMemcpyAsync(b->params.p, b->params0.p, b->params.n * fb, hipMemcpyDefault, b->stream));   // (params0 may be pinned host memory: zero-copy staging)
            HIP_TRY(hipMemsetAsync(b->adam_m.p, 0, b->adam_m.n * fb, b->stream));
            HIP_TRY(hipMemsetAsync(b->adam_v.p, 0, b->adam_v.n * fb, b->stream));
        }
        if (dense_losses) {
            rc = bf_fit_with_scans(b, n_iters, h, hd, io);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(b->ev[1], b->stream));
            if (want_v) {
                rc = bf_launch_mesh(m, &b->scratch, b->F, b->state.p, b->vraw.p, b->vout.p, b->xpart.p, b->joints.p, nullptr, b->stream, b->ev[2], nullptr);
                if (rc) return rc;
            } else HIP_TRY(hipEventRecord(b->ev[2], b->stream));
        } else {
            // reference-literal schedule: every iteration evaluates the whole mesh (smplify.py:179-190)
            for (int it = 0; it < n_iters; ++it) {
                HIP_TRY(bf_fit_launch(&m->fit, &io, &hd, 1, 0, b->adam_tab.p, b->steps_done + it, b->fit_smem, b->stream, nullptr));
                rc = bf_launch_mesh(m, &b->scratch, b->F, b->state.p, b->vraw.p, b->vout.p, b->xpart.p, b->joints.p, nullptr, b->stream, nullptr, nullptr);
                if (rc) return rc;
            }
            HIP_TRY(hipEventRecord(b->ev[1], b->stream));
            HIP_TRY(hipEventRecord(b->ev[2], b->stream));
        }
        if (fetch) {
            HIP_TRY(hipMemcpyAsync(b->cur ? b->h_res_b : b->h_res, b->cur ? b->res_b.p : b->res.p,
                                   ((want_v || dense) ? b->res.n : b->res_small) * fb, hipMemcpyDeviceToHost, b->stream));
        }
    }
    b->fetched = fetch;
    HIP_TRY(hipEventRecord(b->ev[3], b->stream));
    b->ring_n += 1;
    b->steps_done += n_iters;
    b->timed = true;
    b->have_result = want_v || dense;
    return BF_OK;
}

int bf_loss_grad(bf_batch *b, const bf_hyper *hyper, float *terms, float *grads) {
    if (!b) return fail(BF_ERR_INVALID, "bf_loss_grad: null batch");
    bf_model *m = b->m;
    HIP_TRY(hipSetDevice(m->device));
    bf_hyper h;
    if (hyper) h = *hyper; else bf_hyper_default(&h);
    int rc = ensure_adam_tab(b, h, 1);
    if (rc) return rc;
    HyperDev hd = bf_to_dev(h);
    rc = bf_guard_arena(b);
    if (rc) return rc;
    FrameIO io = bf_frame_io(b, true);
    if (m->kp_dense) { rc = bf_dense_loss_grad(b, h, hd, io); if (rc) return rc; }
    else HIP_TRY(bf_fit_launch(&m->fit, &io, &hd, 1, 1, b->adam_tab.p, 0, b->fit_smem, b->stream, nullptr));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    if (terms) HIP_TRY(hipMemcpy(terms, b->terms.p, b->terms.n * sizeof(float), hipMemcpyDeviceToHost));
    if (grads) HIP_TRY(hipMemcpy(grads, b->grads.p, b->grads.n * sizeof(float), hipMemcpyDeviceToHost));
    return BF_OK;
}

int bf_batch_sync(bf_batch *b) {
    if (!b) return fail(BF_ERR_INVALID, "bf_batch_sync: null batch");
    HIP_TRY(hipSetDevice(b->m->device));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    return BF_OK;
}

/* The result of the fit issued BEFORE the last one, while the last one is still running: frame i's rtn_dict is read under
 * frame i+1's fit (the serial loop of apps/genebody_fitting.py:183-192 as a two-deep pipeline).  Needs both fits issued with
 * BF_FIT_RESET | BF_FIT_FETCH | BF_FIT_NOTIME on the keypoint-only path (their results then alternate between the two arenas);
 * waits only for that result's hand-over, never for the stream.  params[F,n_params] and the outputs of bf_batch_get_result. */
int bf_batch_get_previous(bf_batch *b, float *params, float *vertices, float *joints, float *full_pose, float *loss_terms) {
    if (!b) return fail(BF_ERR_INVALID, "bf_batch_get_previous: null batch");
    const bf_model *m = b->m;
    HIP_TRY(hipSetDevice(m->device));
    const int k = b->cur ^ 1;
    if (b->fit_seq < 2 || b->arena_seq[k] != b->fit_seq - 2 || b->arena_seq[b->cur] != b->fit_seq - 1 || !b->arena_fetched[k])
        return fail(BF_ERR_INVALID, "bf_batch_get_previous: the previous fit's result is not held in the other arena (both fits need "
                                    "BF_FIT_RESET | BF_FIT_FETCH | BF_FIT_NOTIME on the keypoint-only path)");
    if ((vertices || joints) && !b->arena_has_v[k]) return fail(BF_ERR_INVALID, "bf_batch_get_previous: no mesh was evaluated");
    { int rt_ = bf_flush_tail(b); if (rt_) return rt_; }
    HIP_TRY(hipEventSynchronize(b->ev_copied[k]));
    const float *h = k ? b->h_res_b : b->h_res;
    if (params) std::memcpy(params, h + b->res_off[0], b->res_cnt[0] * sizeof(float));
    if (loss_terms) std::memcpy(loss_terms, h + b->res_off[1], b->res_cnt[1] * sizeof(float));
    if (joints) std::memcpy(joints, h + b->res_off[3], b->res_cnt[3] * sizeof(float));
    if (vertices) std::memcpy(vertices, h + b->res_off[4], b->res_cnt[4] * sizeof(float));
    if (full_pose) {
        const size_t stride = bf_state_stride(m->nj, m->npf, m->nb);
        for (int f = 0; f < b->F; ++f) {
            StateView v = bf_state_view(const_cast<float *>(h) + b->res_off[2] + (size_t)f * stride, m->nj, m->npf, m->nb);
            std::memcpy(full_pose + (size_t)f * 3 * m->nj, v.theta, sizeof(float) * 3 * m->nj);
        }
    }
    return BF_OK;
}

int bf_batch_get_result(bf_batch *b, float *vertices, float *joints, float *full_pose, float *loss_terms) {
    if (!b) return fail(BF_ERR_INVALID, "bf_batch_get_result: null batch");
    const bf_model *m = b->m;
    HIP_TRY(hipSetDevice(m->device));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    if ((vertices || joints) && !b->have_result)
        return fail(BF_ERR_INVALID, "bf_batch_get_result: no mesh was evaluated (BF_FIT_NO_VERTICES or no bf_fit yet)");
    if (b->fetched) {
        if (vertices) std::memcpy(vertices, b->h_vout, b->vout.n * sizeof(float));
        if (joints) std::memcpy(joints, b->h_joints, b->joints.n * sizeof(float));
        if (loss_terms) std::memcpy(loss_terms, b->h_terms, b->terms.n * sizeof(float));
    } else {
        if (vertices) HIP_TRY(hipMemcpy(vertices, b->vout.p, b->vout.n * sizeof(float), hipMemcpyDeviceToHost));
        if (joints) HIP_TRY(hipMemcpy(joints, b->joints.p, b->joints.n * sizeof(float), hipMemcpyDeviceToHost));
        if (loss_terms) HIP_TRY(hipMemcpy(loss_terms, b->terms.p, b->terms.n * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (full_pose) {
        const size_t stride = bf_state_stride(m->nj, m->npf, m->nb);
        std::vector<float> st((size_t)b->F * stride);
        if (b->fetched) std::memcpy(st.data(), b->h_state, st.size() * sizeof(float));
        else HIP_TRY(hipMemcpy(st.data(), b->state.p, st.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int f = 0; f < b->F; ++f) {
            StateView v = bf_state_view(st.data() + (size_t)f * stride, m->nj, m->npf, m->nb);
            std::memcpy(full_pose + (size_t)f * 3 * m->nj, v.theta, sizeof(float) * 3 * m->nj);
        }
    }
    return BF_OK;
}

int bf_batch_export_params_dev(bf_batch *b, void *dst_dev) {
    if (!b || !dst_dev) return fail(BF_ERR_INVALID, "bf_batch_export_params_dev: null argument");
    HIP_TRY(hipSetDevice(b->m->device));
    HIP_TRY(hipMemcpyAsync(dst_dev, b->params.p, b->params.n * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    return BF_OK;
}

int bf_batch_last_timing(bf_batch *b, float ms[4]) {
    if (!b || !ms) return fail(BF_ERR_INVALID, "bf_batch_last_timing: null argument");
    if (!b->timed) return fail(BF_ERR_INVALID, "bf_batch_last_timing: no bf_fit recorded yet");
    HIP_TRY(hipSetDevice(b->m->device));
    HIP_TRY(hipEventSynchronize(b->ev[3]));
    HIP_TRY(hipEventElapsedTime(&ms[0], b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms[1], b->ev[1], b->ev[2]));
    HIP_TRY(hipEventElapsedTime(&ms[2], b->ev[2], b->ev[3]));
    HIP_TRY(hipEventElapsedTime(&ms[3], b->ev[0], b->ev[3]));
    return BF_OK;
}

int bf_batch_timing_reset(bf_batch *b) {
    if (!b) return fail(BF_ERR_INVALID, "bf_batch_timing_reset: null batch");
    b->ring_n = 0;
    return BF_OK;
}

int bf_batch_timing_sum(bf_batch *b, float ms[4], int32_t *n_calls) {
    if (!b || !ms || !n_calls) return fail(BF_ERR_INVALID, "bf_batch_timing_sum: null argument");
    HIP_TRY(hipSetDevice(b->m->device));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    int n = std::min(b->ring_n, (int)bf_batch::kRing);
    double acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        hipEvent_t *e = b->ring.data() + (size_t)i * 4;
        float t01, t12, t23, t03;
        HIP_TRY(hipEventElapsedTime(&t01, e[0], e[1]));
        HIP_TRY(hipEventElapsedTime(&t12, e[1], e[2]));
        HIP_TRY(hipEventElapsedTime(&t23, e[2], e[3]));
        HIP_TRY(hipEventElapsedTime(&t03, e[0], e[3]));
        acc[0] += t01; acc[1] += t12; acc[2] += t23; acc[3] += t03;
    }
    for (int k = 0; k < 4; ++k) ms[k] = (float)acc[k];
    *n_calls = n;
    return BF_OK;
}

/* The full-mesh forward's OWN duration (bench.py's roofline_mesh): `reps` launches of the single-frame kernel on frame 0's pose state, each
 * leaving first-workgroup-start .. last-workgroup-end on the device's 100 MHz wall clock - no event record, no launch gap in the figure.
 * us[0..2] = mean / min / max in microseconds.  Single-frame SMPL-sized models only (the kernel bf_fit uses there). */
int bf_batch_mesh_span(bf_batch *b, int reps, float us[3]) {
    if (!b || reps <= 0 || !us) return fail(BF_ERR_INVALID, "bf_batch_mesh_span: bad argument");
    bf_model *m = b->m;
    if (bf_mesh_use_multi(m->npf, 1)) return fail(BF_ERR_UNSUPPORTED, "bf_batch_mesh_span: this model's single-frame forward is bf_mesh_multi_kernel");
    HIP_TRY(hipSetDevice(m->device));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    DevBuf<unsigned long long> d_span;
    HIP_TRY(d_span.alloc(2));
    double sum = 0.0, lo = 1e30, hi = 0.0;
    for (int r = 0; r < reps; ++r) {
        const unsigned long long init[2] = {~0ull, 0ull};
        HIP_TRY(hipMemcpy(d_span.p, init, sizeof init, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(bf_mesh_span_kernel, dim3(m->mesh.n_tiles, 1), dim3(BF_MESH_TILE * 3 * BF_MESH_RG), m->mesh_smem, b->stream, m->mesh,
                           (const float *)b->state.p, b->vraw.p, b->vout.p, b->xpart.p, d_span.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(b->stream));
        unsigned long long got[2];
        HIP_TRY(hipMemcpy(got, d_span.p, sizeof got, hipMemcpyDeviceToHost));
        const double t = (double)(got[1] - got[0]) * 0.01;          // 100 MHz ticks -> us
        sum += t; lo = std::min(lo, t); hi = std::max(hi, t);
    }
    us[0] = (float)(sum / reps); us[1] = (float)lo; us[2] = (float)hi;
    return BF_OK;
}

/* test hook: the first-iteration intermediates of frame 0 dumped by the last kernel launch */
int bf_batch_debug_dump(bf_batch *b, float *dst, int n) {
    if (!b || !dst || n <= 0 || n > 8192) return fail(BF_ERR_INVALID, "bf_batch_debug_dump: bad argument");
    HIP_TRY(hipSetDevice(b->m->device));
    { int rs_ = bf_sync_all(b); if (rs_) return rs_; }
    HIP_TRY(hipMemcpy(dst, b->debug.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return BF_OK;
}

}  // extern "C"
