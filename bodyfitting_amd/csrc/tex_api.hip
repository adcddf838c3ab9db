// Host side of the texture-fitting loop (reference smplify/texture_fitting.py:240-275; kernels: tex_kernels.hip).
#include "bf_host.h"

struct TexView { float R[9], t[3], K[9], orig; };
extern "C" __global__ void bf_tex_project_kernel(int, const float *, TexView, float *);
extern "C" __global__ void bf_tex_face_kernel(int, const int *, const float *, int, int, float *, int *, int *, int *, int, int);
extern "C" __global__ void bf_tex_raster_kernel(int, int, const float *, const int *, const int *, const float *, int, float, float, float, float,
                                                float, float *, float *, int);
extern "C" __global__ void bf_tex_compose_kernel(int, int, const float *, float *);
extern "C" __global__ void bf_tex_depth_kernel(int, int, const float *, float *);
extern "C" __global__ void bf_tex_loss_kernel(int, const float *, const float *, float *, double *);
extern "C" __global__ void bf_tex_backward_kernel(int, int, int, int, const float *, const float *, int, const float *, float *);
extern "C" __global__ void bf_tex_backward_large_kernel(int, int, int, const float *, const float *, int, const float *, float *);
extern "C" __global__ void bf_tex_adam_kernel(size_t, float *, float *, float *, const float *, float, float, float, float, float, float);
extern "C" __global__ void bf_grid_scan_kernel(int *, int *, int);

#define BF_TEX_TILE 8
#define BF_TEX_REC 20

struct bf_texmesh {
    int nv = 0, nf = 0;
    DevBuf<float> verts, tex, pv, frec, pix, rgb, m, v, grad;
    DevBuf<int> faces, tile_start, cursor, tile_list;
    int *h_total = nullptr;      // pinned: the number of tile-list entries the last render needed (read after the caller's sync)
    bool adam = false;
    void release() {
        verts.release(); tex.release(); pv.release(); frec.release(); pix.release(); rgb.release(); m.release(); v.release(); grad.release();
        faces.release(); tile_start.release(); cursor.release(); tile_list.release();
        if (h_total) { (void)hipHostFree(h_total); h_total = nullptr; }
        nv = nf = 0; adam = false;
    }
};

struct bf_texfit {
    int device = 0, out = 0, is = 0, tiles = 0, ts = 0, aa = 1, steps = 0;
    float near = 0.f, far = 100.f, bg[3] = {1.f, 1.f, 1.f};
    hipStream_t stream = nullptr;
    bf_texmesh mesh[3];               // 0 = target, 1 = fitted, 2 = scratch of bf_texfit_render_ndc
    DevBuf<float> image[3], grad_image, depth_image;
    DevBuf<double> partial;
    double *h_partial = nullptr;      // pinned copy of the loss partials
};

static TexView make_view(const float *R, const float *t, const float *K, float orig) {
    TexView V;
    std::memcpy(V.R, R, sizeof V.R); std::memcpy(V.t, t, sizeof V.t); std::memcpy(V.K, K, sizeof V.K);
    V.orig = orig;
    return V;
}

// renders mesh `which` from the view into image[which] (device); leaves pix / frec of the mesh for the backward pass.  Nothing
// here waits for the device: the tile lists are written into the capacity at hand and the number of entries they needed goes to
// pinned memory - tex_lists_fit() after the caller's next synchronisation says whether the render has to be repeated with more.
static int tex_render(bf_texfit *x, int which, const TexView &V) {
    bf_texmesh &M = x->mesh[which];
    if (!M.nf) return fail(BF_ERR_INVALID, "bf_texfit: no mesh set for this slot");
    const int is = x->is, tiles = x->tiles, ntile = tiles * tiles, cap = (int)M.tile_list.n;
    hipLaunchKernelGGL(bf_tex_project_kernel, dim3((M.nv + 255) / 256), dim3(256), 0, x->stream, M.nv, (const float *)M.verts.p, V, M.pv.p);
    HIP_TRY(hipMemsetAsync(M.tile_start.p, 0, (size_t)(ntile + 1) * sizeof(int), x->stream));
    hipLaunchKernelGGL(bf_tex_face_kernel, dim3((M.nf + 255) / 256), dim3(256), 0, x->stream, M.nf, (const int *)M.faces.p, (const float *)M.pv.p,
                       is, tiles, M.frec.p, M.tile_start.p, (int *)nullptr, (int *)nullptr, 0, cap);
    hipLaunchKernelGGL(bf_grid_scan_kernel, dim3(1), dim3(1024), 0, x->stream, M.tile_start.p, M.cursor.p, ntile + 1);
    HIP_TRY(hipMemcpyAsync(M.h_total, M.tile_start.p + ntile, sizeof(int), hipMemcpyDeviceToHost, x->stream));
    hipLaunchKernelGGL(bf_tex_face_kernel, dim3((M.nf + 255) / 256), dim3(256), 0, x->stream, M.nf, (const int *)M.faces.p, (const float *)M.pv.p,
                       is, tiles, M.frec.p, M.tile_start.p, M.cursor.p, M.tile_list.p, 1, cap);
    hipLaunchKernelGGL(bf_tex_raster_kernel, dim3((ntile + 3) / 4), dim3(256), 0, x->stream, is, tiles, (const float *)M.frec.p,
                       (const int *)M.tile_start.p, (const int *)M.tile_list.p, (const float *)M.tex.p, x->ts, x->near, x->far, x->bg[0], x->bg[1],
                       x->bg[2], M.pix.p, M.rgb.p, cap);
    hipLaunchKernelGGL(bf_tex_compose_kernel, dim3((3 * x->out * x->out + 255) / 256), dim3(256), 0, x->stream, x->out, x->aa,
                       (const float *)M.rgb.p, x->image[which].p);
    HIP_TRY(hipGetLastError());
    return BF_OK;
}

// after a synchronisation: did the last render of mesh `which` fit its tile lists?  If not the lists are grown (-> false: render again)
static int tex_lists_fit(bf_texfit *x, int which, bool *fit) {
    bf_texmesh &M = x->mesh[which];
    const size_t need = (size_t)std::max(*M.h_total, 0);
    *fit = need <= M.tile_list.n;
    if (!*fit) {
        M.tile_list.release();
        HIP_TRY(M.tile_list.alloc(need + need / 2 + 1024));
    }
    return BF_OK;
}

// renders both meshes and the loss partials + dL/dimage; ONE synchronisation, which also brings the tile-list totals: a render
// whose lists did not fit is repeated (rare: the capacity grows by half beyond what was needed)
static int tex_forward(bf_texfit *x, const TexView &V) {
    for (int attempt = 0; attempt < 3; ++attempt) {
        int rc = tex_render(x, 0, V);
        if (!rc) rc = tex_render(x, 1, V);
        if (rc) return rc;
        const int n = 3 * x->out * x->out, nb = (n + 255) / 256;
        hipLaunchKernelGGL(bf_tex_loss_kernel, dim3(nb), dim3(256), 0, x->stream, n, (const float *)x->image[0].p, (const float *)x->image[1].p,
                           x->grad_image.p, x->partial.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(x->h_partial, x->partial.p, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, x->stream));
        HIP_TRY(hipStreamSynchronize(x->stream));
        bool fit0 = true, fit1 = true;
        rc = tex_lists_fit(x, 0, &fit0);
        if (!rc) rc = tex_lists_fit(x, 1, &fit1);
        if (rc) return rc;
        if (fit0 && fit1) return BF_OK;
    }
    return fail(BF_ERR_HIP, "bf_texfit: the tile lists keep overflowing");
}

// dL/dtextures of mesh 1 from the forward pass's dL/dimage (M.grad is fully rewritten)
static int tex_backward(bf_texfit *x) {
    bf_texmesh &M = x->mesh[1];
    hipLaunchKernelGGL(bf_tex_backward_kernel, dim3(M.nf), dim3(64), (size_t)x->ts * x->ts * x->ts * 3 * sizeof(float), x->stream, M.nf, x->is,
                       x->out, x->aa, (const float *)M.pix.p, (const float *)M.frec.p, x->ts, (const float *)x->grad_image.p, M.grad.p);
    hipLaunchKernelGGL(bf_tex_backward_large_kernel, dim3((x->is * x->is + 255) / 256), dim3(256), 0, x->stream, x->is, x->out, x->aa,
                       (const float *)M.pix.p, (const float *)M.frec.p, x->ts, (const float *)x->grad_image.p, M.grad.p);
    HIP_TRY(hipGetLastError());
    return BF_OK;
}

static int tex_read_loss(bf_texfit *x, double *loss) {       // (the partials arrived with tex_forward's synchronisation)
    const int nb = (3 * x->out * x->out + 255) / 256;
    double tot = 0.0;
    for (int i = 0; i < nb; ++i) tot += x->h_partial[i];
    *loss = tot;
    return BF_OK;
}

extern "C" {

void bf_texfit_destroy(bf_texfit *x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    if (x->stream) { (void)hipStreamSynchronize(x->stream); (void)hipStreamDestroy(x->stream); }
    x->mesh[0].release(); x->mesh[1].release(); x->mesh[2].release();
    if (x->h_partial) (void)hipHostFree(x->h_partial);
    delete x;
}

int bf_texfit_create(int device, int image_size, int texture_size, float near, float far, const float *background, int anti_aliasing,
                     bf_texfit **out) {
    if (!out || image_size <= 0 || image_size > 4096 || texture_size < 2 || texture_size > 16) return fail(BF_ERR_INVALID, "bf_texfit_create: bad argument");
    *out = nullptr;
    if (device < 0 || device >= bf_device_count()) return fail(BF_ERR_NO_DEVICE, "bf_texfit_create: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    auto *x = new bf_texfit();
    x->device = device; x->out = image_size; x->aa = anti_aliasing ? 1 : 0; x->is = image_size * (x->aa ? 2 : 1); x->ts = texture_size;
    x->tiles = (x->is + BF_TEX_TILE - 1) / BF_TEX_TILE;
    x->near = near; x->far = far;
    if (background) std::memcpy(x->bg, background, sizeof x->bg);
    const size_t n = (size_t)3 * image_size * image_size;
    bool ok = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) == hipSuccess && x->image[0].alloc(n) == hipSuccess &&
              x->image[1].alloc(n) == hipSuccess && x->image[2].alloc(n) == hipSuccess && x->depth_image.alloc(n / 3) == hipSuccess &&
              x->grad_image.alloc(n) == hipSuccess && x->partial.alloc((n + 255) / 256) == hipSuccess &&
              hipHostMalloc((void **)&x->h_partial, ((n + 255) / 256) * sizeof(double)) == hipSuccess;
    if (!ok) { bf_texfit_destroy(x); return fail(BF_ERR_HIP, "bf_texfit_create: device allocation failed"); }
    *out = x;
    return BF_OK;
}

// which: 0 = the target (the textured scan), 1 = the mesh whose textures are fitted (SMPL+D).  textures[nf][ts][ts][ts][3]
// (neural_renderer's per-face texture cubes, load_obj.py / load_textures).  Setting mesh 1 resets the Adam state.
static int tex_set_mesh(bf_texfit *x, int which, int n_verts, const float *verts, int n_faces, const int32_t *faces, const float *textures);

int bf_texfit_set_mesh(bf_texfit *x, int which, int n_verts, const float *verts, int n_faces, const int32_t *faces, const float *textures) {
    if (!x || which < 0 || which > 1) return fail(BF_ERR_INVALID, "bf_texfit_set_mesh: bad argument");
    return tex_set_mesh(x, which, n_verts, verts, n_faces, faces, textures);
}

static int tex_set_mesh(bf_texfit *x, int which, int n_verts, const float *verts, int n_faces, const int32_t *faces, const float *textures) {
    if (!x || n_verts <= 0 || n_faces <= 0 || !verts || !faces || !textures)
        return fail(BF_ERR_INVALID, "bf_texfit_set_mesh: bad argument");
    for (int i = 0; i < n_faces * 3; ++i)
        if (faces[i] < 0 || faces[i] >= n_verts) return fail(BF_ERR_INVALID, "bf_texfit_set_mesh: face index out of range");
    HIP_TRY(hipSetDevice(x->device));
    HIP_TRY(hipStreamSynchronize(x->stream));
    bf_texmesh &M = x->mesh[which];
    M.release();
    M.nv = n_verts; M.nf = n_faces;
    const size_t ntex = (size_t)n_faces * x->ts * x->ts * x->ts * 3, npx = (size_t)x->is * x->is, ntile = (size_t)x->tiles * x->tiles;
    HIP_TRY(M.verts.upload(std::vector<float>(verts, verts + (size_t)n_verts * 3)));
    HIP_TRY(M.faces.upload(std::vector<int>(faces, faces + (size_t)n_faces * 3)));
    HIP_TRY(M.tex.upload(std::vector<float>(textures, textures + ntex)));
    HIP_TRY(M.pv.alloc((size_t)n_verts * 3)); HIP_TRY(M.frec.alloc((size_t)n_faces * BF_TEX_REC));
    HIP_TRY(M.pix.alloc(npx * 5)); HIP_TRY(M.rgb.alloc(npx * 3));
    HIP_TRY(M.tile_start.alloc(ntile + 1)); HIP_TRY(M.cursor.alloc(ntile + 1));
    HIP_TRY(M.tile_list.alloc((size_t)n_faces * 4 + ntile + 1024));         // (first guess; grown when a render says so)
    HIP_TRY(hipHostMalloc((void **)&M.h_total, sizeof(int)));
    *M.h_total = 0;
    if (which == 1) {
        HIP_TRY(M.m.alloc(ntex)); HIP_TRY(M.v.alloc(ntex)); HIP_TRY(M.grad.alloc(ntex));
        HIP_TRY(bf_memset_sync(M.m.p, 0, ntex * sizeof(float))); HIP_TRY(bf_memset_sync(M.v.p, 0, ntex * sizeof(float)));
        M.adam = true;
        x->steps = 0;
    }
    return BF_OK;
}

// Renderer.render_texture (neural_renderer/renderer.py:294-346: nr.rasterize_rgbad on faces that are already in normalised device
// coordinates - the OBJ's `vt` lines mapped to [-1, 1], z = 1) = what render_texture_map (smplify/texture_fitting.py:149-151,298)
// turns into the UV-space texture image smpl.png.  ndc[n_verts][3], faces[n_faces][3], textures[n_faces][ts][ts][ts][3] ->
// rgb[3][image_size][image_size], depth[image_size][image_size] (far where nothing was drawn); either output may be NULL.
// No projection: the vertices go to the rasteriser as they are.
int bf_texfit_render_ndc(bf_texfit *x, int n_verts, const float *ndc, int n_faces, const int32_t *faces, const float *textures, float *rgb, float *depth) {
    if (!x) return fail(BF_ERR_INVALID, "bf_texfit_render_ndc: null handle");
    int rc = tex_set_mesh(x, 2, n_verts, ndc, n_faces, faces, textures);
    if (rc) return rc;
    TexView V{};
    V.orig = -1.f;                                    // (bf_tex_project_kernel: pass the vertices through)
    for (int attempt = 0; attempt < 3; ++attempt) {
        rc = tex_render(x, 2, V);
        if (rc) return rc;
        hipLaunchKernelGGL(bf_tex_depth_kernel, dim3((x->out * x->out + 255) / 256), dim3(256), 0, x->stream, x->out, x->aa,
                           (const float *)x->mesh[2].pix.p, x->depth_image.p);
        HIP_TRY(hipGetLastError());
        if (rgb) HIP_TRY(hipMemcpyAsync(rgb, x->image[2].p, x->image[2].n * sizeof(float), hipMemcpyDeviceToHost, x->stream));
        if (depth) HIP_TRY(hipMemcpyAsync(depth, x->depth_image.p, x->depth_image.n * sizeof(float), hipMemcpyDeviceToHost, x->stream));
        HIP_TRY(hipStreamSynchronize(x->stream));
        bool fit = true;
        rc = tex_lists_fit(x, 2, &fit);
        if (rc) return rc;
        if (fit) { x->mesh[2].release(); return BF_OK; }
    }
    return fail(BF_ERR_HIP, "bf_texfit_render_ndc: the tile lists keep overflowing");
}

// Renderer.render_rgb (neural_renderer/renderer.py:174-232, camera_mode='projection', ambient light 1, fill_back=False):
// R[9], t[3] world-to-camera, K[9], orig_size -> rgb[3][image_size][image_size] (host)
int bf_texfit_render(bf_texfit *x, int which, const float *R, const float *t, const float *K, float orig_size, float *rgb) {
    if (!x || which < 0 || which > 1 || !R || !t || !K || !rgb) return fail(BF_ERR_INVALID, "bf_texfit_render: bad argument");
    HIP_TRY(hipSetDevice(x->device));
    for (int attempt = 0; attempt < 3; ++attempt) {
        int rc = tex_render(x, which, make_view(R, t, K, orig_size));
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(rgb, x->image[which].p, x->image[which].n * sizeof(float), hipMemcpyDeviceToHost, x->stream));
        HIP_TRY(hipStreamSynchronize(x->stream));
        bool fit = true;
        rc = tex_lists_fit(x, which, &fit);
        if (rc) return rc;
        if (fit) return BF_OK;
    }
    return fail(BF_ERR_HIP, "bf_texfit_render: the tile lists keep overflowing");
}

// One iteration of texture_fitting.py:262-270: render both meshes from the view, loss = sum |scan_img - smpl_img|, backward to
// the fitted mesh's textures, one Adam step (lr; torch defaults otherwise).  *loss receives the loss of THIS view before the step.
int bf_texfit_step(bf_texfit *x, const float *R, const float *t, const float *K, float orig_size, float lr, double *loss) {
    if (!x || !R || !t || !K) return fail(BF_ERR_INVALID, "bf_texfit_step: bad argument");
    if (!x->mesh[1].adam) return fail(BF_ERR_INVALID, "bf_texfit_step: set the mesh to fit (slot 1) first");
    HIP_TRY(hipSetDevice(x->device));
    int rc = tex_forward(x, make_view(R, t, K, orig_size));
    if (!rc) rc = tex_backward(x);
    if (rc) return rc;
    bf_texmesh &M = x->mesh[1];
    x->steps += 1;
    const double b1 = 0.9, b2 = 0.999;
    const float step_size = (float)((double)lr / (1.0 - std::pow(b1, x->steps))), bc2_sqrt = (float)std::sqrt(1.0 - std::pow(b2, x->steps));
    const size_t ntex = M.tex.n;
    hipLaunchKernelGGL(bf_tex_adam_kernel, dim3((unsigned)((ntex + 255) / 256)), dim3(256), 0, x->stream, ntex, M.tex.p, M.m.p, M.v.p, M.grad.p,
                       step_size, bc2_sqrt, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), 1e-8f);
    HIP_TRY(hipGetLastError());
    return loss ? tex_read_loss(x, loss) : BF_OK;
}

// loss and dL/dtextures of the fitted mesh from this view, no step (what `loss.backward()` leaves in smpl_t.grad, :266-269)
int bf_texfit_loss_grad(bf_texfit *x, const float *R, const float *t, const float *K, float orig_size, double *loss, float *grad) {
    if (!x || !R || !t || !K || !grad) return fail(BF_ERR_INVALID, "bf_texfit_loss_grad: bad argument");
    if (!x->mesh[1].adam) return fail(BF_ERR_INVALID, "bf_texfit_loss_grad: set the mesh to fit (slot 1) first");
    HIP_TRY(hipSetDevice(x->device));
    int rc = tex_forward(x, make_view(R, t, K, orig_size));
    if (!rc) rc = tex_backward(x);
    if (rc) return rc;
    bf_texmesh &M = x->mesh[1];
    HIP_TRY(hipMemcpyAsync(grad, M.grad.p, M.grad.n * sizeof(float), hipMemcpyDeviceToHost, x->stream));
    HIP_TRY(hipStreamSynchronize(x->stream));
    return loss ? tex_read_loss(x, loss) : BF_OK;
}

int bf_texfit_get_textures(bf_texfit *x, float *textures) {
    if (!x || !textures || !x->mesh[1].nf) return fail(BF_ERR_INVALID, "bf_texfit_get_textures: bad argument");
    HIP_TRY(hipSetDevice(x->device));
    HIP_TRY(hipStreamSynchronize(x->stream));
    HIP_TRY(hipMemcpy(textures, x->mesh[1].tex.p, x->mesh[1].tex.n * sizeof(float), hipMemcpyDeviceToHost));
    return BF_OK;
}

}  // extern "C"
