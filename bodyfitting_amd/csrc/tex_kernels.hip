// Texture fitting (reference smplify/texture_fitting.py:240-275): the slice of neural_renderer its loop exercises, for gfx950.
// Compiled with -ffp-contract=off: the reference kernels' float32 operation order is kept literally (an edge test that
// flips on a contracted fma moves a pixel from one face to another).
//
//   bf_tex_project_kernel   neural_renderer/projection.py:6-42, zero distortion: world -> (u, v in [-1,1], z)
//   bf_tex_face_kernel      forward_face_index_map_cuda_kernel_1 (cuda/rasterize_cuda_kernel.cu:24-68): per face the nine projected
//                           coordinates, the back-face test, the inverted pixel-space triangle; + the 8x8-pixel tiles its bounding
//                           box touches (count pass / fill pass of the tile lists)
//   bf_tex_raster_kernel    forward_face_index_map_cuda_kernel_2 (:70-174) + forward_texture_sampling (:177-252) +
//                           forward_background (rasterize.py:181-190): one wave per tile, lane = pixel.  The reference walks ALL
//                           faces for every pixel; here a pixel walks the faces of its tile (staged through LDS 64 at a time) and
//                           keeps the lexicographic (depth, face index) minimum - the reference's strict `<` in face order - so the
//                           result does not depend on the order of the list.
//   bf_tex_compose_kernel   permute, vertical flip, 2x2 average pooling (rasterize.py:300-318) -> image[3][is][is]
//   bf_tex_loss_kernel      sum |a - b| (texture_fitting.py:266) and its derivative sign(b - a)
//   bf_tex_backward_kernel  backward_textures_cuda_kernel (:498-540) through pooling / flip / background mask: atomicAdd of
//                           sampling weight x dL/drgb into the face's texture cube (sampling indices / weights recomputed from
//                           the stored weights and depth: 20 bytes per pixel kept instead of 64)
//   bf_tex_adam_kernel      torch.optim.Adam (defaults) on every texel
#include "bf_internal.h"

#define BF_TEX_TILE 8
#define BF_TEX_REC 20            // floats per face record
#define BF_TEX_GATHER_MAX 4096   // faces whose pixel box is larger go through the per-pixel atomic path of the backward pass

struct TexView { float R[9], t[3], K[9], orig; };

extern "C" __global__ void __launch_bounds__(256)
bf_tex_project_kernel(int nv, const float *__restrict__ verts, TexView V, float *__restrict__ pv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    const float a = verts[i * 3], b = verts[i * 3 + 1], c = verts[i * 3 + 2];
    // vertices @ R^T + t : (a R00 + b R01) + c R02, then + t  (row-vector times matrix, k ascending)
    const float x = ((a * V.R[0] + b * V.R[1]) + c * V.R[2]) + V.t[0];
    const float y = ((a * V.R[3] + b * V.R[4]) + c * V.R[5]) + V.t[1];
    const float z = ((a * V.R[6] + b * V.R[7]) + c * V.R[8]) + V.t[2];
    const float x_ = x / (z + 1e-9f), y_ = y / (z + 1e-9f);
    float u = (x_ * V.K[0] + y_ * V.K[1]) + V.K[2];
    float w = (x_ * V.K[3] + y_ * V.K[4]) + V.K[5];
    w = V.orig - w;
    u = 2.f * (u - V.orig / 2.f) / V.orig;
    w = 2.f * (w - V.orig / 2.f) / V.orig;
    pv[i * 3] = u; pv[i * 3 + 1] = w; pv[i * 3 + 2] = z;
}

// face record (BF_TEX_REC floats): nine projected coordinates (x0 y0 z0 x1 y1 z1 x2 y2 z2) | nine entries of the inverted
// triangle | the pixel box that holds every pixel the face can own: x0 | x1 << 16, y0 | y1 << 16 (empty: x1 < x0)
// pass 0: the record + count the tiles of the box; pass 1: write the face into their lists (cursor = running start)
extern "C" __global__ void __launch_bounds__(256)
bf_tex_face_kernel(int nf, const int *__restrict__ faces, const float *__restrict__ pv, int is, int tiles, float *__restrict__ frec,
                   int *__restrict__ tile_count, int *__restrict__ cursor, int *__restrict__ tile_list, int pass, int cap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nf) return;
    float *rec = frec + (size_t)i * BF_TEX_REC;
    float f[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *v = pv + (size_t)faces[i * 3 + c] * 3;
        f[c * 3] = v[0]; f[c * 3 + 1] = v[1]; f[c * 3 + 2] = v[2];
    }
    if (pass == 0) { rec[18] = __int_as_float(1); rec[19] = __int_as_float(1); }          // (empty box until shown otherwise)
    if ((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])) return;            // back side: never drawn
    float p[3][2];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int d = 0; d < 2; ++d) p[n][d] = 0.5f * (f[3 * n + d] * is + is - 1);
    // pixels whose centre can pass the three edge tests lie inside the triangle's pixel-space bounding box (one pixel of slack)
    const float xmin = fminf(p[0][0], fminf(p[1][0], p[2][0])), xmax = fmaxf(p[0][0], fmaxf(p[1][0], p[2][0]));
    const float ymin = fminf(p[0][1], fminf(p[1][1], p[2][1])), ymax = fmaxf(p[0][1], fmaxf(p[1][1], p[2][1]));
    if (!(xmax >= -1.f && ymax >= -1.f && xmin <= (float)is && ymin <= (float)is)) return;       // (also drops NaN boxes)
    const int x0 = max((int)floorf(fmaxf(xmin, 0.f)) - 1, 0), x1 = min((int)ceilf(fminf(xmax, (float)is)) + 1, is - 1);
    const int y0 = max((int)floorf(fmaxf(ymin, 0.f)) - 1, 0), y1 = min((int)ceilf(fminf(ymax, (float)is)) + 1, is - 1);
    if (pass == 0) {
        float inv[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                        p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                        p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
        const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
#pragma unroll
        for (int k = 0; k < 9; ++k) { rec[k] = f[k]; rec[9 + k] = inv[k] / den; }
        rec[18] = __int_as_float(x0 | (x1 << 16)); rec[19] = __int_as_float(y0 | (y1 << 16));
    }
    for (int ty = y0 / BF_TEX_TILE; ty <= y1 / BF_TEX_TILE; ++ty)
        for (int tx = x0 / BF_TEX_TILE; tx <= x1 / BF_TEX_TILE; ++tx) {
            const int tile = ty * tiles + tx;
            if (pass == 0) atomicAdd(tile_count + tile + 1, 1);
            else { const int slot = atomicAdd(cursor + tile, 1); if (slot < cap) tile_list[slot] = i; }      // (cap: the host re-runs the pass with a larger list when the total said so)
        }
}

// texture sampling of one pixel (forward_texture_sampling, kernel.cu:205-250): the 8 corner indices and weights
__device__ __forceinline__ void tex_corners(const float w[3], float depth, const float *__restrict__ frec, int ts, int idx[8], float wt[8]) {
    float tif[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = w[k] * (ts - 1) * (depth / frec[3 * k + 2]);
        v = fmaxf(v, 0.f);
        v = fminf(v, ts - 1 - 1e-4f);
        tif[k] = v;
    }
#pragma unroll
    for (int pn = 0; pn < 8; ++pn) {
        float ww = 1.f;
        int ti[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int fl = (int)tif[k];
            if (((pn >> k) & 1) == 0) { ww *= 1.f - (tif[k] - fl); ti[k] = fl; }
            else { ww *= tif[k] - fl; ti[k] = fl + 1; }
        }
        idx[pn] = ti[0] * ts * ts + ti[1] * ts + ti[2];
        wt[pn] = ww;
    }
}

// grid (tiles * tiles / 4), 256 threads: wave = tile, lane = pixel (8 x 8).  pix[is][is] = (w0, w1, w2, depth, face) per pixel,
// rgb[is][is][3] with the background filled in.
extern "C" __global__ void __launch_bounds__(256)
bf_tex_raster_kernel(int is, int tiles, const float *__restrict__ frec, const int *__restrict__ tile_start, const int *__restrict__ tile_list,
                     const float *__restrict__ textures, int ts, float near, float far, float bg0, float bg1, float bg2,
                     float *__restrict__ pix, float *__restrict__ rgb, int cap) {
    __shared__ float s_f[4][64][19];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tile = blockIdx.x * 4 + wv;
    if (tile >= tiles * tiles) return;                     // (wave-uniform)
    const int ty = tile / tiles, tx = tile - ty * tiles;
    const int yi = ty * BF_TEX_TILE + (lane >> 3), xi = tx * BF_TEX_TILE + (lane & 7);
    const float yp = (2.f * yi + 1 - is) / is, xp = (2.f * xi + 1 - is) / is;
    float depth_min = far, wmin[3] = {0.f, 0.f, 0.f};
    int fmin = -1;
    const int s0 = min(tile_start[tile], cap), s1 = min(tile_start[tile + 1], cap);
    for (int base = s0; base < s1; base += 64) {
        const int n = min(64, s1 - base);
        __builtin_amdgcn_wave_barrier();
        if (lane < n) {
            const int fn = tile_list[base + lane];
            const float *src = frec + (size_t)fn * BF_TEX_REC;
#pragma unroll
            for (int k = 0; k < 18; ++k) s_f[wv][lane][k] = src[k];
            s_f[wv][lane][18] = __int_as_float(fn);
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < n; ++j) {
            const float *face = s_f[wv][j], *inv = face + 9;
            const int fn = __float_as_int(face[18]);
            /* check [py, px] is inside the face */
            if (((yp - face[1]) * (face[3] - face[0]) < (xp - face[0]) * (face[4] - face[1])) ||
                ((yp - face[4]) * (face[6] - face[3]) < (xp - face[3]) * (face[7] - face[4])) ||
                ((yp - face[7]) * (face[0] - face[6]) < (xp - face[6]) * (face[1] - face[7])))
                continue;
            float w[3];
            w[0] = inv[0] * xi + inv[1] * yi + inv[2];
            w[1] = inv[3] * xi + inv[4] * yi + inv[5];
            w[2] = inv[6] * xi + inv[7] * yi + inv[8];
            float wsum = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { w[k] = fminf(fmaxf(w[k], 0.f), 1.f); wsum += w[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) w[k] /= wsum;
            const float zp = 1.f / (w[0] / face[2] + w[1] / face[5] + w[2] / face[8]);
            if (zp <= near || far <= zp) continue;
            if (zp < depth_min || (zp == depth_min && fmin >= 0 && fn < fmin)) {      // first strictly nearer face in face order
                depth_min = zp; fmin = fn; wmin[0] = w[0]; wmin[1] = w[1]; wmin[2] = w[2];
            }
        }
    }
    if (yi >= is || xi >= is) return;
    const size_t o = (size_t)yi * is + xi;
    float px[3] = {bg0, bg1, bg2};
    if (fmin >= 0) {
        int idx[8];
        float wt[8];
        tex_corners(wmin, depth_min, frec + (size_t)fmin * BF_TEX_REC, ts, idx, wt);
        const float *tex = textures + (size_t)fmin * ts * ts * ts * 3;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int pn = 0; pn < 8; ++pn)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[k] += wt[pn] * (tex[idx[pn] * 3 + k] * 1.0f);      // (x 1: ambient light, lighting.py:33-37)
        // forward_background: rgb * mask + (1 - mask) * background with mask = 1
        px[0] = acc[0] * 1.f + 0.f * bg0; px[1] = acc[1] * 1.f + 0.f * bg1; px[2] = acc[2] * 1.f + 0.f * bg2;
    }
    rgb[o * 3] = px[0]; rgb[o * 3 + 1] = px[1]; rgb[o * 3 + 2] = px[2];
    float *pp = pix + o * 5;
    pp[0] = wmin[0]; pp[1] = wmin[1]; pp[2] = wmin[2]; pp[3] = depth_min; pp[4] = __int_as_float(fmin);
}

// image[c][y][x] (out x out) from rgb[is][is][3]: vertical flip, then (aa) the mean of the 2 x 2 block
extern "C" __global__ void __launch_bounds__(256)
bf_tex_compose_kernel(int out, int aa, const float *__restrict__ rgb, float *__restrict__ image) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * out * out) return;
    const int c = i / (out * out), r = i - c * out * out, y = r / out, x = r - y * out;
    const int is = aa ? out * 2 : out;
    auto at = [&](int yy, int xx) { return rgb[((size_t)(is - 1 - yy) * is + xx) * 3 + c]; };      // flipped row yy
    image[i] = aa ? (at(2 * y, 2 * x) + at(2 * y, 2 * x + 1) + at(2 * y + 1, 2 * x) + at(2 * y + 1, 2 * x + 1)) * 0.25f : at(y, x);
}

// partial[block] = sum |a - b| over the block's elements (fixed order inside a block; the host adds the blocks in order);
// grad[i] = sign(b - a)
extern "C" __global__ void __launch_bounds__(256)
bf_tex_loss_kernel(int n, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ grad, double *__restrict__ partial) {
    __shared__ double s[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (i < n) {
        const float d = b[i] - a[i];
        v = (double)fabsf(d);                        // |a - b| in float32 as torch forms it; the sum in double
        grad[i] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

// dL/drgb of one super-sampled pixel, through the pooling and the flip
__device__ __forceinline__ void tex_pixel_grad(int yi, int xi, int is, int out, int aa, const float *__restrict__ grad_image, float g[3]) {
    const int yf = is - 1 - yi, oy = aa ? yf >> 1 : yf, ox = aa ? xi >> 1 : xi;
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = grad_image[((size_t)c * out + oy) * out + ox] * (aa ? 0.25f : 1.f);
}

// backward_textures, gathered per face: one wave per face walks the face's pixel box, adds the pixels it owns into the face's
// texture cube in LDS and stores the cube (every texel of every face is written: no clearing pass, no global atomics).
// Dynamic LDS: ts^3 * 3 floats.  Faces with a box above BF_TEX_GATHER_MAX pixels store zeros and are left to
// bf_tex_backward_large_kernel.
extern "C" __global__ void __launch_bounds__(64)
bf_tex_backward_kernel(int nf, int is, int out, int aa, const float *__restrict__ pix, const float *__restrict__ frec, int ts,
                       const float *__restrict__ grad_image, float *__restrict__ grad_tex) {
    extern __shared__ float cube[];
    const int fn = blockIdx.x, lane = threadIdx.x, n = ts * ts * ts * 3;
    const float *rec = frec + (size_t)fn * BF_TEX_REC;
    for (int i = lane; i < n; i += 64) cube[i] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const int bx = __float_as_int(rec[18]), by = __float_as_int(rec[19]);
    const int x0 = bx & 0xffff, x1 = bx >> 16, y0 = by & 0xffff, y1 = by >> 16, W = x1 - x0 + 1, H = y1 - y0 + 1;
    if (W > 0 && H > 0 && W * H <= BF_TEX_GATHER_MAX) {
        for (int p = lane; p < W * H; p += 64) {
            const int yi = y0 + p / W, xi = x0 + p % W;
            const float *pp = pix + ((size_t)yi * is + xi) * 5;
            if (__float_as_int(pp[4]) != fn) continue;
            float g[3];
            tex_pixel_grad(yi, xi, is, out, aa, grad_image, g);
            const float w[3] = {pp[0], pp[1], pp[2]};
            int idx[8];
            float wt[8];
            tex_corners(w, pp[3], rec, ts, idx, wt);
#pragma unroll
            for (int pn = 0; pn < 8; ++pn)
#pragma unroll
                for (int c = 0; c < 3; ++c) atomicAdd(cube + idx[pn] * 3 + c, wt[pn] * g[c]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    float *gt = grad_tex + (size_t)fn * n;
    for (int i = lane; i < n; i += 64) gt[i] = cube[i];
}

// the faces the gather kernel left out (box above BF_TEX_GATHER_MAX pixels): per pixel, atomicAdd as the reference does
extern "C" __global__ void __launch_bounds__(256)
bf_tex_backward_large_kernel(int is, int out, int aa, const float *__restrict__ pix, const float *__restrict__ frec, int ts,
                             const float *__restrict__ grad_image, float *__restrict__ grad_tex) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= is * is) return;
    const float *pp = pix + (size_t)i * 5;
    const int fn = __float_as_int(pp[4]);
    if (fn < 0) return;
    const float *rec = frec + (size_t)fn * BF_TEX_REC;
    const int bx = __float_as_int(rec[18]), by = __float_as_int(rec[19]);
    if (((bx >> 16) - (bx & 0xffff) + 1) * ((by >> 16) - (by & 0xffff) + 1) <= BF_TEX_GATHER_MAX) return;
    const int yi = i / is, xi = i - yi * is;
    float g[3];
    tex_pixel_grad(yi, xi, is, out, aa, grad_image, g);
    const float w[3] = {pp[0], pp[1], pp[2]};
    int idx[8];
    float wt[8];
    tex_corners(w, pp[3], rec, ts, idx, wt);
    float *gt = grad_tex + (size_t)fn * ts * ts * ts * 3;
#pragma unroll
    for (int pn = 0; pn < 8; ++pn)
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(gt + idx[pn] * 3 + c, wt[pn] * g[c]);
}

// torch.optim.Adam, single-tensor form
extern "C" __global__ void __launch_bounds__(256)
bf_tex_adam_kernel(size_t n, float *__restrict__ p, float *__restrict__ m, float *__restrict__ v, const float *__restrict__ g,
                   float step_size, float bc2_sqrt, float omb1, float beta2, float omb2, float eps) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * omb1;             // lerp_(grad, 1 - beta1): the weight is formed in double on the host, like torch's python float
    const float vi = v[i] * beta2 + gi * gi * omb2;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi; v[i] = vi;
}
