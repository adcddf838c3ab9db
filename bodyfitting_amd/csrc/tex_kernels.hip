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
    if (V.orig < 0.f) { pv[i * 3] = a; pv[i * 3 + 1] = b; pv[i * 3 + 2] = c; return; }      // already normalised device coordinates (UV-space render)
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

// ---- the per-face / per-pixel arithmetic of the rasteriser, in this file's own terms ---------------------------------------------
// (The VALUES follow neural_renderer's float32 operation order - rasterize_cuda_kernel.cu:38-63,110-137,209-240 - because a render has
// to agree with it pixel for pixel: an edge test that rounds the other way hands a pixel to the neighbouring face.  This file is
// compiled without fused multiply-adds for the same reason.)
struct TexTri { float x[3], y[3], z[3]; };       // a face's corners: normalised device coordinates + depth

__device__ __forceinline__ TexTri tex_tri(const float *f9) {
    TexTri t;
#pragma unroll
    for (int c = 0; c < 3; ++c) { t.x[c] = f9[3 * c]; t.y[c] = f9[3 * c + 1]; t.z[c] = f9[3 * c + 2]; }
    return t;
}

// the face shows its back when its signed area is negative: (c2 - c0) x (c1 - c0) compared as two products
__device__ __forceinline__ bool tex_back_facing(const TexTri &t) {
    return (t.y[2] - t.y[0]) * (t.x[1] - t.x[0]) < (t.y[1] - t.y[0]) * (t.x[2] - t.x[0]);
}

// A pixel centre (xp, yp) lies outside the face when it is on the wrong side of one of the three directed edges a -> b.
__device__ __forceinline__ bool tex_outside(const TexTri &t, float xp, float yp) {
    bool out = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int b = (a + 1) % 3;
        out = out || (yp - t.y[a]) * (t.x[b] - t.x[a]) < (xp - t.x[a]) * (t.y[b] - t.y[a]);
    }
    return out;
}

// Rows of the inverse of [[x0 x1 x2], [y0 y1 y2], [1 1 1]] over pixel-space corners: row k gives corner k's barycentric weight as
// a x + b y + c.  Row k is the cofactor row of the two OTHER corners taken cyclically, a = k + 1, b = k + 2.
__device__ __forceinline__ void tex_barycentric_rows(const float px[3], const float py[3], float rows[9]) {
    float cof[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        cof[3 * k] = py[a] - py[b];
        cof[3 * k + 1] = px[b] - px[a];
        cof[3 * k + 2] = px[a] * py[b] - px[b] * py[a];
    }
    const float det = px[2] * cof[6] + px[0] * cof[0] + px[1] * cof[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) rows[k] = cof[k] / det;
}

// weights of pixel (xi, yi) clamped to [0, 1] and renormalised; returns the interpolated depth 1 / sum(w_k / z_k)
__device__ __forceinline__ float tex_weights(const float rows[9], const TexTri &t, int xi, int yi, float w[3]) {
    float total = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w[k] = fminf(fmaxf(rows[3 * k] * xi + rows[3 * k + 1] * yi + rows[3 * k + 2], 0.f), 1.f);
        total += w[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] /= total;
    return 1.f / (w[0] / t.z[0] + w[1] / t.z[1] + w[2] / t.z[2]);
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
    const TexTri tri = tex_tri(f);
    if (tex_back_facing(tri)) return;                                                      // never drawn
    float px[3], py[3];                                                                    // corners in pixel units of the super-sampled image
#pragma unroll
    for (int n = 0; n < 3; ++n) { px[n] = 0.5f * (tri.x[n] * is + is - 1); py[n] = 0.5f * (tri.y[n] * is + is - 1); }
    // pixels whose centre can pass the three edge tests lie inside the triangle's pixel-space bounding box (one pixel of slack)
    const float xmin = fminf(px[0], fminf(px[1], px[2])), xmax = fmaxf(px[0], fmaxf(px[1], px[2]));
    const float ymin = fminf(py[0], fminf(py[1], py[2])), ymax = fmaxf(py[0], fmaxf(py[1], py[2]));
    if (!(xmax >= -1.f && ymax >= -1.f && xmin <= (float)is && ymin <= (float)is)) return;       // (also drops NaN boxes)
    const int x0 = max((int)floorf(fmaxf(xmin, 0.f)) - 1, 0), x1 = min((int)ceilf(fminf(xmax, (float)is)) + 1, is - 1);
    const int y0 = max((int)floorf(fmaxf(ymin, 0.f)) - 1, 0), y1 = min((int)ceilf(fminf(ymax, (float)is)) + 1, is - 1);
    if (pass == 0) {
        float rows[9];
        tex_barycentric_rows(px, py, rows);
#pragma unroll
        for (int k = 0; k < 9; ++k) { rec[k] = f[k]; rec[9 + k] = rows[k]; }
        rec[18] = __int_as_float(x0 | (x1 << 16)); rec[19] = __int_as_float(y0 | (y1 << 16));
    }
    for (int ty = y0 / BF_TEX_TILE; ty <= y1 / BF_TEX_TILE; ++ty)
        for (int tx = x0 / BF_TEX_TILE; tx <= x1 / BF_TEX_TILE; ++tx) {
            const int tile = ty * tiles + tx;
            if (pass == 0) atomicAdd(tile_count + tile + 1, 1);
            else { const int slot = atomicAdd(cursor + tile, 1); if (slot < cap) tile_list[slot] = i; }      // (cap: the host re-runs the pass with a larger list when the total said so)
        }
}

// Texture sampling of one pixel: position inside the face's ts^3 texture cube = barycentric weight x (ts - 1), perspective-corrected by
// depth / corner depth and kept inside the cube; the colour is the trilinear blend of the 8 texels around it.  idx / wt: their
// indices and weights, corner bit k set = the upper texel along axis k.
__device__ __forceinline__ void tex_corners(const float w[3], float depth, const float *__restrict__ frec, int ts, int idx[8], float wt[8]) {
    int cell[3];
    float hi[3], lo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pos = fminf(fmaxf(w[k] * (ts - 1) * (depth / frec[3 * k + 2]), 0.f), ts - 1 - 1e-4f);
        cell[k] = (int)pos;
        hi[k] = pos - cell[k];
        lo[k] = 1.f - hi[k];
    }
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const int u0 = corner & 1, u1 = (corner >> 1) & 1, u2 = corner >> 2;
        idx[corner] = (cell[0] + u0) * ts * ts + (cell[1] + u1) * ts + (cell[2] + u2);
        wt[corner] = (u0 ? hi[0] : lo[0]) * (u1 ? hi[1] : lo[1]) * (u2 ? hi[2] : lo[2]);
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
            const float *face = s_f[wv][j];
            const int fn = __float_as_int(face[18]);
            const TexTri tri = tex_tri(face);
            if (tex_outside(tri, xp, yp)) continue;
            float w[3];
            const float zp = tex_weights(face + 9, tri, xi, yi, w);
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
        for (int corner = 0; corner < 8; ++corner)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[k] += wt[corner] * (tex[idx[corner] * 3 + k] * 1.0f);      // (x 1: ambient light, lighting.py:33-37)
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

// depth[y][x] (out x out) from pix[is][is][5]'s depth (far where no face was drawn): the same flip and 2 x 2 mean as the colours
extern "C" __global__ void __launch_bounds__(256)
bf_tex_depth_kernel(int out, int aa, const float *__restrict__ pix, float *__restrict__ depth) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= out * out) return;
    const int y = i / out, x = i - y * out, is = aa ? out * 2 : out;
    auto at = [&](int yy, int xx) { return pix[((size_t)(is - 1 - yy) * is + xx) * 5 + 3]; };
    depth[i] = aa ? (at(2 * y, 2 * x) + at(2 * y, 2 * x + 1) + at(2 * y + 1, 2 * x) + at(2 * y + 1, 2 * x + 1)) * 0.25f : at(y, x);
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
            for (int corner = 0; corner < 8; ++corner)
#pragma unroll
                for (int c = 0; c < 3; ++c) atomicAdd(cube + idx[corner] * 3 + c, wt[corner] * g[c]);
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
    for (int corner = 0; corner < 8; ++corner)
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(gt + idx[corner] * 3 + c, wt[corner] * g[c]);
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
