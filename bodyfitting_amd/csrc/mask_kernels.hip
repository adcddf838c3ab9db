// Silhouette loss for gfx950: reference smplify/loss.py:85-130 `multview_mask_loss`, with its gradient
// w.r.t. the (every 4th, loss.py:99) body vertices.
//
//   per mask view:  uv = project(verts[::4]);  inside = 0 <= uv < imsize (both axes)
//     contour term  sum over contour points c of  coeff_c * min_{inside s} |uv_s - c|,
//                   coeff_c = 10 if the mask is empty at trunc(uv_argmin) else 1        (loss.py:110-119)
//     binary term   10 * sum over ALL sampled verts of bilinear(1 - mask)(uv)               (loss.py:123-128,
//                   grid_sample: bilinear, zeros padding, align_corners=False as torch 2.x evaluates it)
//
//   bf_mask_project_kernel   thread = sampled vertex x view: uv, inside flag, binary term + d/duv
//   bf_mask_contour_kernel   thread = contour point: exact nearest inside vertex (first minimum, like
//                            torch.min), its weight, the unit direction
//   bf_mask_gather_kernel    workgroup = (64 sampled vertices, view): lists the contour points that chose them IN CONTOUR
//                            ORDER (deterministic; no atomics), sums per vertex, maps d/duv back through the projection
//   bf_mask_gsum_kernel      adds the views in view order into dL/dvertex
// Distances are exact (a - b)^2 sums; torch.cdist switches to the |a|^2+|b|^2-2ab form for these sizes,
// which is noisier (about 1e-2 px at 512 px) - see DESIGN.md.
#include "bf_internal.h"
#include "loss_bodies.h"

namespace {
__device__ inline float mk_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
}  // namespace

// grid (ceil(Ns/256), M, F).  uvi[F][M][Ns] = (u, v, inside, 1/pix_z);  duvb[F][M][Ns][2] = d(binary term)/duv * weight
extern "C" __global__ void __launch_bounds__(256)
bf_mask_project_kernel(MaskIO K, const float *__restrict__ vout, const float *__restrict__ proj_all, float *__restrict__ uvi,
                       float *__restrict__ duvb, float *__restrict__ loss_part) {
    __shared__ float sred[4];
    const int s = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y, f = blockIdx.z;
    float lval = 0.f;
    if (s < K.ns) {
        const float *X = vout + ((size_t)f * K.nv + (size_t)s * K.sstride) * 3;
        lval = bf_mask_project_one(K, X[0], X[1], X[2], proj_all, f, m, s, uvi, duvb);
    }
    lval = mk_wave_sum(lval);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = lval;
    __syncthreads();
    if (threadIdx.x == 0)
        loss_part[((size_t)f * K.n_masks + m) * K.part_stride + blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

// grid (ceil(16 Cmax/256), M, F).  For contour point c: choice[F][M][Cmax] = sampled vertex (or -1),
// cgrad[F][M][Cmax][2] = weight * coeff * (uv - c) / |uv - c|.  SIXTEEN lanes per contour point, each scanning every 16th
// sampled vertex; the group is merged with the lexicographic (distance, index) minimum = torch.min's first minimum.
extern "C" __global__ void __launch_bounds__(256)
bf_mask_contour_kernel(MaskIO K, const float *__restrict__ uvi, int *__restrict__ choice, float *__restrict__ cgrad,
                       float *__restrict__ loss_part) {
    __shared__ float4 tile[256];
    __shared__ float sred[4];
    bf_mask_contour_body<256>(blockIdx.x, blockIdx.y, blockIdx.z, tile, sred, K, uvi, choice, cgrad, loss_part);
}

// grid (ceil(Ns/64), M, F), 256 threads.  gpart[f][m][s][3] = dL/dvertex 4 s from mask view m (one workgroup per view and
// block of 64 sampled vertices: the views run in parallel; bf_mask_gsum_kernel adds them in view order).  The workgroup
// walks the contour 1,024 points at a time (four per thread); a point that chose one of ITS vertices is appended - in contour order, by ballot
// ranks - to a short list in LDS (a block of 64 vertices is chosen by ~70 of the ~3000 points), and the 64 vertex lanes add
// up their entries of the list in list order = contour order (deterministic; no atomics).  The first version compared every
// point with every vertex of the block: 64 compares per point instead of one.
extern "C" __global__ void __launch_bounds__(256)
bf_mask_gather_kernel(MaskIO K, const float *__restrict__ proj_all, const float *__restrict__ uvi, const float *__restrict__ duvb,
                      const int *__restrict__ choice, const float *__restrict__ cgrad, float *__restrict__ gpart) {
    constexpr int CAP = 2048, SUB = 4;                     // SUB x 256 contour points per step: two barriers per 1,024 points, not per 256
    __shared__ int s_v[CAP];                               // vertex (0..63 inside the block) of a listed point
    __shared__ float2 s_g[CAP];
    __shared__ int s_wcnt[SUB * 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, s0 = blockIdx.x * 64, m = blockIdx.y, f = blockIdx.z;
    const int vm = f * K.n_masks + m;
    const int cnt = K.contour_count[vm];
    // (what the closing step needs is requested now: it depends on nothing the walk produces)
    const int s_fin = s0 + tid;
    float fin_tu = 0.f, fin_tv = 0.f;
    float4 fin_r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 64 && s_fin < K.ns) {
        fin_tu = duvb[((size_t)vm * K.ns + s_fin) * 2]; fin_tv = duvb[((size_t)vm * K.ns + s_fin) * 2 + 1];
        fin_r = ((const float4 *)uvi)[(size_t)vm * K.ns + s_fin];
    }
    float du = 0.f, dv = 0.f;                              // (vertex lanes: tid < 64)
    int n = 0;                                             // (uniform) entries in the list
    auto drain = [&]() {
        if (tid < 64)
            for (int i = 0; i < n; ++i)
                if (s_v[i] == tid) { du += s_g[i].x; dv += s_g[i].y; }
        n = 0;
        __syncthreads();
    };
    // (the next step's points are requested before this step's are handled)
    const int *chp = choice + (size_t)vm * K.cmax;
    const float2 *cgp = (const float2 *)cgrad + (size_t)vm * K.cmax;
    int ch_n[SUB];
    float2 g_n[SUB];
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        const int c = j * 256 + tid;
        ch_n[j] = c < cnt ? chp[c] : -1;
        g_n[j] = c < cnt ? cgp[c] : make_float2(0.f, 0.f);
    }
    for (int base = 0; base < cnt; base += SUB * 256) {
        int ch[SUB];
        float2 g[SUB];
        bool mine[SUB];
        unsigned long long mk[SUB];
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            ch[j] = ch_n[j]; g[j] = g_n[j];
            const int c2 = base + SUB * 256 + j * 256 + tid;
            ch_n[j] = c2 < cnt ? chp[c2] : -1;
            g_n[j] = c2 < cnt ? cgp[c2] : make_float2(0.f, 0.f);
            mine[j] = ch[j] >= s0 && ch[j] < s0 + 64;
            mk[j] = __ballot(mine[j]);
            if (lane == 0) s_wcnt[j * 4 + wv] = __popcll(mk[j]);
        }
        __syncthreads();
        int off = n;                                       // contour order: sub-tile after sub-tile, wave after wave, lane after lane
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            int mine_off = off;
            for (int w = 0; w < wv; ++w) mine_off += s_wcnt[j * 4 + w];
            if (mine[j]) { const int o = mine_off + __popcll(mk[j] & ((1ull << lane) - 1ull)); s_v[o] = ch[j] - s0; s_g[o] = g[j]; }
            off += s_wcnt[j * 4] + s_wcnt[j * 4 + 1] + s_wcnt[j * 4 + 2] + s_wcnt[j * 4 + 3];
        }
        n = off;
        __syncthreads();
        if (n > CAP - SUB * 256) drain();
    }
    drain();
    const int s = s_fin;
    if (tid < 64 && s < K.ns) {
        float tu = fin_tu, tv = fin_tv;
        tu += du; tv += dv;
        float4 r = fin_r;
        const float *P = proj_all + ((size_t)f * K.n_views + K.view_index[m]) * 12;
        float q0 = tu * r.w, q1 = tv * r.w, q2 = -(tu * r.x + tv * r.y) * r.w;
        float *o = gpart + ((size_t)vm * K.ns + s) * 3;
        o[0] = P[0] * q0 + P[4] * q1 + P[8] * q2;
        o[1] = P[1] * q0 + P[5] * q1 + P[9] * q2;
        o[2] = P[2] * q0 + P[6] * q1 + P[10] * q2;
    }
}

// grid (ceil(Ns/256), F).  dvout[f][4 s] += sum over the mask views, in view order; other vertices untouched
extern "C" __global__ void __launch_bounds__(256)
bf_mask_gsum_kernel(MaskIO K, const float *__restrict__ gpart, float *__restrict__ dvout) {
    const int s = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (s >= K.ns) return;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int m = 0; m < K.n_masks; ++m) {
        const float *p = gpart + (((size_t)f * K.n_masks + m) * K.ns + s) * 3;
        g0 += p[0]; g1 += p[1]; g2 += p[2];
    }
    float *o = dvout + ((size_t)f * K.nv + (size_t)s * K.sstride) * 3;
    o[0] += g0; o[1] += g1; o[2] += g2;
}

// grid (F): loss[f] = sum over views / blocks of the partials (fixed order)
extern "C" __global__ void bf_mask_loss_kernel(MaskIO K, const float *__restrict__ loss_part, float *__restrict__ loss) {
    const int f = blockIdx.x;
    if (threadIdx.x != 0) return;
    float tot = 0.f;
    for (int m = 0; m < K.n_masks; ++m) {
        const int vm = f * K.n_masks + m;
        const int nb = K.proj_blocks + (K.contour_count[vm] * 16 + 255) / 256;         // (16 lanes per contour point)
        for (int b = 0; b < nb; ++b) tot += loss_part[(size_t)vm * K.part_stride + b];
    }
    loss[f] = tot;
}

// ---- extract_countours (smplify/loss.py:73-83) on the device ----------------------------------------------------------------
// cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_NONE) is Suzuki-Abe border following (CVGIP 30, 1985, Algorithm 1); this
// kernel runs it per mask: ONE WAVE PER MASK, the image as three bit planes in LDS (foreground | marked | negative mark:
// 96 KB for 512 x 512; images too large for LDS use the same planes in global memory).  The lanes build the planes by ballot, then the wave
// scans the rows a 32-pixel word at a time (outer-border starts = foreground & ~(left neighbour) & ~marked, by bit tricks) and
// follows every external border once, marking it and writing its points; the longest one (first on ties) is kept.  A walk step reads the 3 x 3 neighbourhood as three word pairs and finishes in registers.
// External = not inside a hole of another component: a start is skipped while the last marked pixel met on its row carries a
// positive mark (between the left and the right edge of a traced border).  See oracle/contour_oracle.py for the restatement
// this is tested against, and for which contour the reference keeps.
namespace {
struct ContourPlanes {
    unsigned *fg, *mk, *ng;
    int H, W, wpr;                         // wpr = 32-bit words per row
    __device__ bool get(const unsigned *pl, int x, int y) const {
        return x >= 0 && y >= 0 && x < W && y < H && ((pl[y * wpr + (x >> 5)] >> (x & 31)) & 1u);
    }
    // the eight neighbours of (x, y) as a bit mask: bit c = direction code c (0 east, then counter-clockwise on the screen).
    // Called by the whole wave with the same (x, y): lane c < 8 fetches neighbour c, the ballot is the mask.
    __device__ unsigned around(int x, int y) const {
        const int c = threadIdx.x & 7;
        const int dx = (c == 0 || c == 1 || c == 7) ? 1 : ((c == 3 || c == 4 || c == 5) ? -1 : 0);
        const int dy = (c >= 1 && c <= 3) ? -1 : ((c >= 5) ? 1 : 0);
        return (unsigned)__ballot(threadIdx.x < 8 && get(fg, x + dx, y + dy)) & 0xffu;
    }
    __device__ void mark(int x, int y, bool negative) {
        const int w = y * wpr + (x >> 5);
        const unsigned b = 1u << (x & 31);
        if (negative) { mk[w] |= b; ng[w] |= b; }
        else if (!(mk[w] & b)) mk[w] |= b;                 // (a positive mark never replaces an earlier mark)
    }
};

// Follow the outer border that starts at (x0, y0) - the whole wave in step, every lane with the same state (the lanes share
// the neighbourhood fetch; lane 0 alone marks and writes).  out != null: write the points (at most cap).  Returns the length.
__device__ int contour_follow(ContourPlanes &P, int x0, int y0, float *out, int cap, bool set_marks) {
    const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1}, DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    unsigned nb = P.around(x0, y0);
    // first neighbour clockwise, starting after the left one: codes 3, 2, 1, 0, 7, 6, 5
    int first = -1;
    for (int k = 0, s = 3; k < 7; ++k, s = (s - 1) & 7)
        if (nb & (1u << s)) { first = s; break; }
    if (first < 0) {
        if (threadIdx.x == 0) {
            if (set_marks) P.mark(x0, y0, true);
            if (out && cap > 0) { out[0] = (float)x0; out[1] = (float)y0; }
        }
        return 1;
    }
    const int x1 = x0 + DX[first], y1 = y0 + DY[first];
    int x3 = x0, y3 = y0, s = first, n = 0;
    for (;;) {
        // first neighbour counter-clockwise, starting after the direction we came from; `passed_east`: code 8 was examined
        int t = s + 1;
        while (!(nb & (1u << (t & 7)))) ++t;
        const bool passed_east = t >= 9;
        if (threadIdx.x == 0) {
            if (set_marks) P.mark(x3, y3, passed_east);
            if (out && n < cap) { out[n * 2] = (float)x3; out[n * 2 + 1] = (float)y3; }
        }
        ++n;
        const int x4 = x3 + DX[t & 7], y4 = y3 + DY[t & 7];
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) return n;
        x3 = x4; y3 = y4;
        s = ((t & 7) + 4) & 7;
        nb = P.around(x3, y3);
    }
}
}  // namespace

// grid (n_masks), 256 threads: four waves build the bit planes, wave 0 then follows the borders alone.
// masks[n][H][W] (non-zero = foreground) -> count[n] = length of the kept contour (also when it
// exceeds cap: the caller retries with more room), count[n + m] = which half of xy[m][2][cap][2] holds its points (x, y).  planes_global: 3 * H * wpr words per mask,
// or null when the planes fit the dynamic LDS given to the launch.
extern "C" __global__ void __launch_bounds__(256)
bf_contour_kernel(const unsigned char *__restrict__ masks, int H, int W, int cap, int select, float *__restrict__ xy, int *__restrict__ count,
                  unsigned *planes_global) {
    extern __shared__ unsigned s_planes[];
    const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ContourPlanes P;
    P.H = H; P.W = W; P.wpr = (W + 31) >> 5;
    const int plane = H * P.wpr;
    P.fg = planes_global ? planes_global + (size_t)m * 3 * plane : s_planes;
    P.mk = P.fg + plane; P.ng = P.mk + plane;
    const unsigned char *img = masks + (size_t)m * H * W;
    // bit planes: 64 consecutive pixels per step, one coalesced byte load per lane, the ballot is two finished words; the steps are
    // dealt out over the four waves (the loop is a chain of load -> ballot -> store round trips: 1,024 of them for 512 x 512 on one wave)
    const int chunks = (W + 63) >> 6;
    for (int it = wave * 4; it < H * chunks; it += 16) {
        unsigned long long bits[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = it + k, y = i / chunks, x = (i - y * chunks) * 64 + lane;
            const bool on = i < H * chunks && x < W && img[(size_t)y * W + x] != 0;
            bits[k] = __ballot(on);
        }
        if (lane < 8) {
            const int k = lane >> 1, half = lane & 1, i = it + k, y = i / chunks, wi = (i - y * chunks) * 2 + half;
            if (i < H * chunks && wi < P.wpr) {
                const int w = y * P.wpr + wi;
                P.fg[w] = (unsigned)(bits[k] >> (32 * half)); P.mk[w] = 0; P.ng[w] = 0;
            }
        }
    }
    __threadfence_block();                 // (the planes of a large image live in global memory: the other waves' words, before wave 0 reads them)
    __syncthreads();
    if (wave != 0) return;
    // from here on every lane runs the same scalar program (one wave: lock-step, LDS accesses in program order)
    // Outer-border starts are rare (one per component, plus the skipped ones next to holes), so the words are screened 64
    // at a time - lane l looks at word base + l - and only a word with a start is handled, serially.  A walk changes marks,
    // so the screen is repeated from the handled word on.  Points are written while walking, into the half of the slab that
    // does not hold the border kept so far.
    int best_len = 0, best_half = 0;
    float *slab = xy + (size_t)m * 2 * cap * 2;
    for (int base = 0; base < plane; base += 64) {
        int cursor = 0;
        for (;;) {
            const int wl = base + lane;
            bool any = false;
            if (wl < plane && lane >= cursor) {
                const unsigned fgw = P.fg[wl];
                const unsigned left = (wl % P.wpr) ? P.fg[wl - 1] >> 31 : 0u;
                any = (fgw & ~((fgw << 1) | left) & ~P.mk[wl]) != 0u;
            }
            const unsigned long long has = __ballot(any);
            if (!has) break;
            const int L = __ffsll((long long)has) - 1, w = base + L, y = w / P.wpr, wi = w - y * P.wpr;
            cursor = L + 1;
            const unsigned fg = P.fg[w], carry = wi ? P.fg[w - 1] >> 31 : 0u;
            bool inside = false;                           // sign of the last marked pixel met on this row before the word
            for (int k = w - 1; k >= y * P.wpr; --k) {
                const unsigned mk = P.mk[k];
                if (mk) { inside = !((P.ng[k] >> (31 - __clz(mk))) & 1u); break; }
            }
            int pos = 0;                                   // bits below pos have been passed (their marks are in `inside`)
            for (;;) {
                const unsigned mk = P.mk[w], ng = P.ng[w], from = pos < 32 ? ~0u << pos : 0u;
                const unsigned starts = fg & ~((fg << 1) | carry) & ~mk & from;
                const int b = starts ? __ffs(starts) - 1 : 32;
                const unsigned passed = mk & from & (b < 32 ? (1u << b) - 1u : ~0u);      // marked pixels in [pos, b)
                if (passed) inside = !((ng >> (31 - __clz(passed))) & 1u);
                if (b == 32) break;
                pos = b + 1;
                if (inside) continue;                      // inside a hole of a traced component: not external
                const int len = contour_follow(P, wi * 32 + b, y, slab + (size_t)(1 - best_half) * cap * 2, cap, true);
                // which external border is kept (bf_contour_select): the last one the scan meets (0), the first (1), the longest (2)
                const bool keep = select == 0 ? true : (select == 1 ? best_len == 0 : len > best_len);
                if (keep) { best_len = len; best_half = 1 - best_half; }
                inside = !((P.ng[w] >> b) & 1u);           // the start pixel carries a mark now
            }
        }
    }
    if (lane == 0) { count[m] = best_len; count[gridDim.x + m] = best_half; }
}
