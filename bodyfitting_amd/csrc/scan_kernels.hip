// Scan closest-point search and the dense (all-vertex) reverse pass for gfx950.
//
//   bf_nearest_kernel     closest point on the scan surface per query vertex, uniform-grid pruned
//                         (replaces thirdparty/mesh_grid/mesh_grid_kernel.cu:239-353, incl. its
//                         per-triangle rule :12-109 with the argmin-coefficient edge fallback)
//   bf_pc_partial_kernel / bf_pc_grad_kernel
//                         point_cloud_loss_mesh_grid (smplify/loss.py:233-242): one Frobenius norm,
//                         scaled 5 * imsize / scan_height (smplify.py:206,210), and its gradient
//   bf_mesh_bwd_kernel / bf_ext_reduce_kernel
//                         reverse of the full-mesh forward: dL/dvertices -> dL/d(pose feature, chain
//                         matrices, betas, transl, scale), per 32-vertex tile then summed in tile order
//
// The grid itself (cell lists in CSR form, cells from MeshGridSearcher.set_mesh,
// utils/mesh_grid_searcher.py:56-79) is built once per scan by grid_kernels.hip, with triangles in face
// order inside every cell (the reference fills its lists with atomicCAS in arbitrary order).
#include <atomic>
#include "bf_internal.h"
#include "loss_bodies.h"
#include "joints_body.h"
#include "../../include/bodyfit.h"
#include <cstdlib>
#include "nearest_rule_ref.h"

namespace {

// closest point of triangle (p0,p1,p2), given relative to the query, by the reference's rule (mesh_grid_kernel.cu:12-109).
// Everything stays in registers: the edge the fallback picks is chosen with selects, not by indexing a local array.
__device__ __forceinline__ float closest_rule(const float *p0, const float *p1, const float *p2, float *coeff) {
    const float e1x = p1[0] - p0[0], e1y = p1[1] - p0[1], e1z = p1[2] - p0[2];
    const float e2x = p2[0] - p0[0], e2y = p2[1] - p0[1], e2z = p2[2] - p0[2];
    const float a11 = e1x * e1x + e1y * e1y + e1z * e1z;
    const float a12 = e1x * e2x + e1y * e2y + e1z * e2z;
    const float a22 = e2x * e2x + e2y * e2y + e2z * e2z;
    const float b1 = -(p0[0] * e1x + p0[1] * e1y + p0[2] * e1z);
    const float b2 = -(p0[0] * e2x + p0[1] * e2y + p0[2] * e2z);
    const float det = a11 * a22 - a12 * a12;
    const bool ok = det > 1e-12f * a11 * a22 && det > 0.f;
    int i;
    if (ok) {
        // (v_rcp_f32, 1 ulp, instead of two IEEE divisions: ~20 of the rule's ~110 instructions; the barycentrics are fp32-noisy
        //  at that level anyway - the in-plane solve loses digits with the triangle's aspect ratio)
        const float idet = __builtin_amdgcn_rcpf(det);
        const float u = (b1 * a22 - b2 * a12) * idet, v = (a11 * b2 - a12 * b1) * idet;
        const float c0 = 1.f - u - v;
        i = c0 > u ? 1 : 0;
        const float ci = i ? u : c0;
        i = ci > v ? 2 : i;                            // most negative coefficient (mesh_grid_kernel.cu:82-83)
        if ((i == 2 ? v : ci) >= 0.f) {
            coeff[0] = c0; coeff[1] = u; coeff[2] = v;
            const float x0 = c0 * p0[0] + u * p1[0] + v * p2[0];
            const float x1 = c0 * p0[1] + u * p1[1] + v * p2[1];
            const float x2 = c0 * p0[2] + u * p1[2] + v * p2[2];
            return x0 * x0 + x1 * x1 + x2 * x2;
        }
    } else {
        // degenerate triangle: the vertex opposite its longest edge (mesh_grid_kernel.cu:40-45)
        const float l0 = (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
        const float l1 = a22, l2 = a11;
        i = l0 < l1 ? 1 : 0;
        i = (i == 0 ? l0 : l1) < l2 ? 2 : i;
    }
    // edge (j, k) opposite corner i: i=0 -> (1,2), i=1 -> (2,0), i=2 -> (0,1)
    const bool i0 = i == 0, i1 = i == 1;
    const float pjx = i0 ? p1[0] : (i1 ? p2[0] : p0[0]), pjy = i0 ? p1[1] : (i1 ? p2[1] : p0[1]), pjz = i0 ? p1[2] : (i1 ? p2[2] : p0[2]);
    const float pkx = i0 ? p2[0] : (i1 ? p0[0] : p1[0]), pky = i0 ? p2[1] : (i1 ? p0[1] : p1[1]), pkz = i0 ? p2[2] : (i1 ? p0[2] : p1[2]);
    const float dx = pkx - pjx, dy = pky - pjy, dz = pkz - pjz;
    const float dd = dx * dx + dy * dy + dz * dz;
    const float t = dd > 0.f ? -(pjx * dx + pjy * dy + pjz * dz) * __builtin_amdgcn_rcpf(dd) : 0.5f;
    float cj = 1.f - t, ck = t;
    if (cj < 0.f) { cj = 0.f; ck = 1.f; }                 // same test order as the reference (:89-98)
    else if (ck < 0.f) { cj = 1.f; ck = 0.f; }
    coeff[0] = i0 ? 0.f : (i1 ? ck : cj);
    coeff[1] = i0 ? cj : (i1 ? 0.f : ck);
    coeff[2] = i0 ? ck : (i1 ? cj : 0.f);
    const float x0 = cj * pjx + ck * pkx, x1 = cj * pjy + ck * pky, x2 = cj * pjz + ck * pkz;
    return x0 * x0 + x1 * x1 + x2 * x2;
}

__device__ inline float blk_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

}  // namespace

// wave-wide minimum of a float / an int on the DPP path (all lanes get it)
__device__ inline float nn_wave_min_f(float v) {
    auto step = [&](int t) { v = fminf(v, __int_as_float(t)); };
    step(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xf, 0xf, false));
    step(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xf, 0xf, false));
    step(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x124, 0xf, 0xf, false));
    step(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x128, 0xf, 0xf, false));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}
__device__ inline int nn_wave_min_i(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// grid (ceil(n / 4), F), 256 threads: ONE WAVE PER QUERY.  points[F][n][3] -> face[F][n], pts[F][n][3], bary[F][n][3]
// (bary may be null).  The wave walks the expanding L-infinity shells of the uniform grid together: the cells of a shell
// are spread over the lanes (box lower bound, list bounds: one memory latency for the whole shell instead of one per
// cell), then the triangle lists of the surviving cells are laid end to end over the lanes (one packed 48-byte record per
// lane).  The result is the lexicographic (distance, face id) minimum, which is what the reference's "first strictly
// closer triangle in face order" amounts to, so it does not depend on how the work is split; a cell is only skipped
// when its box lies strictly beyond the best distance so far, and the reference's stop test `best < (L step)^2` is
// applied to the merged value after every shell.
// Shells 0 and 1 are one step over the 27-cell cube (the reference can never stop after shell 0: `best < 0`): the list
// bounds of all 27 cells are requested at once, the HOME cell's list is walked first, and only then are the 26
// neighbours pruned - with the distance the home cell gave.  `warm`: face[] still holds this query's answer of the
// previous call - a real candidate; it rides on the last lane of the home-cell pass (its vertices arrive while the
// cell bounds do), so the warm start costs no pass of its own.
//
// RULE selects the per-triangle arithmetic: BF_NEAREST_REFERENCE = the reference's own (nearest_rule_ref.h: Gram matrix, bordered
// KKT system, pivoted elimination with its absolute rank tests, IEEE divisions, nothing fused) - the default, face ids / coefficients /
// points equal to oracle/nearest_ref.c wherever no two faces tie bit for bit; BF_NEAREST_FAST = the 2 x 2 normal equations with
// v_rcp_f32 (same mathematics, ~half the instructions, other last bits: 4.7 % of config 5's queries then pick the other face of a
// shared edge - DESIGN 2.3).
// The reference rule's regular paths are straight-line code (nearest_rule_ref.h); a triangle on which the reference would take a rank
// decision - a scan has a handful of degenerate / collinear triangles (coincident vertices after the OBJ's four decimals), and every
// query that walks a cell holding one meets it: hundreds of evaluations per launch at config 5's size - goes through the general
// routines right there: a call on the lanes concerned, its systems in the wave's slot of an LDS array (a version with private arrays
// needed 104 registers and scratch memory; a second kernel doing such queries again cost 55 - 85 us per launch for a few hundred of
// them, one wave-latency each).
#ifndef BF_NN_PASSES
#define BF_NN_PASSES 3
#endif
constexpr int NN_PASSES = BF_NN_PASSES;   // passes of 64 list entries per trip through the screen, their loads in flight together (same box,
                                           // config 5 size, warm / moved 5 mm / cold us: 2 -> 117 / 165 / 208, 3 -> 114 / 156 / 216, 4 -> 121 / 166 / 234, 5 -> 122 / 169 / 240)
constexpr int NN_QCAP = 64 * (NN_PASSES + 1);   // a wave's queue of screened records
#ifdef BF_NEAREST_STATS
// diagnostic build (make CXXFLAGS+=-DBF_NEAREST_STATS): [0] queries [1] searches (1 + retries) [2] groups of cell lists [3] trips through
// the screen [4] screen passes [5] records screened [6] rule passes [7] records through the rule
__device__ unsigned long long bf_nearest_stats[8];
extern "C" int bf_nearest_stats_read(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(bf_nearest_stats), sizeof(bf_nearest_stats)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(bf_nearest_stats), z, sizeof z); }
    return 0;
}
#define NSTAT(i, v) do { if (lane == 0) atomicAdd(&bf_nearest_stats[i], (unsigned long long)(v)); } while (0)
#else
#define NSTAT(i, v) do { } while (0)
#endif
template <int RULE>
__device__ __forceinline__ void nearest_body(const ScanDev *__restrict__ scans, const float *__restrict__ points, int n,
                  int *face, float *pts, float *__restrict__ bary, int warm, int id, int f, float *scr, int *queue) {
    const int lane = threadIdx.x & 63;
    const ScanDev S = scans[f];
    const size_t o = (size_t)f * n + id;
    const float *q = points + o * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    // The query is wave-uniform and lives in scalar registers - and a vector instruction with a scalar operand issues at 4 cycles per
    // SIMD where the same instruction on vector registers issues at 2 beside the other waves' work (profiles/r06_issue_rate.md).
    // -DBF_NN_QVGPR subtracts it from a copy in vector registers in the per-lane arithmetic (box distances, the screen, the rule's
    // corners; what steers the walk stays scalar): 37 fewer slow-pipe instructions of 524 in the ISA - and 3 more registers, 82, one
    // more than six waves per SIMD allow.  Measured round 6, same box, exact hints / hints moved 5 mm: off 109.4 / 139.0 us, on at five
    // waves 119.1 / 152.4, on and held to 80 registers 114.8 / 147.1, on with two screen passes per trip (73 registers) 111.6 / 140.4.  Off.
    float qvx = qx, qvy = qy, qvz = qz;
#ifdef BF_NN_QVGPR
    asm volatile("" : "+v"(qvx), "+v"(qvy), "+v"(qvz));
#endif
    int cx = (int)floorf((qx - S.ox) / S.step), cy = (int)floorf((qy - S.oy) / S.step), cz = (int)floorf((qz - S.oz) / S.step);
    cx = min(max(cx, 0), S.nx - 1); cy = min(max(cy, 0), S.ny - 1); cz = min(max(cz, 0), S.nz - 1);
    // the reference's shell limit (mesh_grid_kernel.cu:254-257, 262): per axis x > size - x ? x : size - x, shells L < that - on the
    // side where the home cell is past the middle this is one short of the far wall, and a query metres outside the grid (the only
    // kind that walks that far) never sees the last layer of cells.  Kept.
    const int maxL = max(max(cx > S.nx - cx ? cx : S.nx - cx, cy > S.ny - cy ? cy : S.ny - cy), cz > S.nz - cz ? cz : S.nz - cz) - 1;
    // scr: this lane's slot of the wave's [20][LANES] systems of the general routines; queue: the wave's NN_QCAP records that passed the
    // screen and wait for the rule (both in LDS, the kernel's)
    constexpr int PASSES = NN_PASSES;
    // `warm`: pts[] still holds this query's nearest point of the previous call: its squared distance from the query as it is now,
    // a shade enlarged, is (almost always) an upper bound of the answer.  It is only a guess - the search below is run with it and
    // CHECKED against what it found (the last lines of the loop).
    float U = 3.0e38f;
    if (warm) {
        const float ux = pts[o * 3] - qx, uy = pts[o * 3 + 1] - qy, uz = pts[o * 3 + 2] - qz;
        // (the slack covers what the rule's own value may lie above the true distance of a regular triangle, also for a query ON the
        //  surface; a NaN is no guess)
        const float u2 = (ux * ux + uy * uy + uz * uz) * 1.01f + 1e-7f;
        U = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(u2 < 3.0e38f ? u2 : 3.0e38f)));
    }
    // ---- shells 0 and 1 are one step over the 27-cell cube, lane c = (dz+1)*9 + (dy+1)*3 + (dx+1), the home cell is lane 13 (the
    // reference can never stop after shell 0: `best < 0`): list bounds and box distances of all 27 cells, once
    int st, cnt;
    float d2 = 0.f;
    bool cell_ok;
    {
        const int dz = lane / 9 - 1, rem = lane % 9, dy = rem / 3 - 1, dx = rem % 3 - 1;
        const int x = cx + dx, y = cy + dy, z = cz + dz;
        cell_ok = lane < 27 && x >= 0 && x < S.nx && y >= 0 && y < S.ny && z >= 0 && z < S.nz && (maxL >= 1 || lane == 13);
        st = 0; cnt = 0;
        if (cell_ok) {
            const int cell = (x * S.ny + y) * S.nz + z;
            st = S.cell_start[cell];
            cnt = S.cell_start[cell + 1] - st;
            float lo, e;
            lo = S.ox + S.step * x; e = qvx < lo ? lo - qvx : (qvx > lo + S.step ? qvx - lo - S.step : 0.f); d2 += e * e;
            lo = S.oy + S.step * y; e = qvy < lo ? lo - qvy : (qvy > lo + S.step ? qvy - lo - S.step : 0.f); d2 += e * e;
            lo = S.oz + S.step * z; e = qvz < lo ? lo - qvz : (qvz > lo + S.step ? qvz - lo - S.step : 0.f); d2 += e * e;
        }
        cell_ok = cell_ok && cnt > 0;
    }
    const int st27 = st, cnt27 = cnt;
    // this lane's best so far (distance, face, coefficients); merged at the end
    float best, bc0, bc1, bc2, gbest;
    int bface, brec;                                          // (brec: the best triangle's record, for the result's corners)
    NSTAT(0, 1);
    for (;;) {
        NSTAT(1, 1);
        best = 3.0e38f; bc0 = 0.f; bc1 = 0.f; bc2 = 0.f;
        bface = 0x7fffffff; brec = 0;
        gbest = 3.0e38f;                                    // (wave-uniform) best distance the rule has returned so far
        float B = U;                                        // (wave-uniform) min(U, gbest): what cells and records are screened against
        int qn = 0;                                         // (wave-uniform) records in the queue
        // The walk is a nest of loops with ONE copy of the screen and ONE of the rule (inlined per call site they took the kernel from
        // 69 to 202 registers): cell sets (the home cell / the cube / 64 cells of a shell) > groups of up to six cell lists laid end to
        // end > trips of PASSES x 64 list entries through the screen, each followed by the rule on the queue when it holds 64 records -
        // or, on the extra trip that ends a cell set's last group, whatever it holds (`force`).
        // (a guess of a cell's size or more prunes none of the 27 cells: the home cell goes first then too - its records screened
        //  against the guess - and what it gives prunes the others)
        const bool cold = !(U < 3.0e38f);
        int phase = U < S.step * S.step ? 1 : 0, L = 1, chunk = 0;
        st = st27; cnt = cnt27;
        for (;;) {
            // ---- the next cell set: lanes whose (st, cnt) lists are to be walked, and whether the queue is emptied after it
            unsigned long long cells;
            bool flush_after = true;
            if (phase == 0) cells = __ballot(cell_ok && lane == 13);                                   // no bound yet: the HOME cell's list first, through the rule ...
            else if (phase == 1) cells = __ballot(cell_ok && (U < S.step * S.step || lane != 13) && !(B < d2));      // ... and the cube's other cells pruned with what it gave (or with U)
            else {
                // shell L: 64 cells of its cube per set; only the shell (max |d| == L), in bounds, not beyond the bound
                const int side = 2 * L + 1, ncube = side * side * side;
                const int c = chunk * 64 + lane;
                const int dz = c / (side * side) - L, rem = c % (side * side), dy = rem / side - L, dx = rem % side - L;
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                bool use = c < ncube && max(max(abs(dx), abs(dy)), abs(dz)) == L && x >= 0 && x < S.nx && y >= 0 && y < S.ny && z >= 0 && z < S.nz;
                st = 0; cnt = 0;
                if (use) {
                    float lo, e, dd = 0.f;
                    lo = S.ox + S.step * x; e = qvx < lo ? lo - qvx : (qvx > lo + S.step ? qvx - lo - S.step : 0.f); dd += e * e;
                    lo = S.oy + S.step * y; e = qvy < lo ? lo - qvy : (qvy > lo + S.step ? qvy - lo - S.step : 0.f); dd += e * e;
                    lo = S.oz + S.step * z; e = qvz < lo ? lo - qvz : (qvz > lo + S.step ? qvz - lo - S.step : 0.f); dd += e * e;
                    use = !(B < dd);
                    if (use) {
                        const int cell = (x * S.ny + y) * S.nz + z;
                        st = S.cell_start[cell];
                        cnt = S.cell_start[cell + 1] - st;
                    }
                }
                cells = __ballot(use && cnt > 0);
                flush_after = (chunk + 1) * 64 >= ncube;
            }
            do {
                // up to six lists end to end: their bounds once, as wave-uniform values (base[c] = start - offset of list c), so that the
                // entry -> record map of a pass is compare / select steps instead of a readlane pair per cell and pass
                int base[6], off[7];
                off[0] = 0;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int src = cells ? __ffsll((long long)cells) - 1 : 0;
                    const int n0 = cells ? __builtin_amdgcn_readlane(cnt, src) : 0;
                    base[c] = __builtin_amdgcn_readlane(st, src) - off[c];
                    off[c + 1] = off[c] + n0;
                    cells &= cells - 1;
                }
                const int total = off[6];
                int nlists = 1;
#pragma unroll
                for (int c = 1; c < 6; ++c) nlists += off[c + 1] > off[c] ? 1 : 0;
                NSTAT(2, 1);
                for (int e0 = 0;; e0 += 64 * PASSES) {
                    const int npass = min(PASSES, (total - e0 + 63) >> 6);       // (<= 0 on the extra trip)
                    if (npass > 0) {
                        NSTAT(3, 1); NSTAT(4, npass);
                        int rec[PASSES];
                        float4 lo[PASSES], hi[PASSES];
#pragma unroll
                        for (int k = 0; k < PASSES; ++k)
                            if (k < npass) {                // (wave-uniform; every lane loads: one without an entry reads the group's first record)
                                const int e = e0 + k * 64 + lane;
                                int d = base[0];
                                // (wave-uniform BRANCHES, not selects - round 6: flattened by the compiler, the chain was a compare, two
                                //  selects and a move for each of the five possible further lists in every pass, whatever the group holds;
                                //  the empty asm keeps the blocks from being if-converted again.  Same box, exact hints 110.5 -> 107.4 us)
                                if (nlists > 1) { asm volatile(""); d = e >= off[1] ? base[1] : d;
                                    if (nlists > 2) { asm volatile(""); d = e >= off[2] ? base[2] : d;
                                        if (nlists > 3) { asm volatile(""); d = e >= off[3] ? base[3] : d;
                                            if (nlists > 4) { asm volatile(""); d = e >= off[4] ? base[4] : d;
                                                if (nlists > 5) { asm volatile(""); d = e >= off[5] ? base[5] : d; } } } } }
                                rec[k] = e < total ? e + d : -1;
                                const size_t r = (size_t)(e < total ? e + d : base[0]) * 2;
                                lo[k] = S.cell_box[r]; hi[k] = S.cell_box[r + 1];
                            }
                        // THE SCREEN.  A record goes to the rule only if its triangle's bounding box is not beyond B.  What makes that exact:
                        // the distance the reference's rule returns (the multiplier of its KKT system) is never below the true squared
                        // distance by more than 2e-7 x the largest squared corner distance - for every shape, needles and coincident corners
                        // included, where it may be far ABOVE it or NaN, and a NaN never wins a `<` (oracle/nearest_ref.c over millions of
                        // pairs: tests/test_nearest_ref_oracle.py) - and the distance to the box is a lower bound of the true
                        // one.  With lb2 = squared distance to the box and fb2 = to its far corner, lb2 * 0.999 - 1e-5 * fb2 > B  =>  the
                        // rule's value for this triangle is > B >= the final minimum: it can neither win nor tie.  The far corner is within
                        // a box diagonal of the near point, fb2 <= 2 lb2 + 2 diag^2, so the test is made with the record's own margin
                        // m = 2.1e-5 diag^2 (grid_kernels.hip): lb2 * 0.998 - m > B.
#pragma unroll
                        for (int k = 0; k < PASSES; ++k)
                            if (k < npass) {
                                const float lx = lo[k].x - qvx, ly = lo[k].y - qvy, lz = lo[k].z - qvz, hx = hi[k].x - qvx, hy = hi[k].y - qvy, hz = hi[k].z - qvz;
                                const float ex = fmaxf(fmaxf(lx, -hx), 0.f), ey = fmaxf(fmaxf(ly, -hy), 0.f), ez = fmaxf(fmaxf(lz, -hz), 0.f);
                                const float lb2 = ex * ex + ey * ey + ez * ez;
                                const bool pass = rec[k] >= 0 && !(lb2 * 0.998f - lo[k].w > B);
                                const unsigned long long m = __ballot(pass);
                                if (pass) queue[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = rec[k];
                                qn += __popcll(m);
                            }
                    }
                    const bool force = npass <= 0 && !cells && flush_after;
                    while (qn >= 64 || (force && qn > 0)) {
                        // THE RULE on the first (up to 64) records of the queue
                        const int count = min(qn, 64);
                        NSTAT(6, 1); NSTAT(7, count);
                        __builtin_amdgcn_wave_barrier();
                        if (lane < count) {
                            const int rq = queue[lane];
                            const size_t r = (size_t)rq * 3;
                            const float4 q0 = S.cell_pack[r], q1 = S.cell_pack[r + 1], q2 = S.cell_pack[r + 2];
                            const float p[9] = {q0.x - qvx, q0.y - qvy, q0.z - qvz, q0.w - qvx, q1.x - qvy, q1.y - qvz, q1.z - qvx, q1.w - qvy, q2.x - qvz};
                            const int t = __float_as_int(q2.y);
                            float co[3];
                            float dist;
                            if (RULE == BF_NEAREST_FAST) dist = closest_rule(p, p + 3, p + 6, co);
                            else dist = nrule::nearest_proj_general(p, co, scr);
                            if (dist < best || (dist == best && t < bface)) { best = dist; bface = t; brec = rq; bc0 = co[0]; bc1 = co[1]; bc2 = co[2]; }
                        }
                        gbest = fminf(gbest, nn_wave_min_f(best));
                        B = fminf(U, gbest);
                        __builtin_amdgcn_wave_barrier();
                        if (qn > 64) {                      // what is left moves to the front: up to PASSES x 64 records
                            int rest[PASSES];
#pragma unroll
                            for (int k = 0; k < PASSES; ++k) rest[k] = 64 * (k + 1) + lane < qn ? queue[64 * (k + 1) + lane] : 0;
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int k = 0; k < PASSES; ++k)
                                if (64 * (k + 1) + lane < qn) queue[64 * k + lane] = rest[k];
                        }
                        qn = max(qn - 64, 0);
                    }
                    if (npass <= 0) break;
                }
            } while (cells);
            // ---- the set is done (and, if it ended a shell, the queue empty): the stop tests
            if (phase == 0) phase = 1;
            else if (phase == 1) {
                // mesh_grid_kernel.cu:349 after shell 1 (gbest = 3e38 while nothing was found).  A guess below (L step)^2 that the shells up
                // to L have not confirmed is wrong: the triangle behind a right one has its nearest point within L cells, so in one of the
                // cells walked so far, and would have passed the screen.
                if (gbest < S.step * S.step || maxL < 2 || U < S.step * S.step) break;
                phase = 2; L = 2; chunk = 0;
            } else if (flush_after) {
                if (gbest < (float)L * (float)L * S.step * S.step || L >= maxL || U < (float)L * (float)L * S.step * S.step) break;     // mesh_grid_kernel.cu:349
                ++L; chunk = 0;
            } else ++chunk;
        }
        // THE CHECK of the guess U: everything skipped had a rule value above min(U, the best at that time).  If the walk ended with
        // gbest <= U, that is above the final minimum, and so was the value of every stop test - the answer is the unscreened walk's.
        // If not (the guess was not a point of the surface, or the rule's value for its triangle lies further above the true distance
        // than the slack), once more without the guess.
        if (cold || gbest <= U) break;
        U = 3.0e38f;
    }
    // merge: the lexicographic (distance, face id) minimum over the lanes, then its owner's coefficients
    const float dmin = gbest;
    const int fmin_ = nn_wave_min_i(best == dmin ? bface : 0x7fffffff);
    const unsigned long long own = __ballot(best == dmin && bface == fmin_);
    const int wl = __ffsll((long long)own) - 1;
    const bool found = fmin_ != 0x7fffffff && wl >= 0;
    const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bc0), wl < 0 ? 0 : wl));
    const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bc1), wl < 0 ? 0 : wl));
    const float w2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bc2), wl < 0 ? 0 : wl));
    // the winner's corners: its record (a copy of the three vertices), found by its index - one round trip of wave-uniform loads
    // instead of the face -> vertex indices -> vertices chase
    const size_t wr = (size_t)__builtin_amdgcn_readlane(brec, wl < 0 ? 0 : wl) * 3;
    // (no early return for the other lanes: inlined into a loop it becomes a `continue`, and the lanes then run ahead of lane 0)
    if (lane == 0) {
        face[o] = found ? fmin_ : -1;
        float r0 = qx, r1 = qy, r2 = qz;
        if (found) {
            const float4 q0 = S.cell_pack[wr], q1 = S.cell_pack[wr + 1], q2 = S.cell_pack[wr + 2];
            const float v0[3] = {q0.x, q0.y, q0.z}, v1[3] = {q0.w, q1.x, q1.y}, v2[3] = {q1.z, q1.w, q2.x};
            // proj = q + sum c_i (v_i - q), as the reference forms it (:318-329)
            if (RULE != BF_NEAREST_FAST) {                     // ... every product and sum rounded, left to right
                r0 = nrule::project(qx, w0, v0[0] - qx, w1, v1[0] - qx, w2, v2[0] - qx);
                r1 = nrule::project(qy, w0, v0[1] - qy, w1, v1[1] - qy, w2, v2[1] - qy);
                r2 = nrule::project(qz, w0, v0[2] - qz, w1, v1[2] - qz, w2, v2[2] - qz);
            } else {
                r0 = qx + w0 * (v0[0] - qx) + w1 * (v1[0] - qx) + w2 * (v2[0] - qx);
                r1 = qy + w0 * (v0[1] - qy) + w1 * (v1[1] - qy) + w2 * (v2[1] - qy);
                r2 = qz + w0 * (v0[2] - qz) + w1 * (v1[2] - qz) + w2 * (v2[2] - qz);
            }
        }
        pts[o * 3] = r0; pts[o * 3 + 1] = r1; pts[o * 3 + 2] = r2;
        if (bary) { bary[o * 3] = found ? w0 : 0.f; bary[o * 3 + 1] = found ? w1 : 0.f; bary[o * 3 + 2] = found ? w2 : 0.f; }
    }
}

// A launch covers F frames, each with its own scan: the grid is ONE-dimensional and frame-MINOR - workgroup l works on frame l % F, block
// l / F of it.  Workgroups go to the eight XCDs round-robin by their linear id, so with F = 8 (config 5's shard; any F that divides 8 or
// is a multiple of it) a frame's workgroups all land on one XCD and its scan's records (4 + 6 MB) are that XCD's L2 contents instead of
// one eighth of every scan's.
// Waves (= queries) per workgroup.  ONE since round 5: a workgroup's slots - its LDS above all - are held until its slowest wave is done,
// and the queries of a workgroup take very different times (a cold or far query walks shells, a warm one a few cells).  Same box,
// config 5 size, us per launch (exact hints / hints moved 5 mm / cold): 8 waves 125 / 179 / 240, 4 waves 113 / 158 / 222, 2 waves
// 110 / 145 / 216, 1 wave 107 / 138 / 210.
#ifndef NN_WAVES
#define NN_WAVES 1
#endif
#define NN_FRAME(F) ((int)(blockIdx.x % (unsigned)(F)))
#define NN_BLOCK(F) ((int)(blockIdx.x / (unsigned)(F)))
#define NN_WAVE_LDS(RULE)                                                                                                               \
    __shared__ float s_general[RULE == BF_NEAREST_REFERENCE ? NN_WAVES * 20 * nrule::LANES : 1];                                                \
    __shared__ int s_queue[NN_WAVES * NN_QCAP];                                                                                                 \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   /* (told to the compiler as the wave-uniform value it is: the walk's bookkeeping then lives in scalar registers) */ \
    float *scr = s_general + (RULE == BF_NEAREST_REFERENCE ? wave * 20 * nrule::LANES + (threadIdx.x & 63) % nrule::LANES : 0);                          \
    int *queue = s_queue + wave * NN_QCAP;

#ifdef BF_NN_WPE
#define NN_OCC __attribute__((amdgpu_waves_per_eu(BF_NN_WPE, BF_NN_WPE)))
#else
#define NN_OCC
#endif
extern "C" __global__ void __launch_bounds__(64 * NN_WAVES) NN_OCC
bf_nearest_kernel(const ScanDev *__restrict__ scans, const float *__restrict__ points, int n,
                  int *face, float *pts, float *__restrict__ bary, int warm, int n_frames) {
    NN_WAVE_LDS(BF_NEAREST_REFERENCE)
    const int id = NN_BLOCK(n_frames) * NN_WAVES + wave;
    if (id >= n) return;
    nearest_body<BF_NEAREST_REFERENCE>(scans, points, n, face, pts, bary, warm, id, NN_FRAME(n_frames), scr, queue);
}
extern "C" __global__ void __launch_bounds__(64 * NN_WAVES)
bf_nearest_fast_kernel(const ScanDev *__restrict__ scans, const float *__restrict__ points, int n,
                       int *face, float *pts, float *__restrict__ bary, int warm, int n_frames) {
    NN_WAVE_LDS(BF_NEAREST_FAST)
    const int id = NN_BLOCK(n_frames) * NN_WAVES + wave;
    if (id >= n) return;
    nearest_body<BF_NEAREST_FAST>(scans, points, n, face, pts, bary, warm, id, NN_FRAME(n_frames), scr, queue);
}

// nearest_rule_ref.h's division helper and per-triangle rule, exposed for the tests (bf_nearest_selftest_*)
extern "C" __global__ void bf_nearest_quot_kernel(int n, const float *__restrict__ num, const float *__restrict__ den, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = nrule::quot(num[i], nrule::recip(den[i]));
}
// patches[n][9] (corners relative to the query) -> dist[n], coeff[n][3]; general != 0: with the general routines behind the regular
// paths (what the pair of kernels computes), 0: the regular paths alone (-1 where they decline)
extern "C" __global__ void bf_nearest_rule_kernel(int n, const float *__restrict__ patches, float *__restrict__ dist, float *__restrict__ coeff, int general) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    __shared__ float s_general[20 * nrule::LANES];
    float p[9], c[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = patches[(size_t)i * 9 + k];
    dist[i] = general ? nrule::nearest_proj_general(p, c, s_general + threadIdx.x % nrule::LANES) : nrule::nearest_proj_regular(p, c);
    coeff[i * 3] = c[0]; coeff[i * 3 + 1] = c[1]; coeff[i * 3 + 2] = c[2];
}

// which arithmetic bf_nearest_launch uses: BF_NEAREST_RULE=fast|reference in the environment at first use, or bf_nearest_rule_set()
// (read by bf_group's worker threads side by side: an atomic, initialised from the environment exactly once)
static std::atomic<int> &nearest_rule_cell() {
    static std::atomic<int> cell([] {
        const char *e = getenv("BF_NEAREST_RULE");
        return (e && (e[0] == 'f' || e[0] == '1')) ? BF_NEAREST_FAST : BF_NEAREST_REFERENCE;
    }());
    return cell;
}
extern "C" int bf_nearest_rule_get(void) { return nearest_rule_cell().load(std::memory_order_relaxed); }
extern "C" int bf_nearest_rule_set(int rule) {
    if (rule != BF_NEAREST_REFERENCE && rule != BF_NEAREST_FAST) return -1;
    nearest_rule_cell().store(rule, std::memory_order_relaxed);
    return 0;
}
extern "C" void bf_nearest_launch(dim3 grid, hipStream_t stream, const ScanDev *scans, const float *points, int n, int *face, float *pts,
                                  float *bary, int warm) {
    // (grid = (ceil(n / 4), frames) as the callers think of it: one query per wave; launched one-dimensional and frame-minor, NN_FRAME)
    const int F = (int)grid.y;
    const dim3 sgrid((unsigned)((n + NN_WAVES - 1) / NN_WAVES) * F);
    if (bf_nearest_rule_get() == BF_NEAREST_FAST)
        hipLaunchKernelGGL(bf_nearest_fast_kernel, sgrid, dim3(64 * NN_WAVES), 0, stream, scans, points, n, face, pts, bary, warm, F);
    else
        hipLaunchKernelGGL(bf_nearest_kernel, sgrid, dim3(64 * NN_WAVES), 0, stream, scans, points, n, face, pts, bary, warm, F);
}

// grid (nblk, F): partial[f][blk] = sum over this block's vertices of |P - C|^2
extern "C" __global__ void __launch_bounds__(256)
bf_pc_partial_kernel(const float *__restrict__ P, const float *__restrict__ C, int n, float *__restrict__ partial) {
    __shared__ float s[4];
    const int id = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    float a = 0.f;
    if (id < n) {
        const float *p = P + ((size_t)f * n + id) * 3, *c = C + ((size_t)f * n + id) * 3;
        float d0 = p[0] - c[0], d1 = p[1] - c[1], d2 = p[2] - c[2];
        a = d0 * d0 + d1 * d1 + d2 * d2;
    }
    a = blk_wave_sum(a);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)f * gridDim.x + blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

// grid (nblk, F): dvout = weight[f] * (P - C) / ||P - C||_F ;  weight[f] = 5 * imsize / scan_height
// `accumulate` != 0 adds to dvout instead of overwriting it.  loss[f] = weight[f] * norm (block 0 writes it).
extern "C" __global__ void __launch_bounds__(256)
bf_pc_grad_kernel(const float *__restrict__ P, const float *__restrict__ C, int n, const float *__restrict__ partial,
                  const float *__restrict__ weight, float *__restrict__ dvout, float *__restrict__ loss, int accumulate,
                  int *door, int door_target) {
    const int id = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    float tot = 0.f;
    for (int b = 0; b < (int)gridDim.x; ++b) tot += partial[(size_t)f * gridDim.x + b];      // fixed order
    const float norm = sqrtf(tot), w = weight[f];
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss) loss[f] = w * norm;
    // (door: the keypoint workgroups whose dL/dvertices this kernel adds onto ran on the second stream beside the search and count
    //  themselves off in door[BF_DOOR_KP] - they are done ~100 us before this kernel starts; the stream-level join that used to sit
    //  in front of this launch was a wait packet of ~7 us per iteration.  A wait that runs into its time limit raises the door's
    //  error flag and the call fails, as for the other doorbells.)
    if (door) {
        if (threadIdx.x == 0) (void)bf_door_wait(door, BF_DOOR_KP, door_target);
        __syncthreads();
    }
    if (id >= n) return;
    const size_t o = ((size_t)f * n + id) * 3;
    const float k = w / norm;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float g = k * (P[o + c] - C[o + c]);
        dvout[o + c] = accumulate ? dvout[o + c] + g : g;
    }
}

// Reverse of the full-mesh forward for one 32-vertex tile and up to FPW frames.  grid (n_tiles, ceil(F / FPW)), 512 threads.
//   in : dvout[F][NV][3] = dL/d((v + t) s c), vposed[F][NV][3] (pose-blended vertices saved by the forward),
//        vraw[F][NV][3], state
//   out: part[F][n_tiles][EXT] with EXT = npf + nj*12 + nb + 4:
//        dfeat[npf] | per joint 3 rows of (sum w dv (x) vp | sum w dv) | dbeta[nb] | dt[3] ds[1]
// posedirsT is the [3NV][npf] transpose, so that thread p streams row-contiguous memory.
// The tile's posedirsT slice (96 x npf floats - the 61 MB of SMPL-X spread over 328 tiles) is streamed ONCE and every loaded value
// feeds one fma per frame; the per-frame vectors it is dotted with sit frame-minor in LDS (one b128 read for four
// frames).  shapedirs of the tile are staged in LDS, so the beta sums do not chase 96 dependent loads.  Every sum has a
// fixed order that does not depend on FPW or on the frame's position in the batch.
template <int FPW, bool FOLD>
__global__ void __launch_bounds__(512)
bf_mesh_bwd_multi_kernel(MeshTab M, const float *__restrict__ posedirsT, const float *__restrict__ state, int n_frames,
                         const float *__restrict__ dvout, const float *__restrict__ vposed, const float *__restrict__ vraw,
                         float *__restrict__ part, const float *__restrict__ gpart, int n_masks, int n_sampled, int samp_stride, int split, MaskFold G) {
    // split = 2 (one frame per workgroup, the grid smaller than half the machine): TWO workgroups per tile, each streaming half of the
    // tile's posedirsT columns in (a) - that stream is a chain of dependent batches at ~1.3 us each (a kernel starts with cold caches),
    // and more loads in flight per CU made it slower, more CUs do not.  Each half leaves a partial row of its own (the reduction adds the
    // rows in order); half 0 also does (b)-(d), half 1 writes zeros there.
    constexpr int TV = BF_MESH_TILE, COLS = BF_MESH_TILE * 3;
    extern __shared__ __align__(16) float sm[];
    const int nj = M.nj, nb = M.nb, npf = M.npf, nv = M.nv;
    float *s_dvpT = sm;                      // [COLS][FPW]  T_v.R^T dv, frame-minor
    float *s_A = s_dvpT + COLS * FPW;        // [FPW][nj][12]
    float *s_w = s_A + FPW * nj * 12;        // [TV][nj]
    float *s_dv = s_w + TV * nj;             // [FPW][COLS]  dL/dv (model space)
    float *s_vp = s_dv + FPW * COLS;         // [FPW][COLS]
    float *s_sim = s_vp + FPW * COLS;        // [FPW][8]: t[3], s, c
    float *s_sd = s_sim + FPW * 8;           // [COLS][nb]
    float *s_ts = s_sd + COLS * nb;          // [FPW][2][COLS]  dt, ds terms
    const int tile = blockIdx.x / split, half = blockIdx.x - tile * split, fbase = blockIdx.y * FPW, nf = min(FPW, n_frames - fbase);
    const int tid = threadIdx.x, v0 = tile * TV;
    const size_t sstride = bf_state_stride(nj, npf, nb);
    const int nvt = min(TV, nv - v0);
    for (int i = tid; i < FPW * nj * 12; i += 512) {
        const int f = i % FPW, r = i / FPW, j = r / 12, e = r - j * 12, a = e >> 2, b = e & 3;       // (FPW, 12: compile-time divisors)
        float x = 0.f;
        if (f < nf) {
            StateView st = bf_state_view(const_cast<float *>(state) + (fbase + f) * sstride, nj, npf, nb);
            x = b < 3 ? st.GR[j * 9 + a * 3 + b] : st.At[j * 3 + a];
        }
        s_A[f * nj * 12 + r] = x;
    }
    for (int i = tid; i < TV * nj; i += 512) s_w[i] = i < nvt * nj ? M.lbs_weights[(size_t)v0 * nj + i] : 0.f;
    for (int i = tid; i < COLS * nb; i += 512) s_sd[i] = i < nvt * 3 * nb ? M.shapedirs[(size_t)v0 * 3 * nb + i] : 0.f;
    if (tid < FPW * 8) {
        const int f = tid >> 3, l = tid & 7;
        float x = 0.f;
        if (f < nf && l < 5) {
            StateView st = bf_state_view(const_cast<float *>(state) + (fbase + f) * sstride, nj, npf, nb);
            x = l < 3 ? st.t[l] : st.sc[l - 3];
        }
        s_sim[tid] = x;
    }
    __syncthreads();
    for (int i = tid; i < FPW * COLS; i += 512) {
        const int f = i / COLS, c = i - f * COLS;
        const bool ok = f < nf && c < nvt * 3;
        const size_t o = ((size_t)(fbase + f) * nv + v0) * 3 + c;
        const float sc = s_sim[f * 8 + 3] * s_sim[f * 8 + 4];
        float g = ok ? dvout[o] : 0.f;
        if (gpart && ok && (samp_stride == 4 ? ((v0 + c / 3) & 3) == 0 : v0 + c / 3 < n_sampled)) {
            // the silhouette gradient of every 4th vertex (loss.py:99) is still per mask view: add the views here, in view order,
            // exactly as bf_mask_gsum_kernel would have (sum of the views first, then onto dL/dvertex) - one launch less
            const int sidx = samp_stride == 4 ? (v0 + c / 3) >> 2 : v0 + c / 3;
            float gs = 0.f;
            for (int m = 0; m < n_masks; ++m) gs += gpart[(((size_t)(fbase + f) * n_masks + m) * n_sampled + sidx) * 3 + c % 3];
            g += gs;
        } else if (FOLD && G.acc && ok && (samp_stride == 4 ? ((v0 + c / 3) & 3) == 0 : v0 + c / 3 < n_sampled)) {      // (FOLD: an instance of its own - with this path compiled into it the eight-frame instance of config 5 went from 27 to 35 us)
            // the same sum from the contour scan's fixed-point sums: bf_mask_gather_kernel's closing step per view (binary term + contour
            // term back through the projection), the views added in view order
            const int sidx = samp_stride == 4 ? (v0 + c / 3) >> 2 : v0 + c / 3, k = c % 3;
            float gs = 0.f;
            // (eight views at a time: every view's loads are requested before the first is used - a rolled loop paid one memory round trip
            //  per view, ~3 us of this launch; the views are still added in view order)
            constexpr int MB = 8;
            for (int m0 = 0; m0 < n_masks; m0 += MB) {
                unsigned long long au[MB], av[MB];
                float bu[MB], bv[MB], pk[MB][3];
                float4 rr[MB];
#pragma unroll
                for (int e = 0; e < MB; ++e) {
                    const int m = min(m0 + e, n_masks - 1);
                    const size_t o = ((size_t)(fbase + f) * n_masks + m) * n_sampled + sidx;
                    au[e] = G.acc[o * 2]; av[e] = G.acc[o * 2 + 1];
                    bu[e] = G.duvb[o * 2]; bv[e] = G.duvb[o * 2 + 1];
                    rr[e] = ((const float4 *)G.uvi)[o];
                    const float *P = G.proj + ((size_t)(fbase + f) * G.n_views + G.view_index[m]) * 12;
                    pk[e][0] = P[k]; pk[e][1] = P[4 + k]; pk[e][2] = P[8 + k];
                }
#pragma unroll
                for (int e = 0; e < MB; ++e) {
                    if (m0 + e < n_masks) {
                        const float tu = bu[e] + bf_acc_float(au[e]), tv = bv[e] + bf_acc_float(av[e]);
                        const float4 r = rr[e];
                        const float q0 = tu * r.w, q1 = tv * r.w, q2 = -(tu * r.x + tv * r.y) * r.w;
                        gs += pk[e][0] * q0 + pk[e][1] * q1 + pk[e][2] * q2;
                    }
                }
            }
            g += gs;
        }
        s_dv[i] = g * sc;
        s_vp[i] = ok ? vposed[o] : 0.f;
        s_ts[(f * 2) * COLS + c] = g * sc;                                                          // d/dt_k = sum dvout * s c
        s_ts[(f * 2 + 1) * COLS + c] = ok ? g * (vraw[o] + s_sim[f * 8 + c % 3]) * s_sim[f * 8 + 4] : 0.f;   // d/ds = sum dvout . (v + t) c
    }
    __syncthreads();
    // dvp = T_v.R^T dv : item (f, vl), its three components together; with sparse skinning rows (MeshTab::v_nnz = 4 or 8
    // non-zero weights per vertex, the rest exact zeros) only those joints are visited - 4 instead of 55 for SMPL-X
    for (int i = tid; i < FPW * TV; i += 512) {
        const int f = i / TV, vl = i - f * TV;
        const float *A = s_A + f * nj * 12, *dv = s_dv + f * COLS + vl * 3;
        const float d0 = dv[0], d1 = dv[1], d2 = dv[2];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        auto add = [&](int j, float w) {
            const float4 r0 = *(const float4 *)(A + j * 12), r1 = *(const float4 *)(A + j * 12 + 4), r2 = *(const float4 *)(A + j * 12 + 8);
            a0 += w * (r0.x * d0 + r1.x * d1 + r2.x * d2);
            a1 += w * (r0.y * d0 + r1.y * d1 + r2.y * d2);
            a2 += w * (r0.z * d0 + r1.z * d1 + r2.z * d2);
        };
        if (M.v_nnz) {
            const int nnz = M.v_nnz;
            if (vl < nvt)
                for (int q = 0; q < nnz; ++q) {
                    const float w = M.v_nzw[(size_t)(v0 + vl) * nnz + q];
                    if (w != 0.f) add(M.v_nzj[(size_t)(v0 + vl) * nnz + q], w);
                }
        } else {
            for (int j = 0; j < nj; ++j) add(j, s_w[vl * nj + j]);
        }
        s_dvpT[(vl * 3) * FPW + f] = a0; s_dvpT[(vl * 3 + 1) * FPW + f] = a1; s_dvpT[(vl * 3 + 2) * FPW + f] = a2;
    }
    __syncthreads();
    const int EXT = npf + nj * 12 + nb + 4;
    float *out0 = part + ((size_t)fbase * gridDim.x + blockIdx.x) * EXT;
    const int ca = half * (COLS / split), cb = ca + COLS / split;         // this workgroup's columns of (a)
    const size_t fstride = (size_t)gridDim.x * EXT;
    // (a) dfeat partials: thread p streams posedirsT[col][p] once for all frames
    for (int p = tid; p < npf; p += 512) {
        const float *src = posedirsT + (size_t)v0 * 3 * npf + p;
        float acc[FPW];
#pragma unroll
        for (int f = 0; f < FPW; ++f) acc[f] = 0.f;
        for (int c0 = ca; c0 < cb; c0 += 16) {
            float x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = c0 + i < nvt * 3 ? src[(size_t)(c0 + i) * npf] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (c0 + i < nvt * 3) {
                    const float *d = s_dvpT + (c0 + i) * FPW;
                    if constexpr (FPW >= 4) {
#pragma unroll
                        for (int f4 = 0; f4 < FPW; f4 += 4) {
                            const float4 y = *(const float4 *)(d + f4);
                            acc[f4] += x[i] * y.x; acc[f4 + 1] += x[i] * y.y; acc[f4 + 2] += x[i] * y.z; acc[f4 + 3] += x[i] * y.w;
                        }
                    } else {
#pragma unroll
                        for (int f = 0; f < FPW; ++f) acc[f] += x[i] * d[f];
                    }
                }
            }
        }
#pragma unroll
        for (int f = 0; f < FPW; ++f) if (f < nf) out0[f * fstride + p] = acc[f];
    }
    if (half != 0) {                       // (the second half of a split tile: zeros in the other sections of its row)
        for (int i = tid; i < nf * (EXT - npf); i += 512) { const int f = i / (EXT - npf), e = i - f * (EXT - npf); out0[f * fstride + npf + e] = 0.f; }
        return;
    }
    // (b) chain-matrix partials: item (f, j) = all twelve entries sum_v w_vj dv_a [vp_b | 1] of a joint: per vertex one weight,
    // three dv and three vp reads feed twelve products (the entry-per-thread version issued 36 LDS reads for them and was
    // LDS-issue bound: 16 of this kernel's 46 us)
    for (int i = 511 - tid; i < nf * nj; i += 512) {
        const int f = i / nj, j = i - f * nj;
        const float *dv = s_dv + f * COLS, *vp = s_vp + f * COLS;
        float acc[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) acc[e] = 0.f;
        for (int vl = 0; vl < TV; ++vl) {
            const float w = s_w[vl * nj + j];
            const float d0 = dv[vl * 3], d1 = dv[vl * 3 + 1], d2 = dv[vl * 3 + 2];
            const float p0 = vp[vl * 3], p1 = vp[vl * 3 + 1], p2 = vp[vl * 3 + 2];
            const float t0 = w * d0, t1 = w * d1, t2 = w * d2;
            acc[0] += t0 * p0; acc[1] += t0 * p1; acc[2] += t0 * p2; acc[3] += t0 * 1.f;
            acc[4] += t1 * p0; acc[5] += t1 * p1; acc[6] += t1 * p2; acc[7] += t1 * 1.f;
            acc[8] += t2 * p0; acc[9] += t2 * p1; acc[10] += t2 * p2; acc[11] += t2 * 1.f;
        }
        float *o = out0 + f * fstride + npf + j * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) o[e] = acc[e];
    }
    // (c) betas (item (f, l)), transl, scale (item (f, 0..3))
    for (int i = tid; i < nf * nb; i += 512) {
        const int f = i / nb, l = i - f * nb;
        float acc = 0.f;
        for (int c = 0; c < nvt * 3; ++c) acc += s_sd[c * nb + l] * s_dvpT[c * FPW + f];
        out0[f * fstride + npf + nj * 12 + l] = acc;
    }
    for (int i = tid; i < nf * 4; i += 512) {
        const int f = i >> 2, k = i & 3;
        float acc = 0.f;
        if (k < 3) { for (int vl = 0; vl < TV; ++vl) acc += s_ts[(f * 2) * COLS + vl * 3 + k]; }
        else { for (int c = 0; c < COLS; ++c) acc += s_ts[(f * 2 + 1) * COLS + c]; }
        out0[f * fstride + npf + nj * 12 + nb + k] = acc;
    }
}

extern "C" int bf_mesh_bwd_multi_launch(const MeshTab *M, const float *posedirsT, const float *state, int n, const float *dvout,
                                        const float *vposed, const float *vraw, float *part, hipStream_t stream,
                                        const float *gpart, int n_masks, int n_sampled, int samp_stride, int part_rows, int *rows_out,
                                        const MaskFold *fold) {
    MaskFold G = {};
    if (fold) G = *fold;
    // part_rows: rows per frame the partial buffer holds; *rows_out: rows per frame this launch wrote (n_tiles, or 2 n_tiles when split)
    constexpr int COLS = BF_MESH_TILE * 3;
    const int fpw = n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : 8));
    const int split = (fpw == 1 && M->n_tiles * 2 <= part_rows && M->n_tiles * 2 <= 256 && COLS % 32 == 0) ? 2 : 1;
    if (rows_out) *rows_out = M->n_tiles * split;
    const size_t smem = sizeof(float) * ((size_t)COLS * fpw + (size_t)fpw * M->nj * 12 + (size_t)BF_MESH_TILE * M->nj + 2 * (size_t)fpw * COLS +
                                         (size_t)fpw * 8 + (size_t)COLS * M->nb + 2 * (size_t)fpw * COLS);
    const dim3 grid(M->n_tiles * split, (n + fpw - 1) / fpw), block(512);
    if (smem > 64 * 1024) return (int)hipErrorInvalidValue;
    switch (fpw) {
    case 1: if (G.acc) hipLaunchKernelGGL((bf_mesh_bwd_multi_kernel<1, true>), grid, block, smem, stream, *M, posedirsT, state, n, dvout, vposed, vraw, part, gpart, n_masks, n_sampled, samp_stride, split, G);
             else hipLaunchKernelGGL((bf_mesh_bwd_multi_kernel<1, false>), grid, block, smem, stream, *M, posedirsT, state, n, dvout, vposed, vraw, part, gpart, n_masks, n_sampled, samp_stride, split, G);
             break;
    case 2: if (G.acc) hipLaunchKernelGGL((bf_mesh_bwd_multi_kernel<2, true>), grid, block, smem, stream, *M, posedirsT, state, n, dvout, vposed, vraw, part, gpart, n_masks, n_sampled, samp_stride, split, G);
             else hipLaunchKernelGGL((bf_mesh_bwd_multi_kernel<2, false>), grid, block, smem, stream, *M, posedirsT, state, n, dvout, vposed, vraw, part, gpart, n_masks, n_sampled, samp_stride, split, G);
             break;
    case 4: if (G.acc) hipLaunchKernelGGL((bf_mesh_bwd_multi_kernel<4, true>), grid, block, smem, stream, *M, posedirsT, state, n, dvout, vposed, vraw, part, gpart, n_masks, n_sampled, samp_stride, split, G);
             else hipLaunchKernelGGL((bf_mesh_bwd_multi_kernel<4, false>), grid, block, smem, stream, *M, posedirsT, state, n, dvout, vposed, vraw, part, gpart, n_masks, n_sampled, samp_stride, split, G);
             break;
    default: if (G.acc) hipLaunchKernelGGL((bf_mesh_bwd_multi_kernel<8, true>), grid, block, smem, stream, *M, posedirsT, state, n, dvout, vposed, vraw, part, gpart, n_masks, n_sampled, samp_stride, split, G);
             else hipLaunchKernelGGL((bf_mesh_bwd_multi_kernel<8, false>), grid, block, smem, stream, *M, posedirsT, state, n, dvout, vposed, vraw, part, gpart, n_masks, n_sampled, samp_stride, split, G);
             break;
    }
    return (int)hipGetLastError();
}


// grid (ceil(EXT/32), F), 256 threads = 32 outputs x 8 tile chunks: ext[f][i] = sum over the tiles of part[f][tile][i].
// Self-test of the two streams the resident fit launch needs (scan_api.hip): `probe` on the fit stream waits (bounded) for a
// bell that `ring` on the batch stream sets.  If the runtime has put both streams on one hardware queue the ring cannot start while
// the probe runs, and the probe reports 2 instead of 1: the batch then keeps one fit launch per iteration.
extern "C" __global__ void bf_door_probe_kernel(int *door) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    int seen = 0;
    while (!(seen = __hip_atomic_load(door + BF_DOOR_EXT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < 2000000LL)   // 20 ms
        __builtin_amdgcn_s_sleep(8);
    __hip_atomic_store(door + BF_DOOR_TICKET, seen ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" __global__ void bf_door_ring_kernel(int *door) {
    if (threadIdx.x == 0) __hip_atomic_store(door + BF_DOOR_EXT, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// A chunk lane adds its contiguous run of tiles in tile order (loads issued eight at a time: a plain serial loop costs
// one memory latency per tile), then the eight chunk sums are added in chunk order: a fixed order, run to run.
#ifndef BF_RED_UNROLL
#define BF_RED_UNROLL 24
#endif
extern "C" __global__ void __launch_bounds__(8 * BF_RED_COLS)
bf_ext_reduce_kernel(const float *__restrict__ part, int n_tiles, int EXT, float *__restrict__ ext, int ext_stride, int *door, int door_k) {
    // (BF_RED_COLS outputs per workgroup: 64 since the end of round 5 - half as many workgroups ring the door, each behind a device-scope
    //  release that writes its XCD's L2 back, and a wave reads 256 contiguous bytes of a partial row instead of two rows' 128)
    __shared__ float s_c[8][BF_RED_COLS];
    const int li = threadIdx.x % BF_RED_COLS, ch = threadIdx.x / BF_RED_COLS, i = blockIdx.x * BF_RED_COLS + li, f = blockIdx.y;
    const int per = (n_tiles + 7) / 8, t0 = ch * per, t1 = min(n_tiles, t0 + per);
    float acc = 0.f;
    if (i < EXT) {
        const float *p = part + (size_t)f * n_tiles * EXT + i;
        int t = t0;
        for (; t + BF_RED_UNROLL <= t1; t += BF_RED_UNROLL) {          // (loads in flight per thread; the additions stay in row order)
            float v[BF_RED_UNROLL];
#pragma unroll
            for (int q = 0; q < BF_RED_UNROLL; ++q) v[q] = p[(size_t)(t + q) * EXT];
#pragma unroll
            for (int q = 0; q < BF_RED_UNROLL; ++q) acc += v[q];
        }
        for (; t + 8 <= t1; t += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(t + q) * EXT];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; t < t1; ++t) acc += p[(size_t)t * EXT];
    }
    s_c[ch][li] = acc;
    __syncthreads();
    if (ch == 0 && i < EXT) {
        float tot = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) tot += s_c[c][li];
        ext[(size_t)f * ext_stride + i] = tot;
    }
    if (door) {        // the last workgroup to finish rings the fit launch's bell for dense iteration door_k (BfDoor)
        __syncthreads();                                       // every store of this workgroup has reached the XCD's L2
        if (threadIdx.x == 0) {
            // ONE device-scope release per workgroup (it writes the L2's dirty lines back, whoever wrote them; a fence per wave
            // made this kernel 21 us instead of 5 at eight frames)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const int total = gridDim.x * gridDim.y;
            const int t = __hip_atomic_fetch_add(door + BF_DOOR_TICKET, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t == total * door_k - 1) __hip_atomic_store(door + BF_DOOR_EXT, door_k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


extern "C" __global__ void __launch_bounds__(512)
bf_kp_loss_kernel(KpIO Q, const float *__restrict__ jraw, const float *__restrict__ state, const float *__restrict__ proj_all,
                  const float *__restrict__ keypoints, const int *__restrict__ ndiv, const int *__restrict__ lmk_vid,
                  const float *__restrict__ lmk_w, float *__restrict__ ext, float *__restrict__ dvout, float *__restrict__ terms,
                  MeshTab M, const float *__restrict__ vraw, const float *__restrict__ xpart, int *door) {
    extern __shared__ __align__(16) float sm[];
    if (vraw) {           // the joints first (bf_joints_kernel's body: jraw / lmk_vid / lmk_w are then OUTPUTS of this workgroup)
        bf_joints_body<512>(M, state, vraw, xpart, nullptr, nullptr, const_cast<float *>(jraw), const_cast<int *>(lmk_vid), const_cast<float *>(lmk_w),
                            blockIdx.x, sm);
        __syncthreads();
    }
    bf_kp_loss_body(blockIdx.x, sm, Q, jraw, state, proj_all, keypoints, ndiv, lmk_vid, lmk_w, ext, dvout, terms);
    if (door) {           // (on the second stream beside the search: count this workgroup off for bf_pc_grad_kernel - one device-scope release, then the ticket)
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            (void)__hip_atomic_fetch_add(door + BF_DOOR_KP, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The dense keypoint loss and the silhouette loss's contour scan in ONE launch.  grid (contour blocks + 1, M, F), 512 threads:
// block x < gridDim.x - 1 is a contour block (32 contour points, 16 lanes each: bf_mask_contour_kernel's arithmetic, same
// results); block (gridDim.x - 1, 0, f) is frame f's keypoint workgroup.
extern "C" __global__ void __launch_bounds__(512)
bf_kp_contour_kernel(KpIO Q, const float *__restrict__ jraw, const float *__restrict__ state, const float *__restrict__ proj_all,
                     const float *__restrict__ keypoints, const int *__restrict__ ndiv, const int *__restrict__ lmk_vid,
                     const float *__restrict__ lmk_w, float *__restrict__ ext, float *__restrict__ dvout, float *__restrict__ terms,
                     MaskIO K, const float *__restrict__ uvi, int *__restrict__ choice, float *__restrict__ cgrad,
                     float *__restrict__ loss_part, MeshTab M, const float *__restrict__ vraw, const float *__restrict__ xpart) {
    extern __shared__ __align__(16) float sm[];
    __shared__ float4 tile[512];
    __shared__ float sred[8];
    if (blockIdx.x == gridDim.x - 1) {
        if (blockIdx.y == 0) {
            if (vraw) {
                bf_joints_body<512>(M, state, vraw, xpart, nullptr, nullptr, const_cast<float *>(jraw), const_cast<int *>(lmk_vid),
                                    const_cast<float *>(lmk_w), blockIdx.z, sm);
                __syncthreads();
            }
            bf_kp_loss_body(blockIdx.z, sm, Q, jraw, state, proj_all, keypoints, ndiv, lmk_vid, lmk_w, ext, dvout, terms);
        }
        return;
    }
    bf_mask_contour_body<512>(blockIdx.x, blockIdx.y, blockIdx.z, tile, sred, K, uvi, choice, cgrad, loss_part);
}
