// The reference's per-triangle closest-point rule in the reference's OWN float32 arithmetic (device code, gfx950).
//
//   thirdparty/mesh_grid/mesh_grid_kernel.cu:12-109   search_nearest_proj: Gram matrix of the corner vectors, the bordered 4 x 4
//                                                     system solved by solve4, argmin-coefficient edge fallback through solve3,
//                                                     |Lagrange multiplier| as the squared distance
//   thirdparty/mesh_grid/matrix.h:13-112, 114-316     solve3 / solve4: elimination with partial pivoting, rank decisions on
//                                                     |pivot| <= 1e-9 (absolute)
//
// Why the arithmetic and not just the mathematics: the search returns FACE IDS, and on a closed surface a third of the queries have
// their closest point on an edge or a vertex that several faces share.  Which of those faces wins is decided by the last bits of the
// distances the rule returns for them - so the kernel evaluates every product, sum and quotient the reference's source writes, in its
// order, each rounded once (no fused multiply-adds: every operation below is a separate statement or goes through op_*; IEEE
// division).  tests/test_gpu_scan.py holds this file to oracle/nearest_ref.c (the CPU restatement, itself held bit for bit to the
// reference's matrix.h) on face ids, coefficients and points.
//
// Shape of the code.  For every triangle within a metre of its query the first pivot of solve4 is the border row (|G_ij| < 1 = the
// border's entry), after which the 4 x 4 elimination is a 3 x 3 one on (G_1. - G_0., G_2. - G_0., G_0. - G_0 0) with data-dependent
// row order.  `kkt_regular` / `edge_regular` are that path with the row choices as selects - the same operations on the same values
// as the loops of matrix.h, nothing dropped: products by the border's 1 and 0 are exact and written out only where a non-finite
// operand could make a difference.  Whenever a rank decision of the reference would fire (a pivot <= 1e-9, a Gram entry >= 1) they
// decline, and `elim3_general` / `elim4_general` - the routines of matrix.h as loops over LDS-free private arrays, rare and slow -
// take the system from the start.
#pragma once

namespace nrule {

__device__ __forceinline__ float op_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
__device__ __forceinline__ float op_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float op_div(float a, float b) { return a / b; }          // correctly rounded (hipcc default for fp32 '/')
// Correctly rounded quotients without the range scaling of the compiler's expansion (v_div_scale / v_div_fmas / v_div_fixup): the
// same Newton-Raphson core - reciprocal refined once, quotient refined twice with exact fma residuals.  Valid where the regular path
// uses it: divisors are pivots with 1e-9 < |d| < 4 (checked before the division), numerators |n| < 4 (Gram entries below 1, reduced
// by multipliers of magnitude <= 1), so no intermediate leaves the normal range.  tests/test_gpu_scan.py::test_division_helper holds
// it to IEEE division bit for bit on 2^26 operand pairs of that range.
struct Recip { float d, r; };
__device__ __forceinline__ Recip recip(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.f);
    r = __builtin_fmaf(e, r, r);
    return Recip{d, r};
}
__device__ __forceinline__ float quot(float n, const Recip &R) {
    float q = n * R.r;
    float t = __builtin_fmaf(-R.d, q, n);
    q = __builtin_fmaf(t, R.r, q);
    t = __builtin_fmaf(-R.d, q, n);
    return __builtin_fmaf(t, R.r, q);
}
__device__ __forceinline__ float mag(float x) { return x < 0.f ? -x : x; }           // matrix.h:9-11
__device__ __forceinline__ float msub(float x, float m, float y) { const float p = m * y; return x - p; }   // x - m y, two roundings

constexpr float EPS = 1e-9f;                                                         // kernel.cu:14 (scalar_t precision = 1e-9)
// the regular paths decline a little EARLIER than the reference's rank tests fire (pivots <= SAFE instead of <= EPS): declining is
// always right - the general routines then do the whole system - and it keeps every quotient of the regular path inside the range
// in which recip / quot are exact IEEE division (|n / d| < 2^90)
constexpr float SAFE = 1e-8f;

// ---- the general routines (any pivot order, every rank decision): matrix.h as loops.  column-major, A[r + n c] ----------------
// Storage: element i of A at A[i * LANES], of b at b[i * LANES] - the caller's slot in an LDS array shared by the wave's lanes (a
// private array indexed by the pivot lives in scratch memory: ~100 dependent round trips of ~600 cycles per call, measured as 64 us
// for ONE pending query; LDS is ~10x closer).
// Round 6: LANES slots per wave, not 64 - the lanes of a wave that need the general routines (a handful per launch) take turns in
// rounds of LANES (lane l uses slot l % LANES in round l / LANES): 1.25 KB of LDS per wave instead of 5, which was what held the
// closest-point kernel to six waves per SIMD.
#ifndef BF_NRULE_LANES
#define BF_NRULE_LANES 16
#endif
constexpr int LANES = BF_NRULE_LANES;
#ifdef __HIPCC__
__device__ __forceinline__ int general_round() { return (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) / LANES); }
#else
inline int general_round() { return 0; }
#endif
template <int N>
__device__ __noinline__ bool elim_general(float *A_, float *b_) {
    int rank = N, pivot;
    bool valid = true;
    unsigned char permute[4] = {0, 1, 2, 3};
    struct Strided { float *p; __device__ float &operator[](int i) const { return p[i * LANES]; } };
    const Strided A{A_}, b{b_};
    auto at = [&](int r, int c) -> float & { return A[r + N * c]; };
    auto pick = [&](int col, int first) { int p = first; for (int r = first + 1; r < N; ++r) if (mag(at(p, col)) < mag(at(r, col))) p = r; return p; };
    auto exchange = [&](int c1, int c2) { for (int r = 0; r < N; ++r) { const float t = at(r, c1); at(r, c1) = at(r, c2); at(r, c2) = t; } };
    auto stage = [&](int col, int pv) {
        for (int r = col + 1; r < N; ++r) {
            if (pv == r) {                                   // the pivot row and row `col` change places while the old row `col` is reduced
                const float m = op_div(at(col, col), at(r, col));
                for (int c = col + 1; c < N; ++c) { const float t = at(r, c); at(r, c) = msub(at(col, c), m, t); at(col, c) = t; }
                const float t = b[r]; b[r] = msub(b[col], m, t); b[col] = t;
                at(col, col) = at(r, col);
                pv = col;
            } else {
                const float m = op_div(at(r, col), at(pv, col));
                at(r, col) = m;
                for (int c = col + 1; c < N; ++c) at(r, c) = msub(at(r, c), m, at(pv, c));
                b[r] = msub(b[r], m, b[pv]);
            }
        }
    };
    pivot = pick(0, 0);                                      // matrix.h:17-35, 118-151
    for (int last = N - 1; mag(at(pivot, 0)) <= EPS; --last) {
        if (last == 0) { permute[--rank] = 0; break; }
        exchange(0, last);
        permute[--rank] = 0;
        pivot = pick(0, 0);
    }
    if (rank > 0) stage(0, pivot);
    if (N == 3) {                                            // matrix.h:61-86
        if (rank > 1) {
            pivot = pick(1, 1);
            if (mag(A[pivot]) <= EPS) {                      // (the reference reads column 0 here, matrix.h:64,71 - kept)
                if (rank > 2) {
                    exchange(1, 2); permute[--rank] = 1; pivot = pick(1, 1);
                    if (mag(A[pivot]) <= EPS) permute[--rank] = 1;
                } else permute[--rank] = 1;
            }
        }
        if (rank > 1) { stage(1, pivot); if (rank >= 3 && mag(A[8]) <= EPS) permute[--rank] = 2; }
        if (rank >= 3) b[2] = op_div(b[2], A[8]); else if (mag(b[2]) > EPS) valid = false;
        if (rank >= 2) b[1] = op_div(msub(b[1], A[7], b[2]), A[4]); else if (mag(b[1]) > EPS) valid = false;
        if (rank >= 1) b[0] = op_div(msub(msub(b[0], A[6], b[2]), A[3], b[1]), A[0]); else if (mag(b[0]) > EPS) valid = false;
    } else {                                                 // matrix.h:198-301
        if (rank > 1) {
            pivot = pick(1, 1);
            if (mag(at(pivot, 1)) <= EPS) {
                if (rank > 2) {
                    exchange(1, rank - 1); permute[--rank] = 1; pivot = pick(1, 1);
                    if (mag(at(pivot, 1)) <= EPS) {
                        if (rank > 2) { exchange(1, rank - 1); permute[--rank] = 1; pivot = pick(1, 1); }      // (no third look, matrix.h:207-216)
                        else permute[--rank] = 1;
                    }
                } else permute[--rank] = 1;
            }
        }
        if (rank > 1) stage(1, pivot);
        if (rank > 2) {
            pivot = pick(2, 2);
            if (mag(at(pivot, 2)) <= EPS) {
                if (rank > 3) {
                    exchange(2, 3); permute[--rank] = 2; pivot = pick(2, 2);
                    if (mag(at(pivot, 2)) <= EPS) permute[--rank] = 2;
                } else permute[--rank] = 2;
            }
        }
        if (rank > 2) { stage(2, pivot); if (rank > 3 && mag(A[15]) <= EPS) permute[--rank] = 3; }
        if (rank >= 4) b[3] = op_div(b[3], A[15]); else if (mag(b[3]) > EPS) valid = false;
        if (rank >= 3) b[2] = op_div(msub(b[2], A[14], b[3]), A[10]); else if (mag(b[1]) > EPS) valid = false;       // (b[1]: matrix.h:289)
        if (rank >= 2) b[1] = op_div(msub(msub(b[1], A[9], b[2]), A[13], b[3]), A[5]); else if (mag(b[1]) > EPS) valid = false;
        if (rank >= 1) b[0] = op_div(msub(msub(msub(b[0], A[4], b[1]), A[8], b[2]), A[12], b[3]), A[0]); else if (mag(b[0]) > EPS) valid = false;
    }
    for (int u = 1; u < N; ++u)                             // matrix.h:103-110, 306-314
        if (rank <= u && permute[u] != u) { const float t = b[u]; b[u] = b[permute[u]]; b[permute[u]] = t; }
    return valid;
}

// ---- the edge system [G_jj G_jk 1; G_kj G_kk 1; 1 1 0] x = (0, 0, 1) (kernel.cu:46-51, 79-84) -----------------------------------
// regular path: pivot of column 0 is the border row; false when solve3 would take a rank decision.
__device__ __forceinline__ bool edge_regular(float gjj, float gjk, float gkk, float &xj, float &xk, float &lam) {
    const float big = mag(gjj) < mag(gjk) ? gjk : gjj;                      // sequential pivot pick over rows 0, 1 ...
    if (!(mag(big) < 1.f) || !(mag(gkk) < 1.f)) return false;               // ... then against the border's 1 (matrix.h:17-18)
    // after stage 0: row 0 = [1 1 0 | 1]; the other two rows (column 1, column 2 = 1, rhs):
    const float a4 = op_sub(gkk, gjk), b1 = op_sub(0.f, gjk);               // row 1: G_kk - G_kj, -G_kj   (multiplier G_kj / 1)
    const float a5 = op_sub(gjk, gjj), b2 = op_sub(0.f, gjj);               // row 2: G_jk - G_jj, -G_jj   (exchange with the border row)
    const bool low = mag(a4) < mag(a5);                                     // pivot of column 1: row 2 if |a4| < |a5| (matrix.h:62)
    if (!low && mag(gjk) <= EPS) return false;                              // the test that reads column 0: A[1] = G_kj, A[2] = 1
    const float pa = low ? a5 : a4, pb = low ? b2 : b1, xa = low ? a4 : a5, xb = low ? b1 : b2;
    if (!(mag(pa) > SAFE)) return false;
    const Recip rp = recip(pa);
    const float m = quot(xa, rp);
    const float a8 = op_sub(1.f, m);                                        // 1 - m * 1
    const float b8 = msub(xb, m, pb);
    if (!(mag(a8) > SAFE)) return false;                                    // matrix.h:85
    lam = quot(b8, recip(a8));
    xk = quot(op_sub(pb, lam), rp);                                         // (b1 - 1 * lam) / A4
    xj = op_sub(op_sub(1.f, op_mul(0.f, lam)), xk);                         // ((1 - 0 * lam) - 1 * xk) / 1
    return true;
}

// ---- the KKT system [G 1; 1^T 0] x = (0, 0, 0, 1) (kernel.cu:31-38) ------------------------------------------------------------
struct Row { float c1, c2, c3, b; };
__device__ __forceinline__ Row pick_row(bool s, const Row &a, const Row &b) { return Row{s ? a.c1 : b.c1, s ? a.c2 : b.c2, s ? a.c3 : b.c3, s ? a.b : b.b}; }
__device__ __forceinline__ Row reduce(const Row &x, const Row &p, const Recip &rp) {       // x - (x.c1 / p.c1) p, column 1 dropped
    const float m = quot(x.c1, rp);
    return Row{0.f, msub(x.c2, m, p.c2), msub(x.c3, m, p.c3), msub(x.b, m, p.b)};
}

__device__ __forceinline__ bool kkt_regular(float g00, float g01, float g02, float g11, float g12, float g22, float x[4]) {
    float big = mag(g00) < mag(g01) ? g01 : g00;
    big = mag(big) < mag(g02) ? g02 : big;
    if (!(mag(big) < 1.f)) return false;                                    // first pivot = the border row (matrix.h:118-120)
    if (!(mag(g11) < 1.f) || !(mag(g22) < 1.f)) return false;               // (every Gram entry below 1: the range recip / quot are exact in)
    // after stage 0 (multipliers G_0i / 1): rows 1, 2 reduced by the border row, row 3 = old row 0 reduced by it
    const Row r1{op_sub(g11, g01), op_sub(g12, g01), 1.f, op_sub(0.f, g01)};
    const Row r2{op_sub(g12, g02), op_sub(g22, g02), 1.f, op_sub(0.f, g02)};
    const Row r3{op_sub(g01, g00), op_sub(g02, g00), 1.f, op_sub(0.f, g00)};
    // stage 1: pivot row among r1, r2, r3 by |c1|, sequential strict '<' (matrix.h:199-201)
    const bool s2 = mag(r1.c1) < mag(r2.c1);
    const float v12 = s2 ? r2.c1 : r1.c1;
    const bool s3 = mag(v12) < mag(r3.c1);
    if (!(mag(s3 ? r3.c1 : v12) > SAFE)) return false;
    const Row P = s3 ? r3 : (s2 ? r2 : r1);
    const Row X = (!s3 && s2) ? r1 : r2;                                    // position 2: r2, or r1 when r2 is the pivot
    const Row Y = s3 ? r1 : r3;                                             // position 3: r3, or r1 when r3 is the pivot
    const Recip rP = recip(P.c1);
    const Row X1 = reduce(X, P, rP), Y1 = reduce(Y, P, rP);
    // stage 2 (matrix.h:248, 268-278)
    const bool t3 = mag(X1.c2) < mag(Y1.c2);
    const Row P2 = t3 ? Y1 : X1, Z = t3 ? X1 : Y1;
    if (!(mag(P2.c2) > SAFE)) return false;
    const Recip rP2 = recip(P2.c2);
    const float m = quot(Z.c2, rP2);
    const float z3 = msub(Z.c3, m, P2.c3), zb = msub(Z.b, m, P2.b);
    if (!(mag(z3) > SAFE)) return false;                                    // matrix.h:279
    x[3] = quot(zb, recip(z3));
    x[2] = quot(msub(P2.b, P2.c3, x[3]), rP2);
    x[1] = quot(msub(msub(P.b, P.c2, x[2]), P.c3, x[3]), rP);
    x[0] = op_sub(op_sub(op_sub(1.f, x[1]), x[2]), op_mul(0.f, x[3]));      // (((1 - 1 x1) - 1 x2) - 0 x3) / 1
    return true;
}

// The Gram matrix of the corner vectors, upper triangle (kernel.cu:23-30: each entry 0 + three products, left to right)
__device__ __forceinline__ void gram(const float *p, float &g00, float &g01, float &g02, float &g11, float &g12, float &g22) {
    auto dot = [&](int i, int j) {
        float s = op_add(0.f, op_mul(p[i * 3], p[j * 3]));
        s = op_add(s, op_mul(p[i * 3 + 1], p[j * 3 + 1]));
        return op_add(s, op_mul(p[i * 3 + 2], p[j * 3 + 2]));
    };
    g00 = dot(0, 0); g01 = dot(0, 1); g02 = dot(0, 2); g11 = dot(1, 1); g12 = dot(1, 2); g22 = dot(2, 2);
}

// search_nearest_proj (kernel.cu:12-109) where solve4 and solve3 take no rank decision - straight-line code, no branch: every
// "decline" of the regular paths is a flag, the edge system is evaluated for every lane (in a wave of 64 triangles some lane needs
// it anyway) and selects pick the answer.  p = the triangle's corners relative to the query, corner-major.  -> the rule's squared
// distance (never negative), or -1 when the regular paths do not apply (the caller hands the whole query to the kernel built with
// nearest_proj_general: the main kernel then holds no call, no private array and half the registers).
__device__ __forceinline__ float nearest_proj_regular(const float *p, float *coeff) {
    float g00, g01, g02, g11, g12, g22;
    gram(p, g00, g01, g02, g11, g12, g22);
    // ---- solve4 on [G 1; 1^T 0] x = (0, 0, 0, 1), first pivot = the border row (matrix.h:118-120): every Gram entry below 1
    float big = mag(g00) < mag(g01) ? g01 : g00;
    big = mag(big) < mag(g02) ? g02 : big;
    bool bad = !(mag(big) < 1.f) || !(mag(g11) < 1.f) || !(mag(g22) < 1.f);
    // rows 1, 2 reduced by the border row, row 3 = old row 0 reduced by it: (c1, c2, 1 | b)
    const float r1a = op_sub(g11, g01), r1b = op_sub(g12, g01), r1r = op_sub(0.f, g01);
    const float r2a = op_sub(g12, g02), r2b = op_sub(g22, g02), r2r = op_sub(0.f, g02);
    const float r3a = op_sub(g01, g00), r3b = op_sub(g02, g00), r3r = op_sub(0.f, g00);
    // stage 1: pivot row by |c1|, sequential strict '<' (matrix.h:199-201); the two others keep the reference's positions
    const bool s2 = mag(r1a) < mag(r2a);
    const float v12 = s2 ? r2a : r1a;
    const bool s3 = mag(v12) < mag(r3a);
    const bool px = !s3 && s2;                                              // r2 is the pivot
    const float Pa = s3 ? r3a : v12, Pb = s3 ? r3b : (s2 ? r2b : r1b), Pr = s3 ? r3r : (s2 ? r2r : r1r);
    const float Xa = px ? r1a : r2a, Xb = px ? r1b : r2b, Xr = px ? r1r : r2r;   // position 2
    const float Ya = s3 ? r1a : r3a, Yb = s3 ? r1b : r3b, Yr = s3 ? r1r : r3r;   // position 3
    bad = bad || !(mag(Pa) > SAFE);
    const Recip rP = recip(Pa);
    const float mX = quot(Xa, rP), mY = quot(Ya, rP);
    const float X2 = msub(Xb, mX, Pb), X3 = msub(1.f, mX, 1.f), Xq = msub(Xr, mX, Pr);
    const float Y2 = msub(Yb, mY, Pb), Y3 = msub(1.f, mY, 1.f), Yq = msub(Yr, mY, Pr);
    // stage 2 (matrix.h:248, 268-278)
    const bool t3 = mag(X2) < mag(Y2);
    const float Qa = t3 ? Y2 : X2, Q3 = t3 ? Y3 : X3, Qr = t3 ? Yq : Xq;
    const float Za = t3 ? X2 : Y2, Z3 = t3 ? X3 : Y3, Zr = t3 ? Xq : Yq;
    bad = bad || !(mag(Qa) > SAFE);
    const Recip rQ = recip(Qa);
    const float m = quot(Za, rQ);
    const float z3 = msub(Z3, m, Q3), zb = msub(Zr, m, Qr);
    bad = bad || !(mag(z3) > SAFE);                                         // matrix.h:279
    const float x3 = quot(zb, recip(z3));
    const float x2 = quot(msub(Qr, Q3, x3), rQ);
    const float x1 = quot(msub(msub(Pr, Pb, x2), 1.f, x3), rP);
    const float x0 = op_sub(op_sub(op_sub(1.f, x1), x2), op_mul(0.f, x3));  // (((1 - 1 x1) - 1 x2) - 0 x3) / 1
    // ---- the smallest coefficient (kernel.cu:73-74)
    const bool a1 = x0 > x1;
    const float x01 = a1 ? x1 : x0;
    const bool a2 = x01 > x2;
    const bool i0 = !a1 && !a2, i1 = a1 && !a2;                             // i = 0 / 1 / (else) 2
    const bool face = !((a2 ? x2 : x01) < 0.f);
    // ---- the edge opposite corner i: [G_jj G_jk 1; G_kj G_kk 1; 1 1 0] x = (0, 0, 1), (j, k) = (i + 1, i + 2) mod 3 (kernel.cu:76-84)
    const float gjj = i0 ? g11 : (i1 ? g22 : g00);
    const float gkk = i0 ? g22 : (i1 ? g00 : g11);
    const float gjk = i0 ? g12 : (i1 ? g02 : g01);
    const float ebig = mag(gjj) < mag(gjk) ? gjk : gjj;                     // pivot of column 0 against the border's 1 (matrix.h:17-18)
    bool ebad = !(mag(ebig) < 1.f);
    const float a4 = op_sub(gkk, gjk), b1 = op_sub(0.f, gjk);               // row 1: G_kk - G_kj, -G_kj   (multiplier G_kj / 1)
    const float a5 = op_sub(gjk, gjj), b2 = op_sub(0.f, gjj);               // row 2: G_jk - G_jj, -G_jj   (exchange with the border row)
    const bool low = mag(a4) < mag(a5);                                     // pivot of column 1: row 2 if |a4| < |a5| (matrix.h:62)
    // solve3's second-stage singularity test reads column 0 (matrix.h:64, 71): with the border row as first pivot that is A[1] = G_kj
    // when row 1 is the stage's pivot and A[2] = 1 otherwise.  |G_kj| <= 1e-9 - the two corner vectors at right angles - happens
    // about once in 10^5 evaluations (a hundred times per launch at config 5's size), so it is followed here rather than declined: the
    // routine exchanges columns 1 and 2, finds the same row and the same A[1], ends at rank 1 without a division, reports the system
    // inconsistent - which search_nearest_proj ignores on this path (kernel.cu:85) - and leaves x = (1 + G_jj, -G_jj, -G_kj).
    const bool quirk = !low && mag(gjk) <= EPS;
    const float pa = low ? a5 : a4, pb = low ? b2 : b1, xa = low ? a4 : a5, xb = low ? b1 : b2;
    ebad = ebad || (!quirk && !(mag(pa) > SAFE));
    const Recip rp = recip(pa);
    const float em = quot(xa, rp);
    const float a8 = op_sub(1.f, em);                                       // 1 - m * 1
    const float b8 = msub(xb, em, pb);
    ebad = ebad || (!quirk && !(mag(a8) > SAFE));                           // matrix.h:85
    const float lam_r = quot(b8, recip(a8));
    const float ek_r = quot(op_sub(pb, lam_r), rp);                         // (b1 - 1 * lam) / A4
    const float ej_r = op_sub(op_sub(1.f, op_mul(0.f, lam_r)), ek_r);       // ((1 - 0 * lam) - 1 * ek) / 1
    const float ej = quirk ? op_sub(op_sub(1.f, b2), op_mul(0.f, b1)) : ej_r;      // ((1 - 1 b2) - 0 b1) / 1 after the column exchange
    const float ek = quirk ? b2 : ek_r, lam = quirk ? b1 : lam_r;
    bad = bad || (!face && ebad);
    // kernel.cu:86-101
    const bool atk = ej < 0.f, atj = !atk && ek < 0.f;
    const float cj = atk ? 0.f : (atj ? 1.f : ej), ck = atk ? 1.f : (atj ? 0.f : ek);
    const float edist = atk ? gkk : (atj ? gjj : mag(lam));
    coeff[0] = face ? x0 : (i0 ? 0.f : (i1 ? ck : cj));
    coeff[1] = face ? x1 : (i0 ? cj : (i1 ? 0.f : ck));
    coeff[2] = face ? x2 : (i0 ? ck : (i1 ? cj : 0.f));
    return bad ? -1.f : (face ? mag(x3) : edist);
}

// search_nearest_proj (kernel.cu:12-109) with everything: the regular paths where they apply, the general routines of matrix.h where
// the reference takes a rank decision (or a Gram entry reaches 1).  Used by the second kernel only.
// scr: this lane's slot (lane % LANES) of a [20][LANES] float array in LDS (the systems the general routines work on).
__device__ __forceinline__ float nearest_proj_general(const float *p, float *coeff, float *scr) {
    float G[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            float s = op_add(0.f, op_mul(p[i * 3], p[j * 3]));
            s = op_add(s, op_mul(p[i * 3 + 1], p[j * 3 + 1]));
            s = op_add(s, op_mul(p[i * 3 + 2], p[j * 3 + 2]));
            G[j + 3 * i] = s; G[i + 3 * j] = s;
        }
    float x[4];
    bool solved;
    if (!kkt_regular(G[0], G[1], G[2], G[4], G[5], G[8], x)) {
        const float A[16] = {G[0], G[1], G[2], 1.f, G[3], G[4], G[5], 1.f, G[6], G[7], G[8], 1.f, 1.f, 1.f, 1.f, 0.f};
        solved = false;
        const int turn = general_round();
        for (int round = 0; round < 64 / LANES; ++round)
            if (round == turn) {
#pragma unroll
                for (int e = 0; e < 16; ++e) scr[e * LANES] = A[e];
                scr[16 * LANES] = 0.f; scr[17 * LANES] = 0.f; scr[18 * LANES] = 0.f; scr[19 * LANES] = 1.f;
                solved = elim_general<4>(scr, scr + 16 * LANES);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = scr[(16 + e) * LANES];
            }
    } else solved = true;
    int i;
    bool longest = false;
    if (solved) {
        i = x[0] > x[1] ? 1 : 0;                                            // kernel.cu:73-74
        i = (i ? x[1] : x[0]) > x[2] ? 2 : i;
        const float least = i == 2 ? x[2] : (i ? x[1] : x[0]);
        if (!(least < 0.f)) { coeff[0] = x[0]; coeff[1] = x[1]; coeff[2] = x[2]; return mag(x[3]); }
    } else {                                                                // kernel.cu:40-45: the longest edge
        const float l0 = op_sub(op_sub(op_add(G[4], G[8]), G[5]), G[7]);
        const float l1 = op_sub(op_sub(op_add(G[8], G[0]), G[6]), G[2]);
        const float l2 = op_sub(op_sub(op_add(G[0], G[4]), G[1]), G[3]);
        i = l0 < l1 ? 1 : 0;
        i = (i ? l1 : l0) < l2 ? 2 : i;
        longest = true;
    }
    // edge (j, k) = (i + 1, i + 2) mod 3
    const bool i0 = i == 0, i1 = i == 1;
    const float gjj = i0 ? G[4] : (i1 ? G[8] : G[0]);
    const float gkk = i0 ? G[8] : (i1 ? G[0] : G[4]);
    const float gjk = i0 ? G[5] : (i1 ? G[6] : G[1]);                       // G[3 j + k]
    float ej, ek, lam;
    bool ok3 = true;
    if (!edge_regular(gjj, gjk, gkk, ej, ek, lam)) {
        const float gkj = i0 ? G[7] : (i1 ? G[2] : G[3]);                   // G[3 k + j] (the same bits; kept as the reference writes it)
        const float A[9] = {gjj, gjk, 1.f, gkj, gkk, 1.f, 1.f, 1.f, 0.f};
        const int turn = general_round();
        for (int round = 0; round < 64 / LANES; ++round)
            if (round == turn) {
#pragma unroll
                for (int e = 0; e < 9; ++e) scr[e * LANES] = A[e];
                scr[16 * LANES] = 0.f; scr[17 * LANES] = 0.f; scr[18 * LANES] = 1.f;
                ok3 = elim_general<3>(scr, scr + 16 * LANES);
                ej = scr[16 * LANES]; ek = scr[17 * LANES]; lam = scr[18 * LANES];
            }
    }
    float cj, ck, dist;
    if (longest && !ok3) { cj = .5f; ck = .5f; dist = op_div(op_add(gjj, gkk), 2.f); }       // kernel.cu:53-58
    else if (ej < 0.f) { cj = 0.f; ck = 1.f; dist = gkk; }
    else if (ek < 0.f) { cj = 1.f; ck = 0.f; dist = gjj; }
    else { cj = ej; ck = ek; dist = mag(lam); }
    coeff[0] = i0 ? 0.f : (i1 ? ck : cj);
    coeff[1] = i0 ? cj : (i1 ? 0.f : ck);
    coeff[2] = i0 ? ck : (i1 ? cj : 0.f);
    return dist;
}

// proj = q + c0 p0 + c1 p1 + c2 p2, left to right, every product and sum rounded (kernel.cu:318-329)
__device__ __forceinline__ float project(float q, float c0, float a0, float c1, float a1, float c2, float a2) {
    return op_add(op_add(op_add(q, op_mul(c0, a0)), op_mul(c1, a1)), op_mul(c2, a2));
}

}  // namespace nrule
