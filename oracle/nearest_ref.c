/* oracle/nearest_ref.c - TEST INFRASTRUCTURE ONLY (nothing under bodyfitting_amd/ may link, load or call this).
 *
 * A CPU restatement, operation for operation in IEEE float32, of the reference's closest-point search
 * (thirdparty/mesh_grid of generalizable-neural-performer/bodyfitting).  RESTATED, not compiled, reference code:
 *
 *   ref_elim3 / ref_elim4      pivoted elimination with absolute rank tests          matrix.h:13-112 / :114-316
 *   ref_nearest_proj           per-triangle rule: Gram matrix of the corner vectors, bordered 4x4 KKT system,
 *                              argmin-coefficient edge fallback, |multiplier| as the distance
 *                                                                                    mesh_grid_kernel.cu:12-109
 *   ref_search_nearest         the shell walk over the uniform grid: cell order, box pruning, "first strictly
 *                              closer" update, stop test                             mesh_grid_kernel.cu:239-353
 *   ref_nearest_allfaces       the rule over ALL faces in face order (no grid) - what the walk returns wherever
 *                              no two faces tie exactly
 *
 * What "the reference's arithmetic" can mean here: every product, sum and quotient below is rounded once, in the
 * order the reference's SOURCE writes them (compile with -ffp-contract=off).  The reference itself is built by nvcc,
 * whose default (-fmad=true) fuses multiply-adds where it sees fit; that choice is not in the source and cannot be
 * reproduced without nvcc.  `make -C oracle` therefore also builds this file with -ffp-contract=fast -mfma
 * (libnearest_oracle_fma.so): the distance between the two builds is the reference's own latitude, and the tests
 * report the HIP kernel against BOTH.
 *
 * Pinning: matrix.h is plain C++ (no CUDA, no ATen) and compiles where it lies under /root/reference;
 * oracle/matrix_ref_shim.cpp + oracle/Makefile build it into oracle/_ref/libmatrix_ref.so, and
 * tests/test_nearest_ref_oracle.py holds ref_elim3 / ref_elim4 to it BIT FOR BIT (random, rank-deficient and KKT-shaped
 * systems), also through committed vectors (tests/golden/matrix_ref_vectors.npz) where /root/reference is absent.
 * mesh_grid_kernel.cu needs the CUDA toolkit and ATen: unbuildable here, so ref_nearest_proj / ref_search_nearest are
 * pinned by reading only (each block cites its lines) plus the geometric known answers of tests/test_scan_oracle.py.
 *
 * Quirks of the reference that are reproduced on purpose (each marked QUIRK below):
 *   - solve3's second-stage singularity test reads column 0 (A[pivot]) instead of column 1 (matrix.h:64,71);
 *   - solve4 does not look at its pivot again after a second column exchange in stage 1 (matrix.h:207-216);
 *   - solve4's consistency test for a dropped third unknown reads b[1] (matrix.h:289);
 *   - unknowns dropped by a rank decision keep whatever the elimination left in b[];
 *   - the distance returned for a "face" or "edge" answer is |Lagrange multiplier|, not |sum c_i p_i|^2;
 *   - the degenerate fallback returns (G_jj + G_kk) / 2 with coefficients (.5, .5) (kernel.cu:54-58).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

static inline real mag(real x) { return x < 0 ? -x : x; }                 /* matrix.h:9-11 */

/* column-major n x n: equation r, unknown c */
#define AT(A, n, r, c) (A)[(r) + (n) * (c)]

static int pick_row(const real *A, int n, int col, int first) {          /* sequential strict '<' (matrix.h:17-18, 118-120) */
    int p = first;
    for (int r = first + 1; r < n; ++r)
        if (mag(AT(A, n, p, col)) < mag(AT(A, n, r, col))) p = r;
    return p;
}

static void exchange_columns(real *A, int n, int c1, int c2) {
    for (int r = 0; r < n; ++r) { real t = AT(A, n, r, c1); AT(A, n, r, c1) = AT(A, n, r, c2); AT(A, n, r, c2) = t; }
}

/* One elimination stage on column `col` with pivot row `pivot` (matrix.h:37-60, 73-84, 153-196, 223-246, 268-278):
 * rows col+1 .. n-1 in order; when the loop reaches the pivot row itself, rows `col` and `pivot` change places while the
 * old row `col` is reduced ("exchange-eliminate"), and the pivot is row `col` from then on. */
static void eliminate_stage(real *A, real *b, int n, int col, int pivot) {
    for (int r = col + 1; r < n; ++r) {
        if (pivot == r) {
            real m = AT(A, n, col, col) / AT(A, n, r, col);
            AT(A, n, col, col) = m;
            for (int c = col + 1; c < n; ++c) {
                real t = AT(A, n, r, c);
                real prod = m * t;
                AT(A, n, r, c) = AT(A, n, col, c) - prod;
                AT(A, n, col, c) = t;
            }
            {
                real t = b[r];
                real prod = m * t;
                b[r] = b[col] - prod;
                b[col] = t;
            }
            AT(A, n, col, col) = AT(A, n, r, col);
            pivot = col;
        } else {
            real m = AT(A, n, r, col) / AT(A, n, pivot, col);
            AT(A, n, r, col) = m;
            for (int c = col + 1; c < n; ++c) {
                real prod = m * AT(A, n, pivot, c);
                AT(A, n, r, c) = AT(A, n, r, c) - prod;
            }
            real prod = m * b[pivot];
            b[r] = b[r] - prod;
        }
    }
}

/* stage 0 of both routines: while the whole first column is <= eps, bring in the last live column (matrix.h:19-35, 121-151) */
static int first_column(real *A, int n, real eps, int *rank, unsigned char *permute) {
    int pivot = pick_row(A, n, 0, 0);
    for (int last = n - 1; mag(AT(A, n, pivot, 0)) <= eps; --last) {
        if (last == 0) { permute[--*rank] = 0; break; }
        exchange_columns(A, n, 0, last);
        permute[--*rank] = 0;
        pivot = pick_row(A, n, 0, 0);
    }
    return pivot;
}

int ref_elim3(real *A, real *b, real eps) {                               /* matrix.h:13-112 */
    int rank = 3, valid = 1;
    unsigned char permute[3] = {0, 1, 2};
    int pivot = first_column(A, 3, eps, &rank, permute);
    if (rank > 0) {
        eliminate_stage(A, b, 3, 0, pivot);
        if (rank > 1) {
            pivot = pick_row(A, 3, 1, 1);
            if (mag(A[pivot]) <= eps) {                                   /* QUIRK: A[pivot], not A[pivot + 3] (matrix.h:64) */
                if (rank > 2) {
                    exchange_columns(A, 3, 1, 2);
                    permute[--rank] = 1;
                    pivot = pick_row(A, 3, 1, 1);
                    if (mag(A[pivot]) <= eps) permute[--rank] = 1;        /* QUIRK again (matrix.h:71) */
                } else permute[--rank] = 1;
            }
        }
        if (rank > 1) {
            eliminate_stage(A, b, 3, 1, pivot);
            if (rank >= 3 && mag(A[8]) <= eps) permute[--rank] = 2;
        }
    }
    if (rank >= 3) b[2] = b[2] / A[8];
    else if (mag(b[2]) > eps) valid = 0;
    if (rank >= 2) { real p = A[7] * b[2]; b[1] = (b[1] - p) / A[4]; }
    else if (mag(b[1]) > eps) valid = 0;
    if (rank >= 1) { real p2 = A[6] * b[2], p1 = A[3] * b[1]; b[0] = ((b[0] - p2) - p1) / A[0]; }
    else if (mag(b[0]) > eps) valid = 0;
    if (rank <= 1 && permute[1] != 1) { real t = b[1]; b[1] = b[permute[1]]; b[permute[1]] = t; }
    if (rank <= 2 && permute[2] != 2) { real t = b[2]; b[2] = b[permute[2]]; b[permute[2]] = t; }
    return valid;
}

int ref_elim4(real *A, real *b, real eps) {                               /* matrix.h:114-316 */
    int rank = 4, valid = 1;
    unsigned char permute[4] = {0, 1, 2, 3};
    int pivot = first_column(A, 4, eps, &rank, permute);
    if (rank > 0) eliminate_stage(A, b, 4, 0, pivot);
    if (rank > 1) {                                                       /* matrix.h:198-221 */
        pivot = pick_row(A, 4, 1, 1);
        if (mag(AT(A, 4, pivot, 1)) <= eps) {
            if (rank > 2) {
                exchange_columns(A, 4, 1, rank - 1);
                permute[--rank] = 1;
                pivot = pick_row(A, 4, 1, 1);
                if (mag(AT(A, 4, pivot, 1)) <= eps) {
                    if (rank > 2) {
                        exchange_columns(A, 4, 1, rank - 1);
                        permute[--rank] = 1;
                        pivot = pick_row(A, 4, 1, 1);                     /* QUIRK: no third look at the pivot (matrix.h:207-216) */
                    } else permute[--rank] = 1;
                }
            } else permute[--rank] = 1;
        }
    }
    if (rank > 1) eliminate_stage(A, b, 4, 1, pivot);
    if (rank > 2) {                                                       /* matrix.h:247-266 */
        pivot = pick_row(A, 4, 2, 2);
        if (mag(AT(A, 4, pivot, 2)) <= eps) {
            if (rank > 3) {
                exchange_columns(A, 4, 2, 3);
                permute[--rank] = 2;
                pivot = pick_row(A, 4, 2, 2);
                if (mag(AT(A, 4, pivot, 2)) <= eps) permute[--rank] = 2;  /* (the reference's inner `rank > 3` cannot hold here) */
            } else permute[--rank] = 2;
        }
    }
    if (rank > 2) {
        eliminate_stage(A, b, 4, 2, pivot);
        if (rank > 3 && mag(A[15]) <= eps) permute[--rank] = 3;
    }
    if (rank >= 4) b[3] = b[3] / A[15];
    else if (mag(b[3]) > eps) valid = 0;
    if (rank >= 3) { real p = A[14] * b[3]; b[2] = (b[2] - p) / A[10]; }
    else if (mag(b[1]) > eps) valid = 0;                                  /* QUIRK: b[1] (matrix.h:289) */
    if (rank >= 2) { real p2 = A[9] * b[2], p3 = A[13] * b[3]; b[1] = ((b[1] - p2) - p3) / A[5]; }
    else if (mag(b[1]) > eps) valid = 0;
    if (rank >= 1) { real p1 = A[4] * b[1], p2 = A[8] * b[2], p3 = A[12] * b[3]; b[0] = (((b[0] - p1) - p2) - p3) / A[0]; }
    else if (mag(b[0]) > eps) valid = 0;
    if (rank <= 1 && permute[1] != 1) { real t = b[1]; b[1] = b[permute[1]]; b[permute[1]] = t; }
    if (rank <= 2 && permute[2] != 2) { real t = b[2]; b[2] = b[permute[2]]; b[permute[2]] = t; }
    if (rank <= 3 && permute[3] != 3) { real t = b[3]; b[3] = b[permute[3]]; b[permute[3]] = t; }
    return valid;
}

/* the edge opposite corner i of the triangle as a bordered 2 x 2 system (kernel.cu:46-51, 79-84); -> solve3's verdict,
 * x[0..1] the two coefficients, x[2] the multiplier */
static int edge_system(const real G[9], int j, int k, real x[3], real eps) {
    real A[9] = {G[4 * j], G[3 * j + k], 1, G[3 * k + j], G[4 * k], 1, 1, 1, 0};
    x[0] = 0; x[1] = 0; x[2] = 1;
    return ref_elim3(A, x, eps);
}

/* mesh_grid_kernel.cu:12-109.  patch[3*corner + axis] = corner - query.  `path` (may be null) reports the branch taken:
 * 0 face, 1 edge after a negative coefficient, 2 edge after a failed solve4, 3 the (.5, .5) fallback; +4 when the edge answer
 * was clamped to a corner. */
real ref_nearest_proj(const real patch[9], real coeff[3], int *path) {
    const real eps = (real)1e-9;                                          /* kernel.cu:14: scalar_t precision = 1e-9 */
    real G[9];
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            real s = 0;
            for (int k = 0; k < 3; ++k) { real prod = patch[k + i * 3] * patch[k + j * 3]; s = s + prod; }
            G[j + 3 * i] = s; G[i + 3 * j] = s;
        }
    real A[16] = {G[0], G[1], G[2], 1, G[3], G[4], G[5], 1, G[6], G[7], G[8], 1, 1, 1, 1, 0};
    real x[4] = {0, 0, 0, 1};
    int i, j, k, which;
    real e[3];
    if (!ref_elim4(A, x, eps)) {                                          /* kernel.cu:39-71: the longest edge */
        real len[3] = {((G[4] + G[8]) - G[5]) - G[7], ((G[8] + G[0]) - G[6]) - G[2], ((G[0] + G[4]) - G[1]) - G[3]};
        i = len[0] < len[1] ? 1 : 0;
        i = len[i] < len[2] ? 2 : i;
        j = (i + 1) % 3; k = 3 - i - j;
        which = 2;
        if (!edge_system(G, j, k, e, eps)) {
            coeff[i] = 0; coeff[j] = (real).5; coeff[k] = (real).5;
            if (path) *path = 3;
            return (G[4 * j] + G[4 * k]) / 2;
        }
    } else {
        i = x[0] > x[1] ? 1 : 0;                                          /* kernel.cu:73-74: the smallest coefficient */
        i = x[i] > x[2] ? 2 : i;
        if (!(x[i] < 0)) {
            coeff[0] = x[0]; coeff[1] = x[1]; coeff[2] = x[2];
            if (path) *path = 0;
            return mag(x[3]);
        }
        j = (i + 1) % 3; k = 3 - i - j;
        which = 1;
        edge_system(G, j, k, e, eps);                                     /* verdict ignored (kernel.cu:85) */
    }
    coeff[i] = 0;
    if (e[0] < 0) { coeff[j] = 0; coeff[k] = 1; if (path) *path = which + 4; return G[4 * k]; }
    if (e[1] < 0) { coeff[j] = 1; coeff[k] = 0; if (path) *path = which + 4; return G[4 * j]; }
    coeff[j] = e[0]; coeff[k] = e[1];
    if (path) *path = which;
    return mag(e[2]);
}

static inline void ref_patch(const real *verts, const int32_t *tri, const real *q, real patch[9]) {
    for (int d = 0; d < 3; ++d)
        for (int a = 0; a < 3; ++a) patch[a + d * 3] = verts[a + 3 * tri[d]] - q[a];    /* kernel.cu:305-311 */
}

static inline void ref_accept(const real *q, const real patch[9], const real c[3], real *coeff, real *proj) {
    for (int a = 0; a < 3; ++a) {                                         /* kernel.cu:315-330, left to right */
        coeff[a] = c[a];
    }
    for (int a = 0; a < 3; ++a) {
        real s = q[a];
        real p0 = c[0] * patch[a], p1 = c[1] * patch[3 + a], p2 = c[2] * patch[6 + a];
        s = s + p0; s = s + p1; s = s + p2;
        proj[a] = s;
    }
}

/* search_nearest_point_kenerel for every query (mesh_grid_kernel.cu:239-353) on one host thread per call.
 * size[4] = cells per axis and their product; tri_num = inclusive cumulative counts; tri_idx = face id + 1.
 * near_idx / proj / coeff must be zero-filled by the caller like search_nearest_point_cuda does (:405-410).
 * stats (may be null): [0] rule evaluations, [1] cells visited, [2] shells. */
void ref_search_nearest(const int32_t *tri_num, const int32_t *tri_idx, const int32_t *size, const real *gmin, real step,
                        const real *verts, const int32_t *faces, const real *queries, int32_t n_queries,
                        real *coeff_out, real *proj_out, int32_t *near_idx, real *dist_out, int64_t *stats) {
    for (int32_t id = 0; id < n_queries; ++id) {
        const real *q = queries + 3 * id;
        int32_t home[3], lin = 0, maxL = 0, nearest = tri_num[size[3] - 1];
        for (int d = 0; d < 3; ++d) {
            real xf = (q[d] - gmin[d]) / step;
            xf = (xf < 0 ? 0 : (xf >= size[d] ? size[d] - 1 : (real)floor(xf)));
            home[d] = (int32_t)xf;
            int32_t reach = home[d] > size[d] - home[d] ? home[d] : size[d] - home[d];
            if (reach > maxL) maxL = reach;
        }
        real best = -1;
        for (int32_t L = 0; L < maxL; ++L) {
            /* the six faces of the shell in the reference's order: axis a = f % 3 pinned at -L (f < 3) or +L; the other two axes in
             * the order (f+1) % 3 fastest, (f+2) % 3 slowest; an axis already covered by an earlier face loses that end of its range
             * (kernel.cu:269-284, 343-346) */
            for (int f = 0; f < (L == 0 ? 1 : 6); ++f) {
                int ax[2] = {(f + 1) % 3, (f + 2) % 3};
                int lo[2], cnt[2];
                for (int s = 0; s < 2; ++s) {
                    int t = s + 1 + f;                                    /* the reference's d + f */
                    if (t >= 6) { lo[s] = -L + 1; cnt[s] = 2 * L - 1; }
                    else if (t >= 3) { lo[s] = -L + 1; cnt[s] = 2 * L; }
                    else { lo[s] = -L; cnt[s] = 2 * L + 1; }
                }
                for (int k1 = 0; k1 < cnt[1]; ++k1)
                    for (int k0 = 0; k0 < cnt[0]; ++k0) {
                        int off[3];
                        off[f % 3] = f < 3 ? -L : L;
                        off[ax[0]] = lo[0] + k0;
                        off[ax[1]] = lo[1] + k1;
                        real bound = 0;
                        int inside = 1;
                        for (int d = 0; d < 3; ++d) {
                            int32_t y = home[d] + off[d];
                            if (y < 0 || y >= size[d]) { inside = 0; break; }
                            if (off[d] < 0) {
                                real w = step * (y + 1);
                                real e = (q[d] - gmin[d]) - w;
                                real ee = e * e;
                                bound = bound + ee;
                            } else if (off[d] > 0) {
                                real w = step * y;
                                real e = (-q[d] + gmin[d]) + w;
                                real ee = e * e;
                                bound = bound + ee;
                            }
                            lin = d > 0 ? lin * size[d] + y : y;
                        }
                        if (!inside) continue;
                        if (best >= 0 && best < bound) continue;
                        if (stats) ++stats[1];
                        for (int32_t i = lin == 0 ? 0 : tri_num[lin - 1]; i < tri_num[lin]; ++i) {
                            real patch[9], c[3];
                            ref_patch(verts, faces + 3 * (tri_idx[i] - 1), q, patch);
                            real d2 = ref_nearest_proj(patch, c, 0);
                            if (stats) ++stats[0];
                            if (best < 0 || d2 < best) {                 /* first strictly closer (kernel.cu:314) */
                                ref_accept(q, patch, c, coeff_out + 3 * id, proj_out + 3 * id);
                                nearest = tri_idx[i] - 1;
                                best = d2;
                            }
                        }
                    }
            }
            if (stats) ++stats[2];
            {
                real r2 = (real)(L * L) * step;
                r2 = r2 * step;
                if (best >= 0 && best < r2) break;                        /* kernel.cu:349 */
            }
        }
        near_idx[id] = nearest;
        if (dist_out) dist_out[id] = best;
    }
}

/* the rule over all faces in face order, first strictly closer; ties_out[q] (may be null) counts the OTHER faces whose
 * returned distance equals the winner's bit for bit */
void ref_nearest_allfaces(const real *verts, const int32_t *faces, int32_t n_faces, const real *queries, int32_t n_queries,
                          real *coeff_out, real *proj_out, int32_t *near_idx, real *dist_out, int32_t *ties_out) {
    for (int32_t id = 0; id < n_queries; ++id) {
        const real *q = queries + 3 * id;
        real best = -1;
        int32_t nearest = -1, ties = 0;
        for (int32_t t = 0; t < n_faces; ++t) {
            real patch[9], c[3];
            ref_patch(verts, faces + 3 * t, q, patch);
            real d2 = ref_nearest_proj(patch, c, 0);
            if (best < 0 || d2 < best) {
                ref_accept(q, patch, c, coeff_out + 3 * id, proj_out + 3 * id);
                nearest = t; best = d2; ties = 0;
            } else if (d2 == best) ++ties;
        }
        near_idx[id] = nearest;
        if (dist_out) dist_out[id] = best;
        if (ties_out) ties_out[id] = ties;
    }
}

/* the rule for a list of (face, query) pairs: coefficients, distance and branch per pair */
void ref_rule_pairs(const real *verts, const int32_t *faces, const int32_t *face_of_pair, const real *queries, int32_t n_pairs,
                    real *coeff_out, real *dist_out, int32_t *path_out) {
    for (int32_t p = 0; p < n_pairs; ++p) {
        real patch[9];
        int path = 0;
        ref_patch(verts, faces + 3 * face_of_pair[p], queries + 3 * p, patch);
        dist_out[p] = ref_nearest_proj(patch, coeff_out + 3 * p, &path);
        if (path_out) path_out[p] = path;
    }
}

/* thin entry points for the elimination routines (bit-for-bit tests against oracle/_ref/libmatrix_ref.so) */
void ref_elim_batch(int n, real *A, real *b, int32_t count, real eps, int32_t *valid_out) {
    for (int32_t s = 0; s < count; ++s)
        valid_out[s] = n == 3 ? ref_elim3(A + 9 * s, b + 3 * s, eps) : ref_elim4(A + 16 * s, b + 4 * s, eps);
}

int ref_real_bytes(void) { return (int)sizeof(real); }
