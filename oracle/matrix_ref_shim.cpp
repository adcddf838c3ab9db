// oracle/matrix_ref_shim.cpp - TEST INFRASTRUCTURE ONLY.
// C entry points around the REFERENCE's own elimination routines: thirdparty/mesh_grid/matrix.h is plain C++ (its
// __host__/__device__ markers are defined away by the header itself, matrix.h:3-8) and is compiled WHERE IT LIES under
// /root/reference (oracle/Makefile: -I$(REF)/thirdparty/mesh_grid, output oracle/_ref/libmatrix_ref.so).  Nothing of the
// reference is copied into this repository; this file holds no arithmetic.  Used only by tests/ to hold
// oracle/nearest_ref.c's restatement of solve3 / solve4 to the reference bit for bit.
#include <cstdint>
#include "matrix.h"

extern "C" void mref_solve_batch(int n, float *A, float *b, int32_t count, float eps, int32_t *valid_out) {
    for (int32_t s = 0; s < count; ++s)
        valid_out[s] = n == 3 ? (int32_t)solve3<float>(A + 9 * s, b + 3 * s, eps) : (int32_t)solve4<float>(A + 16 * s, b + 4 * s, eps);
}
extern "C" void mref_solve_batch_f64(int n, double *A, double *b, int32_t count, double eps, int32_t *valid_out) {
    for (int32_t s = 0; s < count; ++s)
        valid_out[s] = n == 3 ? (int32_t)solve3<double>(A + 9 * s, b + 3 * s, eps) : (int32_t)solve4<double>(A + 16 * s, b + 4 * s, eps);
}
