// Dependent-chain latency of single fp32 VALU instructions on gfx950, written as inline asm so the compiler cannot fuse, split or
// reorder them: one wave (and 2 / 8 waves per workgroup = 1 / 2 per SIMD), N dependent instructions between two s_memtime reads.
// Build: hipcc -O3 --offload-arch=gfx950 lat2.hip -o lat2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 256
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(float *out, long long *cyc, int which, int slot) {
    float a = out[threadIdx.x] + 1.0f, m = 1.0000001f, c = 1e-9f, b = a + 0.5f;
    v2f p = {a, b}, m2 = {m, m}, c2 = {c, c};
    long long t0 = clock64();
    if (which == 0) { asm volatile(REP256("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(m), "v"(c)); }
    else if (which == 1) { asm volatile(REP256("v_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(m)); }
    else if (which == 2) { asm volatile(REP256("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(c)); }
    else if (which == 3) { asm volatile(REP64("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n") : "+v"(a) : "v"(m), "v"(c)); }
    else if (which == 4) { asm volatile(REP256("v_fmac_f32 %0, %1, %2\n") : "+v"(a) : "v"(m), "v"(c)); }
    else if (which == 5) { asm volatile(REP256("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p) : "v"(m2), "v"(c2)); }
    else if (which == 6) { asm volatile(REP256("v_pk_mul_f32 %0, %0, %1\n") : "+v"(p) : "v"(m2)); }
    else if (which == 7) { asm volatile(REP256("v_pk_add_f32 %0, %0, %1\n") : "+v"(p) : "v"(c2)); }
    else if (which == 8) { asm volatile(REP256("v_mov_b32 %0, %0\n") : "+v"(a)); }
    else if (which == 9) { asm volatile(REP256("v_fma_f32 %0, %1, %2, %0\n") : "+v"(a) : "v"(m), "v"(c)); }            // dependency through the addend only
    else if (which == 10) { asm volatile(REP64("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b) : "v"(m), "v"(c)); }   // two chains
    else if (which == 11) { asm volatile(REP256("v_sqrt_f32 %0, %0\n") : "+v"(a)); }
    else if (which == 12) { asm volatile(REP256("v_rcp_f32 %0, %0\n") : "+v"(a)); }
    else if (which == 13) { asm volatile(REP256("v_max_f32 %0, %0, %1\n") : "+v"(a) : "v"(c)); }
    else if (which == 14) { asm volatile(REP256("v_mul_f32 %0, %0, %1\n s_nop 0\n") : "+v"(a) : "v"(m)); }
    long long t1 = clock64();
    out[threadIdx.x] = a + b + p.x + p.y;
    if (threadIdx.x == 0) cyc[which * 4 + slot] = t1 - t0;
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMemset(out, 0, 4096 * 4);
    hipMalloc(&cyc, 64 * 8); hipMemset(cyc, 0, 64 * 8);
    const char *names[] = {"v_fma_f32 (dep via multiplicand)", "v_mul_f32", "v_add_f32", "v_mul,v_add alternating", "v_fmac_f32 (dep via addend)", "v_pk_fma_f32",
                           "v_pk_mul_f32", "v_pk_add_f32", "v_mov_b32", "v_fma_f32 (dep via addend)", "v_fma_f32 two chains", "v_sqrt_f32", "v_rcp_f32", "v_max_f32",
                           "v_mul_f32 + s_nop", "v_mad_f32"};
    const int nws[] = {1, 4, 8};
    for (int w = 0; w < 15; ++w)
        for (int j = 0; j < 3; ++j)
            for (int rep = 0; rep < 3; ++rep) {      // (the last of three runs counts: the first one fetches the code past a cold instruction cache)
                hipLaunchKernelGGL(k, dim3(1), dim3(64 * nws[j]), 0, 0, out, cyc, w, j);
                hipDeviceSynchronize();
            }
    std::vector<long long> h(64);
    hipMemcpy(h.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
    for (int w = 0; w < 15; ++w) {
        printf("%-36s", names[w]);
        for (int j = 0; j < 3; ++j) printf("  waves=%d: %6lld cyc (%.1f/instr)", nws[j], h[w * 4 + j], (double)h[w * 4 + j] / N);
        printf("\n");
    }
    return 0;
}
