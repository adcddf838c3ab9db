// Micro-benchmarks of single-wave instruction costs on gfx950 (one wave per SIMD unless stated): cycles per
// instruction for dependent / independent chains.  Build: hipcc -O3 --offload-arch=gfx950 lat.hip -o lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N 256
__global__ void k(float *out, long long *cyc, int which, int nw) {
    __shared__ __align__(16) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    float a = out[threadIdx.x], b = a * 0.5f, c = a + 1.f, d = a + 2.f;
    v2f p = {a, b}, q = {c, d}, r = {a, c}, s = {b, d};
    const float m = 1.0000001f;
    const v2f m2 = {m, m};
    long long t0 = clock64();
    if (which == 0) {          // dependent v_fma_f32
#pragma unroll
        for (int i = 0; i < N; ++i) a = __builtin_fmaf(a, m, 1e-9f);
    } else if (which == 1) {   // 4 independent v_fma_f32 chains
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { a = __builtin_fmaf(a, m, 1e-9f); b = __builtin_fmaf(b, m, 1e-9f); c = __builtin_fmaf(c, m, 1e-9f); d = __builtin_fmaf(d, m, 1e-9f); }
    } else if (which == 2) {   // dependent v_pk_fma_f32
#pragma unroll
        for (int i = 0; i < N; ++i) p = p * m2 + m2;
    } else if (which == 3) {   // 4 independent v_pk_fma_f32 chains
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { p = p * m2 + m2; q = q * m2 + m2; r = r * m2 + m2; s = s * m2 + m2; }
    } else if (which == 4) {   // dependent ds_read_b32 (pointer chase, address from data)
        int idx = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { float v = lds[idx]; idx = ((int)v + idx) & 1023; }
        a = idx;
    } else if (which == 5) {   // independent broadcast ds_read_b128, 16 in flight
        const float4 *l4 = (const float4 *)lds;
        float4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { float4 v = l4[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        a = acc.x + acc.y + acc.z + acc.w;
    } else if (which == 6) {   // dependent ds_bpermute
        int x = __float_as_int(a);
        const int addr = ((threadIdx.x * 7) & 63) * 4;
#pragma unroll
        for (int i = 0; i < N / 4; ++i) x = __builtin_amdgcn_ds_bpermute(addr, x) + 1;
        a = __int_as_float(x);
    } else if (which == 7) {   // dependent v_rcp_f32
#pragma unroll
        for (int i = 0; i < N / 4; ++i) a = __builtin_amdgcn_rcpf(a) + 1.0f;
    } else if (which == 8) {   // s_barrier round trips
#pragma unroll
        for (int i = 0; i < N / 4; ++i) __syncthreads();
    } else if (which == 9) {   // dependent DPP add
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { int rr = __builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, true); a += __int_as_float(rr); }
    } else if (which == 10) {  // LDS write -> fence -> read by another lane (round trip inside a wave)
        int l = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { lds[l] = a; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); a = lds[(l + 1) & 63] + 1.f; }
    } else if (which == 11) {  // per-lane-address ds_read_b128 (stride 48 B), independent
        const float4 *l4 = (const float4 *)(lds + (threadIdx.x & 63) * 12);
        float4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { float4 v = l4[(i & 3)]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        a = acc.x + acc.y + acc.z + acc.w;
    } else if (which == 12) {  // dependent v_mul + v_add pairs (plain VALU latency)
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { a = a * m; a = a + 1e-9f; }
    }
    long long t1 = clock64();
    out[threadIdx.x] = a + b + c + d + p.x + p.y + q.x + q.y + r.x + r.y + s.x + s.y;
    if (threadIdx.x == 0) cyc[which * 8 + nw] = t1 - t0;
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMemset(out, 0, 4096 * 4);
    hipMalloc(&cyc, 16 * 8 * 8); hipMemset(cyc, 0, 16 * 8 * 8);
    const char *names[] = {"dep v_fma_f32 x256", "4-way indep v_fma_f32 x256", "dep v_pk_fma_f32 x256", "4-way indep v_pk_fma_f32 x256",
        "dep ds_read_b32 x64", "indep bcast ds_read_b128 x64", "dep ds_bpermute x64", "dep v_rcp+add x64", "s_barrier x64", "dep dpp add x64",
        "lds write-fence-read x64", "indep per-lane ds_read_b128 x64", "dep mul,add x256"};
    const int counts[] = {256, 256, 256, 256, 64, 64, 64, 64, 64, 64, 64, 64, 256};
    int nws[] = {1, 4, 8};
    for (int w = 0; w < 13; ++w)
        for (int j = 0; j < 3; ++j)
            for (int rep = 0; rep < 3; ++rep) {      // (the last of three runs counts: the first one fetches the code past a cold instruction cache)
                hipLaunchKernelGGL(k, dim3(1), dim3(64 * nws[j]), 0, 0, out, cyc, w, j);
                hipDeviceSynchronize();
            }
    std::vector<long long> h(16 * 8);
    hipMemcpy(h.data(), cyc, 16 * 8 * 8, hipMemcpyDeviceToHost);
    for (int w = 0; w < 13; ++w) {
        printf("%-34s", names[w]);
        for (int j = 0; j < 3; ++j) printf("  waves=%d: %6lld cyc (%.1f/op)", nws[j], h[w * 8 + j], (double)h[w * 8 + j] / counts[w]);
        printf("\n");
    }
    return 0;
}
