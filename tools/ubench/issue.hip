// What one SIMD of a gfx950 CU issues: cycles per wave64 vector instruction PER SIMD for streams of INDEPENDENT instructions (and, beside
// them, dependent chains of 1 / 2 / 4) at 1, 2, 4, 6 and 8 waves per SIMD - one workgroup alone on the chip, and every CU busy.
// The issue bounds of bench.py / DESIGN.md (closest-point kernel, fit kernel) are priced with the constants this prints
// (profiles/r06_issue_rate.md).  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/issue.hip -o tools/ubench/issue
//
// Method.  Workgroups of 256 threads = 4 waves, one per SIMD (checked: every wave stores its HW_ID).  k workgroups per CU are forced by
// the dynamic LDS size (floor(160 KB / lds) = k) with a grid of 256 * k workgroups, all resident; a grid-wide arrival counter starts them
// together.  A wave times ITERS trips of four 64-instruction inline-asm blocks (2 KB at most: warm in the instruction cache after the
// first of three launches, the last one counts) with s_memtime.  Per SIMD: all the instructions of its waves / (last end - first start)
// in shader cycles per wave-instruction; beside it the fastest and the slowest single wave's own cycles per instruction (the arbiter
// serves the oldest wave first: the waves of a SIMD do NOT finish together).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at line %d\n", __LINE__); exit(1); } } while (0)
#define ITERS 96
#define BLOCK 256      // instructions per trip of the timed loop: four 64-instruction asm statements (a taken branch costs ~30 cycles)

#define R8(x) x x x x x x x x
// eight destinations, three sources: no instruction reads what another one of the block writes
#define OUT8P "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "+v"(o6), "+v"(o7)
#define OUT8 "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5), "=&v"(o6), "=&v"(o7)
#define VOP3_8(op) R8(op " %0, %8, %9, %10\n" op " %1, %8, %9, %10\n" op " %2, %8, %9, %10\n" op " %3, %8, %9, %10\n" \
                      op " %4, %8, %9, %10\n" op " %5, %8, %9, %10\n" op " %6, %8, %9, %10\n" op " %7, %8, %9, %10\n")
#define VOP2_8(op) R8(op " %0, %8, %9\n" op " %1, %8, %9\n" op " %2, %8, %9\n" op " %3, %8, %9\n" \
                      op " %4, %8, %9\n" op " %5, %8, %9\n" op " %6, %8, %9\n" op " %7, %8, %9\n")
#define VOP1_8(op) R8(op " %0, %8\n" op " %1, %8\n" op " %2, %8\n" op " %3, %8\n" op " %4, %8\n" op " %5, %8\n" op " %6, %8\n" op " %7, %8\n")

enum Kind { K_FMA, K_ADD, K_MUL, K_CNDMASK, K_CMP, K_PKFMA, K_MIX_SALU, K_DPP, K_ADDU, K_MULLO, K_RCP, K_BPERM, K_DEP1, K_DEP2, K_DEP4, K_LDSB128, K_CND_SMASK, K_CND_SGPR, K_CMP_CND, K_MINMAX,
    K2_V_SUB_F32, K2_V_MAX_F32, K2_V_MIN_F32, K2_V_AND_B32, K2_V_OR_B32, K2_V_XOR_B32, K2_V_LSHLREV_B32, K2_V_ASHRREV_I32, K2_V_SUB_U32, K2_V_MUL_U32_U24, K2_V_FMAC_F32, K2_V_MUL_HI_U32, K2_V_MAX_I32, K2_V_LDEXP_F32, K1_V_MOV_B32, K1_V_CVT_F32_I32, K1_V_CVT_I32_F32, K1_V_CVT_F32_U32, K1_V_FLOOR_F32, K1_V_FRACT_F32, K1_V_RNDNE_F32, K1_V_SQRT_F32, K1_V_RSQ_F32, K1_V_EXP_F32, K1_V_NOT_B32, K3_V_MAD_U32_U24, K3_V_MED3_F32, K3_V_MAX3_F32, K3_V_MIN3_F32, K3_V_ADD3_U32, K3_V_LSHL_ADD_U32, K3_V_BFE_U32, K3_V_PERM_B32, K3_V_DIV_FIXUP_F32, K3_V_ALIGNBIT_B32, K3_V_AND_OR_B32, K3_V_XAD_U32, KX_FMA_NEG, KX_ADD_SGPR, KX_ADD_LIT, KX_FMA_LIT, KX_ADD_DPP, KX_MOV_DPP_QP, KX_MOV_DPP_BC, KX_ADD_SDWA, KX_CMP_E64, KX_CMP_3CND, KX_CMP64_3CND, KX_CND_VCC_FMA, KX_CND_VCC_3FMA, KX_FMA_MAX, KX_READFIRST, KX_READLANE, KX_ADD_CO, KX_ADDC_CO, KX_LSHL_B64, KX_MAD_U64, KX_MOV_B64, KX_PK_MUL, KX_SALU, KX_MAX_SALU,
    KY_FMA_ACC, KY_FMAC_E64, KY_CND_E64_VCC, KY_MOV_SGPR, KY_MUL_SGPR, KY_FMA_SGPR, KY_ADD_INL, KY_MUL_INL, KY_SUBREV, KY_LSHL_C, KY_LSHR, KY_ADD_E64, KY_ADD_E64_CLAMP, KY_MUL_OMOD, KY_MAX_E64, KY_MAX_U32, KY_MBCNT, KY_MUL_I24, KY_ADD_F16, KY_FMA_F64, KY_3SRC_DIFF,
    KZ_ACC_NOCONF, KZ_ACC_CONF, KZ_NOACC_CONF, KZ_NOACC_CONF3, KZ_FMAC_NOCONF, KZ_DST_BANK, KZ_ADD_CONF, KZ_ADD_SAME, KZ_ACC_SRC0, KZ_ACC_SRC2_B,
    KB_ADD_1, KB_ADD_2, KB_ADD_3, KB_ADD_4, KB_ADD_5, KB_ADD_8, KB_ADD_9, KB_ADD_12, KB_ADD_16, KB_ADD_32, KB_FMA_2, KB_FMA_4, KB_FMA_8, KB_FMA_9, KB_FMA_16, KB_FMAC_0, KB_FMAC_1, KB_FMAC_8, KB_FMAC_9, KB_FMAC_D8, KB_FMAC_D9,
    KD_INF, KD_NAN, KD_DENORM, KD_RAW1, KD_RAW8, K_COUNT };
static const char *kind_name[K_COUNT] = {
    "v_fma_f32 (independent)", "v_add_f32 (independent)", "v_mul_f32 (independent)", "v_cndmask_b32 vcc (independent)", "v_cmp_lt_f32 -> vcc (independent)",
    "v_pk_fma_f32 (independent; 2 flop-lanes)", "2 v_fma_f32 : 1 s_add_u32 (per instruction of either kind)", "v_mov_b32 dpp row_shr:1 (independent)",
    "v_add_u32 (independent)", "v_mul_lo_u32 (independent)", "v_rcp_f32 (independent)", "ds_bpermute_b32 (independent, 8 in flight)",
    "v_fma_f32, ONE dependent chain", "v_fma_f32, 2 chains interleaved", "v_fma_f32, 4 chains interleaved", "ds_read_b128 broadcast (independent, 8 in flight)",
    "v_cndmask_b32 vcc, vcc written by s_mov_b64", "v_cndmask_b32 (VOP3) on an SGPR pair", "v_cmp_lt_f32 vcc + v_cndmask_b32 vcc pairs (per instruction)", "v_max_f32 / v_min_f32 alternating (independent)",
    "v_sub_f32 (independent)",
    "v_max_f32 (independent)",
    "v_min_f32 (independent)",
    "v_and_b32 (independent)",
    "v_or_b32 (independent)",
    "v_xor_b32 (independent)",
    "v_lshlrev_b32 (independent)",
    "v_ashrrev_i32 (independent)",
    "v_sub_u32 (independent)",
    "v_mul_u32_u24 (independent)",
    "v_fmac_f32 (independent)",
    "v_mul_hi_u32 (independent)",
    "v_max_i32 (independent)",
    "v_ldexp_f32 (independent)",
    "v_mov_b32 (independent)",
    "v_cvt_f32_i32 (independent)",
    "v_cvt_i32_f32 (independent)",
    "v_cvt_f32_u32 (independent)",
    "v_floor_f32 (independent)",
    "v_fract_f32 (independent)",
    "v_rndne_f32 (independent)",
    "v_sqrt_f32 (independent)",
    "v_rsq_f32 (independent)",
    "v_exp_f32 (independent)",
    "v_not_b32 (independent)",
    "v_mad_u32_u24 (independent)",
    "v_med3_f32 (independent)",
    "v_max3_f32 (independent)",
    "v_min3_f32 (independent)",
    "v_add3_u32 (independent)",
    "v_lshl_add_u32 (independent)",
    "v_bfe_u32 (independent)",
    "v_perm_b32 (independent)",
    "v_div_fixup_f32 (independent)",
    "v_alignbit_b32 (independent)",
    "v_and_or_b32 (independent)",
    "v_xad_u32 (independent)",
    "v_fma_f32 with a neg and an abs modifier",
    "v_add_f32 with an SGPR operand",
    "v_add_f32 with a 32-bit literal",
    "v_fmaak_f32 (fma with a literal addend)",
    "v_add_f32 dpp row_shr:1",
    "v_mov_b32 dpp quad_perm",
    "v_mov_b32 dpp row_bcast:15",
    "v_add_f32 sdwa (word select)",
    "v_cmp_lt_f32 (VOP3) -> SGPR pair",
    "1 v_cmp vcc + 3 v_cndmask vcc (per instruction)",
    "1 v_cmp -> SGPR pair + 3 v_cndmask on it (per instruction)",
    "v_cndmask vcc alternating with v_fma_f32 (vcc set once; per instruction)",
    "1 v_cndmask vcc : 3 v_fma_f32 (vcc set once; per instruction)",
    "v_fma_f32 alternating with v_max_f32 (per instruction)",
    "v_readfirstlane_b32",
    "v_readlane_b32 (lane in an SGPR)",
    "v_add_co_u32 -> vcc",
    "v_add_co_u32 + v_addc_co_u32 pairs (64-bit add; per instruction)",
    "v_lshlrev_b64",
    "v_mad_u64_u32",
    "v_mov_b64",
    "v_pk_mul_f32",
    "s_add_u32 (independent SALU only)",
    "1 v_max_f32 : 1 s_add_u32 (per instruction of either kind)",
    "v_fma_f32 d, a, b, d (VOP3, accumulating like v_fmac)",
    "v_fmac_f32_e64",
    "v_cndmask_b32_e64 ..., vcc (VOP3 encoding, vcc set once)",
    "v_mov_b32 v, s",
    "v_mul_f32 v, s, v",
    "v_fma_f32 v, s, v, v",
    "v_add_f32 v, 1.0, v (inline constant)",
    "v_mul_f32 v, 2.0, v (inline constant)",
    "v_subrev_f32",
    "v_lshlrev_b32 v, 2, v (constant shift)",
    "v_lshrrev_b32",
    "v_add_f32_e64 (VOP3 encoding)",
    "v_add_f32_e64 clamp",
    "v_mul_f32_e64 mul:2",
    "v_max_f32_e64 (VOP3 encoding)",
    "v_max_u32",
    "v_mbcnt_lo_u32_b32",
    "v_mul_i32_i24",
    "v_add_f16",
    "v_fma_f64",
    "v_fma_f32 with three DIFFERENT source registers per instruction (bank pattern)",
    "v_fma_f32 d, v16, v17, d with d in bank 3 (accumulate, no bank conflict)",
    "v_fma_f32 d, v16, v17, d with d in bank 0 (accumulate, src0/src2 bank conflict)",
    "v_fma_f32 d, v16, v17, v20 (two sources in bank 0, d elsewhere)",
    "v_fma_f32 d, v16, v20, v24 (three sources in bank 0)",
    "v_fmac_f32 d, v16, v17 with d in bank 3",
    "v_fma_f32 d, v16, v17, v18 with d in bank 0 (dst in a source's bank)",
    "v_add_f32 d, v16, v20 (both sources in bank 0)",
    "v_add_f32 d, v16, v16 (same register twice)",
    "v_fma_f32 d, d, v17, v18 with d in bank 0 (accumulate through src0)",
    "v_fma_f32 d, v17, v18, d with d in bank 0 (sources 1, 2, 0)",
    "v_add_f32 d, v16, v17",
    "v_add_f32 d, v16, v18",
    "v_add_f32 d, v16, v19",
    "v_add_f32 d, v16, v20",
    "v_add_f32 d, v16, v21",
    "v_add_f32 d, v16, v24",
    "v_add_f32 d, v16, v25",
    "v_add_f32 d, v16, v28",
    "v_add_f32 d, v16, v32",
    "v_add_f32 d, v16, v48",
    "v_fma_f32 d, v16, v17, v18",
    "v_fma_f32 d, v16, v17, v20",
    "v_fma_f32 d, v16, v17, v24",
    "v_fma_f32 d, v16, v17, v25",
    "v_fma_f32 d, v16, v17, v32",
    "v_fmac_f32 d, v16, v16 (d in v51..v63 step 4)",
    "v_fmac_f32 d, v16, v17 (d in v51..v63 step 4)",
    "v_fmac_f32 d, v16, v24 (d in v51..v63 step 4)",
    "v_fmac_f32 d, v16, v25 (d in v51..v63 step 4)",
    "v_fmac_f32 d, v16, v17 with d = v24, v32, v40, v48 (d = src0 + 8 n)",
    "v_fmac_f32 d, v16, v17 with d = v25, v33, v41, v49 (d = src1 + 8 n)",
    "v_fma_f32 d, v16, v17, v18 on +inf operands",
    "v_fma_f32 d, v16, v17, v18 on NaN operands",
    "v_fma_f32 d, v16, v17, v18 on denormal operands",
    "v_fmac_f32 chain where every instruction reads the register the previous one wrote as a MULTIPLICAND",
    "v_fmac_f32 d_i, v16, v55 with d_7 = v55 (a multiplicand rewritten every 8th instruction)"};
// wave-instructions per block (the SALU mix counts its 96 instructions: 64 VALU + 32 SALU)
static int kind_count[K_COUNT];   // wave-instructions per 64-slot block: 64, but 96 for the 2:1 SALU mix (filled in main)

struct Rec { long long t0, t1; unsigned hw_id, xcc_id; };

// results are dropped on purpose: a volatile asm stays, and summing them would add dependent VALU work to the timed loop
template <int KIND> __device__ __forceinline__ void block(float &a, float &b, float &c, float &d) {
    float o0 = a, o1 = a, o2 = a, o3 = a, o4 = b, o5 = b, o6 = b, o7 = b;
    if constexpr (KIND == K_FMA) asm volatile(VOP3_8("v_fma_f32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K_ADD) asm volatile(VOP2_8("v_add_f32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K_MUL) asm volatile(VOP2_8("v_mul_f32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K_CNDMASK)
        asm volatile("v_cmp_lt_f32 vcc, %8, %9\n" R8("v_cndmask_b32 %0, %8, %9, vcc\nv_cndmask_b32 %1, %8, %9, vcc\nv_cndmask_b32 %2, %8, %9, vcc\nv_cndmask_b32 %3, %8, %9, vcc\n"
                     "v_cndmask_b32 %4, %8, %9, vcc\nv_cndmask_b32 %5, %8, %9, vcc\nv_cndmask_b32 %6, %8, %9, vcc\nv_cndmask_b32 %7, %8, %9, vcc\n")
                     : OUT8 : "v"(a), "v"(b) : "vcc");
    else if constexpr (KIND == K_CMP) {
        asm volatile(R8(R8("v_cmp_lt_f32 vcc, %0, %1\n")) : : "v"(a), "v"(b) : "vcc");
    } else if constexpr (KIND == K_PKFMA) {
        v2f p0, p1, p2, p3, p4, p5, p6, p7, x = {a, b}, y = {b, c}, z = {c, a};
        asm volatile(R8("v_pk_fma_f32 %0, %8, %9, %10\nv_pk_fma_f32 %1, %8, %9, %10\nv_pk_fma_f32 %2, %8, %9, %10\nv_pk_fma_f32 %3, %8, %9, %10\n"
                        "v_pk_fma_f32 %4, %8, %9, %10\nv_pk_fma_f32 %5, %8, %9, %10\nv_pk_fma_f32 %6, %8, %9, %10\nv_pk_fma_f32 %7, %8, %9, %10\n")
                     : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7) : "v"(x), "v"(y), "v"(z));
    } else if constexpr (KIND == K_MIX_SALU) {
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        asm volatile(R8("v_fma_f32 %0, %12, %13, %14\nv_fma_f32 %1, %12, %13, %14\ns_add_u32 %8, %8, 1\nv_fma_f32 %2, %12, %13, %14\nv_fma_f32 %3, %12, %13, %14\ns_add_u32 %9, %9, 1\n"
                        "v_fma_f32 %4, %12, %13, %14\nv_fma_f32 %5, %12, %13, %14\ns_add_u32 %10, %10, 1\nv_fma_f32 %6, %12, %13, %14\nv_fma_f32 %7, %12, %13, %14\ns_add_u32 %11, %11, 1\n")
                     : OUT8, "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b), "v"(c) : "scc");
    } else if constexpr (KIND == K_DPP)
        asm volatile(R8("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5), "=&v"(o6), "=&v"(o7) : "v"(a));
    else if constexpr (KIND == K_ADDU) asm volatile(VOP2_8("v_add_u32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K_MULLO) asm volatile(VOP2_8("v_mul_lo_u32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K_RCP) asm volatile(VOP1_8("v_rcp_f32") : OUT8 : "v"(a));
    else if constexpr (KIND == K_BPERM) {
        const int addr = ((threadIdx.x * 7) & 63) * 4;
        asm volatile(R8("ds_bpermute_b32 %0, %8, %9\nds_bpermute_b32 %1, %8, %9\nds_bpermute_b32 %2, %8, %9\nds_bpermute_b32 %3, %8, %9\n"
                        "ds_bpermute_b32 %4, %8, %9\nds_bpermute_b32 %5, %8, %9\nds_bpermute_b32 %6, %8, %9\nds_bpermute_b32 %7, %8, %9\ns_waitcnt lgkmcnt(0)\n")
                     : OUT8 : "v"(addr), "v"(a) : "memory");
    } else if constexpr (KIND == K_DEP1) {
        asm volatile(R8(R8("v_fma_f32 %0, %0, %1, %2\n")) : "+v"(d) : "v"(b), "v"(c));
    } else if constexpr (KIND == K_DEP2) {
        asm volatile(R8("v_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3\nv_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3\n"
                        "v_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3\nv_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3\n") : "+v"(d), "+v"(a) : "v"(b), "v"(c));
    } else if constexpr (KIND == K_DEP4) {
        float x = d, y = a;     // (two of the four chains restart every block: 16 deep each, long enough)
        asm volatile(R8("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                        "v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n") : "+v"(x), "+v"(y), "+v"(d), "+v"(a) : "v"(b), "v"(c));
    } else if constexpr (KIND == K_CND_SMASK) {
        asm volatile("s_mov_b64 vcc, 0x5555\n" R8("v_cndmask_b32 %0, %8, %9, vcc\nv_cndmask_b32 %1, %8, %9, vcc\nv_cndmask_b32 %2, %8, %9, vcc\nv_cndmask_b32 %3, %8, %9, vcc\n"
                     "v_cndmask_b32 %4, %8, %9, vcc\nv_cndmask_b32 %5, %8, %9, vcc\nv_cndmask_b32 %6, %8, %9, vcc\nv_cndmask_b32 %7, %8, %9, vcc\n")
                     : OUT8 : "v"(a), "v"(b) : "vcc");
    } else if constexpr (KIND == K_CND_SGPR) {
        unsigned long long m = 0x5555555555555555ull;
        asm volatile(R8("v_cndmask_b32 %0, %8, %9, %10\nv_cndmask_b32 %1, %8, %9, %10\nv_cndmask_b32 %2, %8, %9, %10\nv_cndmask_b32 %3, %8, %9, %10\n"
                        "v_cndmask_b32 %4, %8, %9, %10\nv_cndmask_b32 %5, %8, %9, %10\nv_cndmask_b32 %6, %8, %9, %10\nv_cndmask_b32 %7, %8, %9, %10\n")
                     : OUT8 : "v"(a), "v"(b), "s"(m));
    } else if constexpr (KIND == K_CMP_CND) {
        asm volatile(R8("v_cmp_lt_f32 vcc, %8, %9\nv_cndmask_b32 %0, %8, %9, vcc\nv_cmp_lt_f32 vcc, %9, %8\nv_cndmask_b32 %1, %8, %9, vcc\n"
                        "v_cmp_lt_f32 vcc, %8, %9\nv_cndmask_b32 %2, %8, %9, vcc\nv_cmp_lt_f32 vcc, %9, %8\nv_cndmask_b32 %3, %8, %9, vcc\n")
                     : OUT8 : "v"(a), "v"(b) : "vcc");
    } else if constexpr (KIND == K_MINMAX) {
        asm volatile(R8("v_max_f32 %0, %8, %9\nv_min_f32 %1, %8, %9\nv_max_f32 %2, %8, %9\nv_min_f32 %3, %8, %9\nv_max_f32 %4, %8, %9\nv_min_f32 %5, %8, %9\nv_max_f32 %6, %8, %9\nv_min_f32 %7, %8, %9\n")
                     : OUT8 : "v"(a), "v"(b));
    }
    else if constexpr (KIND == K2_V_SUB_F32) asm volatile(VOP2_8("v_sub_f32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_MAX_F32) asm volatile(VOP2_8("v_max_f32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_MIN_F32) asm volatile(VOP2_8("v_min_f32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_AND_B32) asm volatile(VOP2_8("v_and_b32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_OR_B32) asm volatile(VOP2_8("v_or_b32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_XOR_B32) asm volatile(VOP2_8("v_xor_b32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_LSHLREV_B32) asm volatile(VOP2_8("v_lshlrev_b32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_ASHRREV_I32) asm volatile(VOP2_8("v_ashrrev_i32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_SUB_U32) asm volatile(VOP2_8("v_sub_u32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_MUL_U32_U24) asm volatile(VOP2_8("v_mul_u32_u24") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_FMAC_F32) asm volatile(VOP2_8("v_fmac_f32") : OUT8P : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_MUL_HI_U32) asm volatile(VOP2_8("v_mul_hi_u32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_MAX_I32) asm volatile(VOP2_8("v_max_i32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K2_V_LDEXP_F32) asm volatile(VOP2_8("v_ldexp_f32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == K1_V_MOV_B32) asm volatile(VOP1_8("v_mov_b32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_CVT_F32_I32) asm volatile(VOP1_8("v_cvt_f32_i32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_CVT_I32_F32) asm volatile(VOP1_8("v_cvt_i32_f32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_CVT_F32_U32) asm volatile(VOP1_8("v_cvt_f32_u32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_FLOOR_F32) asm volatile(VOP1_8("v_floor_f32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_FRACT_F32) asm volatile(VOP1_8("v_fract_f32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_RNDNE_F32) asm volatile(VOP1_8("v_rndne_f32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_SQRT_F32) asm volatile(VOP1_8("v_sqrt_f32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_RSQ_F32) asm volatile(VOP1_8("v_rsq_f32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_EXP_F32) asm volatile(VOP1_8("v_exp_f32") : OUT8 : "v"(a));
    else if constexpr (KIND == K1_V_NOT_B32) asm volatile(VOP1_8("v_not_b32") : OUT8 : "v"(a));
    else if constexpr (KIND == K3_V_MAD_U32_U24) asm volatile(VOP3_8("v_mad_u32_u24") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_MED3_F32) asm volatile(VOP3_8("v_med3_f32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_MAX3_F32) asm volatile(VOP3_8("v_max3_f32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_MIN3_F32) asm volatile(VOP3_8("v_min3_f32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_ADD3_U32) asm volatile(VOP3_8("v_add3_u32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_LSHL_ADD_U32) asm volatile(VOP3_8("v_lshl_add_u32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_BFE_U32) asm volatile(VOP3_8("v_bfe_u32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_PERM_B32) asm volatile(VOP3_8("v_perm_b32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_DIV_FIXUP_F32) asm volatile(VOP3_8("v_div_fixup_f32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_ALIGNBIT_B32) asm volatile(VOP3_8("v_alignbit_b32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_AND_OR_B32) asm volatile(VOP3_8("v_and_or_b32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == K3_V_XAD_U32) asm volatile(VOP3_8("v_xad_u32") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == KX_FMA_NEG) asm volatile(R8("v_fma_f32 %0, -%8, |%9|, %10\n" "v_fma_f32 %1, -%8, |%9|, %10\n" "v_fma_f32 %2, -%8, |%9|, %10\n" "v_fma_f32 %3, -%8, |%9|, %10\n" "v_fma_f32 %4, -%8, |%9|, %10\n" "v_fma_f32 %5, -%8, |%9|, %10\n" "v_fma_f32 %6, -%8, |%9|, %10\n" "v_fma_f32 %7, -%8, |%9|, %10\n") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == KX_ADD_SGPR) { float sg = 1.5f; asm volatile(R8("v_add_f32 %0, %8, %9\n" "v_add_f32 %1, %8, %9\n" "v_add_f32 %2, %8, %9\n" "v_add_f32 %3, %8, %9\n" "v_add_f32 %4, %8, %9\n" "v_add_f32 %5, %8, %9\n" "v_add_f32 %6, %8, %9\n" "v_add_f32 %7, %8, %9\n") : OUT8 : "s"(sg), "v"(b)); }
    else if constexpr (KIND == KX_ADD_LIT) asm volatile(R8("v_add_f32 %0, 0x40490fdb, %8\n" "v_add_f32 %1, 0x40490fdb, %8\n" "v_add_f32 %2, 0x40490fdb, %8\n" "v_add_f32 %3, 0x40490fdb, %8\n" "v_add_f32 %4, 0x40490fdb, %8\n" "v_add_f32 %5, 0x40490fdb, %8\n" "v_add_f32 %6, 0x40490fdb, %8\n" "v_add_f32 %7, 0x40490fdb, %8\n") : OUT8 : "v"(a));
    else if constexpr (KIND == KX_FMA_LIT) asm volatile(R8("v_fmaak_f32 %0, %8, %9, 0x40490fdb\n" "v_fmaak_f32 %1, %8, %9, 0x40490fdb\n" "v_fmaak_f32 %2, %8, %9, 0x40490fdb\n" "v_fmaak_f32 %3, %8, %9, 0x40490fdb\n" "v_fmaak_f32 %4, %8, %9, 0x40490fdb\n" "v_fmaak_f32 %5, %8, %9, 0x40490fdb\n" "v_fmaak_f32 %6, %8, %9, 0x40490fdb\n" "v_fmaak_f32 %7, %8, %9, 0x40490fdb\n") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KX_ADD_DPP) asm volatile(R8("v_add_f32_dpp %0, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %1, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %2, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %3, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %4, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %5, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %6, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %7, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KX_MOV_DPP_QP) asm volatile(R8("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : OUT8 : "v"(a));
    else if constexpr (KIND == KX_MOV_DPP_BC) asm volatile(R8("v_mov_b32_dpp %0, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n" "v_mov_b32_dpp %1, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n" "v_mov_b32_dpp %2, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n" "v_mov_b32_dpp %3, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n" "v_mov_b32_dpp %4, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n" "v_mov_b32_dpp %5, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n" "v_mov_b32_dpp %6, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n" "v_mov_b32_dpp %7, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n") : OUT8 : "v"(a));
    else if constexpr (KIND == KX_ADD_SDWA) asm volatile(R8("v_add_f32_sdwa %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_f32_sdwa %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_f32_sdwa %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_f32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_f32_sdwa %4, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_f32_sdwa %5, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_f32_sdwa %6, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" "v_add_f32_sdwa %7, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KX_CMP_E64) { unsigned long long m0, m1, m2, m3; asm volatile(R8("v_cmp_lt_f32 %0, %4, %5\nv_cmp_lt_f32 %1, %4, %5\nv_cmp_lt_f32 %2, %4, %5\nv_cmp_lt_f32 %3, %4, %5\nv_cmp_lt_f32 %0, %5, %4\nv_cmp_lt_f32 %1, %5, %4\nv_cmp_lt_f32 %2, %5, %4\nv_cmp_lt_f32 %3, %5, %4\n") : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(a), "v"(b)); }
    else if constexpr (KIND == KX_CMP_3CND) asm volatile(R8("v_cmp_lt_f32 vcc, %8, %9\nv_cndmask_b32 %0, %8, %9, vcc\nv_cndmask_b32 %1, %8, %9, vcc\nv_cndmask_b32 %2, %8, %9, vcc\nv_cmp_lt_f32 vcc, %9, %8\nv_cndmask_b32 %3, %8, %9, vcc\nv_cndmask_b32 %4, %8, %9, vcc\nv_cndmask_b32 %5, %8, %9, vcc\n") : OUT8 : "v"(a), "v"(b) : "vcc");
    else if constexpr (KIND == KX_CMP64_3CND) { unsigned long long m0, m1; asm volatile(R8("v_cmp_lt_f32 %8, %10, %11\nv_cndmask_b32 %0, %10, %11, %8\nv_cndmask_b32 %1, %10, %11, %8\nv_cndmask_b32 %2, %10, %11, %8\nv_cmp_lt_f32 %9, %11, %10\nv_cndmask_b32 %3, %10, %11, %9\nv_cndmask_b32 %4, %10, %11, %9\nv_cndmask_b32 %5, %10, %11, %9\n") : OUT8, "=&s"(m0), "=&s"(m1) : "v"(a), "v"(b)); }
    else if constexpr (KIND == KX_CND_VCC_FMA) asm volatile("v_cmp_lt_f32 vcc, %8, %9\n" R8("v_cndmask_b32 %0, %8, %9, vcc\nv_fma_f32 %1, %8, %9, %10\nv_cndmask_b32 %2, %8, %9, vcc\nv_fma_f32 %3, %8, %9, %10\nv_cndmask_b32 %4, %8, %9, vcc\nv_fma_f32 %5, %8, %9, %10\nv_cndmask_b32 %6, %8, %9, vcc\nv_fma_f32 %7, %8, %9, %10\n") : OUT8 : "v"(a), "v"(b), "v"(c) : "vcc");
    else if constexpr (KIND == KX_CND_VCC_3FMA) asm volatile("v_cmp_lt_f32 vcc, %8, %9\n" R8("v_cndmask_b32 %0, %8, %9, vcc\nv_fma_f32 %1, %8, %9, %10\nv_fma_f32 %2, %8, %9, %10\nv_fma_f32 %3, %8, %9, %10\nv_cndmask_b32 %4, %8, %9, vcc\nv_fma_f32 %5, %8, %9, %10\nv_fma_f32 %6, %8, %9, %10\nv_fma_f32 %7, %8, %9, %10\n") : OUT8 : "v"(a), "v"(b), "v"(c) : "vcc");
    else if constexpr (KIND == KX_FMA_MAX) asm volatile(R8("v_fma_f32 %0, %8, %9, %10\nv_max_f32 %1, %8, %9\nv_fma_f32 %2, %8, %9, %10\nv_max_f32 %3, %8, %9\nv_fma_f32 %4, %8, %9, %10\nv_max_f32 %5, %8, %9\nv_fma_f32 %6, %8, %9, %10\nv_max_f32 %7, %8, %9\n") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == KX_READFIRST) { int r0, r1, r2, r3; asm volatile(R8("v_readfirstlane_b32 %0, %4\nv_readfirstlane_b32 %1, %4\nv_readfirstlane_b32 %2, %4\nv_readfirstlane_b32 %3, %4\nv_readfirstlane_b32 %0, %5\nv_readfirstlane_b32 %1, %5\nv_readfirstlane_b32 %2, %5\nv_readfirstlane_b32 %3, %5\n") : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3) : "v"(a), "v"(b)); }
    else if constexpr (KIND == KX_READLANE) { int r0, r1, r2, r3, ln = 5; asm volatile(R8("v_readlane_b32 %0, %4, %6\nv_readlane_b32 %1, %4, %6\nv_readlane_b32 %2, %4, %6\nv_readlane_b32 %3, %4, %6\nv_readlane_b32 %0, %5, %6\nv_readlane_b32 %1, %5, %6\nv_readlane_b32 %2, %5, %6\nv_readlane_b32 %3, %5, %6\n") : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3) : "v"(a), "v"(b), "s"(ln)); }
    else if constexpr (KIND == KX_ADD_CO) asm volatile(R8("v_add_co_u32 %0, vcc, %8, %9\n" "v_add_co_u32 %1, vcc, %8, %9\n" "v_add_co_u32 %2, vcc, %8, %9\n" "v_add_co_u32 %3, vcc, %8, %9\n" "v_add_co_u32 %4, vcc, %8, %9\n" "v_add_co_u32 %5, vcc, %8, %9\n" "v_add_co_u32 %6, vcc, %8, %9\n" "v_add_co_u32 %7, vcc, %8, %9\n") : OUT8 : "v"(a), "v"(b) : "vcc");
    else if constexpr (KIND == KX_ADDC_CO) asm volatile(R8("v_add_co_u32 %0, vcc, %8, %9\nv_addc_co_u32 %1, vcc, %8, %9, vcc\nv_add_co_u32 %2, vcc, %8, %9\nv_addc_co_u32 %3, vcc, %8, %9, vcc\nv_add_co_u32 %4, vcc, %8, %9\nv_addc_co_u32 %5, vcc, %8, %9, vcc\nv_add_co_u32 %6, vcc, %8, %9\nv_addc_co_u32 %7, vcc, %8, %9, vcc\n") : OUT8 : "v"(a), "v"(b) : "vcc");
    else if constexpr (KIND == KX_LSHL_B64) { unsigned long long q0, q1, q2, q3, x = 5; asm volatile(R8("v_lshlrev_b64 %0, 2, %4\nv_lshlrev_b64 %1, 2, %4\nv_lshlrev_b64 %2, 2, %4\nv_lshlrev_b64 %3, 2, %4\nv_lshlrev_b64 %0, 3, %4\nv_lshlrev_b64 %1, 3, %4\nv_lshlrev_b64 %2, 3, %4\nv_lshlrev_b64 %3, 3, %4\n") : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(x)); }
    else if constexpr (KIND == KX_MAD_U64) { unsigned long long q0, q1, q2, q3, x = 5; unsigned long long m0; asm volatile(R8("v_mad_u64_u32 %0, %4, %5, %6, %7\nv_mad_u64_u32 %1, %4, %5, %6, %7\nv_mad_u64_u32 %2, %4, %5, %6, %7\nv_mad_u64_u32 %3, %4, %5, %6, %7\nv_mad_u64_u32 %0, %4, %6, %5, %7\nv_mad_u64_u32 %1, %4, %6, %5, %7\nv_mad_u64_u32 %2, %4, %6, %5, %7\nv_mad_u64_u32 %3, %4, %6, %5, %7\n") : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&s"(m0) : "v"(a), "v"(b), "v"(x)); }
    else if constexpr (KIND == KX_MOV_B64) { unsigned long long q0, q1, q2, q3, x = 5; asm volatile(R8("v_mov_b64 %0, %4\nv_mov_b64 %1, %4\nv_mov_b64 %2, %4\nv_mov_b64 %3, %4\nv_mov_b64 %0, %4\nv_mov_b64 %1, %4\nv_mov_b64 %2, %4\nv_mov_b64 %3, %4\n") : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(x)); }
    else if constexpr (KIND == KX_PK_MUL) { v2f p0, p1, p2, p3, x = {a, b}, y = {b, c}; asm volatile(R8("v_pk_mul_f32 %0, %4, %5\nv_pk_mul_f32 %1, %4, %5\nv_pk_mul_f32 %2, %4, %5\nv_pk_mul_f32 %3, %4, %5\nv_pk_mul_f32 %0, %5, %4\nv_pk_mul_f32 %1, %5, %4\nv_pk_mul_f32 %2, %5, %4\nv_pk_mul_f32 %3, %5, %4\n") : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(x), "v"(y)); }
    else if constexpr (KIND == KX_SALU) { int s0 = 0, s1 = 0, s2 = 0, s3 = 0; asm volatile(R8("s_add_u32 %0, %0, 1\ns_add_u32 %1, %1, 1\ns_add_u32 %2, %2, 1\ns_add_u32 %3, %3, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %1, %1, 1\ns_add_u32 %2, %2, 1\ns_add_u32 %3, %3, 1\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc"); }
    else if constexpr (KIND == KX_MAX_SALU) { int s0 = 0, s1 = 0, s2 = 0, s3 = 0; asm volatile(R8("v_max_f32 %0, %12, %13\ns_add_u32 %8, %8, 1\nv_max_f32 %1, %12, %13\ns_add_u32 %9, %9, 1\nv_max_f32 %2, %12, %13\ns_add_u32 %10, %10, 1\nv_max_f32 %3, %12, %13\ns_add_u32 %11, %11, 1\n") : OUT8, "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc"); }
    else if constexpr (KIND == KY_FMA_ACC) asm volatile(R8("v_fma_f32 %0, %8, %9, %0\n" "v_fma_f32 %1, %8, %9, %1\n" "v_fma_f32 %2, %8, %9, %2\n" "v_fma_f32 %3, %8, %9, %3\n" "v_fma_f32 %4, %8, %9, %4\n" "v_fma_f32 %5, %8, %9, %5\n" "v_fma_f32 %6, %8, %9, %6\n" "v_fma_f32 %7, %8, %9, %7\n") : OUT8P : "v"(a), "v"(b));
    else if constexpr (KIND == KY_FMAC_E64) asm volatile(R8("v_fmac_f32_e64 %0, %8, %9\n" "v_fmac_f32_e64 %1, %8, %9\n" "v_fmac_f32_e64 %2, %8, %9\n" "v_fmac_f32_e64 %3, %8, %9\n" "v_fmac_f32_e64 %4, %8, %9\n" "v_fmac_f32_e64 %5, %8, %9\n" "v_fmac_f32_e64 %6, %8, %9\n" "v_fmac_f32_e64 %7, %8, %9\n") : OUT8P : "v"(a), "v"(b));
    else if constexpr (KIND == KY_CND_E64_VCC) asm volatile("v_cmp_lt_f32 vcc, %8, %9\n" R8("v_cndmask_b32_e64 %0, %8, %9, vcc\n" "v_cndmask_b32_e64 %1, %8, %9, vcc\n" "v_cndmask_b32_e64 %2, %8, %9, vcc\n" "v_cndmask_b32_e64 %3, %8, %9, vcc\n" "v_cndmask_b32_e64 %4, %8, %9, vcc\n" "v_cndmask_b32_e64 %5, %8, %9, vcc\n" "v_cndmask_b32_e64 %6, %8, %9, vcc\n" "v_cndmask_b32_e64 %7, %8, %9, vcc\n") : OUT8 : "v"(a), "v"(b) : "vcc");
    else if constexpr (KIND == KY_MOV_SGPR) { float sg = 1.5f; asm volatile(R8("v_mov_b32 %0, %8\n" "v_mov_b32 %1, %8\n" "v_mov_b32 %2, %8\n" "v_mov_b32 %3, %8\n" "v_mov_b32 %4, %8\n" "v_mov_b32 %5, %8\n" "v_mov_b32 %6, %8\n" "v_mov_b32 %7, %8\n") : OUT8 : "s"(sg)); }
    else if constexpr (KIND == KY_MUL_SGPR) { float sg = 1.5f; asm volatile(R8("v_mul_f32 %0, %8, %9\n" "v_mul_f32 %1, %8, %9\n" "v_mul_f32 %2, %8, %9\n" "v_mul_f32 %3, %8, %9\n" "v_mul_f32 %4, %8, %9\n" "v_mul_f32 %5, %8, %9\n" "v_mul_f32 %6, %8, %9\n" "v_mul_f32 %7, %8, %9\n") : OUT8 : "s"(sg), "v"(b)); }
    else if constexpr (KIND == KY_FMA_SGPR) { float sg = 1.5f; asm volatile(R8("v_fma_f32 %0, %8, %9, %10\n" "v_fma_f32 %1, %8, %9, %10\n" "v_fma_f32 %2, %8, %9, %10\n" "v_fma_f32 %3, %8, %9, %10\n" "v_fma_f32 %4, %8, %9, %10\n" "v_fma_f32 %5, %8, %9, %10\n" "v_fma_f32 %6, %8, %9, %10\n" "v_fma_f32 %7, %8, %9, %10\n") : OUT8 : "s"(sg), "v"(b), "v"(c)); }
    else if constexpr (KIND == KY_ADD_INL) asm volatile(R8("v_add_f32 %0, 1.0, %8\n" "v_add_f32 %1, 1.0, %8\n" "v_add_f32 %2, 1.0, %8\n" "v_add_f32 %3, 1.0, %8\n" "v_add_f32 %4, 1.0, %8\n" "v_add_f32 %5, 1.0, %8\n" "v_add_f32 %6, 1.0, %8\n" "v_add_f32 %7, 1.0, %8\n") : OUT8 : "v"(a));
    else if constexpr (KIND == KY_MUL_INL) asm volatile(R8("v_mul_f32 %0, 2.0, %8\n" "v_mul_f32 %1, 2.0, %8\n" "v_mul_f32 %2, 2.0, %8\n" "v_mul_f32 %3, 2.0, %8\n" "v_mul_f32 %4, 2.0, %8\n" "v_mul_f32 %5, 2.0, %8\n" "v_mul_f32 %6, 2.0, %8\n" "v_mul_f32 %7, 2.0, %8\n") : OUT8 : "v"(a));
    else if constexpr (KIND == KY_SUBREV) asm volatile(VOP2_8("v_subrev_f32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_LSHL_C) asm volatile(R8("v_lshlrev_b32 %0, 2, %8\n" "v_lshlrev_b32 %1, 2, %8\n" "v_lshlrev_b32 %2, 2, %8\n" "v_lshlrev_b32 %3, 2, %8\n" "v_lshlrev_b32 %4, 2, %8\n" "v_lshlrev_b32 %5, 2, %8\n" "v_lshlrev_b32 %6, 2, %8\n" "v_lshlrev_b32 %7, 2, %8\n") : OUT8 : "v"(a));
    else if constexpr (KIND == KY_LSHR) asm volatile(VOP2_8("v_lshrrev_b32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_ADD_E64) asm volatile(VOP2_8("v_add_f32_e64") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_ADD_E64_CLAMP) asm volatile(R8("v_add_f32_e64 %0, %8, %9 clamp\n" "v_add_f32_e64 %1, %8, %9 clamp\n" "v_add_f32_e64 %2, %8, %9 clamp\n" "v_add_f32_e64 %3, %8, %9 clamp\n" "v_add_f32_e64 %4, %8, %9 clamp\n" "v_add_f32_e64 %5, %8, %9 clamp\n" "v_add_f32_e64 %6, %8, %9 clamp\n" "v_add_f32_e64 %7, %8, %9 clamp\n") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_MUL_OMOD) asm volatile(R8("v_mul_f32_e64 %0, %8, %9 mul:2\n" "v_mul_f32_e64 %1, %8, %9 mul:2\n" "v_mul_f32_e64 %2, %8, %9 mul:2\n" "v_mul_f32_e64 %3, %8, %9 mul:2\n" "v_mul_f32_e64 %4, %8, %9 mul:2\n" "v_mul_f32_e64 %5, %8, %9 mul:2\n" "v_mul_f32_e64 %6, %8, %9 mul:2\n" "v_mul_f32_e64 %7, %8, %9 mul:2\n") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_MAX_E64) asm volatile(VOP2_8("v_max_f32_e64") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_MAX_U32) asm volatile(VOP2_8("v_max_u32") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_MBCNT) asm volatile(R8("v_mbcnt_lo_u32_b32 %0, -1, %8\n" "v_mbcnt_lo_u32_b32 %1, -1, %8\n" "v_mbcnt_lo_u32_b32 %2, -1, %8\n" "v_mbcnt_lo_u32_b32 %3, -1, %8\n" "v_mbcnt_lo_u32_b32 %4, -1, %8\n" "v_mbcnt_lo_u32_b32 %5, -1, %8\n" "v_mbcnt_lo_u32_b32 %6, -1, %8\n" "v_mbcnt_lo_u32_b32 %7, -1, %8\n") : OUT8 : "v"(a));
    else if constexpr (KIND == KY_MUL_I24) asm volatile(VOP2_8("v_mul_i32_i24") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_ADD_F16) asm volatile(VOP2_8("v_add_f16") : OUT8 : "v"(a), "v"(b));
    else if constexpr (KIND == KY_FMA_F64) { double q0, q1, q2, q3, x = 1.5; asm volatile(R8("v_fma_f64 %0, %4, %4, %4\nv_fma_f64 %1, %4, %4, %4\nv_fma_f64 %2, %4, %4, %4\nv_fma_f64 %3, %4, %4, %4\nv_fma_f64 %0, %4, %4, %4\nv_fma_f64 %1, %4, %4, %4\nv_fma_f64 %2, %4, %4, %4\nv_fma_f64 %3, %4, %4, %4\n") : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(x)); }
    else if constexpr (KIND == KY_3SRC_DIFF) asm volatile(R8("v_fma_f32 %0, %8, %9, %10\nv_fma_f32 %1, %9, %10, %8\nv_fma_f32 %2, %10, %8, %9\nv_fma_f32 %3, %8, %10, %9\nv_fma_f32 %4, %9, %8, %10\nv_fma_f32 %5, %10, %9, %8\nv_fma_f32 %6, %8, %9, %10\nv_fma_f32 %7, %9, %10, %8\n") : OUT8 : "v"(a), "v"(b), "v"(c));
    else if constexpr (KIND == KZ_ACC_NOCONF) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_fma_f32 v27, v16, v17, v27\n" "v_fma_f32 v31, v16, v17, v31\n" "v_fma_f32 v35, v16, v17, v35\n" "v_fma_f32 v39, v16, v17, v39\n" "v_fma_f32 v43, v16, v17, v43\n" "v_fma_f32 v47, v16, v17, v47\n" "v_fma_f32 v51, v16, v17, v51\n" "v_fma_f32 v55, v16, v17, v55\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_ACC_CONF) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_fma_f32 v24, v16, v17, v24\n" "v_fma_f32 v28, v16, v17, v28\n" "v_fma_f32 v32, v16, v17, v32\n" "v_fma_f32 v36, v16, v17, v36\n" "v_fma_f32 v40, v16, v17, v40\n" "v_fma_f32 v44, v16, v17, v44\n" "v_fma_f32 v48, v16, v17, v48\n" "v_fma_f32 v52, v16, v17, v52\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_NOACC_CONF) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_fma_f32 v27, v16, v17, v20\n" "v_fma_f32 v31, v16, v17, v20\n" "v_fma_f32 v35, v16, v17, v20\n" "v_fma_f32 v39, v16, v17, v20\n" "v_fma_f32 v43, v16, v17, v20\n" "v_fma_f32 v47, v16, v17, v20\n" "v_fma_f32 v51, v16, v17, v20\n" "v_fma_f32 v55, v16, v17, v20\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_NOACC_CONF3) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_fma_f32 v27, v16, v20, v16\n" "v_fma_f32 v31, v16, v20, v16\n" "v_fma_f32 v35, v16, v20, v16\n" "v_fma_f32 v39, v16, v20, v16\n" "v_fma_f32 v43, v16, v20, v16\n" "v_fma_f32 v47, v16, v20, v16\n" "v_fma_f32 v51, v16, v20, v16\n" "v_fma_f32 v55, v16, v20, v16\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_FMAC_NOCONF) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_fmac_f32 v27, v16, v17\n" "v_fmac_f32 v31, v16, v17\n" "v_fmac_f32 v35, v16, v17\n" "v_fmac_f32 v39, v16, v17\n" "v_fmac_f32 v43, v16, v17\n" "v_fmac_f32 v47, v16, v17\n" "v_fmac_f32 v51, v16, v17\n" "v_fmac_f32 v55, v16, v17\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_DST_BANK) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_fma_f32 v24, v16, v17, v18\n" "v_fma_f32 v28, v16, v17, v18\n" "v_fma_f32 v32, v16, v17, v18\n" "v_fma_f32 v36, v16, v17, v18\n" "v_fma_f32 v40, v16, v17, v18\n" "v_fma_f32 v44, v16, v17, v18\n" "v_fma_f32 v48, v16, v17, v18\n" "v_fma_f32 v52, v16, v17, v18\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_ADD_CONF) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_add_f32 v27, v16, v20\n" "v_add_f32 v31, v16, v20\n" "v_add_f32 v35, v16, v20\n" "v_add_f32 v39, v16, v20\n" "v_add_f32 v43, v16, v20\n" "v_add_f32 v47, v16, v20\n" "v_add_f32 v51, v16, v20\n" "v_add_f32 v55, v16, v20\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_ADD_SAME) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_add_f32 v27, v16, v16\n" "v_add_f32 v31, v16, v16\n" "v_add_f32 v35, v16, v16\n" "v_add_f32 v39, v16, v16\n" "v_add_f32 v43, v16, v16\n" "v_add_f32 v47, v16, v16\n" "v_add_f32 v51, v16, v16\n" "v_add_f32 v55, v16, v16\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_ACC_SRC0) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_fma_f32 v24, v24, v17, v18\n" "v_fma_f32 v28, v28, v17, v18\n" "v_fma_f32 v32, v32, v17, v18\n" "v_fma_f32 v36, v36, v17, v18\n" "v_fma_f32 v40, v40, v17, v18\n" "v_fma_f32 v44, v44, v17, v18\n" "v_fma_f32 v48, v48, v17, v18\n" "v_fma_f32 v52, v52, v17, v18\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KZ_ACC_SRC2_B) { asm volatile("v_mov_b32 v16, %0\nv_mov_b32 v17, %1\nv_mov_b32 v18, %2\nv_mov_b32 v20, %0\nv_mov_b32 v21, %1\nv_mov_b32 v22, %2\n" R8("v_fma_f32 v24, v17, v18, v24\n" "v_fma_f32 v28, v17, v18, v28\n" "v_fma_f32 v32, v17, v18, v32\n" "v_fma_f32 v36, v17, v18, v36\n" "v_fma_f32 v40, v17, v18, v40\n" "v_fma_f32 v44, v17, v18, v44\n" "v_fma_f32 v48, v17, v18, v48\n" "v_fma_f32 v52, v17, v18, v52\n") : : "v"(a), "v"(b), "v"(c) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_1) { asm volatile(R8("v_add_f32 v51, v16, v17\n" "v_add_f32 v55, v16, v17\n" "v_add_f32 v59, v16, v17\n" "v_add_f32 v63, v16, v17\n" "v_add_f32 v51, v16, v17\n" "v_add_f32 v55, v16, v17\n" "v_add_f32 v59, v16, v17\n" "v_add_f32 v63, v16, v17\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_2) { asm volatile(R8("v_add_f32 v51, v16, v18\n" "v_add_f32 v55, v16, v18\n" "v_add_f32 v59, v16, v18\n" "v_add_f32 v63, v16, v18\n" "v_add_f32 v51, v16, v18\n" "v_add_f32 v55, v16, v18\n" "v_add_f32 v59, v16, v18\n" "v_add_f32 v63, v16, v18\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_3) { asm volatile(R8("v_add_f32 v51, v16, v19\n" "v_add_f32 v55, v16, v19\n" "v_add_f32 v59, v16, v19\n" "v_add_f32 v63, v16, v19\n" "v_add_f32 v51, v16, v19\n" "v_add_f32 v55, v16, v19\n" "v_add_f32 v59, v16, v19\n" "v_add_f32 v63, v16, v19\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_4) { asm volatile(R8("v_add_f32 v51, v16, v20\n" "v_add_f32 v55, v16, v20\n" "v_add_f32 v59, v16, v20\n" "v_add_f32 v63, v16, v20\n" "v_add_f32 v51, v16, v20\n" "v_add_f32 v55, v16, v20\n" "v_add_f32 v59, v16, v20\n" "v_add_f32 v63, v16, v20\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_5) { asm volatile(R8("v_add_f32 v51, v16, v21\n" "v_add_f32 v55, v16, v21\n" "v_add_f32 v59, v16, v21\n" "v_add_f32 v63, v16, v21\n" "v_add_f32 v51, v16, v21\n" "v_add_f32 v55, v16, v21\n" "v_add_f32 v59, v16, v21\n" "v_add_f32 v63, v16, v21\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_8) { asm volatile(R8("v_add_f32 v51, v16, v24\n" "v_add_f32 v55, v16, v24\n" "v_add_f32 v59, v16, v24\n" "v_add_f32 v63, v16, v24\n" "v_add_f32 v51, v16, v24\n" "v_add_f32 v55, v16, v24\n" "v_add_f32 v59, v16, v24\n" "v_add_f32 v63, v16, v24\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_9) { asm volatile(R8("v_add_f32 v51, v16, v25\n" "v_add_f32 v55, v16, v25\n" "v_add_f32 v59, v16, v25\n" "v_add_f32 v63, v16, v25\n" "v_add_f32 v51, v16, v25\n" "v_add_f32 v55, v16, v25\n" "v_add_f32 v59, v16, v25\n" "v_add_f32 v63, v16, v25\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_12) { asm volatile(R8("v_add_f32 v51, v16, v28\n" "v_add_f32 v55, v16, v28\n" "v_add_f32 v59, v16, v28\n" "v_add_f32 v63, v16, v28\n" "v_add_f32 v51, v16, v28\n" "v_add_f32 v55, v16, v28\n" "v_add_f32 v59, v16, v28\n" "v_add_f32 v63, v16, v28\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_16) { asm volatile(R8("v_add_f32 v51, v16, v32\n" "v_add_f32 v55, v16, v32\n" "v_add_f32 v59, v16, v32\n" "v_add_f32 v63, v16, v32\n" "v_add_f32 v51, v16, v32\n" "v_add_f32 v55, v16, v32\n" "v_add_f32 v59, v16, v32\n" "v_add_f32 v63, v16, v32\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_ADD_32) { asm volatile(R8("v_add_f32 v51, v16, v48\n" "v_add_f32 v55, v16, v48\n" "v_add_f32 v59, v16, v48\n" "v_add_f32 v63, v16, v48\n" "v_add_f32 v51, v16, v48\n" "v_add_f32 v55, v16, v48\n" "v_add_f32 v59, v16, v48\n" "v_add_f32 v63, v16, v48\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMA_2) { asm volatile(R8("v_fma_f32 v51, v16, v17, v18\n" "v_fma_f32 v55, v16, v17, v18\n" "v_fma_f32 v59, v16, v17, v18\n" "v_fma_f32 v63, v16, v17, v18\n" "v_fma_f32 v51, v16, v17, v18\n" "v_fma_f32 v55, v16, v17, v18\n" "v_fma_f32 v59, v16, v17, v18\n" "v_fma_f32 v63, v16, v17, v18\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMA_4) { asm volatile(R8("v_fma_f32 v51, v16, v17, v20\n" "v_fma_f32 v55, v16, v17, v20\n" "v_fma_f32 v59, v16, v17, v20\n" "v_fma_f32 v63, v16, v17, v20\n" "v_fma_f32 v51, v16, v17, v20\n" "v_fma_f32 v55, v16, v17, v20\n" "v_fma_f32 v59, v16, v17, v20\n" "v_fma_f32 v63, v16, v17, v20\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMA_8) { asm volatile(R8("v_fma_f32 v51, v16, v17, v24\n" "v_fma_f32 v55, v16, v17, v24\n" "v_fma_f32 v59, v16, v17, v24\n" "v_fma_f32 v63, v16, v17, v24\n" "v_fma_f32 v51, v16, v17, v24\n" "v_fma_f32 v55, v16, v17, v24\n" "v_fma_f32 v59, v16, v17, v24\n" "v_fma_f32 v63, v16, v17, v24\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMA_9) { asm volatile(R8("v_fma_f32 v51, v16, v17, v25\n" "v_fma_f32 v55, v16, v17, v25\n" "v_fma_f32 v59, v16, v17, v25\n" "v_fma_f32 v63, v16, v17, v25\n" "v_fma_f32 v51, v16, v17, v25\n" "v_fma_f32 v55, v16, v17, v25\n" "v_fma_f32 v59, v16, v17, v25\n" "v_fma_f32 v63, v16, v17, v25\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMA_16) { asm volatile(R8("v_fma_f32 v51, v16, v17, v32\n" "v_fma_f32 v55, v16, v17, v32\n" "v_fma_f32 v59, v16, v17, v32\n" "v_fma_f32 v63, v16, v17, v32\n" "v_fma_f32 v51, v16, v17, v32\n" "v_fma_f32 v55, v16, v17, v32\n" "v_fma_f32 v59, v16, v17, v32\n" "v_fma_f32 v63, v16, v17, v32\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMAC_0) { asm volatile(R8("v_fmac_f32 v51, v16, v16\n" "v_fmac_f32 v55, v16, v16\n" "v_fmac_f32 v59, v16, v16\n" "v_fmac_f32 v63, v16, v16\n" "v_fmac_f32 v51, v16, v16\n" "v_fmac_f32 v55, v16, v16\n" "v_fmac_f32 v59, v16, v16\n" "v_fmac_f32 v63, v16, v16\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMAC_1) { asm volatile(R8("v_fmac_f32 v51, v16, v17\n" "v_fmac_f32 v55, v16, v17\n" "v_fmac_f32 v59, v16, v17\n" "v_fmac_f32 v63, v16, v17\n" "v_fmac_f32 v51, v16, v17\n" "v_fmac_f32 v55, v16, v17\n" "v_fmac_f32 v59, v16, v17\n" "v_fmac_f32 v63, v16, v17\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMAC_8) { asm volatile(R8("v_fmac_f32 v51, v16, v24\n" "v_fmac_f32 v55, v16, v24\n" "v_fmac_f32 v59, v16, v24\n" "v_fmac_f32 v63, v16, v24\n" "v_fmac_f32 v51, v16, v24\n" "v_fmac_f32 v55, v16, v24\n" "v_fmac_f32 v59, v16, v24\n" "v_fmac_f32 v63, v16, v24\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMAC_9) { asm volatile(R8("v_fmac_f32 v51, v16, v25\n" "v_fmac_f32 v55, v16, v25\n" "v_fmac_f32 v59, v16, v25\n" "v_fmac_f32 v63, v16, v25\n" "v_fmac_f32 v51, v16, v25\n" "v_fmac_f32 v55, v16, v25\n" "v_fmac_f32 v59, v16, v25\n" "v_fmac_f32 v63, v16, v25\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMAC_D8) { asm volatile(R8("v_fmac_f32 v24, v16, v17\n" "v_fmac_f32 v32, v16, v17\n" "v_fmac_f32 v40, v16, v17\n" "v_fmac_f32 v48, v16, v17\n" "v_fmac_f32 v24, v16, v17\n" "v_fmac_f32 v32, v16, v17\n" "v_fmac_f32 v40, v16, v17\n" "v_fmac_f32 v48, v16, v17\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KB_FMAC_D9) { asm volatile(R8("v_fmac_f32 v25, v16, v17\n" "v_fmac_f32 v33, v16, v17\n" "v_fmac_f32 v41, v16, v17\n" "v_fmac_f32 v49, v16, v17\n" "v_fmac_f32 v25, v16, v17\n" "v_fmac_f32 v33, v16, v17\n" "v_fmac_f32 v41, v16, v17\n" "v_fmac_f32 v49, v16, v17\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KD_INF) { asm volatile("v_mov_b32 v16, 0x7f800000\nv_mov_b32 v17, 1.0\nv_mov_b32 v18, 1.0\n" R8("v_fma_f32 v27, v16, v17, v18\n" "v_fma_f32 v31, v16, v17, v18\n" "v_fma_f32 v35, v16, v17, v18\n" "v_fma_f32 v39, v16, v17, v18\n" "v_fma_f32 v43, v16, v17, v18\n" "v_fma_f32 v47, v16, v17, v18\n" "v_fma_f32 v51, v16, v17, v18\n" "v_fma_f32 v55, v16, v17, v18\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KD_NAN) { asm volatile("v_mov_b32 v16, 0x7fc00000\nv_mov_b32 v17, 1.0\nv_mov_b32 v18, 1.0\n" R8("v_fma_f32 v27, v16, v17, v18\n" "v_fma_f32 v31, v16, v17, v18\n" "v_fma_f32 v35, v16, v17, v18\n" "v_fma_f32 v39, v16, v17, v18\n" "v_fma_f32 v43, v16, v17, v18\n" "v_fma_f32 v47, v16, v17, v18\n" "v_fma_f32 v51, v16, v17, v18\n" "v_fma_f32 v55, v16, v17, v18\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KD_DENORM) { asm volatile("v_mov_b32 v16, 0x00000100\nv_mov_b32 v17, 1.0\nv_mov_b32 v18, 0x00000200\n" R8("v_fma_f32 v27, v16, v17, v18\n" "v_fma_f32 v31, v16, v17, v18\n" "v_fma_f32 v35, v16, v17, v18\n" "v_fma_f32 v39, v16, v17, v18\n" "v_fma_f32 v43, v16, v17, v18\n" "v_fma_f32 v47, v16, v17, v18\n" "v_fma_f32 v51, v16, v17, v18\n" "v_fma_f32 v55, v16, v17, v18\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KD_RAW1) { asm volatile("v_mov_b32 v16, 1.0\nv_mov_b32 v27, 0\nv_mov_b32 v31, 0\n" R8("v_fmac_f32 v27, v16, v31\n" "v_fmac_f32 v31, v16, v27\n" "v_fmac_f32 v27, v16, v31\n" "v_fmac_f32 v31, v16, v27\n" "v_fmac_f32 v27, v16, v31\n" "v_fmac_f32 v31, v16, v27\n" "v_fmac_f32 v27, v16, v31\n" "v_fmac_f32 v31, v16, v27\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == KD_RAW8) { asm volatile("v_mov_b32 v16, 0\nv_mov_b32 v55, 0\n" R8("v_fmac_f32 v27, v16, v55\n" "v_fmac_f32 v31, v16, v55\n" "v_fmac_f32 v35, v16, v55\n" "v_fmac_f32 v39, v16, v55\n" "v_fmac_f32 v43, v16, v55\n" "v_fmac_f32 v47, v16, v55\n" "v_fmac_f32 v51, v16, v55\n" "v_fmac_f32 v55, v16, v55\n") : : : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); }
    else if constexpr (KIND == K_LDSB128) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f q0, q1, q2, q3, q4, q5, q6, q7;
        const int addr = 0;
        asm volatile(R8("ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:16\nds_read_b128 %2, %8 offset:32\nds_read_b128 %3, %8 offset:48\n"
                        "ds_read_b128 %4, %8 offset:64\nds_read_b128 %5, %8 offset:80\nds_read_b128 %6, %8 offset:96\nds_read_b128 %7, %8 offset:112\ns_waitcnt lgkmcnt(0)\n")
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7) : "v"(addr) : "memory");
    }
}

// `arrive` != nullptr: every workgroup of the grid counts in and waits (bounded) for the others before its waves start their clocks
template <int KIND> __global__ __launch_bounds__(1024) void issue_kernel(float *sink, Rec *rec, unsigned *arrive, unsigned expect) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) lds[i] = 1.0f + i;
    if (arrive && threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < (1 << 22); ++spin)
            if (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= expect) break;
    }
    __syncthreads();
    float a = 1.0f + threadIdx.x * 1e-7f, b = 0.999999f, c = 1e-9f, d = 0.5f;
    block<KIND>(a, b, c, d);                     // the block's code in the cache before the clock starts
    __builtin_amdgcn_s_barrier();
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) { block<KIND>(a, b, c, d); block<KIND>(a, b, c, d); block<KIND>(a, b, c, d); block<KIND>(a, b, c, d); }
    const long long t1 = clock64();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        rec[wave].t0 = t0; rec[wave].t1 = t1;
        rec[wave].hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13
        rec[wave].xcc_id = __builtin_amdgcn_s_getreg(20 | (31 << 11));    // HW_REG_XCC_ID
    }
    if (a + d == 12345.678f) sink[threadIdx.x] = a + d + lds[threadIdx.x & 63];
}

typedef void (*kern_t)(float *, Rec *, unsigned *, unsigned);
template <int K> static void fill(kern_t *t) { t[K] = issue_kernel<K>; if constexpr (K + 1 < K_COUNT) fill<K + 1>(t); }

struct Stat { double cyc_median, cyc_min, cyc_max, wave_fast, wave_slow, skew; int simds, waves_per_simd_min, waves_per_simd_max; };

static Stat reduce(const std::vector<Rec> &r, int n_waves, int count) {
    std::map<unsigned long long, std::vector<int>> by_simd;
    for (int w = 0; w < n_waves; ++w) {
        unsigned long long key = ((unsigned long long)(r[w].xcc_id & 0xf) << 32) | (r[w].hw_id & 0xff30u);
        by_simd[key].push_back(w);
    }
    // per SIMD: all the instructions of its waves over the span from the first wave's start to the last wave's end (the waves start
    // together - `skew` = their latest start minus their first, as a fraction of the span - but the arbiter does not serve them evenly:
    // the fastest / slowest wave's own cycles per instruction are reported beside the SIMD's figure)
    std::vector<double> cyc; double skew = 0, wf = 1e30, ws = 0; int wmin = 1 << 30, wmax = 0;
    for (auto &kv : by_simd) {
        long long lo = -1, first = -1, last = -1;
        for (int w : kv.second) {
            const double per = (double)(r[w].t1 - r[w].t0) / ((double)ITERS * 4 * count);
            wf = std::min(wf, per); ws = std::max(ws, per);
            if (lo < 0 || r[w].t0 > lo) lo = r[w].t0;     // latest start
            if (first < 0 || r[w].t0 < first) first = r[w].t0;
            if (last < 0 || r[w].t1 > last) last = r[w].t1;
        }
        cyc.push_back((double)(last - first) / ((double)ITERS * 4 * count * kv.second.size()));
        skew += (double)(lo - first) / (double)(last - first);
        wmin = std::min(wmin, (int)kv.second.size()); wmax = std::max(wmax, (int)kv.second.size());
    }
    std::sort(cyc.begin(), cyc.end());
    Stat s; s.cyc_median = cyc[cyc.size() / 2]; s.cyc_min = cyc.front(); s.cyc_max = cyc.back();
    s.skew = skew / by_simd.size(); s.wave_fast = wf; s.wave_slow = ws; s.simds = (int)by_simd.size(); s.waves_per_simd_min = wmin; s.waves_per_simd_max = wmax;
    return s;
}

int main(int argc, char **argv) {
    kern_t kern[K_COUNT]; fill<0>(kern);
    for (int k = 0; k < K_COUNT; ++k) kind_count[k] = k == K_MIX_SALU ? 96 : 64;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int ks[] = {1, 2, 4, 6, 8};
    const size_t lds_for_k[] = {81 * 1024, 54 * 1024, 40 * 1024, 26 * 1024, 20 * 1024};
    const int max_waves = n_cu * 8 * 4;
    float *sink; Rec *rec; unsigned *arrive;
    CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&rec, sizeof(Rec) * max_waves)); CK(hipMalloc(&arrive, 4));
    for (int k = 0; k < K_COUNT; ++k) CK(hipFuncSetAttribute((const void *)kern[k], hipFuncAttributeMaxDynamicSharedMemorySize, 81 * 1024));
    std::vector<Rec> h(max_waves);
    printf("device: %s, %d CUs, clock %d kHz; ITERS %d x %d-instruction blocks\n\n", prop.gcnArchName, n_cu, prop.clockRate, ITERS, BLOCK);

    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: ONE workgroup on the chip (256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD); mode 1: k 4-wave workgroups on every CU
        printf(mode == 0 ? "## one workgroup alone on the chip (cycles per wave-instruction per SIMD; in brackets what a single wave of it saw)\n\n| instruction stream | 1 wave/SIMD | 2 | 4 |\n|---|---|---|---|\n"
                         : "## every CU busy, k workgroups of 4 waves per CU (median over the chip's SIMDs [min .. max])\n\n"
                           "| instruction stream | 1 wave/SIMD | 2 | 4 | 6 | 8 |\n|---|---|---|---|---|---|\n");
        for (int kind = 0; kind < K_COUNT; ++kind) {
            printf("| %s |", kind_name[kind]);
            for (int j = 0; j < (mode == 0 ? 3 : 5); ++j) {
                const int k = ks[j];
                const int grid = mode == 0 ? 1 : n_cu * k, threads = mode == 0 ? 256 * k : 256;
                const size_t lds = mode == 0 ? 81 * 1024 : lds_for_k[j];
                const int n_waves = grid * threads / 64;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(arrive, 0, 4));
                    hipLaunchKernelGGL(kern[kind], dim3(grid), dim3(threads), lds, 0, sink, rec, mode == 0 ? nullptr : arrive, (unsigned)grid);
                    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
                }
                CK(hipMemcpy(h.data(), rec, sizeof(Rec) * n_waves, hipMemcpyDeviceToHost));
                Stat s = reduce(h, n_waves, kind_count[kind]);
                if (mode == 0) printf(" **%.2f** (a wave: %.2f-%.2f; skew %.3f) |", s.cyc_median, s.wave_fast, s.wave_slow, s.skew);
                else printf(" **%.2f** [%.2f .. %.2f] (a wave: %.2f-%.2f; %d SIMDs x %d-%d waves; skew %.3f) |", s.cyc_median, s.cyc_min, s.cyc_max, s.wave_fast, s.wave_slow, s.simds, s.waves_per_simd_min, s.waves_per_simd_max, s.skew);
            }
            printf("\n");
        }
        printf("\n");
    }
    return 0;
}
