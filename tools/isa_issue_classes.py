"""Static VALU mix of a kernel's ISA by the issue classes measured in profiles/r06_issue_rate.md (tools/ubench/issue.hip).

   usage: isa_issue_classes.py file.s kernel-substring [--blocks]

   gfx950, per SIMD, at two or more waves (one wave alone issues one vector instruction per ~4.2-4.7 cycles whatever its class):
     fast   2 cycles   v_fma/add/sub/mul_f32 (VOP2/VOP3 with VGPR, inline-constant or literal operands, neg/abs modifiers), v_fmaak/fmamk,
                       v_fmac_f32, v_mov_b32, v_and/or/xor/not_b32, v_add/sub/subrev_u32, v_ashrrev_i32, v_lshrrev_b32
     slow   4 cycles   everything else that is not transcendental: v_max/min, v_cmp*, v_cndmask, DPP / SDWA forms of anything, any
                       operand in an SGPR, v_lshlrev_b32 (v_lshrrev_b32 and v_ashrrev_i32 are fast), v_mul_*24 / mul_lo / mul_hi, conversions,
                       floor / fract / rndne, every VOP3-only integer op (mad, add3, lshl_add, bfe, perm, and_or ...), med3 / max3 /
                       min3, packed math (v_pk_*), 64-bit moves / shifts / mads, carry-out adds, v_readfirstlane
     trans  8 cycles   v_rcp / rsq / sqrt / exp / log / sin / cos, v_readlane_b32
     vccrun 16 cycles  a VOP2 v_cndmask_b32 ... vcc directly behind another one (the first of a run is a slow-class instruction)
   The two pipes run side by side: a SIMD's issue time is >= max(2 fast, 4 slow + 8 trans + 16 vccrun) summed over its waves, and a
   wave's own stream is >= ~4.2 cycles per vector instruction.  The counts below are STATIC (every instruction once): they give the mix,
   not the dynamic count - scale by the PMC's SQ_INSTS_VALU per wave."""
import re, sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_lshrrev_b32", "v_add_f16", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_readlane_b32", "v_rcp_iflag_f32"}


def classify(line):
    t = line.split(";")[0].strip()
    op = t.split()[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in TRANS:
        return "trans"
    if op.endswith("_dpp") or op.endswith("_sdwa") or "row_" in t or "quad_perm" in t:
        return "slow"
    if base in FAST:
        # an SGPR operand (s12, s[4:5], vcc, exec, m0) moves the instruction to the slow pipe
        operands = t[len(op):]
        if re.search(r"(^|[ ,\-|])(s\d+|s\[\d+:\d+\]|vcc|exec|m0|ttmp\d+)\b", operands):
            return "slow"
        return "fast"
    return "slow"


def main():
    src = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    per_block = "--blocks" in sys.argv
    start = next(i for i, l in enumerate(src) if re.match(r"^[_A-Za-z]\w*:", l) and key in l)
    end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
    tot = {}
    blocks, label = {}, "entry"
    prev_cnd = False
    ops = {}
    for l in src[start:end + 1]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            label = m.group(1); prev_cnd = False
            continue
        if not t or t[0] in ";." or not re.match(r"^[a-z]", t):
            continue
        op = t.split()[0]
        if op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_accvgpr"):
            c = classify(t)
            is_cnd = op == "v_cndmask_b32_e32"
            if is_cnd and prev_cnd:
                c = "vccrun"
            prev_cnd = is_cnd
            if c == "slow":
                b = re.sub(r"_(e32|e64)$", "", op)
                ops[b] = ops.get(b, 0) + 1
        elif op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_nop"):
            c, prev_cnd = "salu", False
        elif op.startswith("ds_"):
            c, prev_cnd = "lds", False
        elif op.split("_")[0] in ("global", "buffer", "flat", "scratch"):
            c, prev_cnd = "vmem", False
        else:
            c = "other"
        tot[c] = tot.get(c, 0) + 1
        d = blocks.setdefault(label, {})
        d[c] = d.get(c, 0) + 1
    v = {k: tot.get(k, 0) for k in ("fast", "slow", "trans", "vccrun")}
    n = sum(v.values())
    print("%s: %d vector instructions (static): fast %d, slow %d, trans %d, vcc-run cndmask %d; salu %d, lds %d, vmem %d" %
          (key, n, v["fast"], v["slow"], v["trans"], v["vccrun"], tot.get("salu", 0), tot.get("lds", 0), tot.get("vmem", 0)))
    fast_c, slow_c = 2 * v["fast"], 4 * v["slow"] + 8 * v["trans"] + 16 * v["vccrun"]
    print("  cycles per SIMD and pass over the code at >= 2 waves: fast pipe %d, slow pipe %d -> max %d = %.2f per vector instruction"
          " (all at 2: %d; all at 4: %d)" % (fast_c, slow_c, max(fast_c, slow_c), max(fast_c, slow_c) / max(n, 1), 2 * n, 4 * n))
    top = sorted(ops.items(), key=lambda kv: -kv[1])[:14]
    print("  slow-pipe opcodes: " + ", ".join("%s %d" % kv for kv in top))
    if per_block:
        for lab, d in blocks.items():
            nn = sum(d.get(k, 0) for k in ("fast", "slow", "trans", "vccrun"))
            if nn >= 24:
                print("   %-12s valu %4d (fast %4d slow %4d trans %3d vccrun %3d) salu %4d lds %3d vmem %3d" %
                      (lab, nn, d.get("fast", 0), d.get("slow", 0), d.get("trans", 0), d.get("vccrun", 0), d.get("salu", 0), d.get("lds", 0), d.get("vmem", 0)))


if __name__ == "__main__":
    main()
