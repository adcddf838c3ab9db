"""Static instruction counts of a kernel's ISA between its s_barriers, per basic block (bring-up helper).
   usage: isa_phases.py file.s kernel-substring [--loop-only]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end + 1]
def kind(op):
    if op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"): return "bperm"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ldsr"
    if op.startswith("ds_"): return "ldsw"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "bar"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "rdlane"
    if op.startswith("v_pk_"): return "vpk"
    if op.startswith("v_"): return "valu"
    if op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    return "other"
seg, segs, blocks, cur_label = {}, [], [], "entry"
bar = 0
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): 
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur_label = m.group(1)
        continue
    op = t.split()[0]
    if not re.match(r"^[a-z]", op): continue
    k = kind(op)
    extra = "dpp" if ("row_" in t or "quad_perm" in t or "dpp" in op) else None
    key2 = (bar, cur_label)
    d = seg.setdefault(key2, {})
    d[k] = d.get(k, 0) + 1
    if extra: d["dpp"] = d.get("dpp", 0) + 1
    if k == "bar":
        bar += 1
order = []
for k2 in seg:
    order.append(k2)
curbar = -1
for (b, lab) in order:
    d = seg[(b, lab)]
    n = sum(v for kk, v in d.items() if kk != "dpp")
    if b != curbar:
        print("---- after barrier", b)
        curbar = b
    print("  %-12s n=%4d  %s" % (lab, n, " ".join("%s=%d" % kv for kv in sorted(d.items()))))
