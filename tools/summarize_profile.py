#!/usr/bin/env python3
"""Markdown summary of tools/profile_round6.sh's output directory (kernel traces + PMC passes)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
PEAK_F32_MFMA_TF = 157.3       # 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz (v_mfma_f32_32x32x2_f32: 4096 flop / 64 clk)


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


def trace(tag):
    f = one(f"{tag}/**/*kernel_trace.csv")
    if not f:
        return {}
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        g[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    return g


def pmc(tag):
    f = one(f"{tag}/**/*counter_collection.csv")
    if not f:
        return {}
    g = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        g[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return g


def short(n):
    return n.split("(")[0].replace("void ", "")[:46]


def bench_line(tag):
    try:
        for line in open(os.path.join(out, tag + ".log")):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return None


print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
for tag, title in (("trace_cfg2", "config 2: 1 frame x 48 views x 100 iterations (bench.py default)"),
                   ("trace_b32", "32 frames per step (config 4's per-GPU shard), as benchmarked: the mesh tail of a step runs on the second stream under the next step's fit kernel"),
                   ("trace_b256", "256 frames per step (config 4 whole), as benchmarked"),
                   ("trace_b32_serial", "32 frames per step, UN-overlapped (--events --resident: one stream)"),
                   ("trace_b256_serial", "256 frames per step, UN-overlapped (--events --resident: one stream)"),
                   ("trace_cfg35", "config 3 (SMPL-X + silhouette, 200 it) and config 5 (8 x SMPL-X + 83,784-triangle scans, 300 + 300 it; closest point in the reference's own arithmetic)"),
                   ("trace_cfg35_nodoor", "configs 3 and 5 with BF_DENSE_PERSISTENT=0 (one fit launch per iteration, no doorbells): the dense kernels' OWN durations - in the product schedule above the forward mesh kernels wait inside the kernel for the resident fit launch's bell"),
                   ("trace_cfg5_fast_rule", "config 5 with BF_NEAREST_RULE=fast (the 2 x 2 rule of rounds 1-3)")):
    g = trace(tag)
    if not g:
        continue
    b = bench_line(tag)
    print(f"## kernel trace - {title}\n")
    if b and "value" in b:
        print(f"bench line under the profiler: {b['value']:.0f} frames/s, {b['ms_per_step']:.4f} ms/step\n")
    print("| kernel | launches | total ms | avg us | median us | min us |\n|---|---|---|---|---|---|")
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v2 = sorted(v)
        print(f"| `{short(k)}` | {len(v)} | {sum(v) / 1e3:.2f} | {sum(v) / len(v):.1f} | {v2[len(v) // 2]:.1f} | {v2[0]:.1f} |")
    print()
for fr in (32, 256):
    m, fe, wr, tr = pmc(f"pmc_mfma_b{fr}"), pmc(f"pmc_fetch_b{fr}"), pmc(f"pmc_write_b{fr}"), trace(f"trace_b{fr}_serial")
    if not m:
        continue
    print(f"## counters - batched mesh path, {fr} frames per launch (durations: the un-overlapped trace)\n")
    print("| kernel | avg us (trace) | MFMA MOPS_F32 / launch | => GFLOP | TFLOP/s | frac of 157 TF | MFMA busy / SQ busy | FETCH_SIZE KB (x2 corrected) | WRITE_SIZE KB | GB/s |\n|---|---|---|---|---|---|---|---|---|---|")
    for k in m:
        if not any(s in k for s in ("poseblend", "epilogue_batch", "pack_feat", "fit_kernel", "joints", "batch32")):
            continue
        c = m[k]
        mean = lambda x: sum(x) / len(x) if x else 0.0       # noqa: E731
        us = mean(tr.get(k, [])) if tr else 0.0
        mops = mean(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", []))
        # MOPS counter unit: 512 flop per count (guide: MOPS = MFMA ops / 512); a 32x32x2 f32 MFMA = 4096 flop = 8 counts per wave
        gflop = mops * 512 / 1e9
        busy = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / max(mean(c.get("SQ_BUSY_CYCLES", [])), 1.0)
        f_kb = mean(fe.get(k, {}).get("FETCH_SIZE", [])) if fe else 0.0
        w_kb = mean(wr.get(k, {}).get("WRITE_SIZE", [])) if wr else 0.0
        tf = gflop / (us * 1e-6) / 1e3 if us else 0.0
        gbs = (2 * f_kb + w_kb) * 1024 / (us * 1e-6) / 1e9 if us else 0.0
        print(f"| `{short(k)}` | {us:.1f} | {mops:.0f} | {gflop:.3f} | {tf:.1f} | {tf / PEAK_F32_MFMA_TF:.2f} | {busy:.3f} | {2 * f_kb:.0f} | {w_kb:.0f} | {gbs:.0f} |")
    print()
for nn_tag in ("pmc_nearest", "pmc_nearest_fast_rule"):
  nn = pmc(nn_tag)
  for k, c in nn.items():
    if "nearest" in k:
        print(f"({nn_tag}: `{short(k)}`)\n")
        w = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])
        n = len(c["SQ_WAVES"])
        print("## counters - closest-point search (config 5: 8 x 10,475 queries against 83,784-triangle scans; one wave per query)\n")
        print("120 fit iterations (80 with the scan loss) + 120 SMPL+D iterations, twice; per query = per wave:\n")
        print("| launches | VALU / query | SALU / query | VMEM reads / query |\n|---|---|---|---|")
        for name, sl in (("all", slice(0, n)), ("first 20 (the scan loss has just switched on)", slice(0, 20)), ("end of the first fit (launches 60-79)", slice(60, 80)), ("last 20 (SMPL+D)", slice(n - 20, n))):
            row = {x: sum(v[sl]) / max(len(v[sl]), 1) / w for x, v in c.items()}
            print(f"| {name} | {row.get('SQ_INSTS_VALU', 0):.0f} | {row.get('SQ_INSTS_SALU', 0):.0f} | {row.get('SQ_INSTS_VMEM_RD', 0):.1f} |")
        print()
m = pmc("pmc_fetch_cfg2")
w = pmc("pmc_write_cfg2")
if m:
    print("## counters - config 2 (per launch; FETCH_SIZE doubled per the gfx950 correction)\n")
    res = {}
    for k in m:
        f_kb = sum(m[k]["FETCH_SIZE"]) / len(m[k]["FETCH_SIZE"])
        w_kb = sum(w.get(k, {}).get("WRITE_SIZE", [0])) / max(len(w.get(k, {}).get("WRITE_SIZE", [0])), 1)
        print(f"* `{short(k)}`: read {2 * f_kb:.1f} KB, written {w_kb:.1f} KB per launch")
        name = "bf_fit_kernel" if "fit_kernel" in k else ("bf_mesh_kernel" if "bf_mesh_kernel" in k else ("bf_joints_kernel" if "joints" in k else None))
        if name:
            res[name + "_bytes_per_launch"] = (2 * f_kb + w_kb) * 1024
            res[name + "_read_bytes_raw"] = f_kb * 1024
    res["_note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round6.sh); read side doubled per the gfx950 correction; per launch = mean over launches; frames per launch = 1"
    with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
