#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a markdown summary and
pmc_traffic.json.  usage: summarize_profile.py <prof dir>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


def kernel_stats(d):
    rows = []
    for f in find(d, "*kernel_stats.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def kernel_trace(d):
    per = defaultdict(list)
    for f in find(d, "*kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                              r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"),
                                              r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size")))
    return per


def pmc(d, counter):
    per = defaultdict(list)
    for f in find(d, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter:
                    per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def main():
    d = sys.argv[1]
    out = []
    for leg in ("trace", "trace_dense"):
        out.append(f"## rocprofv3 --kernel-trace --stats: {leg} (`bench.py --steps 30 --warmup 5"
                   + (" --dense" if leg.endswith("dense") else "") + "`)\n")
        out.append("| kernel | calls | total ms | avg us | min us | max us | % |")
        out.append("|---|---|---|---|---|---|---|")
        for r in kernel_stats(os.path.join(d, leg)):
            out.append("| {} | {} | {:.3f} | {:.2f} | {:.2f} | {:.2f} | {} |".format(
                r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
        out.append("")
        tr = kernel_trace(os.path.join(d, leg))
        out.append("| kernel | launches | median us | VGPR | AGPR | SGPR | LDS B | grid | wg |")
        out.append("|---|---|---|---|---|---|---|---|---|")
        for k, v in sorted(tr.items()):
            durs = sorted(x[0] for x in v)
            out.append(f"| {k[:60]} | {len(v)} | {durs[len(durs) // 2] / 1e3:.2f} | {v[0][1]} | {v[0][2]} | {v[0][3]} | {v[0][4]} | {v[0][5]} | {v[0][6]} |")
        out.append("")
    traffic = {}
    fetch, write = defaultdict(list), defaultdict(list)
    for leg, keep in (("sparse", ("fit_kernel",)), ("dense", ("bf_mesh_kernel", "bf_joints_kernel"))):
        f = pmc(os.path.join(d, "pmc_fetch_" + leg), "FETCH_SIZE")
        w = pmc(os.path.join(d, "pmc_write_" + leg), "WRITE_SIZE")
        for k in set(f) | set(w):
            if any(x in k for x in keep):
                fetch[k] += f.get(k, [])
                write[k] += w.get(k, [])
    out.append("## HBM traffic per launch (separate --pmc passes: fit kernel from `bench.py --steps 3` = 100-iteration "
               "launches, mesh/joints kernels from `--dense --iters 10`)\n")
    out.append("FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, i.e. exactly 1/2 of a "
               "wide coalesced read stream (MI355X_MICROARCH.md, HBM) - the corrected column doubles it.\n")
    out.append("| kernel | launches | FETCH_SIZE KiB (raw, mean) | read bytes (x2 corrected) | WRITE_SIZE KiB (mean) | write bytes | total bytes / launch |")
    out.append("|---|---|---|---|---|---|---|")
    for k in sorted(set(fetch) | set(write)):
        fr = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
        wr = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
        rb, wb = fr * 1024 * 2, wr * 1024
        out.append(f"| {k[:60]} | {len(fetch.get(k, []))} | {fr:.1f} | {rb:.0f} | {wr:.1f} | {wb:.0f} | {rb + wb:.0f} |")
        short = k.split("(")[0].strip()
        if "fit_kernel" in short:
            short = "bf_fit_kernel"
        traffic[f"{short}_bytes_per_launch"] = rb + wb
        traffic[f"{short}_read_bytes_raw"] = fr * 1024
    print("\n".join(out))
    traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_gpu.sh); read side doubled per "
                        "the gfx950 correction; per launch = mean over launches; frames per launch = 1")
    with open(os.path.join(d, "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)


if __name__ == "__main__":
    main()
