#!/usr/bin/env python3
"""Memory-side traffic of the dense configurations' kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one pass each, of
`tools/bench_configs.py --cfg3` and `--cfg5x`): a markdown table on stdout and the "dense" section of profiles/pmc_traffic.json.
   usage: summarize_dense_traffic.py <dir holding pmc_fetch_cfg3, pmc_write_cfg3, pmc_fetch_cfg5, pmc_write_cfg5> [pmc_traffic.json to update]
FETCH_SIZE is in KB and reports half of a wide streaming read on gfx950 (MI355X_MICROARCH.md, HBM): the read side is doubled."""
import collections, csv, glob, json, os, sys

out = sys.argv[1]
ITERS = {"cfg3": 200, "cfg5": 300}


def pmc(tag):
    g = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in g:
        for r in csv.DictReader(open(f)):
            res[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return res


def short(n):
    return n.split("(")[0].replace("void ", "")[:48]


dense = {}
for cfg in ("cfg3", "cfg5"):
    fe, wr = pmc(f"pmc_fetch_{cfg}"), pmc(f"pmc_write_{cfg}")
    if not fe:
        continue
    fwd = [k for k in fe if "bf_mesh_multi_kernel" in k]
    n_fwd = sum(len(fe[k]["FETCH_SIZE"]) for k in fwd)
    fits = max(n_fwd / ITERS[cfg], 1e-9)          # (every fit iteration of a dense schedule runs one forward mesh launch)
    print(f"## memory-side traffic per launch - {cfg} ({fits:.1f} fits of {ITERS[cfg]} iterations in the pass; FETCH_SIZE doubled)\n")
    print("| kernel | launches / fit | read KB / launch | written KB / launch | MB / fit |\n|---|---|---|---|---|")
    rows, total = {}, 0.0
    for k in sorted(fe, key=lambda k: -sum(fe[k]["FETCH_SIZE"])):
        if "grid_" in k or "rocclr" in k or "contour_kernel" in k and "kp_" not in k:
            continue            # (per-frame input work: scan grids, mask upload, contour extraction - not part of a fit)
        if any(x in k for x in ("bf_disp_", "bf_face_normal", "bf_transpose", "bf_joints")):
            continue            # (the SMPL+D stage, one-off model tables, the result's joints: not the fit either)
        f = fe[k]["FETCH_SIZE"]
        w = wr.get(k, {}).get("WRITE_SIZE", [0.0])
        rd, wt = 2 * sum(f) / len(f) * 1024, sum(w) / max(len(w), 1) * 1024
        per_fit = len(f) / fits
        if "bf_nearest" in k:   # (the SMPL+D stage launches it too: a FIT runs it in its iterations past n // 3 only)
            per_fit = ITERS[cfg] - ITERS[cfg] // 3 - 1
        if per_fit < 0.5:
            continue
        rows[short(k)] = {"launches_per_fit": per_fit, "read_bytes_per_launch": rd, "write_bytes_per_launch": wt}
        total += per_fit * (rd + wt)
        print(f"| `{short(k)}` | {per_fit:.0f} | {rd / 1024:.0f} | {wt / 1024:.0f} | {per_fit * (rd + wt) / 1e6:.1f} |")
    print(f"\ntotal: **{total / 1e9:.3f} GB per fit** ({'1 frame' if cfg == 'cfg3' else '8 frames, one model stream shared'})\n")
    dense[cfg] = {"bytes_per_fit": total, "frames_per_fit": 1 if cfg == "cfg3" else 8, "iters": ITERS[cfg], "kernels": rows}
if len(sys.argv) > 2 and dense:
    try:
        cur = json.load(open(sys.argv[2]))
    except (OSError, ValueError):
        cur = {}
    cur["dense"] = dense
    cur["_note_dense"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/bench_configs.py --cfg3 / --cfg5x (tools/profile_round6.sh); read side doubled "
                          "per the gfx950 correction; the SMPL+D stage's launches of config 5 are in their own rows (disp kernels, and half of bf_nearest_kernel's)")
    json.dump(cur, open(sys.argv[2], "w"), indent=1)
