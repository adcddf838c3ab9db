#!/usr/bin/env python3
"""Timings of the dense-loss paths on one GPU (not part of the bench.py contract): BASELINE config 3
(SMPL-X, 48 views, keypoint + silhouette loss, 200 iterations) and a config-5-shaped shard (frames with a scan:
closest-point loss for 300 iterations, then the SMPL+D stage).  Synthetic data (bodyfitting_amd/synthetic.py),
one JSON object per line.   usage: python tools/bench_configs.py [--cfg3] [--cfg5] [--frames N] [--reps R]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402


# algorithmic bytes per frame-iteration (the same constants bench.py --config 3 | 5 prices its roofline with)
BYTES_SMPLX_FWD = 61_090_200 + 2_514_000 + 2_304_500 + 2_304_500 + 125_700          # 68,338,900: the SMPL-X forward's tensors, each once
BYTES_CFG3_MASK = 15_270_000                                                       # + the sampled vertices' posedirs columns when the silhouette loss is on
BYTES_CFG5_ITER = 136_600_000                                                      # forward + full reverse pass
HBM_PEAK_GBS = 8000.0
# bf_nearest_kernel (reference rule): instructions per query-wave in config 5's loop by PMC (SQ_INSTS_VALU, SQ_INSTS_SALU / queries over
# 120 fit + 120 SMPL+D iterations; profiles/r06_rocprof_summary.md.  Round 5: 894 / 502; the list map's uniform branches moved 42 to the scalar side).
NEAREST_VALU_PER_QUERY = 852
NEAREST_SALU_PER_QUERY = 519
# What a gfx950 SIMD issues (profiles/r06_issue_rate.md, tools/ubench/issue.hip): cycles per wave64 instruction per SIMD at two or more
# waves, by class - the FAST class (fma / add / mul / mov / logic on vector registers) beside the SLOW class (compares, selects, max / min,
# DPP, anything with a scalar operand, ...; transcendentals and v_readlane 8.1), the scalar unit a third pipe.  Rounds 3-5 priced every
# vector instruction at 4.
ISSUE_FAST, ISSUE_SLOW, ISSUE_TRANS, ISSUE_VCCRUN, ISSUE_SALU = 2.1, 4.1, 8.1, 16.0, 4.3
SIMDS, CLOCK_HZ = 1024, 2.4e9
# the kernel's static ISA mix (tools/isa_issue_classes.py on the product build): fractions of its vector instructions
NEAREST_MIX = {"fast": 344 / 947, "slow": 540 / 947, "trans": 59 / 947, "vccrun": 4 / 947}


def nearest_issue_bound(queries):
    """-> (seconds, which pipe): the longest of the three issue pipes over a launch's queries spread evenly over the SIMDs"""
    m = NEAREST_MIX
    pipes = {"slow vector pipe": NEAREST_VALU_PER_QUERY * (m["slow"] * ISSUE_SLOW + m["trans"] * ISSUE_TRANS + m["vccrun"] * ISSUE_VCCRUN),
             "fast vector pipe": NEAREST_VALU_PER_QUERY * m["fast"] * ISSUE_FAST,
             "scalar pipe": NEAREST_SALU_PER_QUERY * ISSUE_SALU}
    which = max(pipes, key=pipes.get)
    return queries * pipes[which] / SIMDS / CLOCK_HZ, which, {k: round(v) for k, v in pipes.items()}


def nearest_dominant(queries, search_s):
    bound_s, which, pipes = nearest_issue_bound(queries)
    return {"name": "bf_nearest_kernel", "bound": "issue: " + which, "queries_per_launch": queries,
            "valu_per_query_wave": NEAREST_VALU_PER_QUERY, "salu_per_query_wave": NEAREST_SALU_PER_QUERY,
            "cycles_per_query_by_pipe": pipes, "ms_per_launch": search_s * 1e3, "issue_bound_ms": bound_s * 1e3,
            "frac": bound_s / search_s if search_s > 0 else None,
            "issue_cycles_per_instruction": {"fast": ISSUE_FAST, "slow": ISSUE_SLOW, "transcendental": ISSUE_TRANS, "salu": ISSUE_SALU},
            "note": "one query per wave, 6 waves per SIMD: frac = queries x the busiest issue pipe's cycles per query (instruction counts by PMC, "
                    "class mix from the ISA, cycles per class MEASURED at this occupancy: profiles/r06_issue_rate.md) / (1,024 SIMDs x 2.4 GHz) / "
                    "the search's device time in the last dense iteration (HIP events).  Rounds 3-5 charged 4 cycles for every vector instruction "
                    "('0.84-0.90'); at 2 for every one it would be half of that"}


# config 5, SURVEY 8(d) by iteration KIND: the scan loss is on for i > iters // 3 (smplify.py:205); before that an iteration is the
# keypoint-only SMPL-X forward
def cfg5_bytes_per_frame(iters):
    kp_only = iters // 3 + 1
    return kp_only * BYTES_SMPLX_FWD + (iters - kp_only) * BYTES_CFG5_ITER


def dense_traffic(cfg, frames, iters):
    """memory-side bytes of one fit by counters (profiles/pmc_traffic.json "dense": FETCH_SIZE doubled + WRITE_SIZE over the fit's kernels,
    tools/profile_round6.sh), if the committed passes were taken at this shard size; else None"""
    try:
        with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
            d = (json.load(f).get("dense") or {}).get(cfg)
    except (OSError, ValueError):
        return None
    return d["bytes_per_fit"] if d and d.get("frames_per_fit") == frames and d.get("iters") == iters else None


def nominal_roofline(bytes_per_fit, fit_s, what, traffic=None):
    gbs = bytes_per_fit / fit_s / 1e9
    return {"bound": "hbm", "kernel": what, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "real_hbm_gbs": traffic / fit_s / 1e9 if traffic else None, "algorithmic_bytes_per_fit": bytes_per_fit,
            "note": "NOMINAL: SURVEY 8(d) bytes per frame-iteration, by iteration kind, x frames over the fit's wall time (every frame charged its own "
                    "stream of the model tensors); `traffic` = the fit's memory-side bytes by counters (profiles/pmc_traffic.json), `real_hbm_gbs` = "
                    "traffic / time: what HBM sees.  The dense configurations are bound by their launch chain's latencies (config 3) and the "
                    "closest-point search's instruction issue (config 5), not by bytes; device_ms_last_iteration has the split"}


def timed(fn, reps):
    fn()                                   # warm (allocations, first-use tables)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def cfg3(reps, n_views=48, iters=200, mask_views=8):
    model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
    dev = N.DeviceModel(model, gmm, device=0)
    mask_frames = list(range(0, n_views, max(1, n_views // mask_views)))[:mask_views]
    prob = S.make_problem_smplx(model, frame=0, n_views=n_views, mask_frames=mask_frames)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, n_views)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    masks = np.array(prob["masks"])
    t0 = time.perf_counter()
    b.set_masks(masks[None], mask_frames, None)            # upload + contour extraction on the device
    t_masks = time.perf_counter() - t0

    def run():
        b.reset(); b.fit(iters); b.sync()
    dt = timed(run, reps)
    b.dense_timing(True); run(); per_class = b.dense_timing(False, read=True)
    out = {"config": "cfg3: 1 frame x %d views, SMPL-X (10475 v, 55 joints, 135 loss joints), keypoint + silhouette loss "
                     "(%d mask views), %d iterations" % (n_views, len(mask_frames), iters),
           "frames_per_s": 1.0 / dt, "ms_per_fit": dt * 1e3, "ms_per_iteration": dt * 1e3 / iters,
           "ms_mask_upload_and_contours": t_masks * 1e3,
           "roofline": nominal_roofline(BYTES_SMPLX_FWD * iters + BYTES_CFG3_MASK * (iters - iters // 3 - 1), dt, "config 3's dense iteration (forward mesh, keypoints + contours with fixed-point gradient sums, reverse mesh, reduce)", dense_traffic("cfg3", 1, iters)),
           "device_ms_last_iteration": per_class, "resident_fit_launch": b.dense_resident()}
    b.close(); dev.close()
    return out


def cfg5(reps, frames=8, n_views=8, iters=300, disp_iters=100):
    model, gmm = S.make_model("smpl", seed=0), S.make_gmm(seed=0)
    dev = N.DeviceModel(model, gmm, device=0)
    items = [S.make_scan_problem(model, frame=f, n_views=n_views) for f in range(frames)]
    scans = [N.Scan(sv, sf) for _, sv, sf in items]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([p for p, _, _ in items])
    b = N.FrameBatch(dev, frames, n_views)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans(scans)

    def run_fit():
        b.reset(); b.fit(iters); b.sync()
    dt = timed(run_fit, reps)

    def run_disp():
        b.fit_displacement(disp_iters); b.sync()
    dd = timed(run_disp, reps)
    out = {"config": "cfg5-shaped shard: %d frames x %d views, SMPL (6890 v) with a 6890-vertex scan each, closest-point loss, "
                     "%d iterations, then %d SMPL+D iterations" % (frames, n_views, iters, disp_iters),
           "frames_per_s_fit": frames / dt, "ms_per_fit": dt * 1e3, "ms_per_iteration": dt * 1e3 / iters,
           "ms_per_displacement_iteration": dd * 1e3 / disp_iters}
    b.close()
    for s in scans:
        s.close()
    dev.close()
    return out


def cfg5x(reps, frames=8, n_views=48, iters=300, disp_iters=300):
    """BASELINE config 5 as stated: SMPL-X, a ~100k-triangle scan per frame, closest-point loss (300 iterations), then
    the SMPL+D stage (300 iterations); one GPU's shard of 8 frames.  The scan upload + grid build (once per frame,
    utils/mesh_grid_searcher.py:56-79) is timed separately and included in the end-to-end figure."""
    model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
    dev = N.DeviceModel(model, gmm, device=0)
    items = [S.make_scan_problem_smplx(model, frame=f, n_views=n_views) for f in range(frames)]
    t0 = time.perf_counter()
    scans = [N.Scan(sv, sf) for _, sv, sf in items]
    t_scan = time.perf_counter() - t0
    for s in scans:
        s.close()
    t0 = time.perf_counter()
    scans = [N.Scan(sv, sf) for _, sv, sf in items]
    t_scan = min(t_scan, time.perf_counter() - t0)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([p for p, _, _ in items])
    b = N.FrameBatch(dev, frames, n_views)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans(scans)

    def run_fit():
        b.reset(); b.fit(iters); b.sync()
    dt = timed(run_fit, reps)

    b.dense_timing(True); run_fit(); per_class = b.dense_timing(False, read=True)

    def run_disp():
        b.fit_displacement(disp_iters); b.sync()
    dd = timed(run_disp, reps)
    total = t_scan + dt + dd
    nv = model["v_template"].shape[0]
    search_s = per_class["closest_point_search"] * 1e-3
    out = {"config": "cfg5: %d frames x %d views, SMPL-X (%d v) with a %d-triangle scan each, closest-point loss, %d iterations, "
                     "then %d SMPL+D iterations" % (frames, n_views, nv, len(items[0][2]), iters, disp_iters),
           "frames_per_s_end_to_end": frames / total, "ms_scan_upload_and_grid": t_scan * 1e3, "ms_per_fit": dt * 1e3,
           "ms_per_iteration": dt * 1e3 / iters, "ms_displacement_stage": dd * 1e3,
           "ms_per_displacement_iteration": dd * 1e3 / disp_iters,
           "roofline": nominal_roofline(cfg5_bytes_per_frame(iters) * frames, dt, "config 5's fit (keypoint-only iterations, then forward mesh, closest-point search, point-cloud loss, reverse mesh, reduce)", dense_traffic("cfg5", frames, iters)),
           "device_ms_last_iteration": per_class, "resident_fit_launch": b.dense_resident(),
           # the iteration's dominant kernel is not bound by bytes: one query per wave, ~894 vector + ~502 scalar instructions per query
           "dominant_kernel": nearest_dominant(frames * nv, search_s)}
    b.close()
    for s in scans:
        s.close()
    dev.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg3", action="store_true"); ap.add_argument("--cfg5", action="store_true")
    ap.add_argument("--cfg5x", action="store_true")
    ap.add_argument("--frames", type=int, default=8); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--iters", type=int, default=300, help="cfg5x: iterations of the fit and of the SMPL+D stage")
    a = ap.parse_args()
    if not (a.cfg3 or a.cfg5 or a.cfg5x):
        a.cfg3 = a.cfg5 = True
    if a.cfg3:
        print(json.dumps(cfg3(a.reps)), flush=True)
    if a.cfg5:
        print(json.dumps(cfg5(a.reps, frames=a.frames)), flush=True)
    if a.cfg5x:
        print(json.dumps(cfg5x(a.reps, frames=a.frames, iters=a.iters, disp_iters=a.iters)), flush=True)
