#!/usr/bin/env python3
"""frames fitted / sec on synthetic 48-view SMPL (6890 v), 100 Adam iterations (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W                        one process drives N GPUs (bf_group)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   one rank per GPU (bf_comm)

One "step" = one complete fit of every GPU's frames, as SURVEY.md 8(d) defines the metric ("includes per-frame input upload
and result download"): the NEXT frame's keypoints and initial estimate go from host memory to the device
(bf_batch_stage_inputs - a different frame every step, cycling over 16 frame sets resident on the host; the reference
uploads the keypoints every iteration, loss.py:160, and gets a new frame every call, apps/genebody_fitting.py:183-192), then
100 iterations of reference smplify/smplify.py:177-213, the final full-mesh forward, the joints, and the copy of parameters /
vertices / joints into pinned host memory (the rtn_dict of smplify.py:216-226).  The model and the cameras (shared by a
subject's frames) are resident in HBM.  `extra.resident_inputs` repeats the run without the per-step upload (the same frame
re-fitted, the figure rounds 1-2 reported).  Default workload = BASELINE config 2 (1 frame per GPU per step); frames are
independent, so with N GPUs every GPU fits its own frames (weak scaling, no data-path collective) and the packed
parameters are all-gathered over RCCL once per job - after the K steps, inside the timed region.

No PyTorch on the measured path, in either launch mode: the host side is numpy + ctypes, RCCL is driven from inside
libbodyfit (csrc/group.hip), and the ranks of a launcher-started job find each other through the file system
(bodyfitting_amd/shard.py).  torch is imported only by the clearly separated `cpu_baseline` leg, to execute the oracle.

The K-step bracket (barrier + device sync on both sides, max over ranks) is repeated --repeats times; `value` and
`ms_per_step` are the MEDIAN bracket, the spread is reported next to them.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import statistics
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from bodyfitting_amd import _lib, native as N, shard, synthetic as S   # noqa: E402

# algorithmic bytes per frame-iteration of the SMPL forward the reference evaluates every
# iteration (posedirs + shapedirs + lbs_weights + J_regressor + J_regressor_extra + v_template, each streamed once).
BYTES_PER_FRAME_ITER = 17_114_760 + 826_800 + 661_440 + 661_440 + 248_040 + 82_680   # 19,595,160
# what ONE launch of the full-mesh kernel must read + write per frame (no J_regressor: pre-contracted)
BYTES_MESH_LAUNCH = 17_114_760 + 826_800 + 661_440 + 82_680 + 2 * 82_680
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
ENGINE_CLOCK_GHZ = 2.4       # MI355X peak engine clock (the fit kernel is a one-CU latency chain: cycles are its natural unit)
# dependent-chain lower bound of one fit iteration, DESIGN.md 4.1: 20 dependent LDS round trips (A 6, B 3, D 3, F 3, I 4 + 1) x ~90
# cycles + 5 s_barriers x ~20 + the dependent arithmetic no layout removes (Rodrigues forward / reverse, the chain rounds,
# projection + GMoF, Adam's sqrt / rcp chains: ~1,400 cycles at the measured dependent-issue costs)
CHAIN_BOUND_CYCLES = 20 * 90 + 5 * 20 + 1400


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="how many times the K-step bracket is timed (median reported)")
    ap.add_argument("--frames-per-gpu", type=int, default=1, help="1 = BASELINE config 2; 32 = config 4's per-GPU shard")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--views", type=int, default=48)
    ap.add_argument("--dense", action="store_true", help="full mesh every iteration (reference-literal schedule)")
    ap.add_argument("--graph", action="store_true", help="replay one hipGraph per step instead of issuing the step's four kernels from the host")
    ap.add_argument("--events", action="store_true", help="keep the HIP event records inside the timed steps (4 per step, ~13 us)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the auxiliary legs (dense schedule, batched shards)")
    ap.add_argument("--cpu-frames", type=int, default=12)
    ap.add_argument("--no-configs", action="store_true", help="skip the config-3 / config-5 legs of `extra`")
    ap.add_argument("--prewarm-s", type=float, default=0.4, help="untimed spin before the W warmup steps (clock ramp, page-in)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5),
                    help="2 (default; 4 = 2 with --frames-per-gpu 32): keypoint-only SMPL.  3: SMPL-X, 48 views + 8 silhouettes, 200 iterations.  "
                         "5: SMPL-X + a ~84k-triangle scan per frame, 300 iterations + 300 SMPL+D iterations (default 8 frames per GPU)")
    ap.add_argument("--resident", action="store_true", help="headline WITHOUT the per-step input upload (re-fit one resident frame set, rounds 1-2)")
    ap.add_argument("--force-ranks", action="store_true", help="run the one-process-per-GPU code path (bf_comm: file rendezvous, ncclCommInitRank, barrier, gather) even with WORLD_SIZE 1")
    ap.add_argument("--force-group", action="store_true", help="drive even ONE GPU through bf_group (the N-GPU code path, incl. its RCCL all-gather with one rank): a 1-GPU box's check of what --gpus N runs")
    ap.add_argument("--frame-sets", type=int, default=16, help="distinct frame sets resident on the host that the steps cycle over")
    return ap.parse_args()


_PROBLEMS = {}


def problem(model, frame, n_views, distinct=None):
    """synthetic frame `frame` (cached; `distinct`: frames repeat modulo that many - generating one takes 76 ms on the host, and the
    scaling legs fit up to 2,048 of them)"""
    key = (id(model), frame if distinct is None else frame % distinct, n_views)
    if key not in _PROBLEMS:
        _PROBLEMS[key] = S.make_problem(model, frame=key[1], n_views=n_views)
    return _PROBLEMS[key]


def pack(model, frames, n_views, distinct=None):
    return N.pack_problem([problem(model, f, n_views, distinct) for f in frames])


def build_batch(dev, model, frames, n_views, distinct=None):
    c2w, K, kp, ndiv, betas, pose = pack(model, frames, n_views, distinct)
    b = N.FrameBatch(dev, len(frames), n_views)
    b.set_cameras(c2w, K)
    b.set_keypoints(kp, ndiv)
    b.set_init(betas, pose)
    return b


class FrameFeed:
    """`n_sets` frame sets resident on the HOST, as packed C-contiguous arrays: set j holds frames [j + lo, j + hi) of a pool of
    n_sets - 1 more frames than the job has slots, so every slot meets a different frame at every step of a cycle."""

    def __init__(self, model, lo, hi, n_views, n_sets):
        probs = [S.make_problem(model, frame=f, n_views=n_views) for f in range(lo, hi + n_sets - 1)]
        n = hi - lo
        self.sets = []
        for j in range(n_sets):
            _, _, kp, ndiv, betas, pose = N.pack_problem(probs[j:j + n])
            self.sets.append(tuple(np.ascontiguousarray(a) for a in (kp, ndiv, betas, pose)))
        self.at = 0

    def next(self):
        s = self.sets[self.at % len(self.sets)]
        self.at += 1
        return s


def run_steps(job, steps, iters, flags, finish=None, feed=None):
    for _ in range(steps):
        if feed is not None:
            job.stage_inputs(*feed.next())                   # this step's frames: host -> device, under the fit in flight
        job.fit(iters, flags=flags | _lib.FIT_RESET)         # re-arm + fit + mesh + joints + fetch, one call (per device)
    if finish is not None:
        return finish()                                      # the job's one collective: gather of the fitted parameters
    job.sync()
    return None


def timed_brackets(job, steps, warmup, iters, flags, repeats, barrier, finish=None, reduce_max=lambda x: x, feed=None):
    """W untimed steps, then `repeats` brackets of exactly K steps: barrier + device sync | K steps (+ the gather) | barrier.
    The cyclic garbage collector is off inside the brackets (a generation-2 pass over the synthetic models' objects is 10 - 30 ms: one
    of those inside a 9 ms bracket was the 3 - 8 x outlier of rounds 2 - 3) and runs between them."""
    run_steps(job, warmup, iters, flags, finish, feed)
    walls, last = [], None
    was_on = gc.isenabled()
    try:
        for _ in range(repeats):
            gc.collect()
            gc.disable()
            barrier()
            t0 = time.perf_counter()
            last = run_steps(job, steps, iters, flags, finish, feed)
            barrier()
            walls.append(reduce_max(time.perf_counter() - t0))
            if was_on:
                gc.enable()
    finally:
        if was_on:
            gc.enable()
    return walls, last


def event_leg(batch, steps, iters, flags):
    """per-kernel device times of the same steps from HIP events on the batch's own stream (bf_batch_timing_sum)"""
    run_steps(batch, 2, iters, flags)
    batch.timing_reset()
    run_steps(batch, steps, iters, flags)
    return batch.timing_sum()


def scaling_legs(mode, model, gmm, dev, n_gpus, rank, views, iters, comm, per_gpu=(32, 256), steps=40):
    """Config 2's job at CU-filling sizes, for the N-GPU modes: `per_gpu` frames on every GPU (32 = BASELINE config 4's shard: 32 of a GPU's
    256 CUs busy; 256 = one frame per CU), one RCCL all-gather per job - and THE SAME TOTAL JOB ON ONE GPU next to it, so that the line
    carries weak scaling (per-GPU work fixed) and strong scaling (total work fixed) of the same code.  Every rank runs this (the gather
    is a collective); rank 0 alone runs the one-GPU job while the others wait at the barrier."""
    legs = []
    flags = _lib.FIT_FETCH | _lib.FIT_NOTIME
    for fb in per_gpu:
        total = fb * n_gpus
        try:
            if mode == "group":
                g = shard.Group(model, gmm, n_frames=total, n_views=views, n_devices=n_gpus)
                c2w, K, kp, ndiv, betas, pose = pack(model, range(total), views, distinct=64)
                g.set_cameras(c2w, K); g.set_keypoints(kp, ndiv); g.set_init(betas, pose)
                w, full = timed_brackets(g, steps, 2, iters, flags, 3, g.sync, g.gather_params)
                g.close()
            else:
                lo, hi = shard.shard_range(total, rank, n_gpus)
                bb = build_batch(dev, model, list(range(lo, hi)), views, distinct=64)
                w, full = timed_brackets(bb, steps, 2, iters, flags, 3, comm.barrier, (lambda: comm.gather_params(bb, total)), comm.max)
                bb.close()
            assert full.shape[0] == total and np.isfinite(full).all()
            wm = statistics.median(w)
            leg = {"frames_per_gpu": fb, "total_frames": total, "n_gpus": n_gpus, "value": total * steps / wm, "unit": "frames/s",
                   "ms_per_step": wm / steps * 1e3, "ms_per_step_spread": spread([x / steps * 1e3 for x in w])}
            if rank == 0:
                one = build_batch(dev, model, list(range(total)), views, distinct=64)
                w1, _ = timed_brackets(one, steps, 2, iters, flags, 3, one.sync)
                one.close()
                w1m = statistics.median(w1)
                leg["same_job_on_one_gpu"] = {"value": total * steps / w1m, "unit": "frames/s", "ms_per_step": w1m / steps * 1e3}
                leg["strong_scaling_speedup"] = w1m / wm
            if mode == "ranks":
                comm.barrier()
            legs.append(leg)
        except Exception as exc:                               # (never let a side leg take the headline line down)
            legs.append({"frames_per_gpu": fb, "error": repr(exc)})
            break
    return legs


def batch_result_bytes(F):
    """params + terms + state + joints + vertices of F SMPL frames (the result arena's slices, 256-byte aligned)"""
    up = lambda n: (n + 63) // 64 * 64
    return 4 * (up(F * 86) + up(F * 4) + up(F * (24 * 9 + 24 * 3 + 24 * 3 + 207 + 72 + 10 + 3 + 2)) + up(F * 49 * 3) + up(F * 6890 * 3))


def spread(xs):
    return {"median": statistics.median(xs), "min": min(xs), "max": max(xs), "n": len(xs)}


def cpu_baseline(model, gmm, n_frames, n_views, iters):
    """Two CPU figures, both one thread (the faster setting for this dispatch-bound loop):
    * `value` (kind "port"): oracle/smplify_oracle.py - the torch-CPU restatement of the reference loop (same ops, autograd,
      Adam) - timed HERE, on this host;
    * `reference`: the unmodified reference imported from /root/reference, timed in the build container when the goldens were
      generated (tests/golden/reference_timing.json - the reference cannot travel to the GPU box)."""
    import torch
    from oracle import smplify_oracle as O
    torch.set_num_threads(1)
    gb = S.gmm_buffers(gmm)
    probs = [S.make_problem(model, frame=f, n_views=n_views) for f in range(n_frames)]
    O.fit(model, gb, probs[0], 2)                      # warm the allocator / op caches
    t0 = time.perf_counter()
    for p in probs:
        O.fit(model, gb, p, iters)
    dt = time.perf_counter() - t0
    out = {"value": n_frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{n_frames} frames x {n_views} views x {iters} iters, oracle/smplify_oracle.py (torch "
                     f"{torch.__version__} CPU, autograd + Adam), {dt:.1f} s on this host ({os.cpu_count()} logical CPUs)"}
    try:
        with open(os.path.join(REPO, "tests", "golden", "reference_timing.json")) as f:
            ref = json.load(f)
        out["reference"] = {"value": ref["frames_per_s"], "unit": "frames/s", "cores": ref["threads"], "kind": "reference",
                            "sample": f"{len(ref['wall_s_per_frame'])} frames, {ref['what']}", "host": ref["host"], "where": ref["where"]}
    except (OSError, ValueError, KeyError):
        pass
    return out


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC passes, or None."""
    try:
        with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


# SURVEY.md 8(d), SMPL-X: posedirs 61,090,200 + shapedirs 2,514,000 + lbs_weights 2,304,500 + J_regressor 2,304,500 + v_template 125,700
BYTES_SMPLX_FWD = 68_338_900
BYTES_CFG3_MASK = 15_270_000          # + the every-4th-vertex posedirs columns of the reverse pass while the silhouette loss is active
BYTES_CFG5_ITER = 136_600_000         # forward + full reverse pass (every vertex carries gradient): the iterations with the scan loss on


def main_dense(a):
    """BASELINE configs 3 and 5 as stated, one GPU or sharded over N (frames are independent: the same partition, the same single
    all-gather of the fitted parameters as config 2 / 4).  A step = the per-frame inputs of every GPU's frames go up (config 3:
    8 silhouettes per frame + contour extraction on the device; config 5: the scan's vertices / faces + the closest-point grid
    built on the device), the fit (config 3: 200 iterations, silhouette loss after 66; config 5: 300, scan loss after 100), for
    config 5 the SMPL+D stage (300 iterations), results in pinned host memory."""
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0")) if env_world > 1 else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if env_world > 1 else 0
    mode = "ranks" if (env_world > 1 or a.force_ranks) else ("group" if (a.gpus > 1 or a.force_group) else "single")
    n_gpus = env_world if mode == "ranks" else a.gpus
    have = _lib.load().bf_device_count()
    if (mode == "group" and a.gpus > have) or (mode == "ranks" and local >= have):
        raise SystemExit(f"bench.py: {a.gpus} GPU(s) requested but {have} visible - refusing to run a smaller job under that name")
    cfg = a.config
    F = a.frames_per_gpu if a.frames_per_gpu != 1 or cfg == 3 else 8          # (config 5's default shard: 8 frames per GPU)
    iters = 200 if cfg == 3 else 300
    n_total = F * n_gpus
    model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
    lo, hi = (0, n_total) if mode == "group" else shard.shard_range(n_total, rank, n_gpus)
    mask_frames = list(range(0, a.views, max(1, a.views // 8)))[:8]
    if cfg == 3:
        probs = [S.make_problem_smplx(model, frame=f, n_views=a.views, mask_frames=mask_frames) for f in range(lo, hi)]
        masks = np.stack([np.array(p["masks"]) for p in probs])
        scans_host = None
    else:
        items = [S.make_scan_problem_smplx(model, frame=f, n_views=a.views) for f in range(lo, hi)]
        probs = [p for p, _, _ in items]
        scans_host = [(sv, sf) for _, sv, sf in items]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem(probs)
    comm = group = None
    rccl_ranks = 1
    if mode == "group":
        group = shard.Group(model, gmm, n_frames=n_total, n_views=a.views, n_devices=n_gpus)
        job, batch, dev = group, group.batches[0], group.models[0]
        rccl_ranks = group.comm_size()
        barrier, finish, reduce_max = group.sync, group.gather_params, (lambda x: x)
        device_of = group.device_of_frame
    else:
        dev = N.DeviceModel(model, gmm, device=local)
        job = batch = N.FrameBatch(dev, hi - lo, a.views)
        device_of = lambda f: local
        if mode == "ranks":
            comm = shard.Comm(rank, n_gpus, local)
            rccl_ranks = comm.size()
            barrier, finish, reduce_max = comm.barrier, (lambda: comm.gather_params(batch, n_total)), comm.max
        else:
            barrier, finish, reduce_max = batch.sync, None, (lambda x: x)
    job.set_cameras(c2w, K); job.set_keypoints(kp, ndiv); job.set_init(betas, pose)
    live = {"scans": None, "next": None, "masks": False}
    parts = {"upload_s": 0.0, "fit_s": 0.0, "disp_s": 0.0, "n": 0}

    def make_scans():
        return [N.Scan(sv, sf, device=device_of(i)) for i, (sv, sf) in enumerate(scans_host)]     # upload + grid build on the device

    # the NEXT step's scans are built by a second host thread while this thread queues the current step's ~3,600 launches (the queue is
    # finite: the issuing thread does not get ahead of the device by a whole step): what a capture's frame loop does with its next frame
    import concurrent.futures
    pool = concurrent.futures.ThreadPoolExecutor(1) if cfg == 5 else None

    def step(timed_parts=False):
        t0 = time.perf_counter()
        if cfg == 3:
            # upload + contour extraction on the device(s): the first step's with the step, the later ones' staged under the fit in
            # flight (the capture's frame loop knows its next frame's silhouettes: bf_batch_stage_masks)
            if live["masks"] and not timed_parts:
                job.stage_masks(masks, mask_frames)
            else:
                job.set_masks(masks, mask_frames, None)
            live["masks"] = True
        else:
            # this step's scans: built at the end of the previous step, under its fit and SMPL+D stage (the frame loop of a capture knows
            # its next frame; a scan's device buffers come from the library's block cache, so building one does not wait for the device)
            nxt = live["next"].result() if live["next"] is not None else None
            live["next"] = None
            new = nxt if nxt is not None and not timed_parts else make_scans()
            if nxt is not None and new is not nxt:
                for sc in nxt:
                    sc.close()
            job.set_scans(new)
            if not timed_parts:
                live["next"] = pool.submit(make_scans)                    # the NEXT step's scans, while this step is issued and runs
            if live["scans"]:
                for sc in live["scans"]:
                    sc.close()
            live["scans"] = new
        if timed_parts:
            job.sync(); t1 = time.perf_counter()
        job.fit(iters, flags=_lib.FIT_FETCH | _lib.FIT_RESET)
        if timed_parts:
            job.sync(); t2 = time.perf_counter()
        if cfg == 5:
            job.fit_displacement(iters)
        if timed_parts:
            job.sync(); t3 = time.perf_counter()
            parts["upload_s"] += t1 - t0; parts["fit_s"] += t2 - t1; parts["disp_s"] += t3 - t2; parts["n"] += 1

    def run(n):
        for _ in range(n):
            step()
        if finish is not None:
            return finish()
        job.sync()
        return None

    steps, warmup, repeats = min(a.steps, 5 if cfg == 3 else 3), min(a.warmup, 1), min(a.repeats, 3)
    run(max(1, warmup))
    walls, gathered = [], None
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        gathered = run(steps)
        barrier()
        walls.append(reduce_max(time.perf_counter() - t0))
    wall = statistics.median(walls)
    mine = batch.get_params()
    assert np.isfinite(mine).all()
    if gathered is not None:
        assert gathered.shape == (n_total, dev.n_params) and np.isfinite(gathered).all()
        assert len(np.unique(gathered[:, 4:14], axis=0)) == n_total
    for _ in range(2):
        step(timed_parts=True)
    per = {k: parts[k] / parts["n"] * 1e3 for k in ("upload_s", "fit_s", "disp_s")}
    # device time of the kernel classes of the last dense iteration of one more fit (HIP events on the batch's stream: bf_batch_dense_timing)
    per_class = None
    try:
        batch.dense_timing(True)
        job.fit(iters, flags=_lib.FIT_FETCH | _lib.FIT_RESET); job.sync()
        per_class = batch.dense_timing(False, read=True)
    except Exception as exc:                                   # (a diagnostic: never take the line down)
        per_class = {"error": repr(exc)}
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_configs as BC
    # SURVEY 8(d) by iteration kind, per frame: config 3 = forward every iteration + the sampled vertices' posedirs columns while the
    # silhouette is on; config 5 = the keypoint-only forward for i <= iters // 3, forward + full reverse pass after that
    bytes_iter = (BYTES_SMPLX_FWD * iters + BYTES_CFG3_MASK * (iters - iters // 3 - 1)) if cfg == 3 else BC.cfg5_bytes_per_frame(iters)
    # counter traffic of one fit (FETCH_SIZE doubled + WRITE_SIZE over the fit's kernels, profiles/pmc_traffic.json "dense"), when the
    # committed passes were taken at this shard size
    dense_pmc = ((pmc_traffic() or {}).get("dense") or {}).get(f"cfg{cfg}")
    traffic = dense_pmc["bytes_per_fit"] if dense_pmc and dense_pmc.get("frames_per_fit") == F and dense_pmc.get("iters") == iters else None
    fit_s = per["fit_s"] * 1e-3
    what = {3: f"{F} frame(s) per GPU per step x {a.views} views (+ 8 silhouettes at 512 x 512), SMPL-X-shaped synthetic model (10,475 v, 55 joints, "
               f"135 output joints, body + hands + face keypoints), keypoint + silhouette loss, {iters} Adam iterations = BASELINE config 3; "
               "per-frame mask upload + contour extraction inside the step",
            5: f"{F} frames per GPU per step x {a.views} views, SMPL-X-shaped synthetic model + a {len(scans_host[0][1]) if scans_host else 0}-triangle scan per frame, "
               f"closest-point loss, {iters} Adam iterations, then {iters} SMPL+D iterations = BASELINE config 5"
               + (" shard" if F == 8 else "") + "; per-frame scan upload + grid build inside the step"}[cfg]
    how = {"single": "1 GPU", "group": f"frames sharded over {n_gpus} GPUs driven by one process (bf_group: one host thread per device, ncclCommInitAll), one RCCL all-gather of the fitted parameters per job",
           "ranks": f"frames sharded over {n_gpus} GPUs, one process per GPU (bf_comm: ncclCommInitRank), one RCCL all-gather of the fitted parameters per job"}[mode]
    out = {"metric": f"frames fitted/sec (BASELINE config {cfg})", "value": n_total * steps / wall, "unit": "frames/s", "n_gpus": n_gpus,
           "steps": steps, "warmup": max(1, warmup), "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "repeats": {"brackets": len(walls), "ms_per_step": spread([w / steps * 1e3 for w in walls])},
           "config": {"workload": what, "frames_per_gpu": F, "views": a.views, "iters": iters, "parallelism": how, "launch_mode": mode,
                      "rccl_ranks": rccl_ranks, "torch_on_measured_path": "torch" in sys.modules, "input_upload_in_step": True},
           "ms_per_step_parts_rank0": {"inputs_ms": per["upload_s"], "fit_ms": per["fit_s"], "displacement_ms": per["disp_s"],
                                       "ms_per_fit_iteration": per["fit_s"] / iters,
                                       "ms_per_displacement_iteration": per["disp_s"] / iters if cfg == 5 else None},
           "roofline": {"bound": "hbm", "kernel": "the dense iteration's launch sequence (no single dominant kernel: see profiles/)",
                        "achieved": bytes_iter * F / fit_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_iter * F / fit_s / 1e9 / HBM_PEAK_GBS,
                        "traffic": traffic, "algorithmic_bytes_per_fit": bytes_iter * F,
                        "real_hbm_gbs": traffic / fit_s / 1e9 if traffic else None,
                        "note": "NOMINAL: SURVEY 8(d) bytes per frame-iteration, by iteration kind, x frames over the fit's wall time on rank 0 - every "
                                "frame charged its own stream of the model tensors.  `traffic` = memory-side bytes of one fit by counters "
                                "(profiles/pmc_traffic.json: FETCH_SIZE doubled + WRITE_SIZE, all its kernels): the frames of a shard share ONE "
                                "stream per launch, so the real rate (`real_hbm_gbs`) is what HBM sees; the configuration is bound by "
                                + ("its launch chain's latencies" if cfg == 3 else "the closest-point search's instruction issue (`dominant_kernel`)")
                                + ", not by bytes"}}
    out["device_ms_last_iteration"] = per_class
    out["resident_fit_launch"] = batch.dense_resident()
    if cfg == 5 and per_class and per_class.get("closest_point_search"):
        nvx, s_search = dev.n_verts, per_class["closest_point_search"] * 1e-3
        out["dominant_kernel"] = BC.nearest_dominant(F * nvx, s_search)
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if live["scans"]:
        job.set_scans(None)
        for sc in live["scans"] + (live["next"].result() if live["next"] is not None else []):
            sc.close()
    if comm is not None:
        comm.barrier(); comm.rendezvous.cleanup(); comm.close()
    if group is not None:
        group.close()
    else:
        batch.close(); dev.close()


def main():
    a = parse()
    if a.config != 2:
        return main_dense(a)
    # stdout carries ONE line, the result.  Libraries underneath (RCCL prints a version banner there) write to file descriptor 1
    # whenever they like, so it points at stderr for the whole run and the result goes out through a saved copy at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0")) if env_world > 1 else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if env_world > 1 else 0
    mode = "ranks" if (env_world > 1 or a.force_ranks) else ("group" if (a.gpus > 1 or a.force_group) else "single")
    n_gpus = env_world if mode == "ranks" else a.gpus
    if mode == "ranks" and a.gpus != env_world and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE {env_world}: using {env_world}", file=sys.stderr)
    have = _lib.load().bf_device_count()
    if (mode == "group" and a.gpus > have) or (mode == "ranks" and local >= have):
        raise SystemExit(f"bench.py: {a.gpus} GPU(s) requested but {have} visible - refusing to run a smaller job under that name")

    model, gmm = S.make_model("smpl", seed=0), S.make_gmm(seed=0)
    F = a.frames_per_gpu
    n_total = F * n_gpus
    graph = a.graph
    no_events = not a.events and not graph and not a.dense
    flags = _lib.FIT_FETCH | (_lib.FIT_DENSE if a.dense else 0) | (_lib.FIT_GRAPH if graph else 0) | (_lib.FIT_NOTIME if no_events else 0)

    comm = group = None
    rccl_ranks = 1
    if mode == "group":
        # one process, N devices: model + batch + stream per device inside libbodyfit, ncclCommInitAll
        group = shard.Group(model, gmm, n_frames=n_total, n_views=a.views, n_devices=n_gpus)
        c2w, K, kp, ndiv, betas, pose = pack(model, range(n_total), a.views)
        group.set_cameras(c2w, K); group.set_keypoints(kp, ndiv); group.set_init(betas, pose)
        rccl_ranks = group.comm_size()
        feed_lo, feed_hi = 0, n_total
        job, batch, dev = group, group.batches[0], group.models[0]
        barrier = group.sync
        finish = group.gather_params
        reduce_max = (lambda x: x)
    else:
        dev = N.DeviceModel(model, gmm, device=local)
        lo, hi = shard.shard_range(n_total, rank, n_gpus)          # distinct frames on every rank
        batch = build_batch(dev, model, list(range(lo, hi)), a.views)
        feed_lo, feed_hi = lo, hi
        job = batch
        if mode == "ranks":
            comm = shard.Comm(rank, n_gpus, local)
            rccl_ranks = comm.size()
            barrier = comm.barrier                                 # device idle + all-reduce over the ranks
            finish = (lambda: comm.gather_params(batch, n_total))
            reduce_max = comm.max
        else:
            barrier = batch.sync
            finish = None
            reduce_max = (lambda x: x)

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < a.prewarm_s:      # not part of W: lets clocks and page-in settle
        run_steps(job, 5, a.iters, flags)
        job.sync()

    stream = not a.resident and not a.dense
    feed = FrameFeed(model, feed_lo, feed_hi, a.views, max(2, a.frame_sets)) if stream else None
    # every frame set of the feed goes through the staging path once before the W warm-up steps (first touch of its pages by the
    # copy into pinned memory), untimed like the pre-warm above; and a bracket shorter than ~50 ms is repeated more often - with the
    # driver's --steps 20 a bracket is 9 ms, and one scheduler hiccup is then the whole bracket: the median of 11 does not flip on one
    if feed is not None:
        run_steps(job, len(feed.sets), a.iters, flags, None, feed)
        job.sync()
    est = None
    repeats = a.repeats
    if a.repeats < 11:
        t0 = time.perf_counter(); run_steps(job, 3, a.iters, flags, None, feed); job.sync(); est = (time.perf_counter() - t0) / 3
        est = reduce_max(est)                              # (ranks mode: every rank must take the same decision - the brackets hold barriers)
        if est * a.steps < 0.05:
            repeats = 11
    walls, gathered = timed_brackets(job, a.steps, a.warmup, a.iters, flags, repeats, barrier, finish, reduce_max, feed)
    wall = statistics.median(walls)

    # sanity: the timed path really produced a fit (and the gather really carried every GPU's frames)
    mine = batch.get_params()
    assert np.isfinite(mine).all() and abs(float(N.split_params(mine[0])["scale"][0]) - 1.0) > 1e-3
    if feed is not None:
        # ... of the frames that were staged LAST: the same frames through the synchronous setters give the same bits
        kp_l, nd_l, be_l, po_l = feed.sets[(feed.at - 1) % len(feed.sets)]
        first = 0 if mode != "group" else group.shards[0][1]
        nb0 = batch.F
        chk = N.FrameBatch(dev, nb0, a.views)
        c2w0, K0, _, _, _, _ = pack(model, range(feed_lo + first, feed_lo + first + nb0), a.views)
        chk.set_cameras(c2w0, K0); chk.set_keypoints(kp_l[first:first + nb0], nd_l[first:first + nb0]); chk.set_init(be_l[first:first + nb0], po_l[first:first + nb0])
        chk.fit(a.iters, flags=_lib.FIT_FETCH)
        want = chk.get_params()
        chk.close()
        assert np.array_equal(want, mine), "streamed inputs: the last step's fit differs from a fit of the same frames set synchronously"
    if gathered is not None:
        lo, hi = (group.shards[0][1], group.shards[0][1] + group.shards[0][2]) if group else shard.shard_range(n_total, rank, n_gpus)
        assert gathered.shape == (n_total, dev.n_params) and np.array_equal(gathered[lo:hi], mine)
        assert np.isfinite(gathered).all() and len(np.unique(gathered[:, 4:14], axis=0)) == n_total     # every frame is a different fit

    per_step = [w / a.steps * 1e3 for w in walls]
    value = n_total * a.steps / wall
    ev = event_leg(batch, max(10, a.steps // 4), a.iters, (flags & ~(_lib.FIT_GRAPH | _lib.FIT_NOTIME)))
    fit_ms = ev["fit_ms"] / max(ev["calls"], 1)
    mesh_ms = ev["mesh_ms"] / max(ev["calls"], 1)
    traffic = pmc_traffic() or {}
    fit_traffic = traffic.get("bf_fit_kernel_bytes_per_launch")
    span = None
    if F == 1 and mode == "single":
        try:
            span = batch.mesh_span(100)                    # the kernel's own duration, measured inside it (no event record, no launch gap)
        except _lib.BodyfitError:
            span = None
    if span:
        mesh_ms = span["mean_us"] * 1e-3
    how = {"single": "1 GPU", "group": f"frames sharded over {n_gpus} GPUs driven by one process (bf_group: ncclCommInitAll), one RCCL all-gather "
                                       f"of the fitted parameters per job",
           "ranks": f"frames sharded over {n_gpus} GPUs, one process per GPU (bf_comm: ncclCommInitRank, id exchanged through the file system), "
                    f"one RCCL all-gather of the fitted parameters per job"}[mode]
    out = {
        "metric": "frames fitted/sec (100 iters, 48 views, SMPL 6890v)",
        "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "repeats": {"brackets": len(walls), "ms_per_step": spread(per_step),
                    "value": spread([n_total * a.steps / w for w in walls])},
        "config": {"workload": (f"{F} frame(s) per GPU per step x {a.views} views x {a.iters} Adam iters, SMPL-shaped "
                                f"synthetic model (6890 v, 24 joints), keypoint-only loss, "
                                + ("per-frame input upload inside the step (a different frame's keypoints + initial estimate every step, "
                                   f"{max(2, a.frame_sets)} frame sets on the host) and result download" if stream else
                                   "inputs resident in HBM (the same frames re-fitted every step)")
                                + (" = BASELINE config 2" if F == 1 else "")
                                + (" = BASELINE config 4 shard" if F == 32 else "")),
                   "frames_per_gpu": F, "views": a.views, "iters": a.iters, "input_upload_in_step": stream,
                   "upload_bytes_per_step_per_gpu": (F * (a.views * 25 * 3 + 86 + 1) * 4) if stream else 0,
                   "download_bytes_per_step_per_gpu": int(batch_result_bytes(F)),
                   "schedule": "dense (full mesh every iteration)" if a.dense else "sparse (gradient-carrying vertices only) + one final full mesh",
                   "submission": ("one hipGraph launch per step" if graph and not a.dense else
                                  "host-issued: fit kernel after fit kernel on the batch stream; the next frame's input transfer, then mesh + joints + result hand-over of the frame before, on the second stream under the fit in flight" + (", no timing records in the timed steps" if no_events else "")),
                   "parallelism": how, "launch_mode": mode, "rccl_ranks": rccl_ranks, "torch_on_measured_path": "torch" in sys.modules},
        "roofline": {
            "bound": "hbm", "kernel": "bf_fit_kernel" if not a.dense else "bf_fit_kernel+bf_mesh_kernel (per iteration)",
            "achieved": BYTES_PER_FRAME_ITER * a.iters * F / (fit_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": BYTES_PER_FRAME_ITER * a.iters * F / (fit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": fit_traffic,
            "avg_launch_ms": fit_ms,
            "algorithmic_bytes_per_launch": BYTES_PER_FRAME_ITER * a.iters * F,
            "note": "nominal bytes = SURVEY 8(d) dense forward stream (19,595,160 B per frame-iteration) x iters x frames; "
                    "the sparse kernel keeps its working set in LDS, see DESIGN.md",
            # what the kernel really is: a latency chain on one CU per frame - report it in its own units too
            "real_hbm_gbs": (fit_traffic / (fit_ms * 1e-3) / 1e9) if fit_traffic else None,
            "latency": {"cycles_per_iteration": fit_ms * 1e-3 * ENGINE_CLOCK_GHZ * 1e9 / a.iters, "clock_ghz": ENGINE_CLOCK_GHZ,
                        "dependent_chain_bound_cycles": CHAIN_BOUND_CYCLES,
                        "frac_of_chain_bound": CHAIN_BOUND_CYCLES / (fit_ms * 1e-3 * ENGINE_CLOCK_GHZ * 1e9 / a.iters),
                        "note": "one workgroup = one frame; bound = 20 dependent LDS round trips x ~90 cycles + 5 s_barriers + ~1,400 cycles of dependent arithmetic per iteration (an estimate, DESIGN.md 4.1)"},
        },
        "roofline_mesh": {
            "bound": "hbm", "kernel": "bf_mesh_kernel", "achieved": BYTES_MESH_LAUNCH * F / (mesh_ms * 1e-3) / 1e9 if mesh_ms > 0 else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": BYTES_MESH_LAUNCH * F / (mesh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if mesh_ms > 0 else None,
            "traffic": traffic.get("bf_mesh_kernel_bytes_per_launch"), "avg_launch_ms": mesh_ms,
            "algorithmic_bytes_per_launch": BYTES_MESH_LAUNCH * F,
            "launch_us_min_max": [span["min_us"], span["max_us"]] if span else None,
            "note": ("avg_launch_ms = first workgroup's start to last workgroup's end on the device's 100 MHz clock, 100 launches "
                     "(bf_batch_mesh_span): the kernel alone, comparable with rocprofv3 --kernel-trace (profiles/)") if span else
                    ("avg_launch_ms is the HIP-event bracket around a ~6 us kernel and includes the record overhead; rocprofv3 "
                     "--kernel-trace gives the kernel alone (profiles/)"),
        },
        "device_ms_per_step": {k: ev[k] / max(ev["calls"], 1) for k in ("fit_ms", "mesh_ms", "tail_ms", "total_ms")},
    }

    legs = None
    if not a.no_extra and mode in ("group", "ranks") and F == 1:
        legs = scaling_legs(mode, model, gmm, dev, n_gpus, rank, a.views, a.iters, comm)
    if rank == 0 and not a.no_extra:
        extra = {}
        if legs is not None:
            extra["scaling_legs"] = legs
            extra["scaling_legs_note"] = ("weak scaling = `value` of a leg against n_gpus x the one-GPU value of ITS per-GPU size; strong scaling = "
                                          "strong_scaling_speedup (the same total job on one GPU / on n_gpus).  Expectation (DESIGN 6): a frame is one "
                                          "workgroup on one CU for ~0.44 ms whatever else runs, so 32 per GPU fills 1/8 of a GPU and the total job on one "
                                          "GPU (256 frames at 8 GPUs) takes ~0.6 ms against ~0.46 ms + the gather sharded: ~1.25x from 8 GPUs; 256 per GPU "
                                          "fills every CU and 2,048 frames on one GPU are eight rounds: ~6-7x from 8 GPUs")
        if mode == "single" and stream:
            w, _ = timed_brackets(batch, a.steps, 3, a.iters, flags, 3, batch.sync)
            wm = statistics.median(w)
            extra["resident_inputs"] = {"value": F * a.steps / wm, "unit": "frames/s", "ms_per_step": wm / a.steps * 1e3,
                                        "note": "the same steps without the per-step input upload (one resident frame set re-fitted): what rounds 1-2 reported as the headline"}
        if mode == "single":
            # the reference-literal schedule on the same workload
            if not a.dense:
                n = max(4, a.steps // 5)
                w, _ = timed_brackets(batch, n, 2, a.iters, _lib.FIT_FETCH | _lib.FIT_DENSE, 1, batch.sync)
                e = event_leg(batch, n, a.iters, _lib.FIT_FETCH | _lib.FIT_DENSE)
                extra["dense_schedule"] = {"value": F * n / w[0], "unit": "frames/s", "ms_per_step": w[0] / n * 1e3,
                                           "device_fit_ms": e["fit_ms"] / e["calls"]}
            # config 4's per-GPU shard and a CU-filling batch: frames are independent workgroups
            for fb in (32, 256, 1024):
                if fb == F:
                    continue
                bb = build_batch(dev, model, list(range(fb)), a.views, distinct=64)      # (64 distinct frames, repeated: 76 ms of host time per synthetic frame)
                # (steps per bracket: a bracket ends with the last step's result download - 21 MB at 256 frames, 0.39 ms - which no next fit hides:
                #  with 10 steps it is 7 % of the bracket, the steady state is what a capture sees)
                n = 50 if fb <= 256 else 20
                e = event_leg(bb, 10, a.iters, _lib.FIT_FETCH)                                  # (with events: device times)
                w, _ = timed_brackets(bb, n, 2, a.iters, _lib.FIT_FETCH | _lib.FIT_NOTIME, 3, bb.sync)
                wg, _ = timed_brackets(bb, n, 2, a.iters, _lib.FIT_FETCH | _lib.FIT_GRAPH, 3, bb.sync)   # (pipelined fetch from 8 frames on)
                wh, wgm = statistics.median(w), statistics.median(wg)
                w = min(wh, wgm)
                extra[f"batch_{fb}_frames"] = {"value": fb * n / w, "unit": "frames/s", "ms_per_step": w / n * 1e3,
                                               "submission": "host-issued kernels" if wh <= wgm else "one hipGraph per step",
                                               "ms_per_step_host_issued": wh / n * 1e3, "ms_per_step_graph": wgm / n * 1e3,
                                               "device_fit_ms": e["fit_ms"] / e["calls"], "device_mesh_ms": e["mesh_ms"] / e["calls"]}
                bb.close()
            # the table-driven instance of the fit kernel (fit_kernel<0,0,0,0,false>): what a model gets whose selector vertices have more
            # than 4 bones - here the same model with a fifth, small weight on its 11 selector vertices
            try:
                m5 = dict(model)
                w = np.array(model["lbs_weights"], np.float32).copy()
                for v in np.asarray(model["selector_ids"])[:11]:
                    while (w[v] > 0).sum() < 5:
                        j = int(np.argmax(w[v])); k = int(np.flatnonzero(w[v] == 0)[0])
                        w[v, j] -= 0.03; w[v, k] += 0.03
                m5["lbs_weights"] = w
                dev5 = N.DeviceModel(m5, gmm, device=local)
                b5 = build_batch(dev5, m5, [0], a.views)
                w5, _ = timed_brackets(b5, a.steps, 3, a.iters, _lib.FIT_FETCH | _lib.FIT_NOTIME, 3, b5.sync)
                e5 = event_leg(b5, max(10, a.steps // 4), a.iters, _lib.FIT_FETCH)
                wm5 = statistics.median(w5)
                extra["table_driven_instance"] = {"value": a.steps / wm5, "unit": "frames/s", "ms_per_step": wm5 / a.steps * 1e3,
                                                  "device_fit_ms": e5["fit_ms"] / e5["calls"],
                                                  "cycles_per_iteration": e5["fit_ms"] / e5["calls"] * 1e-3 * ENGINE_CLOCK_GHZ * 1e9 / a.iters,
                                                  "note": "inputs resident; 5 bones on the selector vertices -> the instance without compile-time sizes"}
                b5.close(); dev5.close()
            except Exception as exc:
                extra["table_driven_instance_error"] = repr(exc)
            # the dense-loss configurations of BASELINE.json (3 and 5 as stated) on this GPU: tools/bench_configs.py
            if not a.no_configs:
                try:
                    sys.path.insert(0, os.path.join(REPO, "tools"))
                    import bench_configs as BC
                    extra["config_3"] = BC.cfg3(2)
                    extra["config_5"] = BC.cfg5x(2)
                except Exception as exc:                       # (never let a side leg take the headline line down)
                    extra["configs_error"] = repr(exc)
        elif mode == "group" and F != 32 and legs is None:
            # BASELINE config 4 as stated: 32 frames per GPU, sharded by the same group API
            try:
                g4 = shard.Group(model, gmm, n_frames=32 * n_gpus, n_views=a.views, n_devices=n_gpus)
                c2w, K, kp, ndiv, betas, pose = pack(model, range(32 * n_gpus), a.views)
                g4.set_cameras(c2w, K); g4.set_keypoints(kp, ndiv); g4.set_init(betas, pose)
                n = 10
                w, full = timed_brackets(g4, n, 2, a.iters, _lib.FIT_FETCH | _lib.FIT_NOTIME, 3, g4.sync, g4.gather_params)
                assert full.shape[0] == 32 * n_gpus and np.isfinite(full).all()
                wm = statistics.median(w)
                extra["config_4"] = {"workload": f"{32 * n_gpus} frames = 32 per GPU x {n_gpus} GPUs, 48 views, 100 iters, one RCCL all-gather per job",
                                     "value": 32 * n_gpus * n / wm, "unit": "frames/s", "ms_per_step": wm / n * 1e3,
                                     "ms_per_step_spread": spread([x / n * 1e3 for x in w])}
                g4.close()
            except Exception as exc:
                extra["config_4_error"] = repr(exc)
        if extra:
            out["extra"] = extra
    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, gmm, a.cpu_frames, a.views, a.iters)
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.barrier()
        comm.rendezvous.cleanup()
        comm.close()
    if group is not None:
        group.close()
    else:
        batch.close()
        dev.close()


if __name__ == "__main__":
    main()
