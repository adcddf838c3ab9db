#!/usr/bin/env python3
"""frames fitted / sec on synthetic 48-view SMPL (6890 v), 100 Adam iterations (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete fit of this rank's frames: re-arm the batch on the device, 100 iterations
of reference smplify/smplify.py:177-213, the final full-mesh forward, the joints, and the copy of
parameters / vertices / joints into pinned host memory (the rtn_dict of smplify.py:216-226).
Inputs (cameras, keypoints, initial estimate, model) are already resident in HBM when the timed
region starts.  N=1 default workload = BASELINE config 2 (1 frame per GPU per step); frames are
independent, so for N>1 every rank fits its own frames (weak scaling, no data-path collective) and
the packed parameters are all-gathered over RCCL once per job (after the K timed steps, inside the timed region).

PyTorch appears here only for torch.distributed (barrier, RCCL all-gather) and - in the clearly
separated `cpu_baseline` leg - to execute the oracle; the measured path is numpy + ctypes + HIP.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from bodyfitting_amd import _lib, native as N, synthetic as S   # noqa: E402

# algorithmic bytes per frame-iteration of the SMPL forward the reference
# evaluates every iteration (posedirs + shapedirs + lbs_weights + J_regressor + J_regressor_extra +
# v_template, each streamed once).
BYTES_PER_FRAME_ITER = 17_114_760 + 826_800 + 661_440 + 661_440 + 248_040 + 82_680   # 19,595,160
# what ONE launch of the full-mesh kernel must read + write per frame (no J_regressor: pre-contracted)
BYTES_MESH_LAUNCH = 17_114_760 + 826_800 + 661_440 + 82_680 + 2 * 82_680
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-per-gpu", type=int, default=1, help="1 = BASELINE config 2; 32 = config 4's per-GPU shard")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--views", type=int, default=48)
    ap.add_argument("--dense", action="store_true", help="full mesh every iteration (reference-literal schedule)")
    ap.add_argument("--graph", action="store_true", help="replay one hipGraph per step instead of issuing the step's four kernels from the host")
    ap.add_argument("--no-graph", action="store_true", help="(default since the step is four kernels and nothing else; kept for old command lines)")
    ap.add_argument("--events", action="store_true", help="keep the HIP event records inside the timed steps (4 per step, ~13 us)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the auxiliary legs (dense schedule, batched shards)")
    ap.add_argument("--cpu-frames", type=int, default=12)
    ap.add_argument("--no-configs", action="store_true", help="skip the config-3 / config-5 legs of `extra`")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) or gloo (validation on a box with fewer GPUs than ranks)")
    ap.add_argument("--same-device", action="store_true", help="every rank uses GPU 0 (validation of the N>1 code path on a 1-GPU box)")
    ap.add_argument("--prewarm-s", type=float, default=0.4, help="untimed spin before the W warmup steps (clock ramp, page-in)")
    return ap.parse_args()


def build_batch(dev, model, frames, n_views):
    probs = [S.make_problem(model, frame=f, n_views=n_views) for f in frames]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem(probs)
    b = N.FrameBatch(dev, len(frames), n_views)
    b.set_cameras(c2w, K)
    b.set_keypoints(kp, ndiv)
    b.set_init(betas, pose)
    return b, probs


def run_steps(batch, steps, iters, flags, finish=None):
    for _ in range(steps):
        batch.fit(iters, flags=flags | _lib.FIT_RESET)       # re-arm + fit + mesh + joints + fetch, one call
    batch.sync()
    if finish is not None:
        finish()                                             # the job's one collective: gather of the fitted parameters


def timed_leg(batch, steps, warmup, iters, flags, barrier=lambda: None, finish=None):
    run_steps(batch, warmup, iters, flags, finish)
    batch.timing_reset()
    barrier()
    batch.sync()
    t0 = time.perf_counter()
    run_steps(batch, steps, iters, flags, finish)
    barrier()
    t1 = time.perf_counter()
    return t1 - t0, batch.timing_sum()


def cpu_baseline(model, gmm, n_frames, n_views, iters):
    """The oracle (torch-CPU restatement of the reference loop: same ops, autograd, Adam) timed on
    this host; 1 thread, which is the faster setting for this dispatch-bound loop."""
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
    return {"value": n_frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n_frames} frames x {n_views} views x {iters} iters, oracle/smplify_oracle.py (torch "
                      f"{torch.__version__} CPU, autograd + Adam), {dt:.1f} s"}


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC passes, or None."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    gather = None
    if a.same_device:
        local = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.dist_backend)
    if a.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    model, gmm = S.make_model("smpl", seed=0), S.make_gmm(seed=0)
    dev = N.DeviceModel(model, gmm, device=local)
    F = a.frames_per_gpu
    frames = list(range(rank * F, rank * F + F))          # distinct frames on every rank
    batch, _ = build_batch(dev, model, frames, a.views)
    a.no_graph = not a.graph
    a.no_events = not a.events and a.no_graph and not a.dense
    flags = _lib.FIT_FETCH | (_lib.FIT_DENSE if a.dense else 0) | (0 if a.no_graph else _lib.FIT_GRAPH) | (_lib.FIT_NOTIME if a.no_events else 0)

    barrier = (lambda: None)
    after = None
    if world > 1 and a.dist_backend == "nccl":
        import torch
        send = torch.empty(F * dev.n_params, dtype=torch.float32, device=f"cuda:{local}")
        recv = torch.empty(world * F * dev.n_params, dtype=torch.float32, device=f"cuda:{local}")

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()

        def after():          # the one collective of the path: final gather of the fitted parameters, once per job
            batch.export_params_dev(send.data_ptr())
            batch.sync()      # (the export runs on the batch's stream, the collective on torch's)
            dist.all_gather_into_tensor(recv, send)
            torch.cuda.synchronize()
        gather = recv
    elif world > 1:
        import torch
        recv = torch.empty(world * F * dev.n_params, dtype=torch.float32)

        def barrier():
            batch.sync()
            dist.barrier()

        def after():          # same gather through host memory (gloo)
            dist.all_gather_into_tensor(recv, torch.from_numpy(batch.get_params().reshape(-1)))
        gather = recv

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < a.prewarm_s:      # not part of W: lets clocks and page-in settle
        run_steps(batch, 5, a.iters, flags)

    wall, ev = timed_leg(batch, a.steps, a.warmup, a.iters, flags, barrier, after)
    if world > 1:
        import torch
        t = torch.tensor([wall], dtype=torch.float64, device=f"cuda:{local}" if a.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # sanity: the timed path really produced a fit (and the gather really carried every rank's frames)
    p = N.split_params(batch.get_params()[0])
    assert np.isfinite(batch.get_params()).all() and abs(float(p["scale"][0]) - 1.0) > 1e-3
    if gather is not None:
        allp = gather.cpu().numpy().reshape(world, F, dev.n_params)
        assert np.array_equal(allp[rank], batch.get_params())

    total_frames = world * F * a.steps
    value = total_frames / wall
    if (not a.no_graph or a.no_events) and not a.dense:
        # inside a hipGraph (or with the event records switched off) the kernels are not bracketed by events: per-kernel
        # device times come from the same steps issued command by command, with events, right after the timed region
        _, ev = timed_leg(batch, max(10, a.steps // 4), 2, a.iters, flags & ~(_lib.FIT_GRAPH | _lib.FIT_NOTIME))
    fit_ms = ev["fit_ms"] / max(ev["calls"], 1)
    mesh_ms = ev["mesh_ms"] / max(ev["calls"], 1)
    traffic = pmc_traffic()
    out = {
        "metric": "frames fitted/sec (100 iters, 48 views, SMPL 6890v)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"{F} frame(s) per GPU per step x {a.views} views x {a.iters} Adam iters, SMPL-shaped "
                                f"synthetic model (6890 v, 24 joints), keypoint-only loss"
                                + (" = BASELINE config 2" if F == 1 else "")
                                + (" = BASELINE config 4 shard" if F == 32 else "")),
                   "frames_per_gpu": F, "views": a.views, "iters": a.iters,
                   "schedule": "dense (full mesh every iteration)" if a.dense else "sparse (gradient-carrying vertices only) + one final full mesh",
                   "submission": ("host-issued: 4 kernels per step (fit, mesh, joints, publish)" + (", no event records in the timed steps" if a.no_events else ""))
                                 if (a.no_graph or a.dense) else "one hipGraph launch per step",
                   "parallelism": f"frames sharded over {world} GPU(s), one RCCL all-gather of the fitted parameters per job" if world > 1 else "1 GPU"},
        "roofline": {
            "bound": "hbm", "kernel": "bf_fit_kernel" if not a.dense else "bf_fit_kernel+bf_mesh_kernel (per iteration)",
            "achieved": BYTES_PER_FRAME_ITER * a.iters * F / (fit_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": BYTES_PER_FRAME_ITER * a.iters * F / (fit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": (traffic or {}).get("bf_fit_kernel_bytes_per_launch"),
            "avg_launch_ms": fit_ms,
            "algorithmic_bytes_per_launch": BYTES_PER_FRAME_ITER * a.iters * F,
            "note": "nominal bytes = SURVEY 8(d) dense forward stream (19,595,160 B per frame-iteration) x iters x frames; "
                    "the sparse kernel keeps its working set in LDS, see DESIGN.md",
        },
        "roofline_mesh": {
            "bound": "hbm", "kernel": "bf_mesh_kernel", "achieved": BYTES_MESH_LAUNCH * F / (mesh_ms * 1e-3) / 1e9 if mesh_ms > 0 else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": BYTES_MESH_LAUNCH * F / (mesh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if mesh_ms > 0 else None,
            "traffic": (traffic or {}).get("bf_mesh_kernel_bytes_per_launch"), "avg_launch_ms": mesh_ms,
            "algorithmic_bytes_per_launch": BYTES_MESH_LAUNCH * F,
            "note": "avg_launch_ms is the HIP-event bracket around a ~6 us kernel and includes the record overhead; rocprofv3 "
                    "--kernel-trace gives 6.1 us for it (profiles/r01_rocprof_final.md), i.e. 3.1 TB/s = 0.38 of peak",
        },
        "device_ms_per_step": {k: ev[k] / max(ev["calls"], 1) for k in ("fit_ms", "mesh_ms", "tail_ms", "total_ms")},
    }

    if rank == 0 and world == 1 and not a.no_extra:
        extra = {}
        # the reference-literal schedule on the same workload
        if not a.dense:
            w, e = timed_leg(batch, max(4, a.steps // 5), 2, a.iters, _lib.FIT_FETCH | _lib.FIT_DENSE)
            n = max(4, a.steps // 5)
            extra["dense_schedule"] = {"value": F * n / w, "unit": "frames/s", "ms_per_step": w / n * 1e3,
                                       "device_fit_ms": e["fit_ms"] / e["calls"]}
        # config 4's per-GPU shard and a CU-filling batch: frames are independent workgroups
        for fb in (32, 256, 1024):
            if fb == F:
                continue
            bb, _ = build_batch(dev, model, list(range(fb)), a.views)
            n = 10
            _, e = timed_leg(bb, n, 2, a.iters, _lib.FIT_FETCH)                                  # (with events: device times)
            w, _ = timed_leg(bb, n, 2, a.iters, _lib.FIT_FETCH | _lib.FIT_RESET | _lib.FIT_NOTIME)
            wg, _ = timed_leg(bb, n, 2, a.iters, _lib.FIT_FETCH | _lib.FIT_GRAPH)        # (pipelined fetch from 8 frames on)
            w = min(w, wg)
            extra[f"batch_{fb}_frames"] = {"value": fb * n / w, "unit": "frames/s", "ms_per_step": w / n * 1e3,
                                           "device_fit_ms": e["fit_ms"] / e["calls"], "device_mesh_ms": e["mesh_ms"] / e["calls"]}
            bb.close()
        # the dense-loss configurations of BASELINE.json (3 and 5 as stated) on this GPU: tools/bench_configs.py
        if not a.no_configs:
            try:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import bench_configs as BC
                extra["config_3"] = BC.cfg3(2)
                extra["config_5"] = BC.cfg5x(2)
            except Exception as exc:                       # (never let a side leg take the headline line down)
                extra["configs_error"] = repr(exc)
        out["extra"] = extra
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, gmm, a.cpu_frames, a.views, a.iters)
    batch.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
