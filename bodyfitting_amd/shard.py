"""Frame sharding over the GPUs of one node.

Frames are independent (the reference fits them in a serial Python loop,
apps/genebody_fitting.py:183-192), so frame f goes to rank f // ceil(F/world) in contiguous blocks,
model and cameras are replicated, and nothing is exchanged during the fit.  The single collective is
the final all-gather of the packed parameters [frames_per_rank, n_params] - RCCL over xGMI when the
process group is "nccl", gloo on CPU in the tests.  torch.distributed is plumbing here; the fit
itself never sees torch.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of frames owned by `rank`; blocks differ by at most one frame."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_frames, world):
    return [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]


def gather_params(local, n_frames, dist=None, device=None, batch=None):
    """All-gather the per-rank parameter blocks into the full [n_frames, n_params] array on every rank.

    `local` is this rank's [frames_here, n_params] numpy block.  With `batch` (a native.FrameBatch)
    and a CUDA `device`, the send buffer is filled device-to-device straight from the library
    (bf_batch_export_params_dev) and the gather runs over RCCL."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local, dtype=np.float32)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = shard_sizes(n_frames, world)
    n_params = np.asarray(local).shape[1] if batch is None else batch.model.n_params
    cap = max(sizes)                                   # ragged shards are padded to the largest
    dev = device if device is not None else "cpu"
    send = torch.zeros(cap * n_params, dtype=torch.float32, device=dev)
    if batch is not None and str(dev).startswith("cuda"):
        batch.export_params_dev(send.data_ptr())
    else:
        send[: sizes[rank] * n_params] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32).reshape(-1))
    recv = torch.empty(world * cap * n_params, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    full = recv.cpu().numpy().reshape(world, cap, n_params)
    return np.concatenate([full[r, : sizes[r]] for r in range(world)], 0)
