"""The N>1 path on CPU.  The partition and the gather's pad / unpack bookkeeping are host arithmetic inside libbodyfit
(bf_shard_range / bf_shard_capacity / bf_shard_unpack, csrc/group.hip) - the same functions bf_group_gather_params and
bf_comm_gather_params use around their ncclAllGather.  Here two CPU processes run that bookkeeping with gloo standing in
for RCCL as the transport, and exchange the RCCL-id-sized blob through the file rendezvous the ranks use on the GPU box."""
import os
import socket

import numpy as np
import pytest

from bodyfitting_amd import shard


def test_partition_covers_every_frame_once():
    for n, w in ((256, 8), (64, 8), (7, 2), (3, 4), (1, 2), (32, 1), (9, 8)):
        blocks = [shard.shard_range(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [b[1] - b[0] for b in blocks]
        assert max(sizes) - min(sizes) <= 1 and max(sizes) == shard.shard_capacity(n, w)
        assert sizes == sorted(sizes, reverse=True)             # earlier shards take the remainder
    # BASELINE configs 4 and 5: 256 -> 32 per GPU, 64 -> 8 per GPU
    assert shard.shard_sizes(256, 8) == [32] * 8 and shard.shard_sizes(64, 8) == [8] * 8


def test_unpack_drops_the_padding_of_ragged_shards():
    n, w, width = 7, 2, 5
    cap = shard.shard_capacity(n, w)
    full = np.arange(n * width, dtype=np.float32).reshape(n, width)
    gathered = np.full((w, cap, width), -1.0, np.float32)
    for r in range(w):
        lo, hi = shard.shard_range(n, r, w)
        gathered[r] = shard.pad_block(full[lo:hi], n, w)
    np.testing.assert_array_equal(shard.unpack(gathered, n, w), full)


def _params(frames, width=86):
    return np.stack([np.arange(width, dtype=np.float32) + 1000.0 * f for f in frames]) if len(frames) else np.zeros((0, width), np.float32)


def _worker(rank, world, port, n_frames, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="pytest")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # what a rank does around the collective (csrc/group.hip: bf_comm_gather_params), gloo instead of RCCL in between
    lo, hi = shard.shard_range(n_frames, rank, world)
    send = torch.from_numpy(shard.pad_block(_params(range(lo, hi)), n_frames, world))
    recv = torch.empty(world * send.numel())
    dist.all_gather_into_tensor(recv, send.reshape(-1))
    full = shard.unpack(recv.numpy().reshape(world, *send.shape), n_frames, world)
    np.save(os.path.join(out, f"r{rank}.npy"), full)
    # the 128-byte RCCL id travels through the file rendezvous; so do the timings' max and the barriers
    rdzv = shard.FileRendezvous(rank, world, key=f"pytest-{port}", root=out)
    uid = rdzv.broadcast("rccl-unique-id", bytes(range(128)) if rank == 0 else None)
    assert uid == bytes(range(128))
    walls = rdzv.all_gather("wall", repr(1.0 + rank).encode())
    assert max(float(w) for w in walls) == float(world)
    rdzv.cleanup()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 8])
def test_gather_world2_gloo(tmp_path, n_frames):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    want = _params(range(n_frames))
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"r{r}.npy"), want)


def test_contour_blocks_of_the_dense_setters_world2():
    """bf_group_set_masks hands device s the masks of its frames and the contour points from xy_first[s] on
    (bf_shard_contour_offsets - host arithmetic, the same call the group makes)."""
    import ctypes as C
    from bodyfitting_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for n_frames, world, n_masks in ((7, 2, 4), (8, 2, 8), (64, 8, 8), (3, 2, 1)):
        counts = rng.integers(0, 50, size=n_frames * n_masks).astype(np.int32)
        out = (C.c_int64 * (world + 1))()
        _lib.check(lib.bf_shard_contour_offsets(n_frames, world, n_masks, _lib.iptr(counts), out), "bf_shard_contour_offsets")
        for s in range(world):
            lo, hi = shard.shard_range(n_frames, s, world)
            assert out[s] == counts[: lo * n_masks].sum()
            assert out[s + 1] - out[s] == counts[lo * n_masks: hi * n_masks].sum()
        assert out[world] == counts.sum()
    bad = np.array([3, -1], np.int32)
    assert lib.bf_shard_contour_offsets(2, 2, 1, _lib.iptr(bad), (C.c_int64 * 3)()) != 0


def test_rendezvous_keys_separate_restarts_and_reused_names(tmp_path, monkeypatch):
    """a restarted worker group (TORCHELASTIC_RESTART_COUNT) must not read the previous group's RCCL id; a barrier name used twice
    does not pass on the first use's files; the directory is private and goes away with cleanup"""
    monkeypatch.setenv("MASTER_PORT", "29999")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    a = shard.FileRendezvous(0, 1, root=str(tmp_path))
    a.broadcast("rccl-unique-id", b"old")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    b = shard.FileRendezvous(0, 1, root=str(tmp_path))
    assert b.dir != a.dir and not os.path.exists(os.path.join(b.dir, "rccl-unique-id.0"))
    assert (os.stat(b.dir).st_mode & 0o777) == 0o700
    b.barrier("x"); b.barrier("x")
    assert sorted(f for f in os.listdir(b.dir) if f.startswith("barrier-x")) == ["barrier-x-1.0", "barrier-x-2.0"]
    b.cleanup()
    assert not os.path.exists(b.dir)
    os.chmod(a.dir, 0o755)
    with pytest.raises(PermissionError):
        shard.FileRendezvous(0, 1, key=os.path.basename(a.dir).split("-", 3)[3], root=str(tmp_path))
