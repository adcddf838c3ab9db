"""The N>1 path on CPU: world-size-2 gloo run of the frame partition and the final gather."""
import os
import socket
import sys

import numpy as np
import pytest

from bodyfitting_amd import shard


def test_partition_covers_every_frame_once():
    for n, w in ((256, 8), (64, 8), (7, 2), (3, 4), (1, 2)):
        blocks = [shard.shard_range(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1


def _worker(rank, world, port, n_frames, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_frames, rank, world)
    local = np.stack([np.arange(86, dtype=np.float32) + 1000.0 * f for f in range(lo, hi)]) if hi > lo else np.zeros((0, 86), np.float32)
    full = shard.gather_params(local, n_frames, dist=dist)
    np.save(os.path.join(out, f"r{rank}.npy"), full)
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
    want = np.stack([np.arange(86, dtype=np.float32) + 1000.0 * f for f in range(n_frames)])
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"r{r}.npy"), want)
