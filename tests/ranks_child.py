"""One rank of a bf_comm job, run in a FRESH process (tests/test_gpu_group.py::test_ranks_in_fresh_processes): model + batch on the
rank's own device, file rendezvous, ncclCommInitRank, barrier, max over the ranks, fit, the one all-gather, cleanup - what
`python -m torch.distributed.run ... bench.py` does per rank, without the launcher."""
import os
import sys
import traceback

import numpy as np


def rank_main(rank, world, root, key, out_pattern, n_frames, n_views, iters):
    try:
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if repo not in sys.path:
            sys.path.insert(0, repo)
        from bodyfitting_amd import native as N, shard, synthetic as S
        model, gmm = S.make_model("smpl", seed=0), S.make_gmm(seed=0)
        dev = N.DeviceModel(model, gmm, device=rank)                       # one GPU per rank: LOCAL_RANK = RANK on one node
        lo, hi = shard.shard_range(n_frames, rank, world)
        probs = [S.make_problem(model, frame=f, n_views=n_views) for f in range(lo, hi)]
        c2w, K, kp, ndiv, betas, pose = N.pack_problem(probs)
        b = N.FrameBatch(dev, hi - lo, n_views)
        b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
        rdzv = shard.FileRendezvous(rank, world, key=key, root=root)
        comm = shard.Comm(rank, world, rank, rendezvous=rdzv)
        comm.barrier()
        top = comm.max(rank + 0.5)
        b.fit(iters)
        full = comm.gather_params(b, n_frames)                             # stream-ordered behind the fit
        np.savez(out_pattern % rank, full=full, mine=b.get_params(), lo=lo, hi=hi, top=top, size=comm.size())
        comm.barrier()
        comm.close(); b.close(); dev.close()
        if rank == 0:
            rdzv.cleanup()
    except BaseException:                                                  # the parent reads the reason from the file
        with open((out_pattern % rank) + ".err", "w") as fh:
            fh.write(traceback.format_exc())
        raise
