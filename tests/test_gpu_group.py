"""bf_group / bf_comm (csrc/group.hip) on the one GPU of the box: the group API with one device and the per-rank
communicator with world size 1 run the same code as N devices do - model + batch per device, bf_fit on the device's own
stream, and the final ncclAllGather over RCCL, stream-ordered behind the fit - and must give what a plain FrameBatch gives."""
import numpy as np
import pytest

from conftest import load_golden
from bodyfitting_amd import _lib, native as N, shard, synthetic as S

pytestmark = pytest.mark.gpu
PARAMS = ("global_transl", "scale", "pose", "betas", "global_orient")


@pytest.fixture(scope="module")
def job(smpl_model):
    probs = [S.make_problem(smpl_model, frame=f, n_views=48) for f in range(5)]
    return probs, N.pack_problem(probs)


def test_group_of_one_device_fits_and_gathers(smpl_model, gmm, dev_model, job):
    probs, (c2w, K, kp, ndiv, betas, pose) = job
    g = shard.Group(smpl_model, gmm, n_frames=5, n_views=48, n_devices=1)
    assert g.shards == [(0, 0, 5)] and g.n_params == 86
    g.set_cameras(c2w, K); g.set_keypoints(kp, ndiv); g.set_init(betas, pose)
    g.fit(100, flags=_lib.FIT_FETCH)
    full = g.gather_params()                       # no host sync between the fit and the collective
    assert g.comm_size() == 1
    b = N.FrameBatch(dev_model, 5, 48)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    b.fit(100)
    np.testing.assert_array_equal(full, b.get_params())
    np.testing.assert_array_equal(g.batches[0].get_result()[0], b.get_result()[0])
    for f in range(4):
        gold = load_golden(f"cfg2_48view_100it_f{f}.npz")
        got = N.split_params(full[f])
        for n in PARAMS:
            np.testing.assert_allclose(got[n], gold[f"it100_{n}"], rtol=0, atol=1e-4, err_msg=f"frame {f} {n}")
    # a second job on the same group: re-armed on the device, gathered again
    g.fit(100, flags=_lib.FIT_FETCH | _lib.FIT_RESET | _lib.FIT_GRAPH)
    np.testing.assert_array_equal(g.gather_params(), full)
    b.close()
    g.close()


def test_group_refuses_devices_that_are_not_there(smpl_model, gmm):
    have = _lib.load().bf_device_count()
    with pytest.raises(_lib.BodyfitError, match="visible"):
        shard.Group(smpl_model, gmm, n_frames=64, n_views=8, n_devices=have + 1)
    with pytest.raises(_lib.BodyfitError):
        shard.Group(smpl_model, gmm, n_frames=4, n_views=8, devices=[0, 0])


def test_comm_world1_barrier_max_and_gather(dev_model, job, tmp_path):
    probs, (c2w, K, kp, ndiv, betas, pose) = job
    rdzv = shard.FileRendezvous(0, 1, key="pytest-world1", root=str(tmp_path))
    comm = shard.Comm(0, 1, 0, rendezvous=rdzv)
    assert comm.size() == 1
    comm.barrier()
    assert comm.max(3.25) == 3.25
    b = N.FrameBatch(dev_model, 5, 48)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    b.fit(20)
    full = comm.gather_params(b, 5)
    np.testing.assert_array_equal(full, b.get_params())
    with pytest.raises(_lib.BodyfitError, match="partition"):
        comm.gather_params(b, 6)
    b.close()
    comm.close()
