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


@pytest.mark.parametrize("threads", ["0", "1"])
def test_group_with_scans_and_masks_equals_a_plain_batch(threads, monkeypatch):
    """(threads = 1: the device's calls are issued from its worker thread, as in an N-device group - errors included)
    bf_group_set_scans / set_masks / stage_inputs / fit_displacement hand every device its block and run the dense loops: with
    one device the results are a plain FrameBatch's, bit for bit (the same code path as N devices, each from its own host thread)"""
    monkeypatch.setenv("BF_GROUP_THREADS", threads)
    model = S.make_model("smpl", seed=0, nv=690)
    gmm = S.make_gmm(seed=0)
    items = [S.make_scan_problem(model, frame=f, n_views=8) for f in range(3)]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([p for p, _, _ in items])
    g = shard.Group(model, gmm, n_frames=3, n_views=8, n_devices=1)
    # an error raised on the worker thread reaches the caller with its message (staged inputs need a fresh start)
    g.set_cameras(c2w, K); g.stage_inputs(kp, ndiv, betas, pose)
    with pytest.raises(_lib.BodyfitError, match="BF_FIT_RESET"):
        g.fit(5, flags=_lib.FIT_FETCH)
    g.fit(5, flags=_lib.FIT_FETCH | _lib.FIT_RESET)
    assert [g.device_of_frame(f) for f in range(3)] == [0, 0, 0]
    g.set_cameras(c2w, K); g.set_keypoints(kp, ndiv); g.set_init(betas, pose)
    scans = [N.Scan(sv, sf, device=g.device_of_frame(f)) for f, (_, sv, sf) in enumerate(items)]
    g.set_scans(scans)
    g.fit(30, flags=_lib.FIT_FETCH)
    g.fit_displacement(5)
    got_p, got_d = g.gather_params(), g.batches[0].get_displacement()
    dev = N.DeviceModel(model, gmm, device=0)
    b = N.FrameBatch(dev, 3, 8)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans(scans)
    b.fit(30, flags=_lib.FIT_FETCH)
    b.fit_displacement(5)
    np.testing.assert_array_equal(got_p, b.get_params())
    np.testing.assert_array_equal(got_d, b.get_displacement())
    # a scan on the wrong device is refused before anything is attached
    g.set_scans(None); b.set_scans(None)
    # silhouettes: contours extracted on the device, and handed over by the caller
    mask_frames = [1, 3, 5, 7]
    probs = [S.make_problem(model, frame=f, n_views=8, mask_frames=mask_frames) for f in range(3)]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem(probs)
    masks = np.stack([np.array(p["masks"]) for p in probs])
    for contours in (None, "host"):
        if contours == "host":
            from oracle.contour_oracle import border_pixels_rowmajor_all as extract
            contours = [extract(np.array(p["masks"]) > 128) for p in probs]
        g.set_cameras(c2w, K); g.stage_inputs(kp, ndiv, betas, pose); g.set_masks(masks, mask_frames, contours)
        g.fit(15, flags=_lib.FIT_FETCH | _lib.FIT_RESET)
        b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_masks(masks, mask_frames, contours)
        b.fit(15, flags=_lib.FIT_FETCH)
        np.testing.assert_array_equal(g.gather_params(), b.get_params())
    # the NEXT step's silhouettes staged under the fit in flight (bf_group_stage_masks: every device its block) == set between the fits
    probs2 = [S.make_problem(model, frame=f + 3, n_views=8, mask_frames=mask_frames) for f in range(3)]
    _, _, kp2, ndiv2, betas2, pose2 = N.pack_problem(probs2)
    masks2 = np.stack([np.array(p["masks"]) for p in probs2])
    g.set_masks(masks, mask_frames, None)                    # (device contours: what staging continues from)
    g.stage_inputs(kp, ndiv, betas, pose); g.fit(15, flags=_lib.FIT_FETCH | _lib.FIT_RESET)
    g.stage_masks(masks2, mask_frames)
    g.stage_inputs(kp2, ndiv2, betas2, pose2); g.fit(15, flags=_lib.FIT_FETCH | _lib.FIT_RESET)
    b.set_keypoints(kp2, ndiv2); b.set_init(betas2, pose2); b.set_masks(masks2, mask_frames, None)
    b.fit(15, flags=_lib.FIT_FETCH)
    np.testing.assert_array_equal(g.gather_params(), b.get_params())
    for s in scans:
        s.close()
    b.close(); dev.close(); g.close()


def test_two_devices_when_the_box_has_them(smpl_model, gmm, job):
    """N > 1 over real RCCL: skipped on the 1-GPU boxes the builder gets; runs wherever two devices are visible"""
    if _lib.load().bf_device_count() < 2:
        pytest.skip("one visible device")
    probs, (c2w, K, kp, ndiv, betas, pose) = job
    g = shard.Group(smpl_model, gmm, n_frames=5, n_views=48, n_devices=2)
    assert [s[2] for s in g.shards] == [3, 2] and g.comm_size() == 2
    g.set_cameras(c2w, K); g.set_keypoints(kp, ndiv); g.set_init(betas, pose)
    g.fit(100, flags=_lib.FIT_FETCH)
    full0, full1 = g.gather_params(0), g.gather_params(1)
    np.testing.assert_array_equal(full0, full1)
    for f in range(4):
        gold = load_golden(f"cfg2_48view_100it_f{f}.npz")
        got = N.split_params(full0[f])
        for n in PARAMS:
            np.testing.assert_allclose(got[n], gold[f"it100_{n}"], rtol=0, atol=1e-4, err_msg=f"frame {f} {n}")
    g.close()


@pytest.mark.parametrize("world", [1, 2])
def test_ranks_in_fresh_processes(world, tmp_path, job):
    """ranks mode end to end, as the benchmark contract launches it: `world` FRESH processes (forked from a server that was started
    before this process made a HIP call, conftest.FRESH), one GPU each: file rendezvous of the RCCL id, ncclCommInitRank, barrier,
    max over the ranks, fit of the rank's block of frames, the one all-gather.  Every rank must hold the same gathered parameters,
    and they must be the bits of one process fitting all frames.  world = 2 runs wherever two devices are visible."""
    import conftest
    import ranks_child
    if conftest.FRESH is None:
        pytest.skip("no fork server")
    if _lib.load().bf_device_count() < world:
        pytest.skip("%d visible device(s)" % _lib.load().bf_device_count())
    n_frames, n_views, iters = 5, 8, 20
    pattern = str(tmp_path / "rank%d.npz")
    procs = [conftest.FRESH.Process(target=ranks_child.rank_main, args=(r, world, str(tmp_path), "pytest-fresh-%d" % world, pattern, n_frames, n_views, iters))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for r, p in enumerate(procs):
        if p.is_alive():
            p.terminate()
        err = tmp_path / ("rank%d.npz.err" % r)
        assert p.exitcode == 0, "rank %d: exit code %s\n%s" % (r, p.exitcode, err.read_text() if err.exists() else "")
    outs = [np.load(pattern % r) for r in range(world)]
    model, gmm = S.make_model("smpl", seed=0), S.make_gmm(seed=0)
    dev = N.DeviceModel(model, gmm, device=0)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([S.make_problem(model, frame=f, n_views=n_views) for f in range(n_frames)])
    b = N.FrameBatch(dev, n_frames, n_views)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    b.fit(iters)
    want = b.get_params()
    b.close(); dev.close()
    for r, o in enumerate(outs):
        assert int(o["size"]) == world and float(o["top"]) == world - 0.5
        np.testing.assert_array_equal(o["full"], want, err_msg="rank %d" % r)
        np.testing.assert_array_equal(o["mine"], want[int(o["lo"]):int(o["hi"])])
