"""Batches of frames through the C ABI: BASELINE config 4's shard sizes (32 frames per GPU, 256 frames) against the
single-frame path and the reference goldens, and the life-time rules of several batches sharing one model
(per-batch LDS size, per-batch MFMA scratch, setters that wait for a fit in flight, scans replaced on a reused batch)."""
import numpy as np
import pytest

from conftest import load_golden
from bodyfitting_amd import _lib, native as N, synthetic as S

pytestmark = pytest.mark.gpu
PARAMS = ("global_transl", "scale", "pose", "betas", "global_orient")
FIT_TOL = 1e-4          # north star: fitted beta / theta / transl within 1e-4 abs of the reference CPU path


def _batch(dev_model, problems):
    c2w, K, kp, ndiv, betas, pose = N.pack_problem(problems)
    b = N.FrameBatch(dev_model, len(problems), c2w.shape[1])
    b.set_cameras(c2w, K)
    b.set_keypoints(kp, ndiv)
    b.set_init(betas, pose)
    return b


@pytest.fixture(scope="module")
def problems(smpl_model):
    return [S.make_problem(smpl_model, frame=f, n_views=48) for f in range(256)]


@pytest.fixture(scope="module")
def singles(dev_model, problems):
    """frames fitted one at a time (the path the cfg-2 goldens pin): params, vertices, joints"""
    out = {}
    for f in (0, 1, 2, 3, 15, 16, 31, 100, 255):
        b = _batch(dev_model, [problems[f]])
        b.fit(100)
        v, j, fp, _ = b.get_result()
        out[f] = (b.get_params()[0], v[0], j[0], fp[0])
        b.close()
    return out


@pytest.mark.parametrize("n_frames", [32, 256])
@pytest.mark.parametrize("flags", [_lib.FIT_DEFAULT, _lib.FIT_FETCH | _lib.FIT_RESET | _lib.FIT_GRAPH],
                         ids=["host-issued", "graph+pipelined-fetch"])
def test_cfg4_shard_equals_single_frames_and_goldens(dev_model, problems, singles, n_frames, flags):
    """bf_fit with F = 32 (config 4's per-GPU shard) and F = 256 (the whole of config 4): from 16 frames on the final mesh
    is pack_feat -> fp32-MFMA GEMM -> batched epilogue and the result fetch is pipelined.  Parameters are bit for bit
    those of the frame fitted alone; vertices / joints agree to fp32 round-off of the other summation tree; frames 0-3 hold
    the reference goldens at 1e-4."""
    b = _batch(dev_model, problems[:n_frames])
    for _ in range(2 if flags else 1):                        # (the graph path twice: both result arenas)
        b.fit(100, flags=flags)
    params = b.get_params()
    verts, joints, full_pose, terms = b.get_result()
    b.close()
    for f, (p1, v1, j1, fp1) in singles.items():
        if f >= n_frames:
            continue
        np.testing.assert_array_equal(params[f], p1, err_msg=f"frame {f}")
        np.testing.assert_array_equal(full_pose[f], fp1, err_msg=f"frame {f}")
        np.testing.assert_allclose(verts[f], v1, atol=2e-6, err_msg=f"frame {f}")
        np.testing.assert_allclose(joints[f], j1, atol=2e-6, err_msg=f"frame {f}")
    for f in range(4):
        g = load_golden(f"cfg2_48view_100it_f{f}.npz")
        got = N.split_params(params[f])
        for n in PARAMS:
            np.testing.assert_allclose(got[n], g[f"it100_{n}"], rtol=0, atol=FIT_TOL, err_msg=f"frame {f} {n}")
        np.testing.assert_allclose(joints[f], g["joints"], atol=FIT_TOL)
        np.testing.assert_allclose(verts[f][::53], g["vertices_sample"], atol=FIT_TOL)
        np.testing.assert_allclose(full_pose[f], g["full_pose"], atol=FIT_TOL)
    assert np.isfinite(verts).all() and np.isfinite(terms).all()
    # size-independent property: every frame's loss went down from its initial value by a wide margin
    assert (terms.sum(1) > 0).all()


def test_two_live_batches_with_different_view_counts(dev_model, smpl_model):
    """The fit kernel's LDS carve depends on the view count: a 48-view and an 8-view batch of ONE model, used alternately
    (what SMPLify._batches does), each get their own launch size."""
    p48 = S.make_problem(smpl_model, frame=1, n_views=48)
    p8 = S.make_problem(smpl_model, frame=5, n_views=8)
    want = {}
    for key, p in (("a", p48), ("b", p8)):
        b = _batch(dev_model, [p])
        b.fit(40)
        want[key] = (b.get_params(), b.get_result()[0])
        b.close()
    a = _batch(dev_model, [p48])
    b = _batch(dev_model, [p8])               # created last: the model-level size (the old bug) would now be the 8-view one
    for rounds in range(2):
        a.fit(40, flags=_lib.FIT_RESET)
        b.fit(40, flags=_lib.FIT_RESET)
        a.fit(40, flags=_lib.FIT_RESET | _lib.FIT_GRAPH | _lib.FIT_FETCH)
        np.testing.assert_array_equal(a.get_params(), want["a"][0])
        np.testing.assert_array_equal(a.get_result()[0], want["a"][1])
        np.testing.assert_array_equal(b.get_params(), want["b"][0])
        np.testing.assert_array_equal(b.get_result()[0], want["b"][1])
    a.close()
    b.close()


def test_two_mfma_batches_of_one_model_interleave(dev_model, problems):
    """>= 16 frames: the pose-blend GEMM's scratch (pose_off, featT) belongs to the batch, so two batches of one model
    queued back to back on their own streams do not overwrite each other's pose offsets"""
    a = _batch(dev_model, problems[:16])
    b = _batch(dev_model, problems[16:40])
    a.fit(10); b.fit(10)
    want = [(x.get_params(), x.get_result()[0]) for x in (a, b)]
    for _ in range(3):                        # no sync between the calls: the two streams overlap
        a.fit(10, flags=_lib.FIT_RESET)
        b.fit(10, flags=_lib.FIT_RESET)
    for x, (p, v) in zip((a, b), want):
        np.testing.assert_array_equal(x.get_params(), p)
        np.testing.assert_array_equal(x.get_result()[0], v)
    a.close()
    b.close()


def test_setters_wait_for_a_fit_in_flight(dev_model, smpl_model):
    """bf_fit is asynchronous; uploading the NEXT frame's cameras / keypoints right behind it must not change the fit
    that is still reading the old ones"""
    p0 = S.make_problem(smpl_model, frame=0, n_views=48)
    p1 = S.make_problem(smpl_model, frame=1, n_views=48)
    g = load_golden("cfg2_48view_100it_f0.npz")
    c2w1, K1, kp1, nd1, _, _ = N.pack_problem([p1])
    b = _batch(dev_model, [p0])
    b.fit(100)                                # queued, not finished
    b.set_keypoints(kp1 * 0.5, nd1)
    b.set_cameras(c2w1[:, ::-1].copy(), K1)
    got = N.split_params(b.get_params()[0])
    for n in PARAMS:
        np.testing.assert_allclose(got[n], g[f"it100_{n}"], rtol=0, atol=FIT_TOL, err_msg=n)
    b.close()


def test_reused_batch_refits_with_a_scan_of_another_height():
    """5 * imsize / scan_height (smplify.py:206,210) follows the scan attached for THIS fit and this call's imsize: a batch
    kept between SMPLify.__call__s gets a new scan (and possibly another image size) every call"""
    model = S.make_model("smpl", seed=0, nv=690)
    dev = N.DeviceModel(model, S.make_gmm(seed=0), device=0)
    items = [S.make_scan_problem(model, frame=f, n_views=8, scan_scale=sc) for f, sc in ((0, 1.0), (1, 0.5))]
    scans = [N.Scan(sv, sf) for _, sv, sf in items]
    packed = [N.pack_problem([p]) for p, _, _ in items]
    hyp = [N.make_hyper(imsize=512), N.make_hyper(imsize=256)]

    def run(b, i):
        c2w, K, kp, ndiv, betas, pose = packed[i]
        b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans([scans[i]])
        b.fit(15, hyp[i])
        return b.get_params().copy()

    fresh = []
    for i in range(2):
        b = N.FrameBatch(dev, 1, 8)
        fresh.append(run(b, i))
        b.close()
    b = N.FrameBatch(dev, 1, 8)
    for i in (0, 1, 0):
        np.testing.assert_array_equal(run(b, i), fresh[i])
    b.close()
    for s in scans:
        s.close()
    dev.close()
