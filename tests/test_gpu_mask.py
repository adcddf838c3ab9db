"""HIP silhouette loss (loss.py:85-130) through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from bodyfitting_amd import native as N
from bodyfitting_amd import synthetic as S
from bodyfitting_amd.contours import extract_contours
from oracle import smplify_oracle as O
from test_mask_oracle import MASK_FRAMES, mask_inputs

pytestmark = pytest.mark.gpu
PARAMS = ("global_transl", "scale", "pose", "betas", "global_orient")


def _batch(dev_model, prob):
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev_model, 1, c2w.shape[1])
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    masks = np.array(prob["masks"])[None]
    contours = [extract_contours(np.array(prob["masks"]) > 128)]
    b.set_masks(masks, [prob["use_frames"].index(f) for f in prob["mask_frames"]], contours)
    return b


def test_mask_loss_value_and_gradient(dev_model, smpl_model):
    """bf_batch_mask_loss vs the fp64 oracle (exact distances) and vs the reference golden (noisy cdist)"""
    g = load_golden("mask_loss_f0.npz")
    prob, contours, masks, w2cs, Ks = mask_inputs(smpl_model, torch.float64)
    b = _batch(dev_model, prob)
    params = {"global_transl": g["transl"], "scale": np.array([float(g["scale"])]), "pose": prob["init_pose"][0, 3:],
              "betas": prob["init_betas"][0], "global_orient": prob["init_pose"][0, :3]}
    b.set_params(N.pack_params(params)[None])
    loss, dv = b.mask_loss()
    # same vertices on the oracle side (fp64 forward of the same parameters)
    m = O.to_torch_model(smpl_model, torch.float64)
    out = O.smpl_forward(m, torch.tensor(prob["init_betas"], dtype=torch.float64), torch.tensor(prob["init_pose"][:, :3], dtype=torch.float64),
                         torch.tensor(prob["init_pose"][:, 3:], dtype=torch.float64))
    verts = ((out["vertices"][0] + torch.tensor(g["transl"], dtype=torch.float64)) * float(g["scale"]) * 0.3).detach().requires_grad_(True)
    want = O.multview_mask_loss(contours, masks, verts, w2cs, Ks, imsize=512)
    want.backward()
    assert float(loss[0]) == pytest.approx(float(want), rel=2e-5)
    assert float(loss[0]) == pytest.approx(float(g["loss"]), rel=5e-5)
    gw = verts.grad.numpy()
    assert np.all(dv[0].reshape(-1, 3)[np.arange(6890) % 4 != 0] == 0)          # only every 4th vertex (loss.py:99)
    err = np.abs(dv[0] - gw)
    assert np.mean(err < 1e-4 * np.abs(gw).max()) > 0.998                        # (an argmin tie may flip a few)
    assert np.mean(np.abs(dv[0][::4] - g["grad_sampled"]) < 2e-3 * np.abs(gw).max()) > 0.99
    b.close()


def test_mask_fit_first_steps_and_progress(dev_model, smpl_model, gmm_bufs):
    g = load_golden("mask_fit_8view_30it.npz")
    prob = S.make_problem(smpl_model, frame=0, n_views=8, mask_frames=MASK_FRAMES)
    b = _batch(dev_model, prob)
    l0 = b.mask_loss()[0][0]
    b.fit(30)
    got = N.split_params(b.get_params()[0])
    # the silhouette loop is ill-conditioned (tests/test_mask_oracle.py): trajectories agree to ~1e-1 at 30 steps
    res = O.fit(smpl_model, gmm_bufs, prob, 30)
    for n in PARAMS:
        assert np.abs(got[n] - res[n if n != "global_transl" else "raw_transl"]).max() < 0.3, n
    assert np.isfinite(b.get_params()).all()
    assert b.mask_loss()[0][0] < 0.7 * l0                    # and the silhouette term really went down
    b.close()
    # keypoint-only prefix is exact: 11 steps of 30 == golden
    b = _batch(dev_model, prob)
    b.fit(33 // 3)          # n_iters = 11 -> threshold 3: iterations 4..10 would use the mask; use the plain batch instead
    b.close()
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    p = N.FrameBatch(dev_model, 1, 8)
    p.set_cameras(c2w, K); p.set_keypoints(kp, ndiv); p.set_init(betas, pose)
    p.fit(11)
    got = N.split_params(p.get_params()[0])
    for n in PARAMS:
        np.testing.assert_allclose(got[n], g[f"it11_{n}"], rtol=0, atol=1e-4)
    p.close()
