"""Host side of the texture-fitting loop: the view schedule of bodyfitting_amd.texture_fitting against the oracle's literal
restatement of utils/renderer.py:7-25 and smplify/texture_fitting.py:63-83,235-263."""
import numpy as np
import pytest

from bodyfitting_amd import texture_fitting as TF
from oracle import texfit_oracle as TO


@pytest.mark.parametrize("gl", [False, True])
@pytest.mark.parametrize("n", [1, 18, 36])
def test_ring_views_match_the_reference_construction(gl, n):
    c = np.array([0.1, 0.9, -0.05])
    for a, b in zip(TO.gen_cam_views(c, n, 2.3, gl), TF.gen_cam_views(c, n, 2.3, gl)):
        np.testing.assert_allclose(a, b, atol=1e-12)
    assert len(TF.gen_cam_views(c, n, 2.3, gl)) == n


def test_ring_cameras_look_at_the_centre_from_dist():
    c = np.array([0.3, -0.2, 1.0])
    for w2c in TF.gen_cam_views(c, 7, 1.5, gl=True):
        p = w2c @ np.append(c, 1.0)
        np.testing.assert_allclose(p[:3], [0, 0, 1.5], atol=1e-12)          # centre straight ahead (+z after the GL flip)
        np.testing.assert_allclose(w2c[:3, :3] @ w2c[:3, :3].T, np.eye(3), atol=1e-12)


def test_sphere_views_match_and_are_rigid():
    rng = np.random.default_rng(0)
    for _ in range(20):
        rad, th, ph, t = rng.uniform(0.5, 3), rng.uniform(0.05, np.pi - 0.05), rng.uniform(0, 2 * np.pi), rng.standard_normal(3)
        a, b = TO.sphere2rot(rad, th, ph, t), TF.sphere2rot(rad, th, ph, t)
        np.testing.assert_allclose(a, b, atol=1e-12)
        np.testing.assert_allclose(b[:3, :3].T @ b[:3, :3], np.eye(3), atol=1e-12)
        np.testing.assert_allclose(np.linalg.norm(b[:3, 3] - t), rad, rtol=1e-12)
        np.testing.assert_allclose((np.linalg.inv(b) @ np.append(t, 1.0))[:3], [0, 0, rad], atol=1e-9)


def test_schedule_is_five_rounds_of_the_ring_then_random_views():
    tf = TF.TextureFitting(iter_num=100, seed=3)
    c, d = np.zeros(3), 2.0
    ring = TF.gen_cam_views(c, TF.ROUND_VIEWS, d, gl=True)
    for i in (0, 17, 18, 89):
        assert tf.view(i, ring, c, d) is ring[i % 18]
    p = tf.view(90, ring, c, d)
    assert not any(p is r for r in ring)
    np.testing.assert_allclose(np.linalg.norm(np.linalg.inv(p)[:3, 3]), d, rtol=1e-12)


def test_scene_bound():
    v = np.array([[0, 0, 0], [1, 2, 3], [-1, 0.4, 1]], np.float32)
    c, d = TF.scene_bound(v)
    np.testing.assert_allclose(c, [0, 1, 1.5]); assert d == pytest.approx(2 / 0.8)


def test_textures_shape_is_checked():
    with pytest.raises(ValueError):
        TF._mesh((np.zeros((3, 3)), np.array([[0, 1, 2]]), np.zeros((1, 4, 4, 3))))
