"""Texture fitting on the GPU (bf_texfit_*, bodyfitting_amd.texture_fitting) against oracle/texfit_oracle.py - the numpy
restatement of the slice of neural_renderer that smplify/texture_fitting.py:240-275 runs.  The kernels keep the reference's
float32 operation order (-ffp-contract=off), so renders are compared bit for bit; the texture gradient is an atomicAdd sum
(any order on the device, as in the reference) and gets a float32 summation tolerance."""
import numpy as np
import pytest

from bodyfitting_amd import texture_fitting as TF
from oracle import texfit_oracle as TO
from texfit_cases import blob_pair, icosphere

pytestmark = pytest.mark.gpu

IS = 32
K32 = np.array([[IS, 0, IS // 2], [0, IS, IS // 2], [0, 0, 1]], np.float32)
GRAD_ATOL = 2e-6          # float32 sums of <= a few hundred terms of magnitude <= 0.25 in a different order


def _setup(ts=4, aa=True, level=2):
    scan, fit = blob_pair(level=level, ts=ts)
    center, dist = TF.scene_bound(scan[0])
    r = TF.Renderer(IS, ts, near=0.0, far=2 * dist, anti_aliasing=aa)
    r.set_mesh(r.TARGET, scan); r.set_mesh(r.FITTED, fit)
    views = TF.gen_cam_views(center, 18, dist, gl=True)
    views.append(np.linalg.inv(TF.sphere2rot(dist, 0.7, 2.1, t=center)))
    views.append(np.linalg.inv(TF.sphere2rot(dist, 2.6, 5.0, t=center)))
    return r, scan, fit, views, dist


def _oracle_render(mesh, pose, dist, aa=True, keep=None):
    return TO.render(*mesh, K32, pose[:3, :3], pose[:3, 3], IS, IS, 0.0, np.float32(2 * dist), anti_aliasing=aa, keep=keep)


@pytest.mark.parametrize("aa", [True, False])
def test_render_is_the_oracle_render_bit_for_bit(aa):
    r, scan, fit, views, dist = _setup(aa=aa)
    for vi in (0, 5, 11, 18, 19):
        for which, mesh in ((r.TARGET, scan), (r.FITTED, fit)):
            got = r.render_rgb(which, views[vi])
            want = _oracle_render(mesh, views[vi], dist, aa)
            assert (want < 1).any()                                     # the mesh is in view
            np.testing.assert_array_equal(got, want, err_msg=f"view {vi} mesh {which}")
    r.close()


@pytest.mark.parametrize("ts", [2, 4, 6])
def test_loss_and_texture_gradient_match_the_oracle(ts):
    r, scan, fit, views, dist = _setup(ts=ts)
    for vi in (2, 19):
        loss, grad = r.loss_grad(views[vi])
        keep = {}
        a = _oracle_render(scan, views[vi], dist)
        b = _oracle_render(fit, views[vi], dist, keep=keep)
        assert loss == pytest.approx(float(np.abs(a - b).astype(np.float64).sum()), rel=1e-12)
        want = TO.texture_grad(np.sign(b - a).astype(np.float32), keep, len(fit[1]), ts, IS)
        assert np.abs(want).max() > 0.1
        np.testing.assert_allclose(grad, want, atol=GRAD_ATOL, rtol=0)
        assert ((grad != 0) == (want != 0)).mean() > 0.9999
    # the gradient buffer is cleared between calls: a second call returns the same thing
    l2, g2 = r.loss_grad(views[19])
    assert l2 == loss
    np.testing.assert_allclose(g2, grad, atol=GRAD_ATOL, rtol=0)
    r.close()


def test_adam_steps_follow_the_oracle_loop():
    r, scan, fit, views, dist = _setup()
    o = TO.TextureFit(scan, fit, IS, 0.0, np.float32(2 * dist), lr=1e-2)
    n = 8
    for i in range(n):
        pose = views[(i * 5) % len(views)]
        got = r.step(pose, 1e-2)
        want, _, _ = o.step(K32, pose[:3, :3], pose[:3, 3], IS)
        assert got == pytest.approx(want, rel=1e-5), f"iteration {i}"
    tex, ref = r.textures(), o.mesh[2]
    # a texel whose gradient is a near-cancelling sum moves by +-lr or not at all depending on the summation order (Adam
    # normalises the step by |g|): allow a handful of those, everything else agrees to float32 rounding
    close = np.abs(tex - ref) < 1e-5
    assert close.mean() > 0.999, close.mean()
    assert np.abs(tex - ref).max() <= n * 1e-2 * 1.01
    assert np.abs(tex - fit[2]).max() > 0.02                         # and the textures did move
    r.close()


def test_fit_recovers_the_scan_colours():
    """the loop of TextureFitting.__call__ on the ring + random views: the L1 image loss falls and the fitted renders approach the scan's"""
    scan, fit = blob_pair(level=3, ts=4)
    tf = TF.TextureFitting(render_img_size=64, lrate=1e-2, iter_num=150, seed=0)
    tex, losses = tf.fit(fit, scan)
    assert tex.shape == fit[2].shape and len(losses) == 150
    first, last = losses[:18].mean(), losses[-18:].mean()
    assert last < 0.45 * first, (first, last)
    assert np.isfinite(tex).all()
    # per-face mean colour moved towards the scan's (same topology, so faces correspond)
    before = np.abs(fit[2].mean((1, 2, 3)) - scan[2].mean((1, 2, 3))).mean()
    after = np.abs(tex.mean((1, 2, 3)) - scan[2].mean((1, 2, 3))).mean()
    assert after < 0.6 * before, (before, after)


def test_full_size_render_properties():
    """512 x 512 (1024 x 1024 super-sampled), SMPL-sized and scan-sized meshes: every pixel is a convex mixture of texels and
    background, rotating the ring by one view changes the image, and a second render of the same view is identical."""
    v, f = icosphere(5)                                               # 20480 faces
    rng = np.random.default_rng(1)
    v = (v * np.array([0.45, 0.8, 0.4], np.float32) + np.array([0, 0.9, 0], np.float32)).astype(np.float32)
    tex = rng.uniform(0.2, 0.7, (len(f), 4, 4, 4, 3)).astype(np.float32)
    center, dist = TF.scene_bound(v)
    r = TF.Renderer(512, 4, near=0.0, far=2 * dist)
    r.set_mesh(r.TARGET, (v, f, tex))
    ring = TF.gen_cam_views(center, 18, dist, gl=True)
    a, a2, b = r.render_rgb(r.TARGET, ring[0]), r.render_rgb(r.TARGET, ring[0]), r.render_rgb(r.TARGET, ring[1])
    np.testing.assert_array_equal(a, a2)
    assert (a != b).mean() > 0.1
    fg = np.all(a < 1.0, 0)
    assert 0.15 < fg.mean() < 0.6
    assert a[:, fg].min() >= 0.2 - 1e-6 and a.max() <= 1.0
    interior = np.all(a <= 0.7 + 1e-6, 0)
    assert interior.sum() > 0.9 * fg.sum()                            # only the silhouette edge mixes with the white background
    r.close()


def test_faces_with_a_large_pixel_box_take_the_per_pixel_path():
    """two triangles that fill the view (boxes far above BF_TEX_GATHER_MAX = 4096 pixels) next to a fan of small ones"""
    rng = np.random.default_rng(5)
    big_v = np.array([[-0.9, -0.9, 2.0], [0.9, -0.9, 2.0], [0.9, 0.9, 2.0], [-0.9, 0.9, 2.0]], np.float32)
    small_v, small_f = icosphere(1)
    v = np.concatenate([big_v, small_v * 0.3 + np.array([0, 0, 1.5], np.float32)]).astype(np.float32)
    f = np.concatenate([np.array([[0, 1, 2], [0, 2, 3]], np.int32), small_f + 4])
    K = np.array([[64, 0, 32], [0, 64, 32], [0, 0, 1]], np.float32)
    pv = TO.project(v, K, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), 64)[f]
    back = (pv[:, 2, 1] - pv[:, 0, 1]) * (pv[:, 1, 0] - pv[:, 0, 0]) < (pv[:, 1, 1] - pv[:, 0, 1]) * (pv[:, 2, 0] - pv[:, 0, 0])
    f[:2][back[:2]] = f[:2][back[:2]][:, ::-1]                       # the two big ones face the camera; the sphere keeps its winding
    ts = 3
    tex_a = rng.uniform(0, 1, (len(f), ts, ts, ts, 3)).astype(np.float32)
    tex_b = rng.uniform(0, 1, (len(f), ts, ts, ts, 3)).astype(np.float32)
    r = TF.Renderer(64, ts, near=0.0, far=10.0)
    r.set_mesh(r.TARGET, (v, f, tex_a)); r.set_mesh(r.FITTED, (v, f, tex_b))
    loss, grad = r.loss_grad(np.eye(4))
    keep = {}
    a = TO.render(v, f, tex_a, K, np.eye(3), np.zeros(3), 64, 64, 0.0, 10.0)
    b = TO.render(v, f, tex_b, K, np.eye(3), np.zeros(3), 64, 64, 0.0, 10.0, keep=keep)
    np.testing.assert_array_equal(r.render_rgb(r.FITTED, np.eye(4)), b)
    want = TO.texture_grad(np.sign(b - a).astype(np.float32), keep, len(f), ts, 64)
    assert np.abs(want[:2]).max() > 10 and np.abs(want[2:]).max() > 0.1          # both kinds of face own pixels
    np.testing.assert_allclose(grad, want, atol=2e-4, rtol=1e-5)                  # (sums of thousands of terms on the big faces)
    assert loss == pytest.approx(float(np.abs(a - b).astype(np.float64).sum()), rel=1e-12)
    r.close()


def test_tile_lists_grow_when_a_render_needs_more_than_the_first_guess():
    """two faces that cover all 4,096 tiles of a 256 x 256 render (8,192 list entries against a first capacity of ~5,100): the
    render is repeated with larger lists behind the caller's back and comes out complete"""
    v = np.array([[-3, -3, 2.0], [3, -3, 2.0], [3, 3, 2.0], [-3, 3, 2.0]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    K = np.array([[256, 0, 128], [0, 256, 128], [0, 0, 1]], np.float32)
    pv = TO.project(v, K, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), 256)[f]
    back = (pv[:, 2, 1] - pv[:, 0, 1]) * (pv[:, 1, 0] - pv[:, 0, 0]) < (pv[:, 1, 1] - pv[:, 0, 1]) * (pv[:, 2, 0] - pv[:, 0, 0])
    f[back] = f[back][:, ::-1]
    tex = np.broadcast_to(np.array([0.25, 0.5, 0.75], np.float32), (2, 4, 4, 4, 3)).copy()
    r = TF.Renderer(256, 4, near=0.0, far=10.0)
    r.set_mesh(r.TARGET, (v, f, tex)); r.set_mesh(r.FITTED, (v, f, 0 * tex))
    img = r.render_rgb(r.TARGET, np.eye(4))
    np.testing.assert_allclose(img, np.array([0.25, 0.5, 0.75], np.float32)[:, None, None] * np.ones((3, 256, 256), np.float32), atol=1e-6)
    np.testing.assert_array_equal(r.render_rgb(r.TARGET, np.eye(4)), img)
    loss = r.step(np.eye(4), 1e-2)                                   # the fitted mesh's lists overflow inside a step as well
    assert loss == pytest.approx(256 * 256 * 1.5, rel=1e-6)
    assert r.step(np.eye(4), 1e-2) < loss
    r.close()


def test_nothing_in_front_of_the_camera_renders_the_background():
    r, scan, fit, views, dist = _setup()
    away = views[0].copy()
    away[2, 3] -= 10 * dist                                           # the whole mesh behind the near plane
    img = r.render_rgb(r.TARGET, away)
    assert (img == 1.0).all()
    loss, grad = r.loss_grad(away)
    assert loss == 0.0 and not grad.any()
    r.close()


def test_far_plane_and_background_colour():
    scan, _ = blob_pair()
    center, dist = TF.scene_bound(scan[0])
    pose = TF.gen_cam_views(center, 18, dist, gl=True)[3]
    r = TF.Renderer(IS, 4, near=0.0, far=dist, background=(0.1, 0.2, 0.3))      # far plane through the centre
    r.set_mesh(r.TARGET, scan)
    got = r.render_rgb(r.TARGET, pose)
    want = TO.render(*scan, K32, pose[:3, :3], pose[:3, 3], IS, IS, 0.0, np.float32(dist), background=(0.1, 0.2, 0.3))
    np.testing.assert_array_equal(got, want)
    assert np.allclose(got[:, 0, 0], [0.1, 0.2, 0.3])
    r.close()


def test_errors():
    from bodyfitting_amd import _lib
    r = TF.Renderer(IS, 4, near=0.0, far=5.0)
    with pytest.raises(_lib.BodyfitError):
        r.step(np.eye(4), 1e-2)                                       # no mesh yet
    with pytest.raises(_lib.BodyfitError):
        r.render_rgb(0, np.eye(4))
    scan, fit = blob_pair()
    bad = (scan[0], scan[1] + 1000, scan[2])
    with pytest.raises(_lib.BodyfitError):
        r.set_mesh(0, bad)
    with pytest.raises(ValueError):
        r.set_mesh(0, blob_pair(ts=2)[0])
    with pytest.raises(_lib.BodyfitError):
        TF.Renderer(IS, 64, near=0.0, far=5.0)
    r.close()


def test_uv_texture_image_matches_the_oracle():
    """bf_texfit_render_ndc / Renderer.render_texture / render_texture_map (neural_renderer/renderer.py:294-346,
    smplify/texture_fitting.py:149-151,298): the UV-space image of the fitted textures, bit for bit the numpy restatement's -
    colours and depth, anti-aliased and not - for the textures as set and after some fitting steps."""
    from texfit_cases import blob_pair, uv_atlas
    from bodyfitting_amd import texture_fitting as TF
    from oracle import texfit_oracle as TO
    scan, fit = blob_pair(level=1, ts=4, seed=2)
    nf = len(fit[1])
    uv, uvf = uv_atlas(nf)
    rng = np.random.default_rng(9)
    tex = rng.uniform(0, 1, fit[2].shape).astype(np.float32)
    for aa in (False, True):
        r = TF.Renderer(48, 4, near=0.0, far=4.0, anti_aliasing=aa)
        r.set_mesh(r.TARGET, scan)
        r.set_mesh(r.FITTED, (fit[0], fit[1], tex))
        rgb, depth = r.render_texture(uv, uvf)
        want_rgb, want_depth = TO.render_texture(uv, uvf, tex, 48, 0.0, 4.0, anti_aliasing=aa)
        np.testing.assert_array_equal(rgb, want_rgb)
        np.testing.assert_array_equal(depth, want_depth)
        assert (depth < 4.0).mean() > 0.2 and (depth == 4.0).any()
        # after a few Adam steps the image is that of the stepped textures; the fitting state is untouched by the UV render
        center, dist = TF.scene_bound(scan[0])
        poses = TF.gen_cam_views(center, 4, dist, gl=True)
        for p in poses:
            r.step(p, 1e-2)
        t1 = r.textures()
        img = TF.render_texture_map(r, uv, uvf)
        want, _ = TO.render_texture(uv, uvf, t1, 48, 0.0, 4.0, anti_aliasing=aa)
        np.testing.assert_array_equal(img, TF.to8b(want.transpose(1, 2, 0)[:, :, ::-1]))
        assert img.dtype == np.uint8 and img.shape == (48, 48, 3)
        np.testing.assert_array_equal(r.textures(), t1)
        r.close()
