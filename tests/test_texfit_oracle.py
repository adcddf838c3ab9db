"""The texture-fitting oracle (oracle/texfit_oracle.py: the slice of neural_renderer that smplify/texture_fitting.py:240-275
exercises, restated in numpy - the CUDA extension itself cannot be built here): known answers and finite differences."""
import numpy as np
import pytest

from oracle import texfit_oracle as TO

K16 = np.array([[16.0, 0, 8], [0, 16.0, 8], [0, 0, 1]], np.float32)
EYE = np.eye(3, dtype=np.float32)


def _tri(z=2.0, s=0.6, shift=(0.0, 0.0)):
    # counter-clockwise seen from the camera at the origin looking down +z with y flipped by the projection (v = orig - v)
    v = np.array([[-s + shift[0], -s + shift[1], z], [s + shift[0], -s + shift[1], z], [shift[0], s + shift[1], z]], np.float32)
    return v, np.array([[0, 1, 2]], np.int32)


def _const_tex(nf, ts, color):
    return np.broadcast_to(np.asarray(color, np.float32), (nf, ts, ts, ts, 3)).copy()


def _front(v, f):
    """orient every face so that it survives the back-face cull of forward_face_index_map for the identity camera"""
    pv = TO.project(v, K16, EYE, np.zeros(3), 16)[f]
    back = (pv[:, 2, 1] - pv[:, 0, 1]) * (pv[:, 1, 0] - pv[:, 0, 0]) < (pv[:, 1, 1] - pv[:, 0, 1]) * (pv[:, 2, 0] - pv[:, 0, 0])
    f = f.copy()
    f[back] = f[back][:, ::-1]
    return f


def test_constant_triangle_renders_its_colour_and_the_background():
    v, f = _tri()
    f = _front(v, f)
    img = TO.render(v, f, _const_tex(1, 4, (0.2, 0.5, 0.8)), K16, EYE, np.zeros(3), 16, 16, near=0.0, far=10.0)
    assert img.shape == (3, 16, 16)
    inside = np.all(np.abs(img - np.array([0.2, 0.5, 0.8], np.float32)[:, None, None]) < 1e-6, 0)
    outside = np.all(img == 1.0, 0)
    edge = ~inside & ~outside
    assert inside.sum() > 20 and outside.sum() > 100 and 0 < edge.sum() < 40
    # anti-aliased edge pixels are quarter-step mixtures of the two colours
    mix = (img[0][edge] - 0.2) / 0.8
    np.testing.assert_allclose(mix * 4, np.round(mix * 4), atol=1e-5)
    # the projection flips y: the apex (y = +0.6 in camera space) is drawn towards the bottom rows after NR's own vertical flip
    rows = np.nonzero(inside.any(1))[0]
    cols_top, cols_bot = inside[rows[0]].sum(), inside[rows[-1]].sum()
    assert cols_top != cols_bot


def test_texture_axis_k_follows_the_barycentric_weight_of_corner_k():
    """texture index k = w_k (ts - 1) (kernel.cu:218-223): a texture that is the index along axis 0, divided by ts - 1, renders w_0"""
    v, f = _tri()
    f = _front(v, f)
    ts = 5
    tex = np.zeros((1, ts, ts, ts, 3), np.float32)
    tex[0, :, :, :, 0] = (np.arange(ts, dtype=np.float32) / (ts - 1))[:, None, None]
    keep = {}
    img = TO.render(v, f, tex, K16, EYE, np.zeros(3), 16, 16, near=0.0, far=10.0, anti_aliasing=False, keep=keep)
    fi = keep["face_index"]
    pv = TO.project(v, K16, EYE, np.zeros(3), 16)[f][0]
    _, w, _ = TO.rasterize(pv[None], 16, 0.0, 10.0)
    got = img[0][::-1][fi >= 0]                                   # undo the flip
    np.testing.assert_allclose(got, w[fi >= 0][:, 0], atol=2e-3)  # (clamped at ts - 1 - eps)
    np.testing.assert_allclose(keep["sampling_weight"][fi >= 0].sum(1), 1.0, atol=1e-6)


def test_nearer_triangle_wins_and_backfaces_are_culled():
    v0, f0 = _tri(z=3.0, s=0.9)
    v1, _ = _tri(z=2.0, s=0.4)
    v = np.concatenate([v0, v1])
    f = _front(v, np.array([[0, 1, 2], [3, 4, 5]], np.int32))
    tex = np.concatenate([_const_tex(1, 2, (1, 0, 0)), _const_tex(1, 2, (0, 1, 0))])
    img = TO.render(v, f, tex, K16, EYE, np.zeros(3), 16, 16, near=0.0, far=10.0, anti_aliasing=False)
    centre = img[:, 8, 8]
    np.testing.assert_allclose(centre, [0, 1, 0], atol=1e-6)      # the small near triangle covers the centre
    assert (np.all(np.abs(img - np.array([1, 0, 0], np.float32)[:, None, None]) < 1e-6, 0)).sum() > 10
    flipped = f[:, ::-1]
    img = TO.render(v, flipped, tex, K16, EYE, np.zeros(3), 16, 16, near=0.0, far=10.0, anti_aliasing=False)
    assert np.all(img == 1.0)                                      # everything culled: background only
    img = TO.render(v, f, tex, K16, EYE, np.zeros(3), 16, 16, near=2.5, far=10.0, anti_aliasing=False)
    np.testing.assert_allclose(img[:, 8, 8], [1, 0, 0], atol=1e-6)  # near plane removes the front triangle


def test_texture_gradient_matches_finite_differences():
    rng = np.random.default_rng(0)
    v0, _ = _tri(z=3.0, s=0.9)
    v1, _ = _tri(z=2.0, s=0.5, shift=(0.2, -0.1))
    v = np.concatenate([v0, v1])
    f = _front(v, np.array([[0, 1, 2], [3, 4, 5]], np.int32))
    ts = 3
    tex = rng.uniform(0.2, 0.8, (2, ts, ts, ts, 3)).astype(np.float32)
    target = rng.uniform(0, 1, (3, 8, 8)).astype(np.float32)
    cfg = dict(K=K16 / 2 * np.array([[1, 1, 1], [1, 1, 1], [2, 2, 2]], np.float32), R=EYE, t=np.zeros(3), orig_size=8, image_size=8, near=0.0, far=10.0)

    def loss(tx):
        return float(np.abs(TO.render(v, f, tx, **cfg).astype(np.float64) - target).sum())
    keep = {}
    img = TO.render(v, f, tex, keep=keep, **cfg)
    g = TO.texture_grad(np.sign(img - target).astype(np.float32), keep, 2, ts, 8)
    assert np.abs(g).sum() > 0
    idx = np.argwhere(np.abs(g) > 1e-3)
    for i in idx[rng.choice(len(idx), 12, replace=False)]:
        h = 1e-3
        tp, tm = tex.copy(), tex.copy()
        tp[tuple(i)] += h; tm[tuple(i)] -= h
        fd = (loss(tp) - loss(tm)) / (2 * h)
        assert fd == pytest.approx(float(g[tuple(i)]), rel=2e-2, abs=2e-3)


def test_view_schedule_helpers():
    poses = TO.gen_cam_views(np.array([0.1, 0.9, -0.2]), 18, 2.0, gl=True)
    assert len(poses) == 18
    for p in poses:
        np.testing.assert_allclose(p[:3, :3] @ p[:3, :3].T, np.eye(3), atol=1e-12)
        cam = -p[:3, :3].T @ p[:3, 3]                                # camera centre in the world
        assert np.linalg.norm(cam - [0.1, 0.9, -0.2]) == pytest.approx(2.0)
    c2w = TO.sphere2rot(2.0, 1.0, 0.5, t=[0.1, 0.9, -0.2])
    np.testing.assert_allclose(c2w[:3, :3].T @ c2w[:3, :3], np.eye(3), atol=1e-12)
    look = c2w[:3, 2]
    np.testing.assert_allclose(look, -(c2w[:3, 3] - [0.1, 0.9, -0.2]) / 2.0, atol=1e-12)   # z axis looks at the centre


def test_fit_reduces_the_loss():
    v, f = _tri(z=2.5, s=0.8)
    f = _front(v, f)
    fit = TO.TextureFit((v, f, _const_tex(1, 3, (0.9, 0.1, 0.3))), (v, f, _const_tex(1, 3, (0.5, 0.5, 0.5))), 8, 0.0, 10.0, lr=5e-2)
    Kk = np.array([[8.0, 0, 4], [0, 8.0, 4], [0, 0, 1]], np.float32)
    losses = [fit.step(Kk, EYE, np.zeros(3), 8)[0] for _ in range(12)]
    assert losses[-1] < 0.3 * losses[0]


def test_adam_update_is_torch_adam():
    """TextureFit's update against torch.optim.Adam (lr 1e-2, defaults) fed the same gradients: float32 rounding apart
    (torch's CPU kernels fuse some multiply-adds)"""
    import torch
    rng = np.random.default_rng(0)
    p0 = rng.uniform(0, 1, (5, 4, 4, 4, 3)).astype(np.float32)
    grads = [rng.standard_normal(p0.shape).astype(np.float32) * s for s in (1.0, 1e-3, 10.0, 0.0, 1e-6)]
    tp = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([tp], lr=1e-2)
    fit = TO.TextureFit(None, (None, None, p0.copy()), 16, 0.0, 10.0, lr=1e-2)
    for g in grads:
        opt.zero_grad(); tp.grad = torch.tensor(g); opt.step()
        fit.adam(g)
        np.testing.assert_allclose(fit.mesh[2], tp.detach().numpy(), atol=2e-7, rtol=0)
    assert np.abs(fit.mesh[2] - p0).max() > 0.03


def test_uv_texture_image_known_answers():
    """render_texture / render_texture_map (neural_renderer/renderer.py:294-346, texture_fitting.py:149-151): every face drawn at its
    UV triangle, front and back.  Constant-colour cubes: a pixel inside a face's UV triangle shows that face's colour whichever way
    the triangle is wound (fill back), the gaps show the background, depth is 1 on the faces and `far` elsewhere."""
    from texfit_cases import uv_atlas
    rng = np.random.default_rng(3)
    nf, ts, size = 6, 3, 24
    uv, uvf = uv_atlas(nf, cols=3)
    colours = rng.uniform(0.1, 0.9, (nf, 3)).astype(np.float32)
    tex = np.broadcast_to(colours[:, None, None, None, :], (nf, ts, ts, ts, 3)).copy()
    rgb, depth = TO.render_texture(uv, uvf, tex, size, near=0.0, far=5.0, background=(1, 1, 1), anti_aliasing=False)
    assert rgb.shape == (3, size, size) and depth.shape == (size, size)
    for i in range(nf):
        c = uv[3 * i:3 * i + 3].mean(0)                          # centroid of the UV triangle, in [0, 1]^2
        x, y = int(c[0] * size), size - 1 - int(c[1] * size)     # image row 0 is the top (vertical flip, rasterize.py:304-315)
        np.testing.assert_allclose(rgb[:, y, x], colours[i], atol=1e-6, err_msg=f"face {i}")
        assert abs(depth[y, x] - 1.0) < 1e-6                     # (1 / (w0 + w1 + w2) in float32)
    assert (depth == 5.0).any() and np.all(rgb[:, depth == 5.0] == 1.0)
    # an asymmetric cube: the back side samples the cube with axes 0 and 2 swapped - for a wound-the-other-way triangle that is
    # the same texel as the front side of the un-reversed one
    tex2 = rng.uniform(0, 1, (nf, ts, ts, ts, 3)).astype(np.float32)
    a, _ = TO.render_texture(uv, uvf, tex2, size, 0.0, 5.0, anti_aliasing=False)
    b, _ = TO.render_texture(uv, uvf[:, ::-1], tex2.transpose(0, 3, 2, 1, 4), size, 0.0, 5.0, anti_aliasing=False)
    np.testing.assert_array_equal(a, b)
    # anti-aliasing = the mean of the 2 x 2 block of the double-size render
    big, _ = TO.render_texture(uv, uvf, tex2, 2 * size, 0.0, 5.0, anti_aliasing=False)
    aa, _ = TO.render_texture(uv, uvf, tex2, size, 0.0, 5.0, anti_aliasing=True)
    want = big.reshape(3, size, 2, size, 2)
    np.testing.assert_allclose(aa, (want[:, :, 0, :, 0] + want[:, :, 0, :, 1] + want[:, :, 1, :, 0] + want[:, :, 1, :, 1]) * 0.25, atol=1e-7)
