"""HIP SMPL-X path (smpl_type='smplx': 55 joints, hand PCA, 135 joints with face landmarks) through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from bodyfitting_amd import native as N
from bodyfitting_amd import synthetic as S
from oracle import smplify_oracle as O

pytestmark = pytest.mark.gpu
import ref_drift as RD          # bands of the silhouette loops = K x the reference's own drift under perturbation


@pytest.fixture(scope="module")
def sx():
    model = S.make_model("smplx", seed=0)
    dev = N.DeviceModel(model, S.make_gmm(seed=0), device=0)
    yield model, dev
    dev.close()


def _params(prob, rng=None):
    p = {"global_transl": np.array([0.01, -0.02, 0.015]), "scale": np.array([1.05]), "pose": prob["init_pose"][0, 3:66],
         "betas": np.linspace(-0.4, 0.4, 10), "global_orient": prob["init_pose"][0, :3], "leye_pose": np.array([0.02, -0.01, 0.03]),
         "reye_pose": np.array([-0.02, 0.01, 0.0]), "left_hand_pose": np.linspace(-0.3, 0.3, 6), "right_hand_pose": np.linspace(0.2, -0.2, 6)}
    return {k: np.asarray(v, np.float64) for k, v in p.items()}


def _batch(dev, prob):
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    assert kp.shape[2] == 135
    b = N.FrameBatch(dev, 1, c2w.shape[1])
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    return b


def test_forward_vertices_and_135_joints(sx):
    model, dev = sx
    assert dev.n_params == 98 and dev.n_loss_joints == 135
    prob = S.make_problem_smplx(model, 0, 4)
    P = _params(prob)
    verts, joints = dev.forward_packed(N.pack_params(P)[None])
    m = O.to_torch_model(model, torch.float64)
    t = lambda k: torch.tensor(P[k][None], dtype=torch.float64)          # noqa: E731
    ref = O.smplx_forward(m, t("betas"), t("global_orient"), t("pose"), t("leye_pose"), t("reye_pose"), t("left_hand_pose"), t("right_hand_pose"))
    np.testing.assert_allclose(verts[0], ref["vertices"][0].numpy(), atol=3e-6)
    np.testing.assert_allclose(joints[0], ref["joints"][0].numpy(), atol=3e-6)      # incl. 51 static + 17 contour landmarks


def test_small_batch_forward_is_bitwise_the_single_frame_forward(sx):
    """SMPL-X, 5 frames in one launch (486 pose-feature rows streamed once for all of them) vs frame by frame"""
    model, dev = sx
    rng = np.random.default_rng(5)
    prob = S.make_problem_smplx(model, 0, n_views=2)
    base = N.pack_params(_params(prob))
    p = (base[None] + rng.normal(0, 0.05, (5, len(base)))).astype(np.float32)
    verts, joints = dev.forward_packed(p)
    for i in range(5):
        v1, j1 = dev.forward_packed(p[i:i + 1])
        np.testing.assert_array_equal(verts[i], v1[0])
        np.testing.assert_array_equal(joints[i], j1[0])


def test_loss_and_gradient_match_autograd(sx, gmm_bufs):
    model, dev = sx
    prob = S.make_problem_smplx(model, 0, 8)
    P = _params(prob)
    b = _batch(dev, prob)
    b.set_params(N.pack_params(P)[None])
    terms, grads = b.loss_grad()
    loss, t64, g64, _, _, _ = O.smplx_loss_and_grad(model, gmm_bufs, prob, P)
    for i, n in enumerate(("reprojection_loss", "pose_prior_loss", "angle_prior_loss", "shape_prior_loss")):
        assert terms[0, i] == pytest.approx(t64[n], rel=3e-6), n
    got = N.split_params(grads[0])
    for k in O.SMPLX_PARAMS:
        np.testing.assert_allclose(got[k], g64[k], atol=1e-5 * np.abs(g64[k]).max(), err_msg=k)
    b.close()


def test_fit_matches_reference_golden(sx):
    """the reference loop for smplx (hands + face keypoints, the unsqueezed-confidence quirk included): 40 iterations"""
    model, dev = sx
    g = load_golden("smplx_8view_40it.npz")
    prob = S.make_problem_smplx(model, frame=0, n_views=8)
    b = _batch(dev, prob)
    done = 0
    for k in (1, 2, 10, 40):
        b.fit(k - done)
        done = k
        got = N.split_params(b.get_params()[0])
        for n in O.SMPLX_PARAMS:
            np.testing.assert_allclose(got[n], g[f"it{k}_{n}"], rtol=0, atol=1e-4, err_msg=f"it{k} {n}")
    verts, joints, full_pose, _ = b.get_result()
    np.testing.assert_allclose(joints[0], g["joints"], atol=1e-4)
    np.testing.assert_allclose(verts[0][::53], g["vertices_sample"], atol=1e-4)
    np.testing.assert_allclose(full_pose[0], g["full_pose"], atol=1e-4)
    b.close()


def test_smplx_with_masks_runs_and_improves(sx):
    """BASELINE config 3 shape: SMPL-X + silhouette loss (ill-conditioned loop: progress, not trajectory, is asserted)"""
    from oracle.contour_oracle import border_pixels_rowmajor_all as extract_contours
    model, dev = sx
    prob = S.make_problem_smplx(model, frame=0, n_views=8, mask_frames=[1, 3, 5, 7])
    b = _batch(dev, prob)
    b.set_masks(np.array(prob["masks"])[None], [1, 3, 5, 7], [extract_contours(np.array(prob["masks"]) > 128)])
    l0 = b.mask_loss()[0][0]
    b.fit(15)
    assert np.isfinite(b.get_params()).all()
    assert b.mask_loss()[0][0] < l0
    g = load_golden("smplx_mask_8view_15it.npz")
    got = N.split_params(b.get_params()[0])
    worst = max(float(np.abs(got[n] - g[f"it15_{n}"]).max()) for n in O.SMPLX_PARAMS)
    print("smplx mask loop: max |param - reference| after 15 steps =", worst)
    band = RD.band(g, load_golden("sens_smplx_mask_8view_15it.npz"), [f"it15_{n}" for n in O.SMPLX_PARAMS])
    print("band = K x the reference's largest drift over the same 15 steps (ten perturbations, tests/ref_drift.py):", band)
    assert worst < band
    b.close()


def test_smplx_scan_fit_and_displacement_stage(gmm_bufs):
    """BASELINE config 5 in small: smpl_type='smplx' with use_mesh (constant scale scan_height / 1.7, closest-point
    loss after num_iters // 3, smplify.py:146-156,205-210) against the oracle loop, then the SMPL+D stage's first step
    against autograd (loss.py:233-288, smplify.py:228-247)"""
    from oracle import mesh_oracle as MO
    model = S.make_model("smplx", seed=0, nv=1200)
    dev = N.DeviceModel(model, S.make_gmm(seed=0), device=0)
    prob, sv, sf = S.make_scan_problem_smplx(model, 0, n_views=4, subdivide=0)
    want = O.fit_smplx(model, gmm_bufs, prob, num_iters=9, scan=(sv, sf))       # iterations 4..8 carry the scan loss
    scan = N.Scan(sv, sf)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, 4)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans([scan])
    b.fit(9)
    got = N.split_params(b.get_params()[0])
    for n in O.SMPLX_PARAMS:
        np.testing.assert_allclose(got[n], want["params"][n], rtol=0, atol=1e-4, err_msg=n)
    verts, joints, _, _ = b.get_result()
    np.testing.assert_allclose(verts[0], want["vertices"], atol=2e-4)
    np.testing.assert_allclose(joints[0], want["joints"], atol=2e-4)
    # SMPL+D, first Adam step: every vertex moves by lr * sign(gradient) = 5e-2 (Adam's normalised first step)
    bv = torch.tensor(verts, dtype=torch.float64)
    disp = torch.zeros_like(bv, requires_grad=True)
    faces_t = torch.as_tensor(np.asarray(model["faces"]), dtype=torch.long)
    ids, cpts, _ = MO.nearest_bruteforce(sv, sf, verts[0])
    tris = sv.astype(np.float64)[sf]
    fn = torch.as_tensor(np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]).astype(np.float32), dtype=torch.float64)
    norms = MO.compute_normal_torch((bv + disp)[0], faces_t)
    c = float((sv[:, 1].max() - sv[:, 1].min()) / 1.7)
    loss = MO.point_cloud_loss(bv + disp, torch.as_tensor(cpts, dtype=torch.float64)) + \
        (MO.normal_loss(fn[torch.as_tensor(ids, dtype=torch.long)], norms) + MO.normal_laplacian_smoothness(norms, faces_t)) * c * 0.1
    loss.backward()
    g = disp.grad.numpy()[0]
    b.fit_displacement(1)
    d = b.get_displacement()[0]
    sure = np.abs(g) > 1e-3 * np.abs(g).max()                                  # (well above Adam's eps: the step is lr * sign)
    np.testing.assert_allclose(d[sure], -0.05 * np.sign(g[sure]), atol=2e-5)
    from bodyfitting_amd import _lib
    m1 = np.empty((1, len(g), 3), np.float32)                                  # Adam's first moment = 0.1 * gradient
    _lib.check(_lib.load().bf_batch_debug_disp_moment(b._h, _lib.fptr(m1)))
    err = np.abs(m1[0] / 0.1 - g) / np.abs(g).max()        # (observed: 2e-4 relative to the largest entry; fp32 normals; a vertex whose
    assert np.mean(err < 3e-4) > 0.995 and err.max() < 2e-2   # closest face flips between the fp32 and fp64 base mesh differs by more)
    b.close(); scan.close(); dev.close()


def _fit_in_subprocess(env, n_iters=12, masks=False):
    """the fitted parameters of a small SMPL-X problem from a fresh process with `env` set (the launch-form switches are read once per process)"""
    import json, os, subprocess, sys
    code = f"""
import json, sys
import numpy as np
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from bodyfitting_amd import native as N, synthetic as S
model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
mf = [1, 3, 5, 7] if {masks!r} else None
prob = S.make_problem_smplx(model, frame=0, n_views=8, mask_frames=mf)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
b = N.FrameBatch(dev, 1, 8)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
if mf: b.set_masks(np.array(prob["masks"])[None], mf, None)
b.fit({n_iters})
print("RESULT " + json.dumps(b.get_params()[0].astype(float).tolist()))
"""
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return np.asarray(json.loads(line[7:]), np.float32)


def test_resident_fit_launch_equals_one_launch_per_iteration():
    """the dense schedule with the fit kernel resident (doorbells) against one fit launch per iteration: the pose state
    both consume is bf_pose_state_kernel's arithmetic, so the fitted parameters are the same bits"""
    a = _fit_in_subprocess({"BF_DENSE_PERSISTENT": "1"})
    b = _fit_in_subprocess({"BF_DENSE_PERSISTENT": "0"})
    assert np.isfinite(a).all() and np.abs(a).sum() > 1
    np.testing.assert_array_equal(a, b)


def test_sub_model_loop_matches_the_full_model_loop():
    """mesh passes on the sampled-first sub-model (a third of the vertices) against the full model: same vertices, same per-vertex
    arithmetic, different tile sums - float32 summation order, before the silhouette's discontinuities can amplify it (kp-only loop)"""
    a = _fit_in_subprocess({"BF_DENSE_SUBMODEL": "1"})
    b = _fit_in_subprocess({"BF_DENSE_SUBMODEL": "0"})
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-5)
    # round 5: the iterations before the dense losses switch on (i <= num_iters // 3: here 0..4 of 12) run on the keypoint-only
    # sub-model - selector vertices, landmark corners, the extra regressor's support - the later ones on the sampled-first one
    c = _fit_in_subprocess({"BF_DENSE_SUBMODEL": "1", "BF_DENSE_SUBMODEL_KP": "0"})
    np.testing.assert_allclose(a, c, rtol=0, atol=2e-5)
    assert np.abs(a - c).max() > 0                     # (another summation order: the switch really selects another sub-model)
    a = _fit_in_subprocess({"BF_DENSE_SUBMODEL": "1"}, n_iters=6, masks=True)          # 2 keypoint-only + 4 silhouette iterations
    b = _fit_in_subprocess({"BF_DENSE_SUBMODEL": "0"}, n_iters=6, masks=True)
    # (four silhouette iterations: as far apart as the reference's own perturbed runs are after their first ten)
    g15 = load_golden("smplx_mask_8view_15it.npz")
    np.testing.assert_allclose(a, b, rtol=0, atol=RD.band(g15, load_golden("sens_smplx_mask_8view_15it.npz"), [f"it15_{n}" for n in O.SMPLX_PARAMS]))


def test_dense_fit_is_the_same_bits_run_to_run(sx):
    """the resident fit launch and the dense kernels exchange doorbells, not data races: a silhouette + keypoint fit repeated on a
    re-armed batch, and on a second batch, gives identical parameters and vertices"""
    from bodyfitting_amd import _lib
    model, dev = sx
    prob = S.make_problem_smplx(model, frame=0, n_views=8, mask_frames=[1, 3, 5, 7])
    runs = []
    for _ in range(2):
        b = _batch(dev, prob)
        b.set_masks(np.array(prob["masks"])[None], [1, 3, 5, 7], None)
        for _ in range(2):
            b.fit(12, flags=_lib.FIT_RESET)
            runs.append((b.get_params().copy(), b.get_result()[0].copy()))
        b.close()
    for p, v in runs[1:]:
        np.testing.assert_array_equal(p, runs[0][0])
        np.testing.assert_array_equal(v, runs[0][1])


def test_sub_model_keeps_every_vertex_of_the_extra_regressor(gmm_bufs):
    """An SMPL-X-kind model WITH a J_regressor_extra (the ABI allows it, models/smpl.py:62-64 is where SMPL gets its own): the dense
    loop's sampled-first sub-model must contain every vertex that carries regressor weight, or the extra joints - here three of the
    loss joints - would be formed from partial sums.  Sub-model on == off (float32 summation order apart)."""
    import os
    model = dict(S.make_model("smplx", seed=0))
    nv = model["v_template"].shape[0]
    rng = np.random.default_rng(11)
    reg = np.zeros((3, nv), np.float32)
    for r in range(3):
        ids = rng.choice(np.arange(nv)[np.arange(nv) % 4 != 0], size=12, replace=False)      # vertices the silhouette sample does NOT contain
        w = rng.uniform(0.2, 1.0, size=12)
        reg[r, ids] = w / w.sum()
    model["J_regressor_extra"] = reg
    jm = np.asarray(model["joint_map"]).copy()
    jm[jm >= 76] += 3                                  # all-joints layout: chain 55 | selector 21 | extra 3 | landmarks
    jm[[1, 8, 15]] = [76, 77, 78]                      # three body loss joints now come from the extra regressor
    model["joint_map"] = jm
    dev = N.DeviceModel(model, S.make_gmm(seed=0), device=0)
    prob = S.make_problem_smplx(S.make_model("smplx", seed=0), frame=0, n_views=8)
    # the regressed loss joints' gradient reaches every vertex of their rows: against autograd of the fp64 restatement
    P = _params(prob)
    b = _batch(dev, prob)
    b.set_params(N.pack_params(P)[None])
    terms, grads = b.loss_grad()
    b.close()
    _, t64, g64, _, _, _ = O.smplx_loss_and_grad(model, gmm_bufs, prob, P)
    assert terms[0, 0] == pytest.approx(t64["reprojection_loss"], rel=3e-6)
    got = N.split_params(grads[0])
    for k in O.SMPLX_PARAMS:
        np.testing.assert_allclose(got[k], g64[k], atol=1e-5 * np.abs(g64[k]).max(), err_msg=k)
    out = {}
    # "1": both sub-models (iterations 0..2 of 8 on the keypoint-only one, 3..7 on the sampled-first one), "kp0": sampled-first only,
    # "0": the full model throughout
    for flag, env in (("1", {"BF_DENSE_SUBMODEL": "1"}), ("kp0", {"BF_DENSE_SUBMODEL": "1", "BF_DENSE_SUBMODEL_KP": "0"}), ("0", {"BF_DENSE_SUBMODEL": "0"})):
        os.environ.update(env)
        try:
            b = _batch(dev, prob)
            terms, grads = b.loss_grad()
            b.fit(8)
            out[flag] = (terms.copy(), grads.copy(), b.get_params().copy())
            b.close()
        finally:
            for k in env:
                del os.environ[k]
    assert np.abs(out["1"][1]).max() > 1.0
    for flag in ("1", "kp0"):
        np.testing.assert_allclose(out[flag][0], out["0"][0], rtol=2e-6)
        np.testing.assert_allclose(out[flag][1], out["0"][1], atol=2e-5 * np.abs(out["0"][1]).max())
        np.testing.assert_allclose(out[flag][2], out["0"][2], atol=2e-5)
    assert np.abs(out["1"][2] - out["kp0"][2]).max() > 0
    dev.close()


def test_staged_inputs_on_the_dense_path(sx):
    """bf_batch_stage_inputs with an SMPL-X batch (135 loss joints: the keypoint loss is a dense kernel, the fit kernel is resident and
    paced by doorbells): frames staged back to back == the same frames through the synchronous setters, bit for bit; views the
    staged frame does not see (confidence 0) and a per-frame divisor travel with it"""
    from bodyfitting_amd import _lib
    model, dev = sx
    probs = [S.make_problem_smplx(model, frame=f, n_views=8) for f in (0, 1, 2)]
    packed = [N.pack_problem([p]) for p in probs]
    packed[1][2][0, 3] = 0.0                                   # frame 1: view 3 without a detection ...
    nd1 = np.array([7], np.int32)                              # ... and the divisor the reference would use (len(use_frames), loss.py:197)
    want = []
    for i, (c2w, K, kp, ndiv, betas, pose) in enumerate(packed):
        b = N.FrameBatch(dev, 1, 8)
        b.set_cameras(packed[0][0], packed[0][1]); b.set_keypoints(kp, nd1 if i == 1 else ndiv); b.set_init(betas, pose)
        b.fit(12)
        want.append((b.get_params().copy(), b.get_result()[0].copy()))
        b.close()
    b = N.FrameBatch(dev, 1, 8)
    b.set_cameras(packed[0][0], packed[0][1])
    for i in (0, 1, 2, 1):
        _, _, kp, ndiv, betas, pose = packed[i]
        b.stage_inputs(kp, nd1 if i == 1 else ndiv, betas, pose)
        b.fit(12, flags=_lib.FIT_RESET | _lib.FIT_FETCH)
        np.testing.assert_array_equal(b.get_params(), want[i][0])
        np.testing.assert_array_equal(b.get_result()[0], want[i][1])
    b.close()


def test_stream_of_smplx_frames_with_silhouettes_equals_the_call_path(sx, tmp_path):
    """SMPLify.stream for smpl_type='smplx' with use_mask (and, for one frame, a scan): the capture's frame loop with the next frame
    prepared under the running fit == SMPLify.__call__ frame by frame, bit for bit (staged inputs + re-arm inside the fit against the
    synchronous setters; a scan built while another frame's fit is in flight)"""
    from bodyfitting_amd import assets
    from bodyfitting_amd.io import save_obj_mesh
    from bodyfitting_amd.smplify import SMPLify
    model, _ = sx
    assets.register_model(model, "smplx", "neutral")
    assets.register_gmm(S.make_gmm(seed=0))
    mask_frames = [1, 3, 5, 7]
    probs = [S.make_problem_smplx(model, frame=f, n_views=8, mask_frames=mask_frames) for f in (0, 2, 1)]
    _, sv, sf = S.make_scan_problem_smplx(model, 3, n_views=8, subdivide=0)
    meshfile = str(tmp_path / "scan.obj")
    save_obj_mesh(meshfile, sv, sf)
    fitter = SMPLify(smpl_type="smplx", num_iters=24, gender="neutral", device=0, debug=False)
    p0 = probs[0]

    def items():
        for i, p in enumerate(probs):
            yield ((p["init_betas"], p["init_pose"]), p["keypoints"], p["masks"], meshfile if i == 1 else None)
    streamed = list(fitter.stream(items(), p0["c2ws"], p0["Ks"], use_frames=p0["use_frames"], imsize=512, mask_frames=mask_frames))
    assert len(streamed) == 3
    for i, (p, res) in enumerate(zip(probs, streamed)):
        one = fitter((p["init_betas"], p["init_pose"]), p0["c2ws"], p0["Ks"], p["keypoints"], use_mask=True, masks=p["masks"],
                     use_frames=p0["use_frames"], mask_frames=mask_frames, imsize=512, use_mesh=(i == 1), meshfile=meshfile if i == 1 else None)
        for key in ("vertices", "joints", "pose", "betas", "global_orient", "global_transl", "scale", "full_pose", "left_hand_pose", "leye_pose"):
            assert np.array_equal(res[key], one[key]), (i, key)
    fitter.close()
