"""BASELINE configs 3 and 5 AT THEIR STATED SIZE through the C ABI (rounds 1-2 held them on reduced models / short loops only).

Config 3: SMPL-X (10,475 v, 55 joints, 135 output joints), 48 views + 8 silhouettes at 512 x 512, 200 iterations - against the
imported reference's run of exactly that (tests/golden/cfg3_smplx_48view_8mask_200it_base.npz) and its own drift under
perturbation (tests/ref_drift.py).  Config 5: 8 SMPL-X frames with a ~84k-triangle scan each, 300 iterations + 300 SMPL+D
iterations - the reference cannot run that size on this container's CPU in useful time (its stand-in searcher is brute force),
so the full-size loop is held by size-independent properties: batch == single frames, run == re-run, resident launch == one
launch per iteration (all bit for bit), and the objective / distance distribution improving by stated factors."""
import os

import numpy as np
import pytest

from conftest import load_golden
from bodyfitting_amd import _lib, native as N, synthetic as S
from oracle import smplify_oracle as O
import ref_drift as RD

pytestmark = pytest.mark.gpu
MASK_FRAMES = list(range(0, 48, 6))


@pytest.fixture(scope="module")
def sx():
    model = S.make_model("smplx", seed=0)
    dev = N.DeviceModel(model, S.make_gmm(seed=0), device=0)
    yield model, dev
    dev.close()


def test_config3_as_stated_against_the_reference(sx):
    from oracle.contour_oracle import border_pixels_rowmajor_all as extract_contours      # (what the golden's cv2 stub returned)
    model, dev = sx
    g = load_golden("cfg3_smplx_48view_8mask_200it_base.npz")
    assert str(g["model_digest"]) == S.model_digest(model)
    prob = S.make_problem_smplx(model, frame=0, n_views=48, mask_frames=MASK_FRAMES)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, 48)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    b.set_masks(np.array(prob["masks"])[None], MASK_FRAMES, [extract_contours(np.array(prob["masks"]) > 128)])
    # the loss the loop computes in its first iteration (multiview_keypoint_loss's dict, loss.py:219-224)
    terms, _ = b.loss_grad()
    for i, n in enumerate(("reprojection_loss", "pose_prior_loss", "angle_prior_loss", "shape_prior_loss")):
        assert terms[0, i] == pytest.approx(float(g[f"it1_loss_{n}"]), rel=3e-5), n
    done, errs = 0, {}
    for k in (1, 66, 67, 68, 200):
        b.fit(k - done, N.make_hyper(dense_after=66))        # 200 // 3: ONE reference loop cut at the golden's snapshots
        done = k
        got = N.split_params(b.get_params()[0])
        errs[k] = max(float(np.abs(got[n] - g[f"it{k}_{n}"]).max()) for n in O.SMPLX_PARAMS)
    ref = {k: RD.cfg3_drift([f"it{k}_{n}" for n in O.SMPLX_PARAMS]) for k in (66, 67, 68, 200)}
    print("config 3 as stated: max |param - reference| per snapshot", errs, "| the reference's own largest drift under ten perturbations", ref)
    for k in (68, 200):
        print("  iteration", k, ":", RD.position(errs[k], RD.cfg3_drifts([f"it{k}_{n}" for n in O.SMPLX_PARAMS])))
    # iterations 1..67 are keypoint-only (the silhouette loss is active for loop index i > 200 // 3 = 66, i.e. from the 68th step on,
    # smplify.py:197): the north-star tolerance
    assert errs[1] < 1e-5 and errs[66] < 1e-4 and errs[67] < 1e-4
    # the first silhouette iteration: a tenth of one Adam step (lr 1e-2) - at most a near-tie flip of a nearest-vertex choice away
    # (the reference moves 1.3e-4 from itself in this step under a one-ulp nudge of the initial pose)
    assert errs[68] < max(1e-3, RD.K * ref[68]), (errs[68], ref[68])
    assert errs[200] < max(RD.FLOOR, RD.K * ref[200]), (errs[200], ref[200])
    verts, joints, full_pose, _ = b.get_result()
    # the 17 dynamic contour landmarks are picked by an integer look-up on the neck's yaw: between two runs of the REFERENCE they jump
    # by 25 cm; the other 118 joints are continuous in the parameters
    base, var = RD.cfg3_variants()
    jd = max(float(np.abs(var[v]["joints"][:118] - base["joints"][:118]).max()) for v in RD.VARIANTS)
    np.testing.assert_allclose(joints[0][:118], g["joints"][:118], atol=max(RD.FLOOR, RD.K * jd))
    np.testing.assert_allclose(verts[0][::53], g["vertices_sample"], atol=max(RD.FLOOR, RD.K * RD.cfg3_drift(["vertices_sample"])))
    # the silhouette term of the objective at the end state, evaluated by the same kernel at both parameter sets
    end_got = float(b.mask_loss()[0][0])
    names = O.SMPLX_PARAMS
    b.set_params(N.pack_params({n: g[f"it200_{n}"] for n in names})[None])
    end_ref = float(b.mask_loss()[0][0])
    ends = []
    for v in RD.VARIANTS:
        b.set_params(N.pack_params({n: var[v][f"it200_{n}"] for n in names})[None])
        ends.append(float(b.mask_loss()[0][0]))
    spread = max(abs(e - end_ref) for e in ends) / end_ref
    print("silhouette loss at the end state: HIP", end_got, "reference", end_ref, "reference variants", ends)
    print("  relative to the reference's:", RD.position(abs(end_got - end_ref) / end_ref, [abs(e - end_ref) / end_ref for e in ends]))
    assert abs(end_got - end_ref) / end_ref < max(0.01, RD.K * spread)
    b.close()


def _cfg5_items(model, n):
    return [S.make_scan_problem_smplx(model, frame=f, n_views=48) for f in range(n)]


def _run_cfg5(dev, items, scans, frames, iters=300, disp=300):
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([items[f][0] for f in frames])
    b = N.FrameBatch(dev, len(frames), 48)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans([scans[f] for f in frames])
    b.fit(iters, flags=_lib.FIT_FETCH)
    params = b.get_params().copy()
    verts = b.get_result()[0].copy()
    b.fit_displacement(disp)
    d = b.get_displacement().copy()
    b.close()
    return params, verts, d


def test_config5_as_stated_is_deterministic_and_improves(sx):
    """8 SMPL-X frames x 48 views, one ~84k-triangle scan per frame, 300 iterations (scan loss after 100) + 300 SMPL+D iterations"""
    model, dev = sx
    items = _cfg5_items(model, 8)
    assert 80_000 < len(items[0][2]) < 90_000
    scans = [N.Scan(sv, sf) for _, sv, sf in items]
    full = _run_cfg5(dev, items, scans, list(range(8)))
    # run == re-run, bit for bit (fixed-order reductions; the resident launch exchanges doorbells, not races)
    again = _run_cfg5(dev, items, scans, list(range(8)))
    for a, b_ in zip(full, again):
        np.testing.assert_array_equal(a, b_)
    # the resident fit launch == one fit launch per iteration, bit for bit
    os.environ["BF_DENSE_PERSISTENT"] = "0"
    try:
        plain = _run_cfg5(dev, items, scans, list(range(8)))
    finally:
        del os.environ["BF_DENSE_PERSISTENT"]
    for a, b_ in zip(full, plain):
        np.testing.assert_array_equal(a, b_)
    # the batch of 8 == eight single-frame fits, bit for bit (frames are independent)
    for f in (0, 3, 7):
        one = _run_cfg5(dev, items, scans, [f])
        np.testing.assert_array_equal(one[0][0], full[0][f])
        np.testing.assert_array_equal(one[1][0], full[1][f])
        np.testing.assert_array_equal(one[2][0], full[2][f])
    # what the loops achieve: point-to-scan distances of the fitted mesh against the keypoint-only fit, and of the SMPL+D mesh
    kp_only = N.FrameBatch(dev, 8, 48)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([it[0] for it in items])
    kp_only.set_cameras(c2w, K); kp_only.set_keypoints(kp, ndiv); kp_only.set_init(betas, pose); kp_only.set_scans(scans)
    kp_only.fit(100, N.make_hyper(dense_after=1000), flags=_lib.FIT_FETCH)          # the first third: no scan loss yet
    v100 = kp_only.get_result()[0]
    kp_only.close()
    ratios = []
    for f in range(8):
        dist = lambda v: np.linalg.norm(v - scans[f].nearest_points(v.astype(np.float32))[0], axis=1)      # noqa: E731
        d100, d300, dd = dist(v100[f]), dist(full[1][f]), dist(full[1][f] + full[2][f])
        print(f"frame {f}: mean distance to the scan after 100 keypoint-only iterations {d100.mean() * 1e3:.2f} mm, after the 300-iteration fit "
              f"{d300.mean() * 1e3:.2f} mm, after SMPL+D {dd.mean() * 1e3:.2f} mm (median {np.median(dd) * 1e3:.2f}, p95 {np.percentile(dd, 95) * 1e3:.2f})")
        ratios.append(float(d300.mean() / d100.mean()))
        assert d300.mean() < 1.2 * d100.mean()                       # the closest-point loss pulls the body onto the scan (below: on average) ...
        assert dd.mean() < 0.6 * d300.mean() and np.median(dd) < 0.4 * np.median(d300)     # ... and the displacement stage closes most of what is left
        assert np.isfinite(full[2][f]).all()
    # (the closest-point term is ~10 % of the objective and competes with the keypoints: frame 2 stays at 13 mm, the others halve)
    print("mean distance after the fit / after the keypoint-only third, per frame:", ratios)
    assert np.mean(ratios) < 0.8 and np.median(ratios) < 0.7
    for s in scans:
        s.close()
