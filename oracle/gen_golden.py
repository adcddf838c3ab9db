"""Generate tests/golden/*.npz by running the UNMODIFIED reference on torch CPU.

TEST INFRASTRUCTURE ONLY - runs in the build container, where /root/reference exists; it is never
executed on the GPU box (nothing there may read /root/reference).  The committed .npz files hold
data only: seeds / small inputs and the reference's outputs.

What is imported from /root/reference, untouched: smplify/smplify.py (SMPLify.__call__ loop),
smplify/loss.py, smplify/prior.py, models/smpl.py, config.py, constants.py.
What is stubbed (absent third-party modules that the hot path imports but never executes for
smpl_type='smpl', keypoint-only): cv2, torchgeometry, mesh_grid, trimesh, imageio, neural_renderer,
torchvision, scipy.misc.face, utils.camera (raises on numpy>=1.24, camera.py:90-92).
What is stood in: `smplx` (oracle/smplx_standin) - see its docstring for the parity consequence.

Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
from __future__ import annotations

import os
import pickle
import sys
import tempfile
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
GOLDEN = os.path.join(REPO, "tests", "golden")


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def install_reference_imports():
    """Make `import smplify.smplify` from /root/reference work in this container."""
    import torch

    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle", "smplx_standin"))
    sys.path.insert(0, REFERENCE)
    _stub("cv2")
    _stub("torchgeometry", angle_axis_to_rotation_matrix=None)
    _stub("mesh_grid", insert_grid_surface=None, cumsum=None, search_nearest_point=None,
          search_inside_mesh=None, search_intersect=None, search_nearest_point_backward=None)
    _stub("trimesh")
    _stub("imageio")
    _stub("neural_renderer")
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.models.resnet = _stub("torchvision.models.resnet")
    tv.transforms = _stub("torchvision.transforms", Normalize=None)
    import scipy.misc
    if not hasattr(scipy.misc, "face"):
        scipy.misc.face = None
    _stub("utils.camera")            # real module body raises on numpy >= 1.24
    try:
        import matplotlib  # noqa: F401  (smplify.py:2 imports matplotlib.dviread)
    except ImportError:
        mpl = _stub("matplotlib")
        mpl.dviread = _stub("matplotlib.dviread")
    torch.set_num_threads(1)


def write_data_dir(tmp, model, gmm):
    """The files the reference opens relative to CWD (config.py:1-2, prior.py:124-128)."""
    os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
    np.save(os.path.join(tmp, "data", "J_regressor_extra.npy"), model["J_regressor_extra"])
    np.save(os.path.join(tmp, "data", "J_regressor_h36m.npy"), model["J_regressor_h36m"])
    with open(os.path.join(tmp, "data", "gmm_08.pkl"), "wb") as f:
        pickle.dump(gmm, f)


def run_reference_fit(problem, num_iters, snapshots=()):
    """Run reference SMPLify.__call__ (smplify.py:84-250) and capture parameter snapshots.

    Snapshots are taken by wrapping torch.optim.Adam.step - the reference code itself is untouched."""
    import torch
    from smplify.smplify import SMPLify

    snaps = {}
    orig_step = torch.optim.Adam.step
    counter = {"n": 0}

    def step(self, *a, **k):
        r = orig_step(self, *a, **k)
        counter["n"] += 1
        if counter["n"] in snapshots:
            g = self.param_groups
            snaps[counter["n"]] = {
                "global_transl": g[0]["params"][0].detach().numpy()[0].copy(),
                "scale": g[1]["params"][0].detach().numpy()[0].copy(),
                "pose": g[2]["params"][0].detach().numpy()[0].copy(),
                "betas": g[3]["params"][0].detach().numpy()[0].copy(),
                "global_orient": g[4]["params"][0].detach().numpy()[0].copy(),
            }
        return r

    torch.optim.Adam.step = step
    try:
        fitter = SMPLify(smpl_type="smpl", num_iters=num_iters, gender="neutral",
                         device=torch.device("cpu"), debug=False)
        net_output = (torch.from_numpy(problem["init_betas"].copy()),
                      torch.from_numpy(problem["init_pose"].copy()))
        t0 = time.perf_counter()
        res = fitter(net_output, problem["c2ws"], problem["Ks"], problem["keypoints"], None,
                     use_frames=problem["use_frames"], imsize=problem["imsize"])
        wall = time.perf_counter() - t0
    finally:
        torch.optim.Adam.step = orig_step
    return res, snaps, wall


def reference_loss_terms(model, gmm, problem, params):
    """reference multiview_keypoint_loss (loss.py:139-230) + autograd gradient at `params`, fp32."""
    import torch
    from models.smpl import SMPL
    from smplify.loss import multiview_keypoint_loss
    from smplify.prior import MaxMixturePrior
    import config

    prior = MaxMixturePrior(prior_folder="data", num_gaussians=8, dtype=torch.float32)
    smpl = SMPL(config.SMPL_MODEL_DIR, batch_size=1, gender="neutral", create_transl=True)
    p = {k: torch.tensor(np.asarray(v, np.float32).reshape(1, -1), requires_grad=True) for k, v in params.items()}
    out = smpl(global_orient=p["global_orient"], body_pose=p["pose"], betas=p["betas"], return_full_pose=True)
    c = problem["constant_scale"]
    mj = (out.joints + p["global_transl"]) * p["scale"] * c
    bv = (out.vertices + p["global_transl"]) * p["scale"] * c
    w2cs = torch.inverse(torch.from_numpy(np.array(problem["c2ws"])).float())
    loss, terms = multiview_keypoint_loss(w2cs, problem["Ks"], problem["keypoints"], mj, p["pose"], p["betas"],
                                          problem["use_frames"], prior, imsize=problem["imsize"])
    loss.backward()
    return (float(loss), {k: float(np.asarray(v).reshape(-1)[0]) for k, v in terms.items()},
            {k: v.grad.numpy()[0].copy() for k, v in p.items()}, mj.detach().numpy()[0], bv.detach().numpy()[0])


class StandInMeshGridSearcher:
    """Replaces utils.mesh_grid_searcher.MeshGridSearcher (whose CUDA extension cannot be built here) inside the imported reference:
    same interface; `set_mesh`'s grid (utils/mesh_grid_searcher.py:56-79, insert_grid_surface) and the search of
    search_nearest_point_kenerel IN THE REFERENCE'S OWN float32 ARITHMETIC (oracle/nearest_ref.c: mesh_grid_kernel.cu:12-109, 239-353 +
    matrix.h, the latter held bit for bit to the reference's header).  FUSED = True selects the build of that restatement in which the
    compiler may fuse multiply-adds, as nvcc's default does for the reference: one of the perturbations of sensitivity_goldens.
    (Rounds 1-3 used a float64 brute-force search by the same rule here.)"""
    FUSED = False

    def __init__(self, verts=None, faces=None, device="cpu"):
        from oracle import mesh_oracle as MO
        self.verts = np.ascontiguousarray(np.asarray(verts), np.float32)
        self.faces = np.ascontiguousarray(np.asarray(faces), np.int32)
        step, num, origin = MO.grid_params(self.verts)
        tri_num, tri_idx = MO.insert_grid_surface(self.verts, self.faces, step, origin, num)
        self.grid = (step, num, origin, tri_num, tri_idx)

    def nearest_points(self, points):
        import torch
        from oracle import nearest_ref as NR
        ids, pts, _, _ = NR.search_nearest(self.verts, self.faces, points.detach().cpu().numpy().reshape(-1, 3), self.grid, fused=self.FUSED)
        return torch.from_numpy(pts), torch.from_numpy(ids)


def run_reference_scan_fit(problem, meshfile, num_iters, snapshots=(), displacement=False):
    """reference SMPLify.__call__ with use_mesh=True (smplify.py:146-156,205-210,228-247)."""
    import torch
    import smplify.smplify as RS

    RS.MeshGridSearcher = StandInMeshGridSearcher
    snaps, disp_snaps = {}, {}
    orig_step = torch.optim.Adam.step
    counter = {"n": 0}

    def step(self, *a, **k):
        r = orig_step(self, *a, **k)
        g = self.param_groups
        if len(g) > 1:
            counter["n"] += 1
            if counter["n"] in snapshots:
                snaps[counter["n"]] = {n: g[i]["params"][0].detach().numpy()[0].copy() for i, n in
                                       enumerate(("global_transl", "scale", "pose", "betas", "global_orient"))}
        else:
            counter["d"] = counter.get("d", 0) + 1
            if counter["d"] in snapshots:
                disp_snaps[counter["d"]] = g[0]["params"][0].detach().numpy()[0].copy()
        return r

    torch.optim.Adam.step = step
    try:
        fitter = RS.SMPLify(smpl_type="smpl", num_iters=num_iters, gender="neutral", device=torch.device("cpu"), debug=False)
        net_output = (torch.from_numpy(problem["init_betas"].copy()), torch.from_numpy(problem["init_pose"].copy()))
        res = fitter(net_output, problem["c2ws"], problem["Ks"], problem["keypoints"], None,
                     use_frames=problem["use_frames"], imsize=problem["imsize"], use_mesh=True, meshfile=meshfile,
                     displacement=displacement)
    finally:
        torch.optim.Adam.step = orig_step
    return res, snaps, disp_snaps


def scan_goldens():
    """config 5 in miniature: reduced 690-vertex model so the brute-force stand-in search stays cheap."""
    import smplx
    from bodyfitting_amd import synthetic as S
    from bodyfitting_amd.io import save_obj_mesh

    model = S.make_model("smpl", seed=0, nv=690)
    gmm = S.make_gmm(seed=0)
    smplx.MODEL_REGISTRY["smpl"] = model
    tmp = tempfile.mkdtemp(prefix="bf_golden_scan_")
    write_data_dir(tmp, model, gmm)
    os.chdir(tmp)
    prob, sv, sf = S.make_scan_problem(model, frame=0, n_views=8)
    meshfile = os.path.join(tmp, "scan.obj")
    save_obj_mesh(meshfile, sv, sf)
    res, snaps, dsn = run_reference_scan_fit(prob, meshfile, 30, snapshots=(1, 11, 12, 20, 30), displacement=True)
    np.savez_compressed(os.path.join(GOLDEN, "scan_nv690_30it.npz"), frame=0, n_views=8, num_iters=30, nv=690,
                        model_digest=S.model_digest(model), scan_verts=sv, scan_faces=sf,
                        vertices=res["vertices"], joints=res["joints"], displacement=res["displacement"],
                        final_global_transl=res["global_transl"], **flat_snaps(snaps),
                        **{f"disp{k}": v for k, v in dsn.items()})
    print("scan golden: max |disp|", np.abs(res["displacement"]).max())


def scan_goldens_long():
    """config 5's iteration counts on the reduced model: the reference loop with use_mesh=True for 300 iterations (scan loss
    after 100) and its SMPL+D stage for 300 more.  The stage is chaotic under round-off, so what the tests hold
    against this golden is the END STATE: fitted parameters of the first loop, and the distribution of point-to-scan distances
    and the normal / laplacian energies of base + displacement after the second."""
    import smplx
    from bodyfitting_amd import synthetic as S
    from bodyfitting_amd.io import save_obj_mesh

    model = S.make_model("smpl", seed=0, nv=690)
    gmm = S.make_gmm(seed=0)
    smplx.MODEL_REGISTRY["smpl"] = model
    tmp = tempfile.mkdtemp(prefix="bf_golden_scan300_")
    write_data_dir(tmp, model, gmm)
    os.chdir(tmp)
    prob, sv, sf = S.make_scan_problem(model, frame=0, n_views=8)
    meshfile = os.path.join(tmp, "scan.obj")
    save_obj_mesh(meshfile, sv, sf)
    t0 = time.perf_counter()
    res, snaps, dsn = run_reference_scan_fit(prob, meshfile, 300, snapshots=(100, 101, 300), displacement=True)
    print("scan 300+300:", time.perf_counter() - t0, "s")
    np.savez_compressed(os.path.join(GOLDEN, "scan_nv690_300it.npz"), frame=0, n_views=8, num_iters=300, nv=690,
                        model_digest=S.model_digest(model), vertices=res["vertices"], joints=res["joints"],
                        displacement=res["displacement"], **flat_snaps(snaps), **{f"disp{k}": v for k, v in dsn.items()})


def install_cv2_contour_stub():
    """cv2.findContours (OpenCV-3 three-value API, as loss.py:79 unpacks it) backed by oracle.contour_oracle.border_pixels_rowmajor."""
    import cv2
    from oracle.contour_oracle import border_pixels_rowmajor as extract_contour

    def findContours(img, mode, method):
        c = extract_contour(np.asarray(img) > 0).astype(np.int32)
        return None, [c[:, None, :]], None

    cv2.findContours = findContours
    cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_NONE = 0, 1


def mask_goldens():
    """(a) reference multview_mask_loss (loss.py:85-130) + autograd at one parameter set; (b) the reference loop
    with use_mask=True (smplify.py:138-144,197-199)."""
    import torch
    import smplx
    from bodyfitting_amd import synthetic as S
    from oracle.contour_oracle import border_pixels_rowmajor_all as extract_contours
    from oracle import smplify_oracle as O

    model = S.make_model("smpl", seed=0)
    gmm = S.make_gmm(seed=0)
    smplx.MODEL_REGISTRY["smpl"] = model
    tmp = tempfile.mkdtemp(prefix="bf_golden_mask_")
    write_data_dir(tmp, model, gmm)
    os.chdir(tmp)
    install_cv2_contour_stub()
    from smplify.loss import multview_mask_loss
    mask_frames = [1, 3, 5, 7]
    prob = S.make_problem(model, frame=0, n_views=8, mask_frames=mask_frames)
    # (a) function level, at the initial estimate
    m = O.to_torch_model(model, torch.float32)
    out = O.smpl_forward(m, torch.tensor(prob["init_betas"]), torch.tensor(prob["init_pose"][:, :3]),
                         torch.tensor(prob["init_pose"][:, 3:]))
    verts = ((out["vertices"] + torch.tensor([[0.01, -0.02, 0.015]])) * 3.3 * 0.3).detach().requires_grad_(True)
    masks = torch.from_numpy((np.array(prob["masks"]) > 128).astype(np.float32))
    contours = [torch.tensor(c[:, None, :], dtype=torch.float32) for c in extract_contours(masks.numpy())]
    w2cs = torch.inverse(torch.from_numpy(np.array(prob["c2ws"])).float())
    idx = [prob["use_frames"].index(f) for f in mask_frames]
    loss = multview_mask_loss(contours, masks, verts, np.zeros((1, 1, 3), np.int32), [w2cs[i] for i in idx],
                              [prob["Ks"][i] for i in idx], mask_frames, imsize=prob["imsize"])
    loss.backward()
    np.savez_compressed(os.path.join(GOLDEN, "mask_loss_f0.npz"), frame=0, n_views=8, mask_frames=np.array(mask_frames),
                        transl=np.array([0.01, -0.02, 0.015], np.float32), scale=np.float32(3.3), loss=float(loss),
                        grad_sampled=verts.grad.numpy()[0, ::4], contour_counts=np.array([len(c) for c in contours]),
                        model_digest=S.model_digest(model))
    print("mask loss", float(loss))
    # (b) the loop
    snaps = {}
    orig_step = torch.optim.Adam.step
    counter = {"n": 0}

    def step(self, *a, **k):
        r = orig_step(self, *a, **k)
        counter["n"] += 1
        if counter["n"] in (1, 11, 12, 20, 30):
            g = self.param_groups
            snaps[counter["n"]] = {n: g[i]["params"][0].detach().numpy()[0].copy() for i, n in
                                   enumerate(("global_transl", "scale", "pose", "betas", "global_orient"))}
        return r

    torch.optim.Adam.step = step
    try:
        from smplify.smplify import SMPLify
        fitter = SMPLify(smpl_type="smpl", num_iters=30, gender="neutral", device=torch.device("cpu"), debug=False)
        res = fitter((torch.from_numpy(prob["init_betas"].copy()), torch.from_numpy(prob["init_pose"].copy())), prob["c2ws"],
                     prob["Ks"], prob["keypoints"], None, use_mask=True, masks=prob["masks"], use_frames=prob["use_frames"],
                     mask_frames=mask_frames, imsize=prob["imsize"])
    finally:
        torch.optim.Adam.step = orig_step
    np.savez_compressed(os.path.join(GOLDEN, "mask_fit_8view_30it.npz"), frame=0, n_views=8, num_iters=30,
                        mask_frames=np.array(mask_frames), joints=res["joints"], vertices_sample=res["vertices"][::53],
                        model_digest=S.model_digest(model), **flat_snaps(snaps))
    print("mask fit done")


def smplx_goldens():
    """BASELINE config 3 in miniature: the reference loop with smpl_type='smplx' (hands + face keypoints,
    smplify.py:57-80,103-128; loss.py:166-181,199-207) on the stand-in smplx.create."""
    import torch
    import smplx
    from bodyfitting_amd import synthetic as S

    model = S.make_model("smplx", seed=0)
    gmm = S.make_gmm(seed=0)
    smplx.MODEL_REGISTRY["smplx"] = model
    tmp = tempfile.mkdtemp(prefix="bf_golden_smplx_")
    os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
    with open(os.path.join(tmp, "data", "gmm_08.pkl"), "wb") as f:
        pickle.dump(gmm, f)
    os.chdir(tmp)
    names = ("global_transl", "scale", "pose", "betas", "global_orient", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")
    for tag, n_views, iters, mask_frames in (("smplx_8view_40it", 8, 40, None), ("smplx_mask_8view_15it", 8, 15, [1, 3, 5, 7])):
        if mask_frames is not None:
            install_cv2_contour_stub()
        prob = S.make_problem_smplx(model, frame=0, n_views=n_views, mask_frames=mask_frames)
        snaps = {}
        orig_step = torch.optim.Adam.step
        counter = {"n": 0}
        want = (1, 2, 6, 10, 20, iters)

        def step(self, *a, **k):
            r = orig_step(self, *a, **k)
            counter["n"] += 1
            if counter["n"] in want:
                g = self.param_groups
                snaps[counter["n"]] = {n: g[i]["params"][0].detach().numpy().reshape(-1).copy() for i, n in enumerate(names)}
            return r

        torch.optim.Adam.step = step
        try:
            from smplify.smplify import SMPLify
            fitter = SMPLify(smpl_type="smplx", num_iters=iters, gender="neutral", device=torch.device("cpu"), debug=False)
            res = fitter((torch.from_numpy(prob["init_betas"].copy()), torch.from_numpy(prob["init_pose"].copy())), prob["c2ws"],
                         prob["Ks"], prob["keypoints"], None, use_frames=prob["use_frames"], imsize=prob["imsize"],
                         use_mask=mask_frames is not None, masks=prob.get("masks"), mask_frames=mask_frames or [0])
        finally:
            torch.optim.Adam.step = orig_step
        np.savez_compressed(os.path.join(GOLDEN, tag + ".npz"), frame=0, n_views=n_views, num_iters=iters,
                            joints=res["joints"], full_pose=res["full_pose"], vertices_sample=res["vertices"][::53],
                            pose=res["pose"], model_digest=S.model_digest(model), **flat_snaps(snaps))
        print(tag, "done")


# Perturbations that change no mathematics: (tag, torch intra-op threads, what is moved by one float32 ulp).  tests/ref_drift.py builds
# the parity bands of the round-off-amplifying loops from how far the IMPORTED reference moves from itself under them.  The first two
# are the ones of rounds 3-5 (their keys in the committed files are unchanged); round 6 added the other eight - a band built on two
# draws of a chaotic system was weak in both directions.
PERTURBATIONS = (("threads8", 8, None), ("ulp", 1, "pose+"), ("threads2", 2, None), ("threads4", 4, None), ("ulp_down", 1, "pose-"),
                 ("ulp_kp", 1, "kp+"), ("ulp_cam", 1, "cam+"), ("ulp_ext", 1, "ext+"), ("t4_ulp_down", 4, "pose-"), ("t2_ulp_kp", 2, "kp+"))
NEW_PERTURBATIONS = PERTURBATIONS[2:]


def _ulp(a, up=True):
    a = np.asarray(a)
    return np.nextafter(a.astype(np.float32), np.float32(np.inf if up else -np.inf)).astype(a.dtype)


def _nudge(problem, what="pose+"):
    """the same problem with every entry of one input moved to the next float32 (1 ulp): the initial pose (up / down), the keypoints'
    pixel coordinates, the intrinsics, or the camera-to-world matrices; None: the problem itself"""
    if what is None:
        return problem
    q = dict(problem)
    if what in ("pose+", "pose-"):
        q["init_pose"] = _ulp(problem["init_pose"], what == "pose+")
    elif what == "ext+":
        q["c2ws"] = [_ulp(c) for c in problem["c2ws"]]
    elif what == "kp+":
        q["keypoints"] = [{k: np.concatenate([_ulp(v[:, :2]), v[:, 2:]], axis=1) for k, v in d.items()} if isinstance(d, dict) else d
                          for d in problem["keypoints"]]
    elif what == "cam+":
        q["Ks"] = [_ulp(K) for K in problem["Ks"]]
    else:
        raise ValueError(what)
    return q


def sensitivity_cfg2_goldens():
    """config 2 under sensitivity_goldens' two perturbations (8 intra-op threads; the initial pose one float32 ulp up): the final
    parameters and rtn_dict's global_transl (= t * s, smplify.py:223) of all four frames.  Frame 3's translation is ill-conditioned -
    this is how far the REFERENCE moves from itself there, the yardstick of tests/test_gpu_parity.py for that field."""
    import torch
    import smplx
    from bodyfitting_amd import synthetic as S
    model = S.make_model("smpl", seed=0)
    gmm = S.make_gmm(seed=0)
    smplx.MODEL_REGISTRY["smpl"] = model
    tmp = tempfile.mkdtemp(prefix="bf_sens_cfg2_")
    write_data_dir(tmp, model, gmm)
    os.chdir(tmp)
    out = {}
    for frame in (0, 1, 2, 3):
        base = S.make_problem(model, frame=frame, n_views=48)
        for tag, threads, what in PERTURBATIONS:
            torch.set_num_threads(threads)
            res, snaps, _ = run_reference_fit(_nudge(base, what), 100, snapshots=(100,))
            out[f"{tag}_f{frame}_final_global_transl"] = res["global_transl"]
            out[f"{tag}_f{frame}_joints"] = res["joints"]
            out.update({f"{tag}_f{frame}_{k}": v for k, v in flat_snaps(snaps).items()})
        torch.set_num_threads(1)
        print("sensitivity: cfg2 frame", frame, "done")
    np.savez_compressed(os.path.join(GOLDEN, "sens_cfg2_48view_100it.npz"), model_digest=S.model_digest(model), **out)


def sensitivity_scan_goldens():
    """The scan loops' share of sensitivity_goldens: 8 threads, one ulp, and - for the closest-point search, whose answers are face
    ids decided by last bits - the restated search built with fused multiply-adds ('fused': what nvcc is free to do to the
    reference's own build)."""
    import torch
    import smplx
    from bodyfitting_amd import synthetic as S
    from bodyfitting_amd.io import save_obj_mesh

    variants = tuple((tag, threads, what, False) for tag, threads, what in PERTURBATIONS) + (("fused", 1, None, True),)
    gmm = S.make_gmm(seed=0)
    # ---- scan loop 300 + SMPL+D 300 on the reduced model (scan_goldens_long) -----------------------------------------
    model = S.make_model("smpl", seed=0, nv=690)
    smplx.MODEL_REGISTRY["smpl"] = model
    tmp = tempfile.mkdtemp(prefix="bf_sens_scan_")
    write_data_dir(tmp, model, gmm)
    os.chdir(tmp)
    base, sv, sf = S.make_scan_problem(model, frame=0, n_views=8)
    meshfile = os.path.join(tmp, "scan.obj")
    save_obj_mesh(meshfile, sv, sf)
    out = {}
    for tag, threads, what, fused in variants:
        torch.set_num_threads(threads)
        StandInMeshGridSearcher.FUSED = fused
        prob = _nudge(base, what)
        res, snaps, dsn = run_reference_scan_fit(prob, meshfile, 300, snapshots=(100, 101, 300), displacement=True)
        out.update({f"{tag}_{k}": v for k, v in flat_snaps(snaps).items()})
        out[f"{tag}_vertices"], out[f"{tag}_joints"], out[f"{tag}_displacement"] = res["vertices"], res["joints"], res["displacement"]
    torch.set_num_threads(1)
    np.savez_compressed(os.path.join(GOLDEN, "sens_scan_nv690_300it.npz"), model_digest=S.model_digest(model), **out)
    print("sensitivity: scan loop done")
    out = {}
    for tag, threads, what, fused in variants:       # the 30 + 30-iteration run of scan_goldens, with its displacement snapshots
        torch.set_num_threads(threads)
        StandInMeshGridSearcher.FUSED = fused
        prob = _nudge(base, what)
        res, snaps, dsn = run_reference_scan_fit(prob, meshfile, 30, snapshots=(1, 11, 12, 20, 30), displacement=True)
        out.update({f"{tag}_{k}": v for k, v in flat_snaps(snaps).items()})
        out.update({f"{tag}_disp{k}": v for k, v in dsn.items()})
        out[f"{tag}_vertices"], out[f"{tag}_displacement"] = res["vertices"], res["displacement"]
    torch.set_num_threads(1)
    StandInMeshGridSearcher.FUSED = False
    np.savez_compressed(os.path.join(GOLDEN, "sens_scan_nv690_30it.npz"), model_digest=S.model_digest(model), **out)
    print("sensitivity: short scan loop done")


def sensitivity_goldens():
    """How far does the REFERENCE move from itself?  The dense-loss loops (silhouette: nearest-vertex choices and 1 <-> 10 weights;
    scan: closest faces; SMPL+D: Adam's normalised steps) amplify round-off, so the parity bands of those loops in tests/ are
    set from what the imported reference does under two perturbations that change no mathematics: torch running 8 intra-op
    threads instead of 1 (other summation orders), and the initial pose moved by one float32 ulp.  Same problems, same
    snapshots as mask_goldens / smplx_goldens / scan_goldens_long; the committed files hold the perturbed runs' outputs."""
    import torch
    import smplx
    from bodyfitting_amd import synthetic as S
    from bodyfitting_amd.io import save_obj_mesh

    variants = PERTURBATIONS
    gmm = S.make_gmm(seed=0)

    def capture(names, want, run):
        snaps, counter = {}, {"n": 0, "d": 0}
        orig_step = torch.optim.Adam.step

        def step(self, *a, **k):
            r = orig_step(self, *a, **k)
            g = self.param_groups
            if len(g) > 1:
                counter["n"] += 1
                if counter["n"] in want:
                    snaps[counter["n"]] = {n: g[i]["params"][0].detach().numpy().reshape(-1).copy() for i, n in enumerate(names)}
            return r
        torch.optim.Adam.step = step
        try:
            res = run()
        finally:
            torch.optim.Adam.step = orig_step
        return res, snaps

    smpl_names = ("global_transl", "scale", "pose", "betas", "global_orient")
    smplx_names = smpl_names + ("leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")

    # ---- silhouette loop, SMPL, 30 iterations (mask_goldens b) ------------------------------------------------------
    model = S.make_model("smpl", seed=0)
    smplx.MODEL_REGISTRY["smpl"] = model
    tmp = tempfile.mkdtemp(prefix="bf_sens_mask_")
    write_data_dir(tmp, model, gmm)
    os.chdir(tmp)
    install_cv2_contour_stub()
    from smplify.smplify import SMPLify
    mask_frames = [1, 3, 5, 7]
    base = S.make_problem(model, frame=0, n_views=8, mask_frames=mask_frames)
    out = {}
    for tag, threads, what in variants:
        torch.set_num_threads(threads)
        prob = _nudge(base, what)
        fitter = SMPLify(smpl_type="smpl", num_iters=30, gender="neutral", device=torch.device("cpu"), debug=False)
        res, snaps = capture(smpl_names, (1, 11, 12, 20, 30), lambda: fitter(
            (torch.from_numpy(prob["init_betas"].copy()), torch.from_numpy(prob["init_pose"].copy())), prob["c2ws"], prob["Ks"],
            prob["keypoints"], None, use_mask=True, masks=prob["masks"], use_frames=prob["use_frames"], mask_frames=mask_frames,
            imsize=prob["imsize"]))
        out.update({f"{tag}_{k}": v for k, v in flat_snaps(snaps).items()})
        out[f"{tag}_joints"] = res["joints"]
    torch.set_num_threads(1)
    np.savez_compressed(os.path.join(GOLDEN, "sens_mask_fit_8view_30it.npz"), model_digest=S.model_digest(model), **out)
    print("sensitivity: mask loop done")

    sensitivity_scan_goldens()

    # ---- silhouette loop, SMPL-X, 15 iterations (smplx_goldens) ------------------------------------------------------
    model = S.make_model("smplx", seed=0)
    smplx.MODEL_REGISTRY["smplx"] = model
    tmp = tempfile.mkdtemp(prefix="bf_sens_smplx_")
    os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
    with open(os.path.join(tmp, "data", "gmm_08.pkl"), "wb") as f:
        pickle.dump(gmm, f)
    os.chdir(tmp)
    base = S.make_problem_smplx(model, frame=0, n_views=8, mask_frames=mask_frames)
    out = {}
    for tag, threads, what in variants:
        torch.set_num_threads(threads)
        prob = _nudge(base, what)
        fitter = SMPLify(smpl_type="smplx", num_iters=15, gender="neutral", device=torch.device("cpu"), debug=False)
        res, snaps = capture(smplx_names, (1, 2, 6, 10, 15), lambda: fitter(
            (torch.from_numpy(prob["init_betas"].copy()), torch.from_numpy(prob["init_pose"].copy())), prob["c2ws"], prob["Ks"],
            prob["keypoints"], None, use_frames=prob["use_frames"], imsize=prob["imsize"], use_mask=True, masks=prob["masks"],
            mask_frames=mask_frames))
        out.update({f"{tag}_{k}": v for k, v in flat_snaps(snaps).items()})
        out[f"{tag}_joints"] = res["joints"]
    torch.set_num_threads(1)
    np.savez_compressed(os.path.join(GOLDEN, "sens_smplx_mask_8view_15it.npz"), model_digest=S.model_digest(model), **out)
    print("sensitivity: smplx mask loop done")


def cfg3_goldens(variants=(("base", 1, None),) + PERTURBATIONS):
    """BASELINE config 3 AS STATED, run by the imported reference: SMPL-X (10,475 vertices, 55 joints, 135 output joints),
    48 views with body + hands + face keypoints, 8 silhouette views at 512 x 512, 200 iterations (the silhouette loss is active
    for i > 66, smplify.py:197).  Snapshots of the optimised parameters after iterations 1 / 66 / 67 / 68 / 200 and the loss values
    the loop computed in those iterations (multiview_keypoint_loss's total + dict, loss.py:219-228; multview_mask_loss's value),
    recorded by wrapping the functions the loop calls - the reference code itself is untouched.  The `threads8` and `ulp`
    variants are the reference's own sensitivity at this size (see sensitivity_goldens)."""
    import torch
    import smplx
    from bodyfitting_amd import synthetic as S

    model = S.make_model("smplx", seed=0)
    gmm = S.make_gmm(seed=0)
    smplx.MODEL_REGISTRY["smplx"] = model
    tmp = tempfile.mkdtemp(prefix="bf_golden_cfg3_")
    os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
    with open(os.path.join(tmp, "data", "gmm_08.pkl"), "wb") as f:
        pickle.dump(gmm, f)
    os.chdir(tmp)
    install_cv2_contour_stub()
    import smplify.smplify as RS
    names = ("global_transl", "scale", "pose", "betas", "global_orient", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")
    mask_frames = list(range(0, 48, 6))
    base = S.make_problem_smplx(model, frame=0, n_views=48, mask_frames=mask_frames)
    want = (1, 66, 67, 68, 200)
    for tag, threads, what in variants:
        torch.set_num_threads(threads)
        prob = _nudge(base, what)
        snaps, losses, counter = {}, {}, {"n": 0}
        orig_step, orig_kp, orig_mask = torch.optim.Adam.step, RS.multiview_keypoint_loss, RS.multview_mask_loss

        def step(self, *a, **k):
            r = orig_step(self, *a, **k)
            counter["n"] += 1
            if counter["n"] in want:
                g = self.param_groups
                snaps[counter["n"]] = {n: g[i]["params"][0].detach().numpy().reshape(-1).copy() for i, n in enumerate(names)}
            return r

        def kp_loss(*a, **k):
            total, d = orig_kp(*a, **k)
            it = counter["n"] + 1                      # (the iteration whose step comes next)
            if it in want:
                losses[it] = {"total": float(total), **{kk: float(np.asarray(v).reshape(-1)[0]) for kk, v in d.items()}}
            return total, d

        def mask_loss(*a, **k):
            v = orig_mask(*a, **k)
            it = counter["n"] + 1
            if it in want:
                losses[it]["mask_loss"] = float(v)
            return v

        torch.optim.Adam.step, RS.multiview_keypoint_loss, RS.multview_mask_loss = step, kp_loss, mask_loss
        t0 = time.perf_counter()
        try:
            fitter = RS.SMPLify(smpl_type="smplx", num_iters=200, gender="neutral", device=torch.device("cpu"), debug=False)
            res = fitter((torch.from_numpy(prob["init_betas"].copy()), torch.from_numpy(prob["init_pose"].copy())), prob["c2ws"],
                         prob["Ks"], prob["keypoints"], None, use_frames=prob["use_frames"], imsize=prob["imsize"],
                         use_mask=True, masks=prob["masks"], mask_frames=mask_frames)
        finally:
            torch.optim.Adam.step, RS.multiview_keypoint_loss, RS.multview_mask_loss = orig_step, orig_kp, orig_mask
        wall = time.perf_counter() - t0
        extra = {}
        for it, d in losses.items():
            for kk, v in d.items():
                extra[f"it{it}_loss_{kk}"] = v
        np.savez_compressed(os.path.join(GOLDEN, f"cfg3_smplx_48view_8mask_200it_{tag}.npz"), frame=0, n_views=48, num_iters=200,
                            mask_frames=np.array(mask_frames), wall_s=wall, threads=threads, joints=res["joints"], full_pose=res["full_pose"],
                            vertices_sample=res["vertices"][::53], model_digest=S.model_digest(model), **flat_snaps(snaps), **extra)
        print("cfg3", tag, wall, "s", {it: d.get("mask_loss") for it, d in losses.items()})
    torch.set_num_threads(1)


def reference_timing(frames=(0, 1, 2)):
    """Wall time of the UNMODIFIED reference loop (SMPLify.__call__, smplify.py:84-250: 48 views, 100 iterations, torch CPU,
    1 thread = its faster setting) on this build container, per frame, without the snapshot hook: the reference-side CPU figure
    bench.py quotes beside its own port's (the reference cannot travel to the GPU box)."""
    import json
    import platform
    import torch
    import smplx
    from bodyfitting_amd import synthetic as S

    model, gmm = S.make_model("smpl", seed=0), S.make_gmm(seed=0)
    smplx.MODEL_REGISTRY["smpl"] = model
    tmp = tempfile.mkdtemp(prefix="bf_golden_time_")
    write_data_dir(tmp, model, gmm)
    os.chdir(tmp)
    run_reference_fit(S.make_problem(model, frame=0, n_views=48), 5)           # warm imports / allocator
    walls = []
    for f in frames:
        _, _, wall = run_reference_fit(S.make_problem(model, frame=f, n_views=48), 100)
        walls.append(wall)
    cpu = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            cpu = next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    out = {"what": "reference SMPLify.__call__ imported from /root/reference, torch CPU, config 2 (1 frame x 48 views x 100 iterations)",
           "wall_s_per_frame": walls, "frames_per_s": len(walls) / sum(walls), "threads": torch.get_num_threads(),
           "host": {"cpu": cpu, "logical_cpus": os.cpu_count(), "machine": platform.machine()}, "torch": torch.__version__,
           "where": "build container (the GPU box has no /root/reference)"}
    with open(os.path.join(GOLDEN, "reference_timing.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("reference timing", out["frames_per_s"], "frames/s", walls)


def openpose_cases():
    """JSON documents for the OpenPose reader: the reference's own sample (openpose/test.json) and variants that reach every
    branch of utils/io_utils.py:138-183 (integer-valued arrays, person ids, several people, empty / zero-confidence parts)."""
    import json
    with open(os.path.join(REFERENCE, "openpose", "test.json")) as f:
        sample = json.load(f)
    rng = np.random.default_rng(42)

    def person(scale=1.0, pid=None, hands=True, face=True):
        def part(n):
            a = np.concatenate([rng.uniform(10, 500, (n, 2)), rng.uniform(0.1, 1.0, (n, 1)) * scale], 1)
            return [round(float(x), 3) for x in a.reshape(-1)]
        out = {"person_id": [-1 if pid is None else pid], "pose_keypoints_2d": part(25)}
        if hands:
            out["hand_left_keypoints_2d"], out["hand_right_keypoints_2d"] = part(21), part(21)
        if face:
            out["face_keypoints_2d"] = part(70)
        out["pose_keypoints_3d"], out["face_keypoints_3d"] = [], []
        return out
    cases = {
        "reference_sample": sample,
        "no_people": {"version": 1.3, "people": []},
        "two_people_second_scores_higher": {"version": 1.3, "people": [person(0.3), person(1.0)]},
        "all_integer_coordinates": {"version": 1.3, "people": [{"person_id": [-1], "pose_keypoints_2d": [float(v) for v in rng.integers(1, 500, 50)]}]},
        "all_zero_part_is_dropped": {"version": 1.3, "people": [dict(person(), hand_left_keypoints_2d=[0.0] * 63)]},
        "person_ids": {"version": 1.3, "people": [person(0.5, pid=3), person(1.0, pid=7)]},
        "ragged_length_is_truncated": {"version": 1.3, "people": [{"person_id": [-1], "pose_keypoints_2d": [float(v) for v in rng.integers(1, 500, 51)] + [7.0, 9.0]}]},
        "only_zero_confidences": {"version": 1.3, "people": [{"person_id": [-1], "pose_keypoints_2d": [1.5, 2.5, 0.0] * 25}]},
    }
    return cases


def openpose_goldens():
    """expected output of the reference's load_openpose (utils/io_utils.py:138-183, imported) on openpose_cases()"""
    import json
    from utils.io_utils import load_openpose

    def enc(v):
        if v is None:
            return None
        if isinstance(v, dict):
            return {"__dict__": [[int(k) if not isinstance(k, str) else k, enc(x)] for k, x in v.items()]}
        if isinstance(v, list):
            return {"__list__": [enc(x) for x in v]}
        a = np.asarray(v)
        return {"dtype": str(a.dtype), "shape": list(a.shape), "data": a.reshape(-1).tolist()}
    tmp = tempfile.mkdtemp(prefix="bf_golden_openpose_")
    out = {}
    for name, doc in openpose_cases().items():
        path = os.path.join(tmp, name + ".json")
        with open(path, "w") as f:
            json.dump(doc, f)
        entry = {"input": doc}
        for only_one in (True, False):
            try:
                entry["only_one" if only_one else "all"] = {"result": enc(load_openpose(path, only_one=only_one))}
            except Exception as exc:                   # (the reference raises on some inputs: that is its behaviour too)
                entry["only_one" if only_one else "all"] = {"raises": type(exc).__name__}
        out[name] = entry
    with open(os.path.join(GOLDEN, "openpose_reader.json"), "w") as f:
        json.dump(out, f)
    print("openpose goldens:", {k: list(v["only_one"]) for k, v in out.items()})


def flat_snaps(snaps):
    out = {}
    for k, d in snaps.items():
        for name, v in d.items():
            out[f"it{k}_{name}"] = v
    return out


def main():
    install_reference_imports()
    import smplx
    from bodyfitting_amd import synthetic as S

    os.makedirs(GOLDEN, exist_ok=True)
    model = S.make_model("smpl", seed=0)
    gmm = S.make_gmm(seed=0)
    smplx.MODEL_REGISTRY["smpl"] = model
    digest = S.model_digest(model)
    tmp = tempfile.mkdtemp(prefix="bf_golden_")
    write_data_dir(tmp, model, gmm)
    os.chdir(tmp)

    meta = dict(model_seed=0, model_digest=digest)

    # ---- config 1: 1 frame, 1 view, 50 iters (plumbing) ------------------------------------
    prob = S.make_problem(model, frame=0, n_views=1)
    res, snaps, wall = run_reference_fit(prob, 50, snapshots=(1, 10, 50))
    np.savez_compressed(os.path.join(GOLDEN, "cfg1_1view_50it.npz"), frame=0, n_views=1, num_iters=50,
                        wall_s=wall, joints=res["joints"], full_pose=res["full_pose"],
                        vertices_sample=res["vertices"][::53], final_global_transl=res["global_transl"],
                        **flat_snaps(snaps), **meta)
    print("cfg1", wall, "s")

    # ---- config 2: 1 frame, 48 views, 100 iters ---------------------------------------------
    for frame in (0, 1, 2, 3):
        prob = S.make_problem(model, frame=frame, n_views=48)
        res, snaps, wall = run_reference_fit(prob, 100, snapshots=(1, 2, 10, 50, 100))
        np.savez_compressed(os.path.join(GOLDEN, f"cfg2_48view_100it_f{frame}.npz"), frame=frame, n_views=48,
                            num_iters=100, wall_s=wall, joints=res["joints"], full_pose=res["full_pose"],
                            vertices_sample=res["vertices"][::53], final_global_transl=res["global_transl"],
                            **flat_snaps(snaps), **meta)
        print("cfg2 frame", frame, wall, "s")

    # ---- ragged input: views without a detection (loss.py:157 skip, :197 divisor) ------------
    prob = S.make_problem(model, frame=5, n_views=8, missing_views=(2, 5))
    res, snaps, wall = run_reference_fit(prob, 20, snapshots=(1, 20))
    np.savez_compressed(os.path.join(GOLDEN, "ragged_8view_20it.npz"), frame=5, n_views=8, num_iters=20,
                        missing_views=np.array([2, 5]), joints=res["joints"],
                        vertices_sample=res["vertices"][::53], **flat_snaps(snaps), **meta)

    # ---- one loss / gradient evaluation (loss.py:219-224 dict + autograd grads) ---------------
    prob = S.make_problem(model, frame=0, n_views=48)
    params = {"global_transl": np.array([0.02, -0.01, 0.03]), "scale": np.array([1.1]),
              "pose": prob["init_pose"][0, 3:], "betas": np.linspace(-0.5, 0.5, 10),
              "global_orient": prob["init_pose"][0, :3]}
    loss, terms, grads, mj, bv = reference_loss_terms(model, gmm, prob, params)
    np.savez_compressed(os.path.join(GOLDEN, "loss_terms_f0.npz"), frame=0, n_views=48, loss=loss,
                        joints=mj, vertices_sample=bv[::53],
                        **{f"term_{k}": v for k, v in terms.items()},
                        **{f"param_{k}": np.asarray(v, np.float32) for k, v in params.items()},
                        **{f"grad_{k}": v for k, v in grads.items()}, **meta)
    print("loss terms", loss, terms)


if __name__ == "__main__":
    if "--main-only" in sys.argv:
        main()
    elif "--scan-only" in sys.argv:
        install_reference_imports()
        scan_goldens()
    elif "--smplx-only" in sys.argv:
        install_reference_imports()
        smplx_goldens()
    elif "--scan-long-only" in sys.argv:
        install_reference_imports()
        scan_goldens_long()
    elif "--openpose-only" in sys.argv:
        install_reference_imports()
        openpose_goldens()
    elif "--sens-cfg2-only" in sys.argv:
        install_reference_imports()
        sensitivity_cfg2_goldens()
    elif "--sens-scan-only" in sys.argv:
        install_reference_imports()
        sensitivity_scan_goldens()
    elif "--sensitivity-only" in sys.argv:
        install_reference_imports()
        sensitivity_goldens()
    elif "--cfg3-only" in sys.argv:
        install_reference_imports()
        cfg3_goldens()
    elif "--cfg3-new-only" in sys.argv:                      # only the perturbations round 6 added (the other files stay as committed)
        install_reference_imports()
        cfg3_goldens(variants=NEW_PERTURBATIONS)
    elif "--timing-only" in sys.argv:
        install_reference_imports()
        reference_timing()
    elif "--mask-only" in sys.argv:
        install_reference_imports()
        mask_goldens()
    else:
        main()
        scan_goldens()
        mask_goldens()
        smplx_goldens()
        scan_goldens_long()
        openpose_goldens()
        sensitivity_goldens()
        sensitivity_cfg2_goldens()
        cfg3_goldens()
        reference_timing()
