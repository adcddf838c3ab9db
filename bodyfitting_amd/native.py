"""Thin numpy-level wrappers over the C ABI: `DeviceModel` (bf_model) and `FrameBatch` (bf_batch).

Host code is plain Python + numpy + ctypes; PyTorch is not involved on this path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .assets import gmm_buffers
from .keypoints import pack_keypoints_smplx

N_LOSS_JOINTS = 25   # SKELETON_LENGTH, reference smplify/loss.py:17


def _f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def model_desc(model, gmm):
    """-> (ModelDesc for bf_model_create / bf_group_create, dict of the model's sizes, the arrays the descriptor points into -
    keep them alive until the call returned)."""
    model_type = model.get("model_type", "smpl")
    means, prec, nllw = gmm_buffers(gmm) if isinstance(gmm, dict) else gmm
    smplx = model_type == "smplx"
    n_betas = 10 if smplx else np.asarray(model["shapedirs"]).shape[2]     # expression dirs stay unused (never optimised)
    if "J_regressor_extra" not in model:
        model = dict(model, J_regressor_extra=np.zeros((0, np.asarray(model["v_template"]).shape[0]), np.float32))
    keep = {
        "v_template": _f32(model["v_template"]), "shapedirs": _f32(np.asarray(model["shapedirs"])[:, :, :n_betas]),
        "posedirs": _f32(model["posedirs"]), "j_regressor": _f32(model["J_regressor"]),
        "lbs_weights": _f32(model["lbs_weights"]), "parents": _i32(model["parents"]),
        "selector_ids": _i32(model["selector_ids"]),
        "j_regressor_extra": _f32(model["J_regressor_extra"]), "joint_map": _i32(model["joint_map"]),
        "gmm_means": _f32(means), "gmm_precisions": _f32(prec), "gmm_nll_weights": _f32(nllw),
    }
    info = {"model_type": model_type}
    info["n_verts"], info["n_joints"] = keep["lbs_weights"].shape
    info["n_betas"] = keep["shapedirs"].shape[2]
    info["n_selector"] = len(keep["selector_ids"])
    info["n_joint_map"] = len(keep["joint_map"])
    info["faces"] = _i32(model["faces"]) if "faces" in model else None
    if keep["posedirs"].shape != (9 * (info["n_joints"] - 1), 3 * info["n_verts"]):
        raise ValueError("posedirs must be [9(NJ-1), 3NV] as smplx stores it")
    d = _lib.ModelDesc()
    d.n_verts, d.n_joints, d.n_betas = info["n_verts"], info["n_joints"], info["n_betas"]
    for name in ("v_template", "shapedirs", "posedirs", "j_regressor", "lbs_weights", "j_regressor_extra",
                 "gmm_means", "gmm_precisions", "gmm_nll_weights"):
        setattr(d, name, _lib.fptr(keep[name]))
    for name in ("parents", "selector_ids", "joint_map"):
        setattr(d, name, _lib.iptr(keep[name]))
    d.n_selector, d.n_extra = info["n_selector"], keep["j_regressor_extra"].shape[0]
    info["n_loss_joints"] = 135 if smplx else N_LOSS_JOINTS               # loss.py:17-19: 25 (+ 42 hands + 68 face)
    d.n_joint_map, d.n_loss_joints = info["n_joint_map"], info["n_loss_joints"]
    if smplx:
        keep.update(pose_mean=_f32(model["pose_mean"]), lhc=_f32(model["left_hand_components"]),
                    rhc=_f32(model["right_hand_components"]), lmk_f=_i32(model["lmk_faces_idx"]),
                    lmk_b=_f32(model["lmk_bary_coords"]), dyn_f=_i32(model["dynamic_lmk_faces_idx"]),
                    dyn_b=_f32(model["dynamic_lmk_bary_coords"]))
        d.model_kind, d.pose_mean, d.n_hand_pca = 1, _lib.fptr(keep["pose_mean"]), keep["lhc"].shape[0]
        d.left_hand_components, d.right_hand_components = _lib.fptr(keep["lhc"]), _lib.fptr(keep["rhc"])
        d.n_lmk_static, d.lmk_faces_idx, d.lmk_bary_coords = len(keep["lmk_f"]), _lib.iptr(keep["lmk_f"]), _lib.fptr(keep["lmk_b"])
        d.n_dyn_rows, d.n_lmk_dynamic = keep["dyn_f"].shape
        d.dynamic_lmk_faces_idx, d.dynamic_lmk_bary_coords = _lib.iptr(keep["dyn_f"]), _lib.fptr(keep["dyn_b"])
        d.neck_joint = int(np.asarray(model["neck_kin_chain"])[0])
    d.gmm_components, d.gmm_dim = keep["gmm_means"].shape
    if info["faces"] is not None:
        keep["faces"] = _i32(info["faces"].reshape(-1, 3))
        d.n_faces, d.faces = len(keep["faces"]), _lib.iptr(keep["faces"])
    return d, info, keep


class DeviceModel:
    """Body model + GMM prior uploaded once to one GPU (replaces the per-frame construction at
    reference smplify/body_fitting.py:82 -> smplify/smplify.py:46-56)."""

    def __init__(self, model, gmm, device=0):
        lib = _lib.load()
        self._lib = lib
        d, info, keep = model_desc(model, gmm)
        self.__dict__.update(info)
        self._h = C.c_void_p()
        _lib.check(lib.bf_model_create(C.byref(d), int(device), C.byref(self._h)), "bf_model_create")
        del keep
        self.device = int(device)
        self.n_params = lib.bf_model_n_params(self._h)
        self.fit_instance = "sized" if lib.bf_model_fit_instance(self._h) else "table-driven"

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bf_model_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def forward(self, betas, global_orient, body_pose):
        """models.smpl.SMPL.forward (reference models/smpl.py:69-83): -> vertices, joints49, joints45."""
        betas = _f32(betas, (-1, self.n_betas))
        n = betas.shape[0]
        orient = _f32(global_orient, (n, 3))
        pose = _f32(body_pose, (n, 3 * (self.n_joints - 1)))
        verts = np.empty((n, self.n_verts, 3), np.float32)
        joints = np.empty((n, self.n_joint_map, 3), np.float32)
        jori = np.empty((n, self.n_joints + self.n_selector, 3), np.float32)
        _lib.check(self._lib.bf_smpl_forward(self._h, n, _lib.fptr(betas), _lib.fptr(orient), _lib.fptr(pose),
                                             _lib.fptr(verts), _lib.fptr(joints), _lib.fptr(jori)), "bf_smpl_forward")
        return verts, joints, jori

    def forward_packed(self, params):
        """vertices / joints (model space) of packed parameter vectors [n, n_params] - any model kind"""
        p = _f32(params, (-1, self.n_params))
        verts = np.empty((len(p), self.n_verts, 3), np.float32)
        joints = np.empty((len(p), self.n_joint_map, 3), np.float32)
        _lib.check(self._lib.bf_model_forward(self._h, len(p), _lib.fptr(p), _lib.fptr(verts), _lib.fptr(joints)), "bf_model_forward")
        return verts, joints


class Scan:
    """A scan mesh with its closest-point grid on the GPU (reference utils/mesh_grid_searcher.py:51-84)."""

    def __init__(self, verts, faces, device=0):
        self._lib = _lib.load()
        v = _f32(verts, (-1, 3))
        f = _i32(np.asarray(faces).reshape(-1, 3))
        self.n_verts, self.n_faces = len(v), len(f)
        self._h = C.c_void_p()
        _lib.check(self._lib.bf_scan_create(int(device), len(v), _lib.fptr(v), len(f), _lib.iptr(f), C.byref(self._h)),
                   "bf_scan_create")
        self.height = float(self._lib.bf_scan_height(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bf_scan_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def grid_info(self):
        dims = np.zeros(3, np.int32)
        os_ = np.zeros(4, np.float32)
        _lib.check(self._lib.bf_scan_grid_info(self._h, _lib.iptr(dims), _lib.fptr(os_)), "bf_scan_grid_info")
        return dims, os_[:3], float(os_[3])

    def grid_lists(self):
        """-> (tri_num int32[cells] inclusive cumsum, tri_idx int32[entries] face id + 1): insert_grid_surface's outputs"""
        dims, _, _ = self.grid_info()
        n = np.zeros(1, np.int32)
        _lib.check(self._lib.bf_scan_grid_lists(self._h, None, None, _lib.iptr(n)), "bf_scan_grid_lists")
        tri_num = np.empty(int(np.prod(dims)), np.int32)
        tri_idx = np.empty(max(int(n[0]), 1), np.int32)
        _lib.check(self._lib.bf_scan_grid_lists(self._h, _lib.iptr(tri_num), _lib.iptr(tri_idx), None), "bf_scan_grid_lists")
        return tri_num, tri_idx[:int(n[0])]

    def inside_mesh(self, points):
        """-> float32[n]: +1 inside the scan surface, -1 outside (MeshGridSearcher.inside_mesh)"""
        p = _f32(points, (-1, 3))
        sign = np.empty(len(p), np.float32)
        _lib.check(self._lib.bf_scan_inside(self._h, len(p), _lib.fptr(p), _lib.fptr(sign)), "bf_scan_inside")
        return sign

    def intersects_any(self, origins, directions):
        """-> bool[n]: does the ray origin + t direction (t >= 0) hit the scan surface (MeshGridSearcher.intersects_any)"""
        o = _f32(origins, (-1, 3))
        d = _f32(directions, (len(o), 3))
        hit = np.empty(len(o), np.uint8)
        _lib.check(self._lib.bf_scan_intersects(self._h, len(o), _lib.fptr(o), _lib.fptr(d), hit.ctypes.data_as(C.POINTER(C.c_uint8))),
                   "bf_scan_intersects")
        return hit.astype(bool)

    def nearest_points(self, points):
        """-> (nearest points [n,3], face ids [n], barycentrics [n,3]) like MeshGridSearcher.nearest_points"""
        p = _f32(points, (-1, 3))
        ids = np.empty(len(p), np.int32)
        pts = np.empty((len(p), 3), np.float32)
        bary = np.empty((len(p), 3), np.float32)
        _lib.check(self._lib.bf_scan_nearest(self._h, len(p), _lib.fptr(p), _lib.iptr(ids), _lib.fptr(pts), _lib.fptr(bary)),
                   "bf_scan_nearest")
        return pts, ids, bary

    def nearest_points_hinted(self, points, hint=None, reps=0):
        """nearest_points with a guess per query (hint[n,3]: e.g. the previous iteration's nearest points; any values are safe - the
        kernel checks the guess).  -> (points, ids, barycentrics[, mean kernel microseconds when reps > 0])"""
        p = _f32(points, (-1, 3))
        h = None if hint is None else _f32(hint, (len(p), 3))
        ids = np.empty(len(p), np.int32)
        pts = np.empty((len(p), 3), np.float32)
        bary = np.empty((len(p), 3), np.float32)
        us = C.c_float(0.0)
        _lib.check(self._lib.bf_scan_nearest_hinted(self._h, len(p), _lib.fptr(p), _lib.fptr(h), _lib.iptr(ids), _lib.fptr(pts), _lib.fptr(bary),
                                                    int(reps), C.byref(us)), "bf_scan_nearest_hinted")
        return (pts, ids, bary, us.value) if reps > 0 else (pts, ids, bary)


    def nearest_points_backward(self, face_ids, bary, dnearest):
        """dL/d(query points) from dL/d(nearest points): SurfaceNearest.backward w.r.t. its first argument (point-to-plane
        where the closest point lies on a face, along the edge on an edge, zero at a corner)"""
        ids = _i32(face_ids)
        b = _f32(bary, (len(ids), 3))
        g = _f32(dnearest, (len(ids), 3))
        out = np.empty((len(ids), 3), np.float32)
        _lib.check(self._lib.bf_scan_nearest_backward(self._h, len(ids), _lib.iptr(ids), _lib.fptr(b), _lib.fptr(g), _lib.fptr(out)),
                   "bf_scan_nearest_backward")
        return out


def set_nearest_rule(rule):
    """The arithmetic of every closest-point search of the process: "reference" (default: search_nearest_proj as the reference's
    source evaluates it in float32, mesh_grid_kernel.cu:12-109 + matrix.h) or "fast" (2 x 2 normal equations, v_rcp_f32).
    -> the rule that was active before."""
    names = {"reference": _lib.NEAREST_REFERENCE, "fast": _lib.NEAREST_FAST}
    lib = _lib.load()
    before = lib.bf_nearest_rule_get()
    if lib.bf_nearest_rule_set(names[rule] if isinstance(rule, str) else int(rule)) != 0:
        raise ValueError("unknown closest-point rule %r" % (rule,))
    return "fast" if before == _lib.NEAREST_FAST else "reference"


def set_mask_fold(mode):
    """How the silhouette loss's contour gradients reach dL/dvertices inside a fit: "sums" (default: fixed-point atomic sums by the
    contour scan, bodyfit.h BF_MASK_FOLD_SUMS) or "gather" (the ordered float32 walk of rounds 2-4).  -> the mode that was active."""
    names = {"sums": 0, "gather": 1}
    lib = _lib.load()
    before = lib.bf_mask_fold_get()
    if lib.bf_mask_fold_set(names[mode] if isinstance(mode, str) else int(mode)) != 0:
        raise ValueError("unknown mask fold mode %r" % (mode,))
    return "gather" if before == 1 else "sums"


def make_hyper(**kw):
    h = _lib.Hyper()
    _lib.load().bf_hyper_default(C.byref(h))
    for k, v in kw.items():
        if not hasattr(h, k):
            raise TypeError(f"unknown hyper-parameter {k!r}")
        setattr(h, k, float(v))
    return h


class FrameBatch:
    """F independent frames x V views resident on the model's GPU."""

    def __init__(self, dev_model: DeviceModel, n_frames, n_views):
        self._lib = dev_model._lib
        self.model = dev_model
        self.F, self.V = int(n_frames), int(n_views)
        self._h = C.c_void_p()
        _lib.check(self._lib.bf_batch_create(dev_model._h, self.F, self.V, C.byref(self._h)), "bf_batch_create")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bf_batch_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- inputs ------------------------------------------------------------------------------
    def set_cameras(self, c2ws, Ks):
        c2w = _f32(c2ws, (self.F, self.V, 4, 4))
        K = _f32(Ks, (self.F, self.V, 3, 3))
        _lib.check(self._lib.bf_batch_set_cameras(self._h, _lib.fptr(c2w), _lib.fptr(K)), "bf_batch_set_cameras")

    def set_keypoints(self, keypoints, n_use_frames=None):
        kp = _f32(keypoints, (self.F, self.V, self.model.n_loss_joints, 3))
        nd = None if n_use_frames is None else _i32(np.broadcast_to(np.asarray(n_use_frames), (self.F,)))
        _lib.check(self._lib.bf_batch_set_keypoints(self._h, _lib.fptr(kp), _lib.iptr(nd)), "bf_batch_set_keypoints")

    def set_init(self, init_betas, init_pose):
        b = _f32(init_betas, (self.F, self.model.n_betas))
        p = _f32(np.asarray(init_pose).reshape(self.F, -1)[:, :72], (self.F, 72))
        _lib.check(self._lib.bf_batch_set_init(self._h, _lib.fptr(b), _lib.fptr(p)), "bf_batch_set_init")

    def stage_inputs(self, keypoints, n_use_frames, init_betas, init_pose):
        """the NEXT frame's keypoints + initial estimate without waiting for the fit in flight (bf_batch_stage_inputs); the next
        fit() must carry FIT_RESET.  Arrays that already are C-contiguous float32 / int32 of the right shape are passed as they are."""
        kp = _f32(keypoints, (self.F, self.V, self.model.n_loss_joints, 3))
        nd = None if n_use_frames is None else _i32(np.broadcast_to(np.asarray(n_use_frames), (self.F,)))
        b = _f32(init_betas, (self.F, self.model.n_betas))
        p = _f32(np.asarray(init_pose).reshape(self.F, -1)[:, :72], (self.F, 72))
        _lib.check(self._lib.bf_batch_stage_inputs(self._h, _lib.fptr(kp), _lib.iptr(nd), _lib.fptr(b), _lib.fptr(p)), "bf_batch_stage_inputs")

    def set_scans(self, scans):
        """one Scan per frame (use_mesh=True); None detaches"""
        if scans is None:
            _lib.check(self._lib.bf_batch_set_scans(self._h, None), "bf_batch_set_scans")
            self._scans = None
            return
        assert len(scans) == self.F
        arr = (C.c_void_p * self.F)(*[s._h for s in scans])
        _lib.check(self._lib.bf_batch_set_scans(self._h, arr), "bf_batch_set_scans")
        self._scans = list(scans)          # keep them alive

    def set_masks(self, masks, view_index, contours=None, contour_select=_lib.CONTOUR_OPENCV_FIRST):
        """masks uint8[F,M,H,W] as loaded; view_index[M]; contours: F lists of M arrays [C,2] (x, y), or None to have them
        extracted from the masks on the device (use_mask=True, smplify.py:138-144); contour_select: which external border of a
        mask with several components is kept then (include/bodyfit.h, BF_CONTOUR_*)"""
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        F, M, H, W = masks.shape
        assert F == self.F
        vi = _i32(view_index)
        mp = masks.ctypes.data_as(C.POINTER(C.c_uint8))
        if contours is None:
            _lib.check(self._lib.bf_batch_set_masks(self._h, M, _lib.iptr(vi), H, W, mp, None, None, int(contour_select)), "bf_batch_set_masks")
            return
        counts = _i32([[len(c) for c in per_frame] for per_frame in contours]).reshape(-1)
        flat = [np.asarray(c, np.float32).reshape(-1, 2) for per_frame in contours for c in per_frame]
        xy = _f32(np.concatenate(flat, 0)) if sum(len(c) for c in flat) else np.zeros((1, 2), np.float32)
        _lib.check(self._lib.bf_batch_set_masks(self._h, M, _lib.iptr(vi), H, W, mp, _lib.iptr(counts), _lib.fptr(xy), 0), "bf_batch_set_masks")

    def stage_masks(self, masks, view_index, contour_select=_lib.CONTOUR_OPENCV_FIRST):
        """the NEXT frame's masks (same views and shape as the ones attached with set_masks(contours=None)): uploaded and
        border-followed under the fit in flight, used by the next fit()"""
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        F, M, H, W = masks.shape
        assert F == self.F
        vi = _i32(view_index)
        _lib.check(self._lib.bf_batch_stage_masks(self._h, M, _lib.iptr(vi), H, W, masks.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                  int(contour_select)), "bf_batch_stage_masks")

    def clear_masks(self):
        """detach the silhouettes (use_mask=False for the next fit)"""
        _lib.check(self._lib.bf_batch_set_masks(self._h, 0, None, 0, 0, None, None, None, 0), "bf_batch_set_masks")

    def mask_loss(self, hyper=None):
        loss = np.empty(self.F, np.float32)
        dv = np.empty((self.F, self.model.n_verts, 3), np.float32)
        hp = C.byref(hyper) if hyper is not None else None
        _lib.check(self._lib.bf_batch_mask_loss(self._h, hp, _lib.fptr(loss), _lib.fptr(dv)), "bf_batch_mask_loss")
        return loss, dv

    def fit_displacement(self, n_iters, hyper=None):
        """SMPL+D stage (smplify.py:228-247) on the vertices of the last fit"""
        hp = C.byref(hyper) if hyper is not None else None
        _lib.check(self._lib.bf_fit_displacement(self._h, int(n_iters), hp), "bf_fit_displacement")

    def get_displacement(self):
        d = np.empty((self.F, self.model.n_verts, 3), np.float32)
        _lib.check(self._lib.bf_batch_get_displacement(self._h, _lib.fptr(d)), "bf_batch_get_displacement")
        return d

    def reset(self):
        """re-arm for another fit of the same inputs (stream-ordered, no host traffic)"""
        _lib.check(self._lib.bf_batch_reset(self._h), "bf_batch_reset")

    def set_params(self, params):
        p = _f32(params, (self.F, self.model.n_params))
        _lib.check(self._lib.bf_batch_set_params(self._h, _lib.fptr(p)), "bf_batch_set_params")

    # -- compute -----------------------------------------------------------------------------
    def fit(self, n_iters, hyper=None, flags=_lib.FIT_DEFAULT):
        hp = C.byref(hyper) if hyper is not None else None
        _lib.check(self._lib.bf_fit(self._h, int(n_iters), hp, int(flags)), "bf_fit")

    def sync(self):
        _lib.check(self._lib.bf_batch_sync(self._h), "bf_batch_sync")

    def loss_grad(self, hyper=None):
        terms = np.empty((self.F, 4), np.float32)
        grads = np.empty((self.F, self.model.n_params), np.float32)
        hp = C.byref(hyper) if hyper is not None else None
        _lib.check(self._lib.bf_loss_grad(self._h, hp, _lib.fptr(terms), _lib.fptr(grads)), "bf_loss_grad")
        return terms, grads

    # -- outputs -----------------------------------------------------------------------------
    def get_params(self):
        p = np.empty((self.F, self.model.n_params), np.float32)
        _lib.check(self._lib.bf_batch_get_params(self._h, _lib.fptr(p)), "bf_batch_get_params")
        return p

    def get_result(self, vertices=True):
        m = self.model
        verts = np.empty((self.F, m.n_verts, 3), np.float32) if vertices else None
        joints = np.empty((self.F, m.n_joint_map, 3), np.float32) if vertices else None
        full_pose = np.empty((self.F, 3 * m.n_joints), np.float32)
        terms = np.empty((self.F, 4), np.float32)
        _lib.check(self._lib.bf_batch_get_result(self._h, _lib.fptr(verts), _lib.fptr(joints), _lib.fptr(full_pose),
                                                 _lib.fptr(terms)), "bf_batch_get_result")
        return verts, joints, full_pose, terms

    def get_previous(self, vertices=True):
        """(params, vertices, joints, full_pose, terms) of the fit issued BEFORE the last one, while the last one runs
        (bf_batch_get_previous): the frame loop as a two-deep pipeline"""
        m = self.model
        params = np.empty((self.F, m.n_params), np.float32)
        verts = np.empty((self.F, m.n_verts, 3), np.float32) if vertices else None
        joints = np.empty((self.F, m.n_joint_map, 3), np.float32) if vertices else None
        full_pose = np.empty((self.F, 3 * m.n_joints), np.float32)
        terms = np.empty((self.F, 4), np.float32)
        _lib.check(self._lib.bf_batch_get_previous(self._h, _lib.fptr(params), _lib.fptr(verts), _lib.fptr(joints), _lib.fptr(full_pose),
                                                   _lib.fptr(terms)), "bf_batch_get_previous")
        return params, verts, joints, full_pose, terms

    def export_params_dev(self, dev_ptr):
        _lib.check(self._lib.bf_batch_export_params_dev(self._h, C.c_void_p(int(dev_ptr))), "bf_batch_export_params_dev")

    def last_timing(self):
        ms = np.zeros(4, np.float32)
        _lib.check(self._lib.bf_batch_last_timing(self._h, _lib.fptr(ms)), "bf_batch_last_timing")
        return {"fit_ms": float(ms[0]), "mesh_ms": float(ms[1]), "tail_ms": float(ms[2]), "total_ms": float(ms[3])}

    def timing_reset(self):
        _lib.check(self._lib.bf_batch_timing_reset(self._h), "bf_batch_timing_reset")

    def timing_sum(self):
        ms = np.zeros(4, np.float32)
        n = C.c_int32(0)
        _lib.check(self._lib.bf_batch_timing_sum(self._h, _lib.fptr(ms), C.byref(n)), "bf_batch_timing_sum")
        return {"fit_ms": float(ms[0]), "mesh_ms": float(ms[1]), "tail_ms": float(ms[2]), "total_ms": float(ms[3]),
                "calls": int(n.value)}

    def mesh_span(self, reps=50):
        """the single-frame full-mesh forward's own duration, measured inside the kernel -> {"mean_us", "min_us", "max_us"}"""
        us = np.zeros(3, np.float32)
        _lib.check(self._lib.bf_batch_mesh_span(self._h, int(reps), _lib.fptr(us)), "bf_batch_mesh_span")
        return {"mean_us": float(us[0]), "min_us": float(us[1]), "max_us": float(us[2])}

    DENSE_CLASSES = ("state_and_forward_mesh", "keypoint_and_silhouette_losses", "closest_point_search", "point_cloud_loss_and_gradient",
                     "reverse_mesh", "partial_block_reduction")

    def dense_timing(self, enable=True, read=False):
        """device milliseconds of the kernel classes of the last dense iteration of the last fit (bf_batch_dense_timing)"""
        ms = np.zeros(6, np.float32) if read else None
        _lib.check(self._lib.bf_batch_dense_timing(self._h, int(bool(enable)), _lib.fptr(ms)), "bf_batch_dense_timing")
        return None if ms is None else dict(zip(self.DENSE_CLASSES, (float(x) for x in ms)))

    def dense_resident(self):
        """True / False: the last dense fit ran with the fit kernel resident / one launch per iteration; None: no dense fit yet"""
        r = self._lib.bf_batch_dense_resident(self._h)
        return None if r < 0 else bool(r)

    def debug_dump(self, n):
        out = np.zeros(n, np.float32)
        _lib.check(self._lib.bf_batch_debug_dump(self._h, _lib.fptr(out), int(n)), "bf_batch_debug_dump")
        return out


SMPLX_EXTRA = (("leye_pose", 3), ("reye_pose", 3), ("left_hand_pose", 6), ("right_hand_pose", 6))


def split_params(packed, n_joints=24, n_betas=10):
    """packed[...,86] (SMPL) or [...,98] (SMPL-X) in optimiser order (reference smplify.py:167-173) -> named blocks."""
    p = np.asarray(packed)
    smplx = p.shape[-1] == 98
    nbp = 63 if smplx else 3 * (n_joints - 1)
    o = 4 + nbp + n_betas
    out = {"global_transl": p[..., 0:3], "scale": p[..., 3:4], "pose": p[..., 4:4 + nbp],
           "betas": p[..., 4 + nbp:o], "global_orient": p[..., o:o + 3]}
    if smplx:
        o += 3
        for name, n in SMPLX_EXTRA:
            out[name] = p[..., o:o + n]
            o += n
    return out


def pack_params(d):
    names = ("global_transl", "scale", "pose", "betas", "global_orient") + tuple(n for n, _ in SMPLX_EXTRA if n in d)
    return np.concatenate([np.asarray(d[k], dtype=np.float32).reshape(-1) for k in names]).astype(np.float32)


def pack_problem(problems):
    """list of synthetic.make_problem dicts -> (c2w[F,V,4,4], K[F,V,3,3], kp[F,V,25,3], ndiv[F], betas, pose)."""
    F, V = len(problems), len(problems[0]["c2ws"])
    c2w = np.stack([np.stack(p["c2ws"]) for p in problems]).astype(np.float32)
    K = np.stack([np.stack(p["Ks"]) for p in problems]).astype(np.float32)
    smplx = any(k is not None and "face" in k for k in problems[0]["keypoints"])
    kp = np.zeros((F, V, 135 if smplx else N_LOSS_JOINTS, 3), np.float32)
    for f, p in enumerate(problems):
        for v, k in enumerate(p["keypoints"]):
            if k is not None:                      # None view: confidence 0 everywhere (loss.py:157)
                kp[f, v] = pack_keypoints_smplx(k) if smplx else k["pose"]
    ndiv = np.array([len(p["use_frames"]) for p in problems], np.int32)
    betas = np.concatenate([p["init_betas"] for p in problems]).astype(np.float32)
    pose = np.concatenate([p["init_pose"] for p in problems]).astype(np.float32)
    return c2w, K, kp, ndiv, betas, pose
