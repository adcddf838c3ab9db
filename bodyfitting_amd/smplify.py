"""`smplify.smplify.SMPLify` of the reference (smplify/smplify.py:18-254) on the HIP path.

Same constructor arguments, same `__call__` signature, same result dict (numpy arrays, batch
dimension squeezed): vertices[NV,3], joints[49,3], pose[69], betas[10], global_orient[3], faces,
global_transl[3] (= t*s, without the constant scale), scale[1], full_pose[72].

Differences, all at the edges: the model / GMM are resolved once per process through
`bodyfitting_amd.assets` instead of being re-read per frame; `net_output` may hold numpy arrays or
anything with `.detach().cpu().numpy()`; `device` is a HIP device index (or a torch.device whose
index is used).  smpl_type='smplx' adds hands + face keypoints (135 joints) and the 98-scalar parameter vector - there is no silent CPU path.
"""
from __future__ import annotations

import numpy as np

from . import assets
from .io import load_obj_mesh
from .native import FrameBatch, Scan, make_hyper, split_params
from .keypoints import pack_keypoints_smplx


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def _device_index(device):
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    idx = getattr(device, "index", None)
    return 0 if idx is None else int(idx)


class SMPLify:
    """Multi-view SMPLify.  One instance can fit any number of frames (`fit_frames`)."""

    def __init__(self, smpl_type="smpl", age="adult", step_size=1e-2, batch_size=1, num_iters=600, gender="male",
                 use_mask=False, device=0, debug=True):
        if smpl_type not in ("smpl", "smplx"):
            raise ValueError(f"unknown smpl_type {smpl_type!r}")
        if age != "adult":
            raise NotImplementedError("age='kid' is out of scope (SURVEY.md 8c)")
        self.smpl_type, self.age, self.gender = smpl_type, age, gender
        self.use_hand_face = smpl_type == "smplx"
        self.use_mask = use_mask
        self.batch_size = batch_size
        self.num_iters = num_iters          # step_size is ignored by the reference too (smplify.py:24,174)
        self.debug = debug
        self.device = _device_index(device)
        self._dev = assets.get_device_model(smpl_type, gender, self.device)
        model = assets.get_model(smpl_type, gender)
        self.smpl_faces = np.asarray(model["faces"]).astype(np.int32).reshape(1, -1, 3)    # smplify.py:82
        self._batches = {}                  # (frames, views) -> FrameBatch: device buffers, stream and pinned mirrors are kept between calls

    def _batch(self, F, V):
        """The reference builds everything anew per frame (body_fitting.py:82); creating and destroying the device side of a
        batch costs ~9 ms - 17x the 100-iteration fit itself - so a batch is kept per shape and re-armed by set_init."""
        b = self._batches.get((F, V))
        if b is None:
            b = self._batches[(F, V)] = FrameBatch(self._dev, F, V)
            b._had_scans = b._had_masks = False
        return b

    def close(self):
        for b in self._batches.values():
            b.close()
        self._batches.clear()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def fit_frames(self, init_betas, init_poses, c2ws, Ks, keypoints, n_use_frames=None, imsize=512,
                   constant_scale=0.3, num_iters=None, flags=0, scans=None, displacement=False, masks=None,
                   mask_view_index=None):
        """Fit F independent frames in one launch.

        init_betas[F,10], init_poses[F,72], c2ws[F,V,4,4], Ks[F,V,3,3], keypoints[F,V,25,3]
        (confidence 0 = no detection); scans: optional list of F (verts, faces) scan meshes (use_mesh=True:
        the point-cloud loss switches on after num_iters // 3 and the constant scale becomes
        scan_height / 1.7, smplify.py:146-156,205-210); displacement: run the SMPL+D stage afterwards
        (smplify.py:228-247); masks: uint8[F,M,H,W] silhouettes of the views mask_view_index[M] (use_mask=True:
        5 * multview_mask_loss after num_iters // 3, smplify.py:138-144,197-199).  Returns a list of F result dicts."""
        init_betas = np.asarray(init_betas, np.float32).reshape(-1, self._dev.n_betas)
        F = init_betas.shape[0]
        c2ws = np.asarray(c2ws, np.float32).reshape(F, -1, 4, 4)
        V = c2ws.shape[1]
        batch = self._batch(F, V)
        dev_scans = []
        disp = None
        ok = False
        try:
            if batch._had_scans and scans is None:
                batch.set_scans(None)
                batch._had_scans = False
            if batch._had_masks and masks is None:
                batch.clear_masks()
                batch._had_masks = False
            batch.set_cameras(c2ws, Ks)
            batch._cams = None
            batch.set_keypoints(keypoints, n_use_frames)
            batch.set_init(init_betas, init_poses)
            if scans is not None:
                dev_scans = [Scan(v, f, device=self.device) for v, f in scans]
                batch.set_scans(dev_scans)
                batch._had_scans = True
            if masks is not None:
                masks = np.asarray(masks, np.uint8).reshape(F, -1, *np.asarray(masks).shape[-2:])
                batch.set_masks(masks, mask_view_index, None)      # contours on the device (loss.py:73-83)
                batch._had_masks = True
            hyper = make_hyper(imsize=imsize, constant_scale=constant_scale)
            n = self.num_iters if num_iters is None else num_iters
            batch.fit(n, hyper, flags | 4)
            params = batch.get_params()
            verts, joints, full_pose, terms = batch.get_result()
            if displacement and scans is not None:
                batch.fit_displacement(n, hyper)
                disp = batch.get_displacement()
            ok = True
        finally:
            if dev_scans:
                batch.set_scans(None)          # the scans of this call go away with it
                batch._had_scans = False
            for sc in dev_scans:
                sc.close()
            if not ok:                         # do not keep a batch in an unknown state
                self._batches.pop((F, V), None)
                batch.close()
        out = []
        for f in range(F):
            p = split_params(params[f], self._dev.n_joints, self._dev.n_betas)
            out.append({
                "vertices": verts[f], "joints": joints[f], "pose": p["pose"].copy(), "betas": p["betas"].copy(),
                "global_orient": p["global_orient"].copy(), "faces": self.smpl_faces[0],
                "global_transl": p["global_transl"] * p["scale"],                      # smplify.py:223
                "scale": p["scale"].copy(), "full_pose": full_pose[f],
                **({k: p[k].copy() for k in ("leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")} if "leye_pose" in p else {}),
                "loss_terms": dict(zip(("reprojection_loss", "pose_prior_loss", "angle_prior_loss", "shape_prior_loss"),
                                       (float(t) for t in terms[f]))),
            })
            if disp is not None:
                out[-1]["displacement"] = disp[f]
        return out

    def _result_dict(self, params_f, verts_f, joints_f, full_pose_f):
        p = split_params(params_f, self._dev.n_joints, self._dev.n_betas)
        return {"vertices": verts_f, "joints": joints_f, "pose": p["pose"].copy(), "betas": p["betas"].copy(),
                "global_orient": p["global_orient"].copy(), "faces": self.smpl_faces[0], "global_transl": p["global_transl"] * p["scale"],
                "scale": p["scale"].copy(), "full_pose": full_pose_f}

    def stream(self, frames, c2ws, Ks, use_frames=None, imsize=512, mask_frames=(0,), displacement=False):
        """The frame loop of apps/genebody_fitting.py:183-192 for ONE capture (fixed cameras): `frames` yields per frame what
        BodyFitting hands `SMPLify.__call__` - `(net_output, keypoints)`, or `(net_output, keypoints, masks)` /
        `(net_output, keypoints, masks, meshfile)` with `masks` the silhouettes of the views `mask_frames` (default [0], as
        `__call__`'s, smplify.py:85; a frame's `masks` = None: no silhouette loss for that frame) and `meshfile` a scan OBJ (None: no scan) - and the results come back in order, as a generator, one frame
        behind: what `__call__` returns for that frame, bit for bit.

        Keypoint-only SMPL: frame i+1's inputs are uploaded and its fit is issued before frame i's result is read, so uploads, fits
        and downloads of consecutive frames overlap (bf_batch_stage_inputs / bf_batch_get_previous).  SMPL-X and the dense losses:
        frame i+1's host work (keypoint packing, OBJ parsing) and its scan's upload and grids happen while frame i is being fitted
        (a scan's device buffers come from the library's block cache, so building one does not wait for the device); the
        first frame's silhouettes go up with its own keypoint-only iterations (bf_batch_set_masks defers the contour extraction), the
        later frames' under the fit before (bf_batch_stage_masks)."""
        from . import _lib
        V = len(c2ws) if use_frames is None else len(use_frames)
        c2w = np.stack([_np(c) for c in c2ws[:V]]).astype(np.float32)[None]
        K = np.stack([_np(k) for k in Ks[:V]]).astype(np.float32)[None]
        batch = self._batch(1, V)
        if batch._had_scans:
            batch.set_scans(None); batch._had_scans = False
        if batch._had_masks:
            batch.clear_masks(); batch._had_masks = False
        batch.set_cameras(c2w, K)
        batch._cams = None
        nl = self._dev.n_loss_joints
        # (the reference looks a mask view up in use_frames only when use_mask is on, smplify.py:141: a keypoint-only or scan-only
        #  stream whose use_frames does not hold view 0 - or that passes mask_frames=None / [] - is fine; the lookup happens at the
        #  first frame that carries masks)
        _mk_idx = []

        def mk_index():
            if not _mk_idx:
                if not mask_frames:
                    raise ValueError("stream(): a frame carries masks but mask_frames is empty - name the views they belong to (default [0], like __call__)")
                views = list(use_frames if use_frames is not None else range(V))
                _mk_idx.append([views.index(f) for f in mask_frames])
            return _mk_idx[0]

        def pack(keypoints):
            kp = np.zeros((1, V, nl, 3), np.float32)
            for i in range(V):
                if keypoints[i] is not None:
                    kp[0, i] = pack_keypoints_smplx(keypoints[i]) if self.use_hand_face else np.asarray(keypoints[i]["pose"], np.float32)[:nl]
            return kp

        frames = iter(frames)
        first = next(frames, None)
        if first is None:
            return
        dense = self.use_hand_face or len(first) > 2 or displacement
        import itertools
        frames = itertools.chain([first], frames)
        if not dense:
            hyper = make_hyper(imsize=imsize, constant_scale=0.3)
            flags = _lib.FIT_RESET | _lib.FIT_FETCH | _lib.FIT_NOTIME
            issued = 0
            for net_output, keypoints in frames:
                betas, poses = (_np(x) for x in net_output)
                batch.stage_inputs(pack(keypoints), [V], betas[:1], poses[:1])
                batch.fit(self.num_iters, hyper, flags)
                issued += 1
                if issued > 1:
                    params, verts, joints, full_pose, _ = batch.get_previous()
                    yield self._result_dict(params[0], verts[0], joints[0], full_pose[0])
            if issued:
                params = batch.get_params()
                verts, joints, full_pose, _ = batch.get_result()
                yield self._result_dict(params[0], verts[0], joints[0], full_pose[0])
            return

        # ---- SMPL-X / silhouettes / scans: one fit in flight, the next frame prepared under it
        in_flight = None                      # (device scan or None, wants displacement)

        def collect(state):
            scan, with_disp = state
            params = batch.get_params()
            verts, joints, full_pose, _ = batch.get_result()
            res = self._result_dict(params[0], verts[0], joints[0], full_pose[0])
            p = split_params(params[0], self._dev.n_joints, self._dev.n_betas)
            res.update({k: p[k].copy() for k in ("leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose") if k in p})
            if with_disp:
                batch.fit_displacement(self.num_iters, state_hyper[0])
                res["displacement"] = batch.get_displacement()[0]
            if scan is not None:
                batch.set_scans(None); batch._had_scans = False
                scan.close()
            return res

        state_hyper = [None]
        mask_shape = None
        try:
            for item in frames:
                net_output, keypoints = item[0], item[1]
                masks = item[2] if len(item) > 2 else None
                meshfile = item[3] if len(item) > 3 else None
                betas, poses = (_np(x) for x in net_output)
                kp = pack(keypoints)
                scan = cscale = None
                if meshfile is not None:      # (parsed, uploaded and indexed while the previous frame's fit runs)
                    sv, sf = load_obj_mesh(meshfile)
                    scan = Scan(sv.astype(np.float32), sf.astype(np.int32), device=self.device)
                mk = None if masks is None else np.stack([np.asarray(m) for m in masks]).astype(np.uint8)[None]
                staged = False
                if mk is not None and in_flight is not None and batch._had_masks and mask_shape == mk.shape:
                    batch.stage_masks(mk, mk_index())      # (binarised, uploaded and border-followed under the previous frame's fit)
                    staged = True
                if in_flight is not None:
                    yield collect(in_flight)
                    in_flight = None
                if scan is not None:
                    batch.set_scans([scan]); batch._had_scans = True
                if staged:
                    pass
                elif mk is not None:
                    batch.set_masks(mk, mk_index(), None); batch._had_masks = True
                    mask_shape = mk.shape
                elif batch._had_masks:
                    batch.clear_masks(); batch._had_masks = False
                hyper = make_hyper(imsize=imsize, constant_scale=0.3)
                state_hyper[0] = hyper
                batch.stage_inputs(kp, [V], betas[:1], poses[:1])
                batch.fit(self.num_iters, hyper, _lib.FIT_RESET | _lib.FIT_FETCH)
                in_flight = (scan, bool(displacement and scan is not None))
            if in_flight is not None:
                yield collect(in_flight)
                in_flight = None
        finally:
            if in_flight is not None and in_flight[0] is not None:      # (the consumer stopped early)
                batch.set_scans(None); batch._had_scans = False
                in_flight[0].close()

    def __call__(self, net_output, c2ws, Ks, keypoints, output_folder=None, use_mask=False, masks=None,
                 use_frames=[0], mask_frames=[0], keyframe=6, imsize=512, use_mesh=False, meshfile=None,
                 displacement=False):
        mk, mk_idx = None, None
        if use_mask:                                                                 # smplify.py:138-144
            mk = np.stack([np.asarray(m) for m in masks])[None]
            mk_idx = [list(use_frames).index(f) for f in mask_frames]
        scans = None
        if use_mesh:
            scan_verts, scan_faces = load_obj_mesh(meshfile)                         # smplify.py:147
            scans = [(scan_verts.astype(np.float32), scan_faces.astype(np.int32))]
        init_betas, init_poses = (_np(x) for x in net_output)
        V = len(use_frames)
        c2w = np.stack([_np(c) for c in c2ws[:V]]).astype(np.float32)
        K = np.stack([_np(k) for k in Ks[:V]]).astype(np.float32)
        nl = self._dev.n_loss_joints
        kp = np.zeros((V, nl, 3), np.float32)
        for i in range(V):
            if keypoints[i] is not None:                                               # loss.py:157
                kp[i] = pack_keypoints_smplx(keypoints[i]) if self.use_hand_face else np.asarray(keypoints[i]["pose"], np.float32)[:nl]
        if mk is None and scans is None and not self.use_hand_face:
            return self._call_staged(init_betas[:1], init_poses[:1], c2w, K, kp, imsize)
        res = self.fit_frames(init_betas[:1], init_poses[:1], c2w[None], K[None], kp[None], n_use_frames=[V],
                              imsize=imsize, scans=scans, displacement=displacement, masks=mk,
                              mask_view_index=mk_idx)[0]                               # divisor loss.py:197
        res.pop("loss_terms")
        return res

    def _call_staged(self, betas, poses, c2w, K, kp, imsize):
        """The keypoint-only SMPL call as a `stream` of one frame: the inputs go up through bf_batch_stage_inputs (one copy kernel on
        the batch's stream instead of three synchronous setters), the fit re-arms itself (BF_FIT_RESET) and the cameras of the
        previous call stay on the device when these are the same arrays' values - a capture's frames share them
        (apps/genebody_fitting.py:134-140).  Same bits as `fit_frames` (tests/test_gpu_parity.py)."""
        from . import _lib
        V = c2w.shape[0]
        batch = self._batch(1, V)
        if batch._had_scans:
            batch.set_scans(None); batch._had_scans = False
        if batch._had_masks:
            batch.clear_masks(); batch._had_masks = False
        cams = getattr(batch, "_cams", None)
        if cams is None or not (np.array_equal(cams[0], c2w) and np.array_equal(cams[1], K)):
            batch.set_cameras(c2w[None], K[None])
            batch._cams = (c2w.copy(), K.copy())
        ok = False
        try:
            batch.stage_inputs(kp[None], [V], betas, poses)
            batch.fit(self.num_iters, make_hyper(imsize=imsize, constant_scale=0.3), _lib.FIT_RESET | _lib.FIT_FETCH | _lib.FIT_NOTIME)
            params = batch.get_params()
            verts, joints, full_pose, _ = batch.get_result()
            ok = True
        finally:
            if not ok:                         # do not keep a batch in an unknown state
                self._batches.pop((1, V), None)
                batch.close()
        return self._result_dict(params[0], verts[0], joints[0], full_pose[0])
