"""Stand-in for the un-vendored `smplx==0.1.13` dependency.  TEST INFRASTRUCTURE ONLY.

The reference imports `smplx` (requirements.txt:5; call sites smplify/smplify.py:51-56,80,179-187,
models/smpl.py:4-6,60,71-72) but the package is not in /root/reference and not installed.  This
module exposes just enough of its API for the *unmodified* reference loop to run on torch CPU in the
build container (oracle/gen_golden.py).  The arithmetic is oracle.smplify_oracle.lbs - a restatement
of the published smplx semantics - so goldens produced through it pin the reference's loop, losses
and priors, but NOT the LBS arithmetic itself ("parity unpinned" at that boundary).

Model tensors do not come from `model_path` (no model files exist here); gen_golden.py registers a
synthetic model dict in MODEL_REGISTRY before the reference constructs its SMPL.
"""
import numpy as np
import torch
import torch.nn as nn

from . import lbs  # noqa: F401  (reference does `from smplx.lbs import vertices2joints`)

MODEL_REGISTRY = {}


class _Output:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23

    def __init__(self, model_path=None, batch_size=1, create_transl=True, gender="neutral", **kwargs):
        super().__init__()
        from oracle import smplify_oracle as O
        model = MODEL_REGISTRY["smpl"]
        self._O = O
        self.faces = np.asarray(model["faces"])
        t = O.to_torch_model(model, torch.float32)
        for name in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
            self.register_buffer(name, t[name])
        self.register_buffer("selector_ids", t["selector_ids"])
        self.parents_list = t["parents"]
        if create_transl:
            self.transl = nn.Parameter(torch.zeros(batch_size, 3), requires_grad=True)

    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, **kwargs):
        m = {"v_template": self.v_template, "shapedirs": self.shapedirs, "posedirs": self.posedirs,
             "J_regressor": self.J_regressor, "lbs_weights": self.lbs_weights, "parents": self.parents_list}
        full_pose = torch.cat([global_orient, body_pose], dim=1)
        vertices, joints = self._O.lbs(betas, full_pose, m)
        joints = torch.cat([joints, vertices[:, self.selector_ids]], dim=1)
        if transl is None and hasattr(self, "transl"):
            transl = self.transl
        if transl is not None:
            joints = joints + transl.unsqueeze(1)
            vertices = vertices + transl.unsqueeze(1)
        return _Output(vertices=vertices, joints=joints, betas=betas, global_orient=global_orient,
                       body_pose=body_pose, full_pose=full_pose)


class SMPLX(nn.Module):
    """`smplx.create(model_type='smplx', use_face_contour=True, joint_mapper=...)` as the reference builds it
    (smplify/smplify.py:59-80): 6 hand PCA components, non-flat hand mean, the caller's JointMapper applied to the
    144 joints (55 chain + 21 selector vertices + 51 static + 17 dynamic-contour landmarks)."""

    def __init__(self, joint_mapper=None, **kwargs):
        super().__init__()
        from oracle import smplify_oracle as O
        model = MODEL_REGISTRY["smplx"]
        self._O = O
        self.faces = np.asarray(model["faces"])
        self._m = O.to_torch_model(model, torch.float32)
        self.joint_mapper = joint_mapper

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                jaw_pose=None, leye_pose=None, reye_pose=None, expression=None, return_full_pose=False, **kwargs):
        out = self._O.smplx_forward(self._m, betas, global_orient, body_pose, leye_pose, reye_pose, left_hand_pose,
                                    right_hand_pose, jaw_pose=jaw_pose, mapped=False)
        joints = out["joints"]
        if self.joint_mapper is not None:
            joints = self.joint_mapper(joints)
        return _Output(vertices=out["vertices"], joints=joints, betas=betas, global_orient=global_orient, body_pose=body_pose,
                       full_pose=out["full_pose"])


def create(model_path=None, model_type="smpl", **kwargs):
    if model_type == "smpl":
        return SMPL(model_path, **kwargs)
    if model_type == "smplx":
        return SMPLX(**kwargs)
    raise NotImplementedError("stand-in smplx.create: model_type %r not provided" % model_type)
