"""`models.smpl.SMPL` of the reference (models/smpl.py:56-90) on the HIP path.

Same constructor / forward keyword names and the same `ModelOutput` fields; tensors are numpy
arrays (the native path has no torch).  `joints` are the 49 joints of JOINT_MAP, `joints_ori` the 45
smplx joints, `vertices` the LBS output in model space.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

import numpy as np

from . import assets


@dataclass
class ModelOutput:                       # models/smpl.py:38-54 (whose __getitem__ forgot to import asdict)
    vertices: np.ndarray = None
    joints: np.ndarray = None
    full_pose: np.ndarray = None
    betas: np.ndarray = None
    expression: np.ndarray = None
    global_orient: np.ndarray = None
    body_pose: np.ndarray = None
    left_hand_pose: np.ndarray = None
    right_hand_pose: np.ndarray = None
    jaw_pose: np.ndarray = None
    joints_ori: np.ndarray = None

    def __getitem__(self, key):
        return asdict(self)[key]


class SMPL:
    def __init__(self, model_path=None, batch_size=1, gender="neutral", age="adult", create_transl=True,
                 kid_template_path=None, device=0, **kwargs):
        if age != "adult":
            raise NotImplementedError("age='kid' needs a newer smplx than the reference pins (SURVEY.md 8c); out of scope")
        self.batch_size = batch_size
        self.gender = gender
        self._dev = assets.get_device_model("smpl", gender, device)
        model = assets.get_model("smpl", gender)
        self.faces = np.asarray(model["faces"]) if "faces" in model else None
        self.J_regressor_extra = np.asarray(model["J_regressor_extra"], dtype=np.float32)
        self.J_regressor_h36m = np.asarray(model["J_regressor_h36m"], dtype=np.float32) if "J_regressor_h36m" in model else None
        self.joint_map = np.asarray(model["joint_map"])
        self.joints = None

    def forward(self, global_orient=None, body_pose=None, betas=None, **kwargs):
        n = np.asarray(betas).reshape(-1, self._dev.n_betas).shape[0]
        verts, joints, jori = self._dev.forward(betas, global_orient, body_pose)
        self.joints = jori
        go = np.asarray(global_orient, np.float32).reshape(n, 3)
        bp = np.asarray(body_pose, np.float32).reshape(n, -1)
        return ModelOutput(vertices=verts, global_orient=go, body_pose=bp, joints=joints, joints_ori=jori,
                           betas=np.asarray(betas, np.float32).reshape(n, -1), full_pose=np.concatenate([go, bp], 1))

    __call__ = forward

    def get_joints_h36m(self, vertices):
        return np.einsum("bik,ji->bjk", np.asarray(vertices), self.J_regressor_h36m)

    def get_joints_ori(self):
        return self.joints
