"""`smplify.body_fitting.BodyFitting` of the reference (smplify/body_fitting.py:44-107) on the HIP path.

Keeps `__init__(options)` / `__call__(images, c2ws, Ks, keypoints, gender=..., keyframe=..., use_frames=...,
use_mask=..., masks=..., mask_frames=..., render_skip=..., output_folder=..., use_mesh=..., meshfile=...,
disp=...)` and the files it writes (`{smpl_type}_parameter.npy` = pickled result dict, `{smpl_type}.obj`).
The HMR initialisation (a ResNet-50 needing weights that do not ship, body_fitting.py:57-75) is out
of scope: the initial (betas, pose) come from `options.init_estimator(image, c2w)` or from the
`net_output=` keyword.
"""
from __future__ import annotations

import os

import numpy as np

from .io import save_obj_mesh
from .smplify import SMPLify


class BodyFitting:
    def __init__(self, options):
        self.options = options
        self.debug = getattr(options, "debug", False)
        self.loadsize = getattr(options, "load_size", 512)
        self.use_mask = getattr(options, "use_mask", False)
        self.smpl_type = getattr(options, "smpl_type", "smpl")
        self.use_hand_face = self.smpl_type == "smplx"
        self.init_estimator = getattr(options, "init_estimator", None)
        self.num_iters = getattr(options, "num_iters", 600)            # smplify.py:26 default
        self._fitters = {}

    def _fitter(self, gender):
        if gender not in self._fitters:      # the reference rebuilds this per call (body_fitting.py:82)
            self._fitters[gender] = SMPLify(smpl_type=self.smpl_type, age=getattr(self.options, "age", "adult"),
                                            gender=gender, use_mask=self.use_mask, num_iters=self.num_iters,
                                            device=getattr(self.options, "device", 0), debug=False)
        return self._fitters[gender]

    def __call__(self, images, c2ws, Ks, keypoints, gender="male", keyframe=25, use_frames=list(range(48)),
                 use_mask=False, masks=None, mask_frames=None, render_skip=12, output_folder=None,
                 use_mesh=False, meshfile=None, disp=False, net_output=None):
        if net_output is None:
            if self.init_estimator is None:
                raise ValueError("no initial estimate: pass net_output=(betas[1,10], pose[1,72]) or set "
                                 "options.init_estimator (the HMR network of the reference is out of scope)")
            net_output = self.init_estimator(images[keyframe], c2ws[keyframe])
        imsize = images[0].shape[0] if images is not None else self.loadsize
        result = self._fitter(gender)(net_output, c2ws, Ks, keypoints, output_folder, use_mask=use_mask, masks=masks,
                                      use_frames=use_frames, mask_frames=mask_frames, keyframe=keyframe, imsize=imsize,
                                      use_mesh=use_mesh, meshfile=meshfile, displacement=disp)
        if output_folder is not None:
            os.makedirs(output_folder, exist_ok=True)
            np.save(os.path.join(output_folder, f"{self.smpl_type}_parameter.npy"), result)
            save_obj_mesh(os.path.join(output_folder, f"{self.smpl_type}.obj"), result["vertices"], result["faces"])
            if disp and "displacement" in result:                                    # body_fitting.py:98-99
                save_obj_mesh(os.path.join(output_folder, f"{self.smpl_type}+d.obj"),
                              result["vertices"] + result["displacement"], result["faces"])
        return result
