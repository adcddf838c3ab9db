"""The body-model files the reference opens, read into the dict `native.model_desc` uploads.

The reference builds its model through smplx (smplify/smplify.py:50-80):

* SMPL   - `models.smpl.SMPL(config.SMPL_MODEL_DIR = 'data/smpl', gender=...)` -> `data/smpl/SMPL_{GENDER}.pkl`, plus
  `data/J_regressor_extra.npy` / `data/J_regressor_h36m.npy` (models/smpl.py:62-63, config.py:1-2) and
  `joint_map = [JOINT_MAP[n] for n in JOINT_NAMES]` (models/smpl.py:61);
* SMPL-X - `smplx.create(model_path='data', model_type='smplx', ext='npz', use_face_contour=True, joint_mapper=...)`
  -> `data/smplx/SMPLX_{GENDER}.npz` with `smpl_to_openpose('smplx', hands, face, contour)` as the joint mapper.

What smplx 0.1.13 [dep] does with the file contents, reproduced here:
  `v_template`, `weights` -> lbs_weights, `f` -> faces as they are; `J_regressor` densified (scipy sparse in the official pickle);
  `posedirs[NV,3,P]` -> reshape(-1, P).T = `[P, 3 NV]`; `kintree_table[0]` -> parents with parents[0] = -1;
  `shapedirs[NV,3,>=10]` -> the first 10 (SMPL-X: + the 10 expression directions: columns 10:20 of the 20-column v1.0 file, 300:310 of
  the 400-column v1.1 file); `hands_components{l,r}[:6]` (num_pca_comps=6, smplify.py:121-122); `pose_mean` = zeros for root, body,
  jaw and eyes, then `hands_mean{l,r}` (flat_hand_mean=False); the landmark tables as stored; the neck chain walked from joint 12
  to the root; the 21 selector vertices from smplx's `vertex_ids` table (layout.VERTEX_IDS), which no model file carries.

The official SMPL pickles hold `chumpy` arrays; chumpy is not a dependency here, so the unpickler maps every `chumpy.*` class to
a shell that keeps the pickled state and hands back its array (`x`) - the "cleaned" pickles smplx recommends load the same way.
No model file exists in this repository or its containers: the loaders are tested against files WRITTEN in these layouts
(`synthetic.write_official_files`), not against the licensed models.
"""
from __future__ import annotations

import os
import pickle

import numpy as np

from . import layout

N_BETAS = 10            # smplx default num_betas; the reference optimises betas[1,10] (smplify.py:103)
N_EXPRESSION = 10       # SMPLX.NUM_EXPR_COEFFS [dep]
N_HAND_PCA = 6          # smplx default num_pca_comps; left/right_hand_pose[1,6] at smplify.py:121-122
SMPLX_NECK = 12         # SMPLX.NECK_IDX [dep]


class _ChumpyShell:
    """Stands in for any chumpy class while unpickling: keeps the state, exposes the numeric payload."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"x": state})

    def __array__(self, dtype=None, copy=None):
        x = self.__dict__.get("x")
        if x is None:
            raise TypeError("chumpy object without a stored array (an expression node): re-save the model with smplx's tools/clean_ch.py")
        return np.asarray(x, dtype=dtype)


class _ModelUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "chumpy" or module.startswith("chumpy."):
            return _ChumpyShell
        return super().find_class(module, name)


def _dense(a, dtype=np.float32):
    """models/smpl.py:31-34 `to_np`: scipy sparse -> dense, then a numpy array of `dtype`."""
    if "scipy.sparse" in str(type(a)):
        a = a.todense()
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _read(path):
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=True, encoding="latin1")
        return {k: z[k] for k in z.files}
    with open(path, "rb") as f:
        return dict(_ModelUnpickler(f, encoding="latin1").load())


def _common(d, model_type, vertex_ids):
    posedirs = _dense(d["posedirs"])
    if posedirs.ndim != 3:
        raise ValueError("posedirs must be [NV,3,P] as the official files store it")
    parents = np.asarray(d["kintree_table"])[0].astype(np.int64).astype(np.int32)      # (the root's parent is stored as 2^32 - 1)
    parents[0] = -1
    faces = np.asarray(d["f"]).astype(np.int64).astype(np.int32)
    out = {
        "model_type": model_type,
        "v_template": _dense(d["v_template"]),
        "posedirs": np.ascontiguousarray(posedirs.reshape(-1, posedirs.shape[-1]).T),
        "J_regressor": _dense(d["J_regressor"]),
        "lbs_weights": _dense(d["weights"]),
        "parents": parents,
        "faces": faces,
        "selector_ids": layout.selector_ids(model_type, vertex_ids),
    }
    nv = out["v_template"].shape[0]
    if out["selector_ids"].max() >= nv:
        raise ValueError(f"selector vertex {int(out['selector_ids'].max())} outside the {nv}-vertex template: not a {model_type} topology")
    return out


def load_smpl(path, extra_regressor=None, h36m_regressor=None, vertex_ids=None):
    """`SMPL_{GENDER}.pkl` (+ the two regressor .npy files of config.py:1-2) -> model dict."""
    d = _read(path)
    m = _common(d, "smpl", vertex_ids)
    shapedirs = _dense(d["shapedirs"])
    m["shapedirs"] = np.ascontiguousarray(shapedirs[:, :, :N_BETAS])
    nv = m["v_template"].shape[0]
    for key, src in (("J_regressor_extra", extra_regressor), ("J_regressor_h36m", h36m_regressor)):
        if src is not None:
            reg = np.load(src) if isinstance(src, (str, os.PathLike)) else src
            m[key] = np.ascontiguousarray(np.asarray(reg, np.float32))          # models/smpl.py:64-65: torch.tensor(..., float32)
            if m[key].shape[1] != nv:
                raise ValueError(f"{key} has {m[key].shape[1]} columns, the template {nv} vertices")
    m["joint_map"] = layout.SMPL_JOINT_MAP.copy()                                # models/smpl.py:61,66
    return m


def load_smplx(path, vertex_ids=None, use_face_contour=True):
    """`SMPLX_{GENDER}.npz` (or .pkl) -> model dict, with the options smplify.py:60-80 passes to smplx.create."""
    d = _read(path)
    m = _common(d, "smplx", vertex_ids)
    shapedirs = _dense(d["shapedirs"])
    n = shapedirs.shape[2]
    if n >= 300 + N_EXPRESSION:                      # v1.1: 300 shape + 100 expression directions
        expr = shapedirs[:, :, 300:300 + N_EXPRESSION]
    elif n >= N_BETAS + N_EXPRESSION:                # v1.0: 10 + 10
        expr = shapedirs[:, :, N_BETAS:N_BETAS + N_EXPRESSION]
    else:
        raise ValueError(f"SMPL-X shapedirs with {n} columns: need the shape and the expression directions")
    m["shapedirs"] = np.ascontiguousarray(np.concatenate([shapedirs[:, :, :N_BETAS], expr], axis=2))
    m["left_hand_components"] = _dense(d["hands_componentsl"])[:N_HAND_PCA].copy()
    m["right_hand_components"] = _dense(d["hands_componentsr"])[:N_HAND_PCA].copy()
    nj = m["lbs_weights"].shape[1]
    pose_mean = np.zeros(3 * nj, np.float32)
    pose_mean[3 * (nj - 30):3 * (nj - 15)] = _dense(d["hands_meanl"]).reshape(-1)      # flat_hand_mean=False [dep]
    pose_mean[3 * (nj - 15):] = _dense(d["hands_meanr"]).reshape(-1)
    m["pose_mean"] = pose_mean
    m["lmk_faces_idx"] = np.asarray(d["lmk_faces_idx"]).astype(np.int32)
    m["lmk_bary_coords"] = _dense(d["lmk_bary_coords"])
    if use_face_contour:
        m["dynamic_lmk_faces_idx"] = np.asarray(d["dynamic_lmk_faces_idx"]).astype(np.int32)
        m["dynamic_lmk_bary_coords"] = _dense(d["dynamic_lmk_bary_coords"])
    chain, j = [], SMPLX_NECK                        # smplx: walk the parents from the neck to the root
    while j != -1:
        chain.append(j)
        j = int(m["parents"][j])
    m["neck_kin_chain"] = np.array(chain, np.int32)
    m["joint_map"] = layout.smpl_to_openpose("smplx", use_hands=True, use_face=True, use_face_contour=use_face_contour,
                                             openpose_format="coco25")                 # smplify.py:60-63
    return m


def find(model_type, gender, folder="data"):
    """The path the reference's smplx call resolves: `data/smpl/SMPL_MALE.pkl` (config.py:4) / `data/smplx/SMPLX_MALE.npz`."""
    g = gender.upper()
    names = ([f"smpl/SMPL_{g}.pkl", f"smpl/SMPL_{g}.npz"] if model_type == "smpl"
             else [f"smplx/SMPLX_{g}.npz", f"smplx/SMPLX_{g}.pkl"])
    for n in names:
        p = os.path.join(folder, n)
        if os.path.exists(p):
            return p
    return None


def load(model_type, gender, folder="data", vertex_ids=None):
    """-> model dict, or None when the folder has no such file."""
    path = find(model_type, gender, folder)
    if path is None:
        return None
    if model_type == "smpl":
        extra, h36m = os.path.join(folder, "J_regressor_extra.npy"), os.path.join(folder, "J_regressor_h36m.npy")
        if not os.path.exists(extra):
            raise FileNotFoundError(f"{extra}: models.smpl.SMPL needs it next to the model (config.py:1, models/smpl.py:62)")
        return load_smpl(path, extra, h36m if os.path.exists(h36m) else None, vertex_ids)
    return load_smplx(path, vertex_ids)
