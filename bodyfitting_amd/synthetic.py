"""Seeded synthetic SMPL-shaped body models and multi-view fitting problems.

No SMPL / SMPL-X model files, GMM prior or datasets exist in the build or GPU
containers (reference README.md:18 asks the user to download them), so every
test, golden fixture and bench input is generated here, with the real tensor
shapes the reference consumes:

  * body model tensors as smplx==0.1.13 stores them (called through reference
    models/smpl.py:56-83): v_template[NV,3], shapedirs[NV,3,NB],
    posedirs[9(NJ-1),3NV], J_regressor[NJ,NV] (dense), lbs_weights[NV,NJ],
    parents[NJ], plus J_regressor_extra[9,NV] (reference models/smpl.py:62-64),
    the 21 VertexJointSelector ids and the 49-entry joint map
    (reference constants.py:13-89);
  * the GMM pose prior dict {means[8,69], covars[8,69,69], weights[8]} that
    reference smplify/prior.py:127-133 unpickles from data/gmm_08.pkl;
  * a ring of calibrated cameras (recipe follows reference utils/renderer.py:7-25,
    expressed directly in the OpenCV convention the loss assumes), OpenPose-25
    keypoints with confidences, and an HMR-like initial estimate.

Everything is numpy + ``numpy.random.default_rng(seed)`` so it is bit-stable on
the build box and on the GPU box (same image).  Seeds: 0 for the model,
``1000 + f`` for frame ``f``.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

from .assets import gmm_buffers                                   # noqa: F401  (product code; re-exported for the tests)
from .keypoints import FACE_MAPPING, pack_keypoints_smplx          # noqa: F401
from .layout import SELECTOR_ORDER, SMPL_JOINT_MAP, smpl_to_openpose   # noqa: F401  (the product's tables: constants.py:71-89, models/utils.py:32-141)

# kinematic trees (smplx 0.1.13 semantics)
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int32)

SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
     15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int32)

# rest-pose joint locations of the capsule humanoid (metres, y up, +x = body left)
_SMPL_REST = np.array([
    [0.000, 0.000, 0.000],    # 0 pelvis
    [0.070, -0.090, 0.000],   # 1 L hip
    [-0.070, -0.090, 0.000],  # 2 R hip
    [0.000, 0.110, -0.020],   # 3 spine1
    [0.100, -0.480, 0.000],   # 4 L knee
    [-0.100, -0.480, 0.000],  # 5 R knee
    [0.000, 0.250, 0.000],    # 6 spine2
    [0.090, -0.900, -0.030],  # 7 L ankle
    [-0.090, -0.900, -0.030],  # 8 R ankle
    [0.000, 0.310, 0.020],    # 9 spine3
    [0.110, -0.960, 0.090],   # 10 L foot
    [-0.110, -0.960, 0.090],  # 11 R foot
    [0.000, 0.520, -0.020],   # 12 neck
    [0.080, 0.420, 0.000],    # 13 L collar
    [-0.080, 0.420, 0.000],   # 14 R collar
    [0.000, 0.610, 0.030],    # 15 head
    [0.190, 0.450, 0.000],    # 16 L shoulder
    [-0.190, 0.450, 0.000],   # 17 R shoulder
    [0.450, 0.450, 0.000],    # 18 L elbow
    [-0.450, 0.450, 0.000],   # 19 R elbow
    [0.700, 0.450, 0.000],    # 20 L wrist
    [-0.700, 0.450, 0.000],   # 21 R wrist
    [0.790, 0.450, 0.000],    # 22 L hand
    [-0.790, 0.450, 0.000],   # 23 R hand
], dtype=np.float64)

# limb radius per child joint (the tube p->j hangs off joint p)
_SMPL_RADIUS = np.array([
    0.0, 0.085, 0.085, 0.120, 0.070, 0.070, 0.125, 0.050, 0.050, 0.125, 0.040, 0.040,
    0.055, 0.070, 0.070, 0.095, 0.055, 0.055, 0.045, 0.045, 0.035, 0.035, 0.030, 0.030])

def _smplx_rest():
    """55-joint rest skeleton: the SMPL body (22 joints) + jaw/eyes + 2x15 finger joints."""
    rest = np.zeros((55, 3))
    rest[:22] = _SMPL_REST[:22]
    rest[22] = [0.000, 0.560, 0.060]    # jaw
    rest[23] = [0.030, 0.640, 0.090]    # left eye
    rest[24] = [-0.030, 0.640, 0.090]   # right eye
    radius = np.zeros(55)
    radius[:22] = _SMPL_RADIUS[:22]
    radius[22:25] = [0.035, 0.012, 0.012]
    # five fingers x three phalanges per hand, fanned out from the wrist
    for side, wrist, base in ((1.0, 20, 25), (-1.0, 21, 40)):
        for f in range(5):
            for k in range(3):
                j = base + 3 * f + k
                spread = (f - 2) * (0.019 + 0.006 * k)              # fanned: the fingers separate towards the tips
                rest[j] = rest[wrist] + [side * (0.085 + 0.028 * k), -0.004 * f, spread]
                radius[j] = 0.008
    return rest, radius


def body_primitives(model_type):
    """The implicit body the template surface is cut from (tools/make_template.py) and the skinning weights are derived from:
    [(a, b, ra, rb, k, joint)] = round cone from a (radius ra) to b (radius rb), blended into the rest with smooth-min width k,
    moved by `joint` (the parent-side joint of the bone it wraps)."""
    if model_type == "smpl":
        J = _SMPL_REST.copy()
    else:
        J, _ = _smplx_rest()
    P = []

    def cone(joint, a, b, ra, rb, k=None):
        a, b = np.asarray(a, float), np.asarray(b, float)
        P.append((a, b, float(ra), float(rb), float(min(ra, rb) * 0.6 if k is None else k), int(joint)))

    dx = np.array([0.055, 0.0, 0.0])
    # torso: two columns side by side (elliptic cross-section), pelvis -> spine1 -> spine2 -> spine3 -> neck
    spine = [0, 3, 6, 9, 12]
    rad = [0.100, 0.098, 0.100, 0.098, 0.060]
    for (i, j), (ri, rj) in zip(zip(spine[:-1], spine[1:]), zip(rad[:-1], rad[1:])):
        w = 1.0 if j != 12 else 0.35
        cone(i, J[i] + dx, J[j] + dx * w, ri, rj, 0.03)
        cone(i, J[i] - dx, J[j] - dx * w, ri, rj, 0.03)
    cone(0, J[0] + [0.0, -0.05, 0.0], J[0] + [0.0, -0.05, 0.0], 0.105, 0.105, 0.03)           # pelvis floor
    for s, hip, knee, ankle, foot in ((1, 1, 4, 7, 10), (-1, 2, 5, 8, 11)):
        cone(0, J[0] + [s * 0.05, -0.04, 0.0], J[hip], 0.100, 0.088, 0.03)
        cone(hip, J[hip], J[knee], 0.086, 0.056, 0.03)
        cone(knee, J[knee], J[ankle], 0.056, 0.040, 0.02)
        cone(ankle, J[ankle], J[foot], 0.042, 0.034, 0.02)
        cone(foot, J[foot], J[foot] + [s * 0.005, -0.012, 0.085], 0.032, 0.024, 0.015)            # toes
        cone(ankle, J[ankle] + [0.0, -0.03, 0.0], J[ankle] + [0.0, -0.055, -0.045], 0.038, 0.036, 0.02)   # heel
    for s, collar, shoulder, elbow, wrist in ((1, 13, 16, 18, 20), (-1, 14, 17, 19, 21)):
        cone(9, J[9] + [s * 0.05, 0.06, 0.0], J[collar], 0.075, 0.062, 0.03)
        cone(collar, J[collar], J[shoulder], 0.062, 0.056, 0.03)
        cone(shoulder, J[shoulder], J[elbow], 0.052, 0.040, 0.02)
        cone(elbow, J[elbow], J[wrist], 0.040, 0.029, 0.015)
    cone(12, J[12], J[15], 0.056, 0.054, 0.02)                                                    # neck
    cone(15, J[15] + [0.0, 0.05, 0.012], J[15] + [0.0, 0.105, 0.005], 0.090, 0.088, 0.02)         # skull
    cone(22 if model_type == "smplx" else 15, J[15] + [0.0, 0.0, 0.045], J[15] + [0.0, -0.035, 0.075], 0.052, 0.040, 0.015)   # jaw / chin
    cone(15, J[15] + [0.0, 0.03, 0.095], J[15] + [0.0, 0.015, 0.115], 0.016, 0.012, 0.008)        # nose
    if model_type == "smpl":
        for s, wrist, hand in ((1, 20, 22), (-1, 21, 23)):
            cone(wrist, J[wrist], J[hand], 0.028, 0.030, 0.012)
            for dz in (-0.018, 0.0, 0.018):                                                         # mitten
                cone(hand, J[hand] + [0.0, 0.0, dz], J[hand] + [s * 0.085, -0.004, dz * 1.5], 0.020, 0.013, 0.008)
            cone(hand, J[hand] + [s * -0.03, 0.0, 0.03], J[hand] + [s * 0.02, 0.0, 0.062], 0.015, 0.011, 0.008)   # thumb
    else:
        for s, wrist, base in ((1, 20, 25), (-1, 21, 40)):
            for f in range(5):
                j0, j1, j2 = base + 3 * f, base + 3 * f + 1, base + 3 * f + 2
                cone(wrist, J[wrist], J[j0], 0.024, 0.0125, 0.010)                                   # palm rays
                cone(j0, J[j0], J[j1], 0.0095, 0.0088, 0.004)
                cone(j1, J[j1], J[j2], 0.0088, 0.0080, 0.004)
                cone(j2, J[j2], J[j2] + (J[j2] - J[j1]) * 0.85, 0.0080, 0.0070, 0.004)
    return P


def load_template(model_type, nv):
    """(verts float64 [nv,3], faces int32 [2nv-4,3]) of the committed closed genus-0 template (tools/make_template.py)"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"template_{model_type}_{nv}.npz")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: generate it with `python tools/make_template.py {model_type} {nv}`")
    d = np.load(path)
    return d["verts"].astype(np.float64), d["faces"].astype(np.int32)


def skinning_weights(verts, prims, nj, top=4):
    """[nv,nj] with `top` non-zeros per row: joint i's pull on a vertex falls off with the vertex's distance to the SURFACE of the
    round cones joint i moves, measured against the nearest cone's - so a vertex of the inner left thigh belongs to the left leg
    although the right leg's bone is as close - over a blend width of 0.3 x the local limb radius."""
    nv = len(verts)
    d = np.full((nv, nj), 1.0e3)
    r_near = np.full(nv, 0.02)
    d_near = np.full(nv, 1.0e3)
    for a, b, ra, rb, _, j in prims:
        ab = b - a
        L2 = float(ab @ ab)
        t = np.clip(((verts - a) @ ab) / L2, 0.0, 1.0) if L2 > 1e-12 else np.zeros(nv)
        r = ra + t * (rb - ra)
        dp = np.linalg.norm(verts - (a + t[:, None] * ab), axis=1) - r
        d[:, j] = np.minimum(d[:, j], dp)
        take = dp < d_near
        d_near = np.where(take, dp, d_near)
        r_near = np.where(take, r, r_near)
    w = np.exp(-(d - d.min(1, keepdims=True)) / (0.3 * r_near)[:, None])
    keep = np.argsort(-w, axis=1, kind="stable")[:, :top]
    out = np.zeros_like(w)
    np.put_along_axis(out, keep, np.take_along_axis(w, keep, 1), 1)
    out[out < 1e-4] = 0.0
    return out / out.sum(1, keepdims=True)



def _nearest_vertex(verts, target, taken):
    d = np.linalg.norm(verts - np.asarray(target)[None], axis=1)
    for idx in np.argsort(d, kind="stable"):
        if int(idx) not in taken:
            taken.add(int(idx))
            return int(idx)
    raise RuntimeError("no free vertex")


def make_gmm(seed=0, n_comp=8, dim=69):
    """GMM prior dict with the keys reference smplify/prior.py:130-133 reads."""
    rng = np.random.default_rng(seed + 7)
    means = rng.normal(0.0, 0.2, size=(n_comp, dim))
    covars = np.empty((n_comp, dim, dim))
    for m in range(n_comp):
        a = rng.normal(0.0, 1.0, size=(dim, dim))
        covars[m] = a @ a.T * 0.01 + 0.5 * np.eye(dim)
    weights = rng.uniform(0.5, 1.5, size=n_comp)
    weights /= weights.sum()
    return {"means": means, "covars": covars, "weights": weights}


def make_model(model_type="smpl", seed=0, nv=None):
    """Synthetic body model with the tensor layout of smplx 0.1.13."""
    rng = np.random.default_rng(seed)
    if model_type == "smpl":
        parents, rest = SMPL_PARENTS, _SMPL_REST.copy()
        nv = 6890 if nv is None else nv
        nb = 10
    elif model_type == "smplx":
        parents = SMPLX_PARENTS
        rest, _ = _smplx_rest()
        nv = 10475 if nv is None else nv
        nb = 20  # 10 betas + 10 expression
    else:
        raise ValueError(f"unknown model type {model_type!r}")
    nj = len(parents)
    # template: one closed genus-0 surface of a human's area with near-uniform triangles (tools/make_template.py)
    verts, faces = load_template(model_type, nv)
    prims = body_primitives(model_type)
    # skinning weights: 4 non-zeros per row, smooth across the joints
    lbs = skinning_weights(verts, prims, nj)

    # joint regressor: mean of the 32 template vertices nearest to each rest joint (dense storage)
    jreg = np.zeros((nj, nv))
    for j in range(nj):
        near = np.argsort(np.linalg.norm(verts - rest[j][None], axis=1), kind="stable")[:32]
        jreg[j, near] = 1.0 / 32.0

    # shape directions: smooth affine deformations about the pelvis + a little per-vertex noise
    centre = verts.mean(0)
    shapedirs = np.empty((nv, 3, nb))
    for l in range(nb):
        a = rng.normal(0.0, 0.012, size=(3, 3))
        a[np.diag_indices(3)] += rng.normal(0.0, 0.02, size=3)
        shapedirs[:, :, l] = (verts - centre) @ a.T
    shapedirs += rng.normal(0.0, 0.0008, size=shapedirs.shape)

    posedirs = rng.normal(0.0, 0.001, size=(9 * (nj - 1), 3 * nv))

    out = {
        "model_type": model_type,
        "v_template": verts.astype(np.float32),
        "shapedirs": shapedirs.astype(np.float32),
        "posedirs": posedirs.astype(np.float32),
        "J_regressor": jreg.astype(np.float32),
        "lbs_weights": lbs.astype(np.float32),
        "parents": parents.copy(),
        "faces": faces,
    }

    taken = set()
    if model_type == "smpl":
        head, lfoot, rfoot, lhand, rhand = rest[15], rest[10], rest[11], rest[22], rest[23]
        targets = [
            head + [0.0, 0.02, 0.11],                                  # nose
            head + [-0.035, 0.05, 0.09], head + [0.035, 0.05, 0.09],   # reye, leye
            head + [-0.09, 0.03, 0.0], head + [0.09, 0.03, 0.0],       # rear, lear
            lfoot + [0.02, -0.03, 0.06], lfoot + [0.05, -0.03, 0.03], rest[7] + [0.0, -0.06, -0.05],
            rfoot + [-0.02, -0.03, 0.06], rfoot + [-0.05, -0.03, 0.03], rest[8] + [0.0, -0.06, -0.05],
        ]
        for f in range(5):
            targets.append(lhand + [0.05, 0.0, (f - 2) * 0.015])
        for f in range(5):
            targets.append(rhand + [-0.05, 0.0, (f - 2) * 0.015])
        out["selector_ids"] = np.array([_nearest_vertex(verts, t, taken) for t in targets], dtype=np.int32)
        out["joint_map"] = SMPL_JOINT_MAP.copy()
        # 9 extra + 17 h36m regressors: sparse random rows (reference models/smpl.py:62-64)
        for name, rows in (("J_regressor_extra", 9), ("J_regressor_h36m", 17)):
            reg = np.zeros((rows, nv))
            for r in range(rows):
                ids = rng.choice(nv, size=16, replace=False)
                w = rng.uniform(0.2, 1.0, size=16)
                reg[r, ids] = w / w.sum()
            out[name] = reg.astype(np.float32)
    else:
        _add_smplx_extras(out, rng, verts, faces, rest, taken)
    return out


def _add_smplx_extras(out, rng, verts, faces, rest, taken):
    """SMPL-X only pieces: hand PCA, pose mean, expression split, landmarks."""
    nv = verts.shape[0]
    head = rest[15]
    targets = [
        head + [0.0, 0.02, 0.11], head + [-0.035, 0.05, 0.09], head + [0.035, 0.05, 0.09],
        head + [-0.09, 0.03, 0.0], head + [0.09, 0.03, 0.0],
        rest[10] + [0.02, -0.03, 0.06], rest[10] + [0.05, -0.03, 0.03], rest[7] + [0.0, -0.06, -0.05],
        rest[11] + [-0.02, -0.03, 0.06], rest[11] + [-0.05, -0.03, 0.03], rest[8] + [0.0, -0.06, -0.05],
    ]
    for base, side in ((25, 1.0), (40, -1.0)):
        for f in range(5):
            targets.append(rest[base + 3 * f + 2] + [side * 0.02, 0.0, 0.0])
    out["selector_ids"] = np.array([_nearest_vertex(verts, t, taken) for t in targets], dtype=np.int32)
    # hand PCA (6 comps, smplify.py:121-122) and the non-flat hand mean that sits in pose_mean
    out["left_hand_components"] = rng.normal(0.0, 0.3, size=(6, 45)).astype(np.float32)
    out["right_hand_components"] = rng.normal(0.0, 0.3, size=(6, 45)).astype(np.float32)
    pose_mean = np.zeros(165)
    pose_mean[75:120] = rng.normal(0.0, 0.1, size=45)
    pose_mean[120:165] = rng.normal(0.0, 0.1, size=45)
    out["pose_mean"] = pose_mean.astype(np.float32)
    # 51 static + 79x17 dynamic contour landmarks as (face id, barycentric) pairs
    head_faces = np.where(np.linalg.norm(verts[faces].mean(1) - head[None], axis=1) < 0.16)[0]
    if len(head_faces) < 68:
        head_faces = np.arange(len(faces))

    def bary(n):
        b = rng.uniform(0.1, 1.0, size=(n, 3))
        return (b / b.sum(1, keepdims=True)).astype(np.float32)

    out["lmk_faces_idx"] = rng.choice(head_faces, size=51).astype(np.int32)
    out["lmk_bary_coords"] = bary(51)
    out["dynamic_lmk_faces_idx"] = rng.choice(head_faces, size=(79, 17)).astype(np.int32)
    out["dynamic_lmk_bary_coords"] = bary(79 * 17).reshape(79, 17, 3)
    out["neck_kin_chain"] = np.array([12, 9, 6, 3, 0], dtype=np.int32)
    # reference models/utils.py:75-94 with use_hands, use_face, use_face_contour -> 135 joints
    out["joint_map"] = smpl_to_openpose("smplx", use_hands=True, use_face=True, use_face_contour=True)


def write_official_files(model, folder, gender="neutral", shape_columns=None):
    """Write `model` (a dict of this module) the way the licensed files are laid out, so that the product's loaders
    (`model_files`, the path apps/genebody_fitting.py takes through smplx) can be tested without them:

      SMPL   -> {folder}/smpl/SMPL_{GENDER}.pkl: a pickled dict with `v_template[NV,3]`, `shapedirs[NV,3,10]`, `posedirs[NV,3,207]`,
                `J_regressor` (scipy.sparse csc), `weights[NV,24]`, `kintree_table[2,24]` (uint32, root parent 2^32 - 1), `f` (uint32),
                + {folder}/J_regressor_extra.npy, J_regressor_h36m.npy (config.py:1-2);
      SMPL-X -> {folder}/smplx/SMPLX_{GENDER}.npz with the same keys + `hands_components{l,r}[45,45]`, `hands_mean{l,r}[45]`,
                `lmk_faces_idx`, `lmk_bary_coords`, `dynamic_lmk_*`; `shape_columns` = 20 (v1.0 layout) or 400 (v1.1: expression at 300).
    Returns (path, vertex_ids): the selector vertices of a synthetic template are not the licensed topology's, so they come back as
    the dict smplx's `vertex_ids` argument takes."""
    import pickle
    import scipy.sparse as sp
    mt = model.get("model_type", "smpl")
    nv, nj = model["lbs_weights"].shape
    P = model["posedirs"].shape[0]
    kin = np.stack([np.asarray(model["parents"]).astype(np.int64) % (1 << 32), np.arange(nj)]).astype(np.uint32)
    d = {
        "v_template": np.asarray(model["v_template"], np.float64),
        "posedirs": np.ascontiguousarray(np.asarray(model["posedirs"], np.float64).T.reshape(nv, 3, P)),
        "J_regressor": sp.csc_matrix(np.asarray(model["J_regressor"], np.float64)),
        "weights": np.asarray(model["lbs_weights"], np.float64),
        "kintree_table": kin,
        "f": np.asarray(model["faces"]).astype(np.uint32),
    }
    vertex_ids = {k: int(v) for k, v in zip(SELECTOR_ORDER, model["selector_ids"])}
    if mt == "smpl":
        d["shapedirs"] = np.asarray(model["shapedirs"], np.float64)
        os.makedirs(os.path.join(folder, "smpl"), exist_ok=True)
        path = os.path.join(folder, "smpl", "SMPL_%s.pkl" % gender.upper())
        with open(path, "wb") as f:
            pickle.dump(d, f, protocol=2)
        np.save(os.path.join(folder, "J_regressor_extra.npy"), np.asarray(model["J_regressor_extra"], np.float64))
        np.save(os.path.join(folder, "J_regressor_h36m.npy"), np.asarray(model["J_regressor_h36m"], np.float64))
        return path, vertex_ids
    sd = np.asarray(model["shapedirs"], np.float64)
    ncol = int(shape_columns or 20)
    full = np.zeros((nv, 3, ncol))
    full[:, :, :10] = sd[:, :, :10]
    e0 = 300 if ncol >= 310 else 10
    full[:, :, e0:e0 + 10] = sd[:, :, 10:20]
    if ncol >= 310:
        full[:, :, 10:300] = 0.125                      # directions the loader must NOT pick up
    d["shapedirs"] = full
    rng = np.random.default_rng(7)
    for side, key in (("l", "left_hand_components"), ("r", "right_hand_components")):
        comp = rng.normal(0.0, 0.3, size=(45, 45))      # the file holds all 45 components; the first six are the model's
        comp[:6] = model[key]
        d["hands_components" + side] = comp
    pm = np.asarray(model["pose_mean"], np.float64)
    d["hands_meanl"], d["hands_meanr"] = pm[3 * (nj - 30):3 * (nj - 15)].copy(), pm[3 * (nj - 15):].copy()
    d["lmk_faces_idx"] = np.asarray(model["lmk_faces_idx"]).astype(np.int64)
    d["lmk_bary_coords"] = np.asarray(model["lmk_bary_coords"], np.float64)
    d["dynamic_lmk_faces_idx"] = np.asarray(model["dynamic_lmk_faces_idx"]).astype(np.int64)
    d["dynamic_lmk_bary_coords"] = np.asarray(model["dynamic_lmk_bary_coords"], np.float64)
    os.makedirs(os.path.join(folder, "smplx"), exist_ok=True)
    path = os.path.join(folder, "smplx", "SMPLX_%s.npz" % gender.upper())
    np.savez(path, **{k: (v.toarray() if hasattr(v, "toarray") else v) for k, v in d.items()})
    return path, vertex_ids


def model_digest(model):
    """Short content hash of the float tensors, stored next to goldens to catch generator drift."""
    h = hashlib.sha256()
    for key in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
        h.update(np.ascontiguousarray(model[key]).tobytes())
    return h.hexdigest()[:16]


# ----------------------------------------------------------------------------------------------
# cameras and per-frame problems
# ----------------------------------------------------------------------------------------------

def ring_cameras(n_views=48, radius=3.2, centre=(0.0, 0.0, 0.0), imsize=512, focal=512.0):
    """`n_views` cameras equally spaced in yaw on a ring, looking at `centre`.

    Returns (c2ws list of [4,4] float32, Ks list of [3,3] float32) the way
    apps/genebody_fitting.py:134-140 hands them to BodyFitting (camera-to-world `RT`, pixel K).
    Camera axes follow OpenCV: x right, y down, z forward.
    """
    centre = np.asarray(centre, dtype=np.float64)
    c2ws, Ks = [], []
    for theta in np.linspace(0.0, 2.0 * np.pi, n_views + 1)[:-1]:
        pos = centre + radius * np.array([np.cos(theta), 0.0, -np.sin(theta)])
        fwd = centre - pos
        fwd /= np.linalg.norm(fwd)
        down = np.array([0.0, -1.0, 0.0])
        right = np.cross(down, fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
        c2ws.append(c2w.astype(np.float32))
        Ks.append(np.array([[focal, 0, imsize / 2], [0, focal, imsize / 2], [0, 0, 1]], dtype=np.float32))
    return c2ws, Ks


def _rodrigues64(rvec):
    a = np.linalg.norm(rvec)
    if a < 1e-12:
        return np.eye(3)
    k = rvec / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def smpl_joints64(model, betas, full_pose):
    """float64 posed model for *data synthesis* only (GT keypoints); not an oracle.

    Returns (vertices[NV,3], joints45[NJ+21,3]).
    """
    vt = model["v_template"].astype(np.float64)
    sd = model["shapedirs"].astype(np.float64)
    nb = min(sd.shape[2], len(betas))
    v_shaped = vt + sd[:, :, :nb] @ np.asarray(betas, dtype=np.float64)[:nb]
    J = model["J_regressor"].astype(np.float64) @ v_shaped
    nj = len(model["parents"])
    R = np.stack([_rodrigues64(r) for r in np.asarray(full_pose, dtype=np.float64).reshape(nj, 3)])
    feat = (R[1:] - np.eye(3)[None]).reshape(-1)
    v_posed = v_shaped + (feat @ model["posedirs"].astype(np.float64)).reshape(-1, 3)
    G = np.zeros((nj, 4, 4))
    for i in range(nj):
        T = np.eye(4)
        T[:3, :3] = R[i]
        p = int(model["parents"][i])
        T[:3, 3] = J[i] if p < 0 else J[i] - J[p]
        G[i] = T if p < 0 else G[p] @ T
    A = G.copy()
    for i in range(nj):
        A[i, :3, 3] -= G[i, :3, :3] @ J[i]
    Tv = np.einsum("vj,jab->vab", model["lbs_weights"].astype(np.float64), A)
    verts = np.einsum("vab,vb->va", Tv[:, :3, :3], v_posed) + Tv[:, :3, 3]
    joints = np.concatenate([G[:, :3, 3], verts[model["selector_ids"]]], 0)
    return verts, joints


def render_mask(verts_world, c2w, K, imsize=512, radius=5):
    """uint8 [imsize,imsize] silhouette (255 = body): union of discs splatted at the projected vertices."""
    from scipy import ndimage
    w2c = np.linalg.inv(np.asarray(c2w, np.float64))
    cam = verts_world @ w2c[:3, :3].T + w2c[:3, 3]
    uvw = cam @ np.asarray(K, np.float64).T
    uv = np.round(uvw[:, :2] / uvw[:, 2:3]).astype(int)
    img = np.zeros((imsize, imsize), bool)
    ok = (uv[:, 0] >= 0) & (uv[:, 0] < imsize) & (uv[:, 1] >= 0) & (uv[:, 1] < imsize)
    img[uv[ok, 1], uv[ok, 0]] = True
    yy, xx = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    img = ndimage.binary_dilation(img, structure=(xx * xx + yy * yy) <= radius * radius)
    return (img * 255).astype(np.uint8)


def make_problem(model, frame=0, n_views=48, imsize=512, constant_scale=0.3, pose_noise=0.1,
                 drop=0.05, missing_views=(), mask_frames=None):
    """One synthetic SMPL frame: cameras, OpenPose-25 keypoints, HMR-like init.

    keypoints[v] is ``{'pose': float32[25,3]}`` (x, y, confidence) like utils/io_utils.py:138-183
    returns, or None for a view listed in `missing_views` (loss.py:157 skips those).
    """
    assert model["model_type"] == "smpl"
    rng = np.random.default_rng(1000 + frame)
    betas_gt = rng.normal(0.0, 0.5, size=10)
    pose_gt = rng.normal(0.0, 0.2, size=72)
    pose_gt[:3] = [0.0, rng.uniform(-np.pi, np.pi), 0.0]
    pose_gt[:3] += rng.normal(0.0, 0.1, size=3)
    transl_gt = rng.normal(0.0, 0.05, size=3)
    scale_gt = rng.uniform(0.9, 1.1) / constant_scale

    verts_gt, joints = smpl_joints64(model, betas_gt, pose_gt)
    op25 = joints[model["joint_map"][:25]]
    world = (op25 + transl_gt) * scale_gt * constant_scale

    # the camera ring belongs to the capture, not to the frame: every frame of the synthetic subject is seen by the same
    # calibrated cameras (BASELINE config 4: "distinct GT poses / keypoints, shared cameras + model")
    c2ws, Ks = ring_cameras(n_views, imsize=imsize, focal=float(imsize), centre=(0.0, 0.05, 0.0))
    keypoints = []
    for v in range(n_views):
        w2c = np.linalg.inv(c2ws[v].astype(np.float64))
        cam = world @ w2c[:3, :3].T + w2c[:3, 3]
        uvw = cam @ Ks[v].astype(np.float64).T
        uv = uvw[:, :2] / uvw[:, 2:3] + rng.normal(0.0, 1.0, size=(25, 2))
        conf = rng.uniform(0.5, 1.0, size=25)
        conf[rng.uniform(size=25) < drop] = 0.0
        kp = np.concatenate([uv, conf[:, None]], 1).astype(np.float32)
        kp[conf == 0.0, :2] = 0.0
        keypoints.append(None if v in missing_views else {"pose": kp})

    init_pose = (pose_gt + rng.normal(0.0, pose_noise, size=72)).astype(np.float32)
    init_betas = np.zeros(10, dtype=np.float32)
    extra = {}
    if mask_frames is not None:        # ground-truth silhouettes for use_mask=True (smplify.py:138-144)
        vw = (verts_gt + transl_gt) * scale_gt * constant_scale
        extra = {"mask_frames": list(mask_frames),
                 "masks": [render_mask(vw, c2ws[v], Ks[v], imsize) for v in mask_frames]}
    return {
        **extra,
        "c2ws": c2ws, "Ks": Ks, "keypoints": keypoints, "imsize": imsize,
        "use_frames": list(range(n_views)),
        "init_betas": init_betas[None], "init_pose": init_pose[None],
        "gt": {"betas": betas_gt, "pose": pose_gt, "transl": transl_gt, "scale": scale_gt},
        "constant_scale": constant_scale,
    }


def _scan_noise(points, rng):
    """what separates a scan from the body under it, in units of `noise` (3 mm): smooth low-frequency relief (wavelengths of 25
    and 16 cm, SURVEY.md 8d: "GT posed mesh + smooth noise") plus a tenth of that as white sensor noise"""
    p = np.asarray(points, np.float64)
    return (0.6 * np.sin(p * 25.0 + rng.uniform(0, 6.28, size=3)) + 0.4 * np.sin(p[:, [1, 2, 0]] * 40.0 + rng.uniform(0, 6.28, size=3))
            + rng.normal(0.0, 0.1, size=p.shape))


def make_scan_problem(model, frame=0, n_views=8, imsize=512, scan_scale=1.0, noise=0.003, pose_noise=0.1):
    """A frame with a scan mesh (use_mesh=True, reference smplify.py:146-156): the scan is the ground-
    truth posed body (the model's own topology) with smooth noise, in a world whose constant scale is
    scan_height / 1.7 by construction.  Returns (problem dict, scan_verts float32[NV,3], scan_faces int32[F,3])."""
    rng = np.random.default_rng(5000 + frame)
    betas_gt = rng.normal(0.0, 0.5, size=10)
    pose_gt = rng.normal(0.0, 0.15, size=72)
    pose_gt[:3] = [0.0, rng.uniform(-np.pi, np.pi), 0.0]
    transl_gt = rng.normal(0.0, 0.05, size=3)
    verts, joints = smpl_joints64(model, betas_gt, pose_gt)
    ext_y = verts[:, 1].max() - verts[:, 1].min()
    scale_gt = 1.7 / ext_y                     # => scan_height / 1.7 == scan_scale exactly
    scan = (verts + transl_gt) * scale_gt * scan_scale
    scan = scan + noise * scan_scale * _scan_noise(scan / scan_scale, rng)
    # the reference reads the scan from an OBJ written with 4 decimals (utils/io_utils.py:185-192)
    scan = np.round(scan, 4).astype(np.float32)
    cscale = float((scan[:, 1].max() - scan[:, 1].min()) / 1.7)
    world = (joints[model["joint_map"][:25]] + transl_gt) * scale_gt * scan_scale
    centre = world.mean(0)
    c2ws, Ks = ring_cameras(n_views, radius=3.2 * scan_scale, imsize=imsize, focal=float(imsize), centre=centre.tolist())
    keypoints = []
    for v in range(n_views):
        w2c = np.linalg.inv(c2ws[v].astype(np.float64))
        cam = world @ w2c[:3, :3].T + w2c[:3, 3]
        uvw = cam @ Ks[v].astype(np.float64).T
        uv = uvw[:, :2] / uvw[:, 2:3] + rng.normal(0.0, 1.0, size=(25, 2))
        conf = rng.uniform(0.5, 1.0, size=25)
        keypoints.append({"pose": np.concatenate([uv, conf[:, None]], 1).astype(np.float32)})
    init_pose = (pose_gt + rng.normal(0.0, pose_noise, size=72)).astype(np.float32)
    prob = {"c2ws": c2ws, "Ks": Ks, "keypoints": keypoints, "imsize": imsize, "use_frames": list(range(n_views)),
            "init_betas": np.zeros((1, 10), np.float32), "init_pose": init_pose[None], "constant_scale": cscale,
            "gt": {"betas": betas_gt, "pose": pose_gt, "transl": transl_gt, "scale": scale_gt}}
    return prob, scan, np.asarray(model["faces"], dtype=np.int32)


def subdivide_mesh(verts, faces, times=1):
    """Midpoint (1 -> 4) subdivision: the scan meshes of BASELINE config 5 have ~100k triangles,
    SMPL-X's 20,908 faces once subdivided give 83,632.  Deterministic: new vertices in sorted-edge order."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    for _ in range(times):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
        e.sort(1)
        key = e[:, 0] * len(v) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        mid = 0.5 * (v[uniq // len(v)] + v[uniq % len(v)])
        m = inv.reshape(3, -1).T + len(v)                     # midpoints of edges (01, 12, 20) per face
        f = np.concatenate([np.stack([f[:, 0], m[:, 0], m[:, 2]], 1), np.stack([f[:, 1], m[:, 1], m[:, 0]], 1),
                            np.stack([f[:, 2], m[:, 2], m[:, 1]], 1), np.stack([m[:, 0], m[:, 1], m[:, 2]], 1)], 0)
        v = np.concatenate([v, mid], 0)
    return v, f.astype(np.int32)


def smplx_full_pose(model, global_orient, body_pose, leye, reye, lhand_pca, rhand_pca, jaw=None):
    """[165] axis-angle vector smplx feeds to lbs: PCA hands, plus pose_mean."""
    jaw = np.zeros(3) if jaw is None else np.asarray(jaw, np.float64).reshape(3)
    lh = np.asarray(lhand_pca, np.float64) @ model["left_hand_components"].astype(np.float64)
    rh = np.asarray(rhand_pca, np.float64) @ model["right_hand_components"].astype(np.float64)
    fp = np.concatenate([np.asarray(global_orient, np.float64).reshape(3), np.asarray(body_pose, np.float64).reshape(63), jaw,
                         np.asarray(leye, np.float64).reshape(3), np.asarray(reye, np.float64).reshape(3), lh, rh])
    return fp + model["pose_mean"].astype(np.float64)


def smplx_joints64(model, betas, full_pose):
    """float64 SMPL-X joints for data synthesis: (vertices, 135 mapped joints, dynamic-contour row)."""
    verts, j76 = smpl_joints64(model, betas, full_pose)
    R = [_rodrigues64(full_pose[3 * j:3 * j + 3]) for j in model["neck_kin_chain"]]
    rel = np.eye(3)
    for r in R:
        rel = r @ rel
    yaw = np.arctan2(-rel[2, 0], np.sqrt(rel[0, 0] ** 2 + rel[1, 0] ** 2))
    y = int(np.round(min(-yaw * 180.0 / np.pi, 39)))
    if y < 0:
        y = 78 if y < -39 else 39 - y
    fidx = np.concatenate([model["lmk_faces_idx"], model["dynamic_lmk_faces_idx"][y]])
    bary = np.concatenate([model["lmk_bary_coords"], model["dynamic_lmk_bary_coords"][y]]).astype(np.float64)
    lm = np.einsum("lfi,lf->li", verts[model["faces"][fidx]], bary)
    joints = np.concatenate([j76, lm], 0)
    return verts, joints[model["joint_map"]], y


def make_problem_smplx(model, frame=0, n_views=48, imsize=512, constant_scale=0.3, pose_noise=0.08, mask_frames=None):
    """One synthetic SMPL-X frame (BASELINE config 3): body + hands + face OpenPose keypoints per view."""
    assert model["model_type"] == "smplx"
    rng = np.random.default_rng(3000 + frame)
    gt = {"betas": rng.normal(0.0, 0.5, size=10), "body_pose": rng.normal(0.0, 0.15, size=63),
          "global_orient": np.array([0.0, rng.uniform(-0.6, 0.6), 0.0]) + rng.normal(0.0, 0.05, size=3),
          "leye": rng.normal(0.0, 0.05, size=3), "reye": rng.normal(0.0, 0.05, size=3),
          "lhand": rng.normal(0.0, 0.5, size=6), "rhand": rng.normal(0.0, 0.5, size=6),
          "transl": rng.normal(0.0, 0.05, size=3), "scale": rng.uniform(0.9, 1.1) / constant_scale}
    fp = smplx_full_pose(model, gt["global_orient"], gt["body_pose"], gt["leye"], gt["reye"], gt["lhand"], gt["rhand"])
    verts, joints, _ = smplx_joints64(model, np.concatenate([gt["betas"], np.zeros(10)]), fp)
    world = (joints + gt["transl"]) * gt["scale"] * constant_scale
    c2ws, Ks = ring_cameras(n_views, imsize=imsize, focal=float(imsize), centre=world[:25].mean(0).tolist())
    inv_face = np.argsort(FACE_MAPPING)
    keypoints = []
    for v in range(n_views):
        w2c = np.linalg.inv(c2ws[v].astype(np.float64))
        cam = world @ w2c[:3, :3].T + w2c[:3, 3]
        uvw = cam @ Ks[v].astype(np.float64).T
        uv = uvw[:, :2] / uvw[:, 2:3] + rng.normal(0.0, 0.7, size=(135, 2))
        conf = rng.uniform(0.5, 1.0, size=135)
        kp = np.concatenate([uv, conf[:, None]], 1).astype(np.float32)
        face70 = np.zeros((70, 3), np.float32)
        face70[:68] = kp[67:][inv_face]                 # back to OpenPose's own face order
        keypoints.append({"pose": kp[:25], "hand_left": kp[25:46], "hand_right": kp[46:67], "face": face70})
    init_pose = np.concatenate([gt["global_orient"], gt["body_pose"]]) + rng.normal(0.0, pose_noise, size=66)
    extra = {}
    if mask_frames is not None:
        vw = (verts + gt["transl"]) * gt["scale"] * constant_scale
        extra = {"mask_frames": list(mask_frames), "masks": [render_mask(vw, c2ws[v], Ks[v], imsize) for v in mask_frames]}
    return {**extra, "c2ws": c2ws, "Ks": Ks, "keypoints": keypoints, "imsize": imsize, "use_frames": list(range(n_views)),
            "init_betas": np.zeros((1, 10), np.float32), "init_pose": np.concatenate([init_pose, np.zeros(6)]).astype(np.float32)[None],
            "gt": gt, "constant_scale": constant_scale}


def make_scan_problem_smplx(model, frame=0, n_views=8, imsize=512, noise=0.003, pose_noise=0.08, subdivide=1):
    """A SMPL-X frame with a scan mesh (BASELINE config 5: use_mesh + SMPL+D): make_problem_smplx's ground truth in
    a world whose constant scale is scan_height / 1.7 (smplify.py:150-156), the scan = the posed ground-truth
    body, `subdivide` times midpoint-subdivided (once: 83,632 triangles), with smooth + white noise, rounded to
    the 4 decimals of the OBJ it would be read from.  Returns (problem dict, scan_verts, scan_faces)."""
    assert model["model_type"] == "smplx"
    rng = np.random.default_rng(7000 + frame)
    gt = {"betas": rng.normal(0.0, 0.5, size=10), "body_pose": rng.normal(0.0, 0.15, size=63),
          "global_orient": np.array([0.0, rng.uniform(-np.pi, np.pi), 0.0]),
          "leye": rng.normal(0.0, 0.05, size=3), "reye": rng.normal(0.0, 0.05, size=3),
          "lhand": rng.normal(0.0, 0.5, size=6), "rhand": rng.normal(0.0, 0.5, size=6),
          "transl": rng.normal(0.0, 0.05, size=3)}
    fp = smplx_full_pose(model, gt["global_orient"], gt["body_pose"], gt["leye"], gt["reye"], gt["lhand"], gt["rhand"])
    verts, joints, _ = smplx_joints64(model, np.concatenate([gt["betas"], np.zeros(10)]), fp)
    gt["scale"] = 1.7 / (verts[:, 1].max() - verts[:, 1].min())
    sv, sf = subdivide_mesh((verts + gt["transl"]) * gt["scale"], model["faces"], subdivide)
    sv = np.round(sv + noise * _scan_noise(sv, rng), 4).astype(np.float32)
    cscale = float((sv[:, 1].max() - sv[:, 1].min()) / 1.7)
    world = (joints + gt["transl"]) * gt["scale"]
    c2ws, Ks = ring_cameras(n_views, radius=3.2, imsize=imsize, focal=float(imsize), centre=world[:25].mean(0).tolist())
    inv_face = np.argsort(FACE_MAPPING)
    keypoints = []
    for v in range(n_views):
        w2c = np.linalg.inv(c2ws[v].astype(np.float64))
        cam = world @ w2c[:3, :3].T + w2c[:3, 3]
        uvw = cam @ Ks[v].astype(np.float64).T
        uv = uvw[:, :2] / uvw[:, 2:3] + rng.normal(0.0, 0.7, size=(135, 2))
        kp = np.concatenate([uv, rng.uniform(0.5, 1.0, size=(135, 1))], 1).astype(np.float32)
        face70 = np.zeros((70, 3), np.float32)
        face70[:68] = kp[67:][inv_face]
        keypoints.append({"pose": kp[:25], "hand_left": kp[25:46], "hand_right": kp[46:67], "face": face70})
    init_pose = np.concatenate([gt["global_orient"], gt["body_pose"]]) + rng.normal(0.0, pose_noise, size=66)
    prob = {"c2ws": c2ws, "Ks": Ks, "keypoints": keypoints, "imsize": imsize, "use_frames": list(range(n_views)),
            "init_betas": np.zeros((1, 10), np.float32),
            "init_pose": np.concatenate([init_pose, np.zeros(6)]).astype(np.float32)[None], "gt": gt, "constant_scale": cscale}
    return prob, sv, sf
