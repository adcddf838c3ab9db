"""Index tables of the parameter / joint layout the reference fits in (product side; the synthetic generator imports them from here).

* `SMPL_JOINT_MAP`  - reference constants.py:13-89: `[JOINT_MAP[n] for n in JOINT_NAMES]` (models/smpl.py:61), 49 entries into the
  54 joints `cat[smplx 45 joints, 9 J_regressor_extra joints]` (models/smpl.py:72-75).
* `smpl_to_openpose` - reference models/utils.py:32-141: the permutation smplx's `JointMapper` applies (smplify.py:60-63 asks for
  "smplx", hands, face and face contour in coco25 order -> 135 joints).
* `VERTEX_IDS` / `selector_ids` - [dep] smplx 0.1.13 `vertex_ids.py` + `VertexJointSelector`: the vertices appended to the chain
  joints.  The official model files do not carry them; smplx ships them as a table, and so does this module.
"""
from __future__ import annotations

import numpy as np

# --- reference constants.py:13-65 (names) and :71-89 (JOINT_MAP) as (name, index) pairs in JOINT_NAMES order ---------------------
_OPENPOSE_25 = (("OP Nose", 24), ("OP Neck", 12), ("OP RShoulder", 17), ("OP RElbow", 19), ("OP RWrist", 21), ("OP LShoulder", 16),
                ("OP LElbow", 18), ("OP LWrist", 20), ("OP MidHip", 0), ("OP RHip", 2), ("OP RKnee", 5), ("OP RAnkle", 8),
                ("OP LHip", 1), ("OP LKnee", 4), ("OP LAnkle", 7), ("OP REye", 25), ("OP LEye", 26), ("OP REar", 27), ("OP LEar", 28),
                ("OP LBigToe", 29), ("OP LSmallToe", 30), ("OP LHeel", 31), ("OP RBigToe", 32), ("OP RSmallToe", 33), ("OP RHeel", 34))
_GROUND_TRUTH_24 = (("Right Ankle", 8), ("Right Knee", 5), ("Right Hip", 45), ("Left Hip", 46), ("Left Knee", 4), ("Left Ankle", 7),
                    ("Right Wrist", 21), ("Right Elbow", 19), ("Right Shoulder", 17), ("Left Shoulder", 16), ("Left Elbow", 18),
                    ("Left Wrist", 20), ("Neck (LSP)", 47), ("Top of Head (LSP)", 48), ("Pelvis (MPII)", 49), ("Thorax (MPII)", 50),
                    ("Spine (H36M)", 51), ("Jaw (H36M)", 52), ("Head (H36M)", 53), ("Nose", 24), ("Left Eye", 26), ("Right Eye", 25),
                    ("Left Ear", 28), ("Right Ear", 27))
JOINT_NAMES = [n for n, _ in _OPENPOSE_25 + _GROUND_TRUTH_24]
JOINT_MAP = dict(_OPENPOSE_25 + _GROUND_TRUTH_24)
SMPL_JOINT_MAP = np.array([JOINT_MAP[n] for n in JOINT_NAMES], dtype=np.int32)

# --- [dep] smplx 0.1.13 vertex_ids.py -------------------------------------------------------------------------------------------
VERTEX_IDS = {
    "smplh": {"nose": 332, "reye": 6260, "leye": 2800, "rear": 4071, "lear": 583,
              "rthumb": 6191, "rindex": 5782, "rmiddle": 5905, "rring": 6016, "rpinky": 6133,
              "lthumb": 2746, "lindex": 2319, "lmiddle": 2445, "lring": 2556, "lpinky": 2673,
              "LBigToe": 3216, "LSmallToe": 3226, "LHeel": 3387, "RBigToe": 6617, "RSmallToe": 6624, "RHeel": 6787},
    "smplx": {"nose": 9120, "reye": 9929, "leye": 9448, "rear": 616, "lear": 6,
              "rthumb": 8079, "rindex": 7669, "rmiddle": 7794, "rring": 7905, "rpinky": 8022,
              "lthumb": 5361, "lindex": 4933, "lmiddle": 5058, "lring": 5169, "lpinky": 5286,
              "LBigToe": 5770, "LSmallToe": 5780, "LHeel": 8846, "RBigToe": 8463, "RSmallToe": 8474, "RHeel": 8635},
}
# VertexJointSelector's order: face, feet, then the finger tips of the left and of the right hand
SELECTOR_ORDER = (["nose", "reye", "leye", "rear", "lear", "LBigToe", "LSmallToe", "LHeel", "RBigToe", "RSmallToe", "RHeel"]
                  + [side + tip for side in "lr" for tip in ("thumb", "index", "middle", "ring", "pinky")])


def selector_ids(model_type="smpl", vertex_ids=None):
    """The 21 vertex ids smplx appends to the chain joints.  smplx.SMPL defaults to the 'smplh' table, SMPL-X to 'smplx';
    `vertex_ids` (a dict with the same keys) overrides the table the way smplx's constructor argument of that name does."""
    table = vertex_ids if vertex_ids is not None else VERTEX_IDS["smplx" if model_type == "smplx" else "smplh"]
    return np.array([table[k] for k in SELECTOR_ORDER], dtype=np.int32)


# --- reference models/utils.py:32-141 ------------------------------------------------------------------------------------------
# body part of the permutation: OpenPose order over the model's chain joints; `tip0` = index of the first appended vertex joint
_BODY_CHAIN = (12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7)
# a hand in OpenPose order: wrist, then per finger (thumb, index, middle, ring, pinky) three chain joints + the tip vertex
_FINGER_OFFSETS = (12, 0, 3, 9, 6)          # chain offset of thumb, index, middle, ring, pinky inside a hand's 15 joints


def smpl_to_openpose(model_type="smplx", use_hands=True, use_face=True, use_face_contour=False, openpose_format="coco25"):
    """Indices that put a model's joints into OpenPose order (reference models/utils.py:32-141; coco25 and coco19 bodies)."""
    fmt = openpose_format.lower()
    if fmt not in ("coco25", "coco19"):
        raise ValueError("Unknown joint format: {}".format(openpose_format))
    if model_type not in ("smpl", "smplh", "smplx"):
        raise ValueError("Unknown model type: {}".format(model_type))
    n_chain = {"smpl": 24, "smplh": 52, "smplx": 55}[model_type]
    n_face_feet = 11 if fmt == "coco25" else 5                # nose, eyes, ears (+ the six foot vertices in coco25)
    body = [n_chain] + list(_BODY_CHAIN) + list(range(n_chain + 1, n_chain + n_face_feet))
    if model_type == "smpl":
        return np.array(body, dtype=np.int32)
    out = list(body)
    if use_hands:
        tip0 = n_chain + n_face_feet
        hand0 = {"smplh": (22, 37), "smplx": (25, 40)}[model_type]
        for h, (wrist, base) in enumerate(zip((20, 21), hand0)):
            out.append(wrist)
            for f, off in enumerate(_FINGER_OFFSETS):
                out += [base + off, base + off + 1, base + off + 2, tip0 + 5 * h + f]
    if use_face and model_type == "smplx":
        face0 = n_chain + n_face_feet + 10
        out += list(range(face0, face0 + 51 + 17 * bool(use_face_contour)))
    return np.array(out, dtype=np.int32)
