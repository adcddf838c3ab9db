"""Packing of OpenPose keypoint dicts into the [joints,3] arrays the C ABI takes (reference smplify/loss.py:157-181)."""
from __future__ import annotations

import numpy as np

FACE_MAPPING = list(range(17, 17 + 51)) + list(range(0, 17))       # reference smplify/loss.py:20


def pack_keypoints_smplx(k, part_sum_confidence=True):
    """OpenPose dict {'pose'[25,3], 'hand_left'[21,3], 'hand_right'[21,3], 'face'[70,3]} -> [135,3] in the order the
    model joints are compared (loss.py:163-181: body | left hand | right hand | face[FACE_MAPPING]); missing parts
    get confidence 0.

    part_sum_confidence reproduces a reference quirk: for the hands and the face the confidence column is NOT
    squeezed (loss.py:168,173,179 vs :162), so `conf**2 * err.sum(-1)` broadcasts to an outer product and every
    joint of a part ends up weighted by the SUM of the part's squared confidences.  The packed confidence of
    those joints is therefore sqrt(sum conf^2) of their part, which makes the ordinary conf_j^2 * rho_j identical."""
    out = np.zeros((135, 3), np.float32)
    if k is None:
        return out
    out[:25] = np.asarray(k["pose"], np.float32)[:25]
    if "hand_left" in k:
        out[25:46] = np.asarray(k["hand_left"], np.float32)
    if "hand_right" in k:
        out[46:67] = np.asarray(k["hand_right"], np.float32)
    if "face" in k:
        out[67:135] = np.asarray(k["face"], np.float32)[FACE_MAPPING]
    if part_sum_confidence:
        for a, b in ((25, 46), (46, 67), (67, 135)):
            out[a:b, 2] = np.sqrt(np.sum(out[a:b, 2].astype(np.float64) ** 2))
    return out
