"""File formats on either side of the loop, without cv2 / imageio / torch."""
from __future__ import annotations

import json
import re

import numpy as np


def save_obj_mesh(mesh_path, verts, faces):
    """utils/io_utils.py:185-192: `v %.4f %.4f %.4f`, 1-based `f %d %d %d`."""
    with open(mesh_path, "w") as f:
        for v in verts:
            f.write("v %.4f %.4f %.4f\n" % (v[0], v[1], v[2]))
        for tri in faces:
            f.write("f %d %d %d\n" % (tri[0] + 1, tri[1] + 1, tri[2] + 1))


def load_openpose(json_name, only_one=True):
    """OpenPose JSON -> {'pose': [25,3], 'hand_left': [21,3], 'hand_right': [21,3], 'face': [70,3]} of
    the highest-scoring person, or None (behaviour of utils/io_utils.py:138-183 for 2-D keypoints)."""
    with open(json_name, "r") as fid:
        d = json.load(fid)
    people = d.get("people", [])
    if not people:
        return None
    parsed = []
    for person in people:
        entry = {}
        for key, val in person.items():
            if "keypoints" not in key:
                continue
            p = np.asarray(val, dtype=np.float64).reshape(-1)
            if p.size == 0:
                continue
            dim = re.findall("([2-9]d)", key)
            dim = 2 if not dim else int(dim[-1][0])
            if p.size % (dim + 1) != 0:
                continue
            p = p.reshape(-1, dim + 1)
            if np.abs(p[:, -1]).max() <= 0:
                continue
            entry[key.replace("_keypoints", "").replace("_%dd" % dim, "")] = p
        parsed.append(entry)
    parsed = [e for e in parsed if e]
    if not parsed:
        return None
    if not only_one:
        return parsed
    scores = [sum(p[:, -1].sum() for p in e.values()) for e in parsed]
    return parsed[int(np.argmax(scores))]


def load_obj_mesh(mesh_file):
    """vertices float64[NV,3], faces int[F,3] (0-based; quads split as the reference does,
    utils/io_utils.py:430-475 with the default flags)."""
    verts, faces = [], []
    with open(mesh_file, "r") as f:
        for line in f:
            if line.startswith("#"):
                continue
            values = line.split()
            if not values:
                continue
            if values[0] == "v":
                verts.append([float(x) for x in values[1:4]])
            elif values[0] == "f":
                idx = [int(x.split("/")[0]) for x in values[1:]]
                faces.append(idx[:3])
                if len(idx) > 3:
                    faces.append([idx[2], idx[3], idx[0]])
    return np.array(verts), np.array(faces) - 1


def load_genebody_cameras(annots, views, crops=None, load_size=512):
    """GeneBody `annots.npy` camera dict -> (Ks [V,3,3] float32, c2ws [V,4,4] float32) for the listed views, as
    apps/genebody_fitting.py:75,134-140 prepares them: `annots` is the file path or the loaded dict (its 'cams' entry
    or the cams dict itself: 'K'[n,3,3], 'RT'[n,4,4] = camera-to-world); view i of `views` reads entry i of the
    arrays (the app enumerates its view list); `crops[i]` = (top, left, bottom, right) of the square crop that was
    resized to load_size: the principal point moves by (left, top) and both rows are rescaled."""
    if isinstance(annots, (str, bytes)) or hasattr(annots, "__fspath__"):
        annots = np.load(annots, allow_pickle=True).item()
    cams = annots["cams"] if "cams" in annots else annots
    Ks, Rts = [], []
    for i, _view in enumerate(views):
        K = np.array(cams["K"][i], dtype=np.float64, copy=True)
        Rt = np.array(cams["RT"][i], dtype=np.float64, copy=True)
        if crops is not None:
            top, left, bottom, right = crops[i]
            K[0, 2] -= left
            K[1, 2] -= top
            K[0, :] *= load_size / float(right - left)
            K[1, :] *= load_size / float(bottom - top)
        Ks.append(K.astype(np.float32))
        Rts.append(Rt.astype(np.float32))
    return np.stack(Ks), np.stack(Rts)
