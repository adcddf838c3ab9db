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


def _openpose_array(key, values):
    """One `*_keypoints*` list -> (short name, array) or None, following utils/io_utils.py:150-169: an array whose values are
    all whole numbers becomes int32 and is then read as bare coordinates [n, dim] (no confidence column) unless it is all
    zero; otherwise rows of (coordinates..., confidence), dropped when every confidence is <= 0; a length that fits neither is
    cut to whole coordinate rows.  dim comes from a `_2d` .. `_9d` suffix of the key (default 2)."""
    flat = np.reshape(values, -1)
    if flat.size == 0:
        return None
    if (flat - np.floor(flat)).max() <= 0:
        flat = flat.astype(np.int32)
    tags = re.findall("([2-9]d)", key)
    dim = int(tags[-1][0]) if tags else 2
    whole = flat.dtype == np.int32
    if flat.size % (dim + 1) == 0 and (not whole or flat.max() == 0):
        rows = flat.reshape(-1, dim + 1)
        if np.abs(rows[:, -1]).max() <= 0:
            return None
    elif flat.size % dim == 0:
        rows = flat.reshape(-1, dim)
    else:
        rows = flat[:(flat.size // dim) * dim].reshape(-1, dim)
    return key.replace("_keypoints", "").replace("_%dd" % dim, ""), rows


def load_openpose(json_name, only_one=True):
    """OpenPose JSON -> {'pose': [25,3], 'hand_left': [21,3], 'hand_right': [21,3], 'face': [70,3]} of the highest-scoring
    person (only_one) or every person - the behaviour of utils/io_utils.py:138-183, quirks included: people without an id are
    collected in a list; a positive `*id*` entry turns the collection into a dict keyed by id (earlier list entries get keys
    -1, -2, ...; a person with id 0, or without an id once the dict exists, is dropped); the best person is the one whose
    summed last columns are strictly largest, starting from entry 0 with score 0; no people at all -> None."""
    with open(json_name, "r") as fid:
        doc = json.load(fid)
    people = doc.get("people", [])
    if len(people) == 0:
        return None
    found = []
    for person in people:
        entry, pid = {}, -1
        for key, val in person.items():
            if "id" in key:
                pid = np.reshape(val, -1)[0]
            elif "keypoints" in key:
                item = _openpose_array(key, val)
                if item is not None:
                    entry[item[0]] = item[1]
        if pid < 0 and isinstance(found, list):
            found.append(entry)
        elif pid > 0:
            if isinstance(found, list):
                found = {-(n + 1): e for n, e in enumerate(found)}
            found[pid] = entry
    if len(found) == 0:
        return None
    if not only_one:
        return found
    best, best_score = 0, 0
    for key, entry in (enumerate(found) if isinstance(found, list) else found.items()):
        score = sum(a[:, -1].sum() for a in entry.values())
        if score > best_score:
            best, best_score = key, score
    return found[best]


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
