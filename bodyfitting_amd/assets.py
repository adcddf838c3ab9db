"""Where body models and the GMM prior come from.

The reference reads `data/smpl/*.pkl` through smplx, `data/J_regressor_extra.npy` and
`data/gmm_08.pkl` relative to the working directory (config.py:1-6, smplify/prior.py:124-128), once
per *frame*.  Here a model is resolved once per process and cached per (type, gender, device):

  1. a dict registered with `register_model(...)` (tests, synthetic benchmarks),
  2. the files the reference itself opens - `data/smpl/SMPL_{GENDER}.pkl` + `data/J_regressor_extra.npy` (config.py:1-4,
     models/smpl.py:56-66) or `data/smplx/SMPLX_{GENDER}.npz` (smplify.py:63-80) - converted as smplx converts them
     (`model_files`),
  3. an `.npz` already in this package's tensor layout under `data/` (`{type}_{gender}.npz`),
  4. otherwise a clear error - nothing is downloaded and nothing is silently replaced.
"""
from __future__ import annotations

import os
import pickle

import numpy as np

from . import model_files

_MODELS = {}
_GMM = {}
_DEVICE_MODELS = {}


def register_model(model, model_type="smpl", gender="neutral"):
    _MODELS[(model_type, gender)] = model
    for k in [k for k in _DEVICE_MODELS if k[:2] == (model_type, gender)]:
        _DEVICE_MODELS.pop(k).close()


def register_gmm(gmm):
    _GMM["gmm"] = gmm
    for k in list(_DEVICE_MODELS):
        _DEVICE_MODELS.pop(k).close()


def _load_npz_model(model_type, gender, folder="data"):
    for name in (f"{model_type}_{gender}.npz", f"{model_type}/{model_type.upper()}_{gender.upper()}.npz"):
        path = os.path.join(folder, name)
        if os.path.exists(path):
            z = np.load(path, allow_pickle=True)
            model = {k: z[k] for k in z.files}
            model.setdefault("model_type", model_type)
            extra = os.path.join(folder, "J_regressor_extra.npy")       # config.py:1
            if "J_regressor_extra" not in model and os.path.exists(extra):
                model["J_regressor_extra"] = np.load(extra)
            return model
    return None


def get_model(model_type="smpl", gender="neutral"):
    for key in ((model_type, gender), (model_type, "neutral")):
        if key in _MODELS:
            return _MODELS[key]
    model = model_files.load(model_type, gender)
    if model is None:
        model = _load_npz_model(model_type, gender)
    if model is None:
        official = "data/smpl/SMPL_%s.pkl + data/J_regressor_extra.npy" % gender.upper() if model_type == "smpl" else "data/smplx/SMPLX_%s.npz" % gender.upper()
        raise FileNotFoundError(
            f"no {model_type}/{gender} body model: place {official} (the files the reference reads) next to the working "
            f"directory, or register a model dict with bodyfitting_amd.assets.register_model()")
    _MODELS[(model_type, gender)] = model
    return model


def get_gmm(prior_folder="data", num_gaussians=8):
    if "gmm" in _GMM:
        return _GMM["gmm"]
    path = os.path.join(prior_folder, "gmm_{:02d}.pkl".format(num_gaussians))    # prior.py:122-124
    if not os.path.exists(path):
        raise FileNotFoundError(f"GMM prior {path!r} not found and none registered (assets.register_gmm)")
    with open(path, "rb") as f:
        gmm = pickle.load(f, encoding="latin1")
    _GMM["gmm"] = {k: np.asarray(gmm[k]) for k in ("means", "covars", "weights")}
    return _GMM["gmm"]


def gmm_buffers(gmm):
    """The three buffers the merged GMM NLL uses (reference smplify/prior.py:143-160).

    Returns float32 ``means[M,D]``, ``precisions[M,D,D]`` and ``nll_weights[M]`` computed in
    float64 and rounded once, exactly as the reference constructor does.
    """
    means = np.asarray(gmm["means"], dtype=np.float32)
    covs32 = np.asarray(gmm["covars"], dtype=np.float32)
    precisions = np.stack([np.linalg.inv(c) for c in covs32]).astype(np.float32)
    sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in gmm["covars"]])
    const = (2.0 * np.pi) ** (69 / 2.0)
    nll_weights = np.asarray(gmm["weights"] / (const * (sqrdets / sqrdets.min())))
    return means, precisions, nll_weights.astype(np.float32)


def get_device_model(model_type="smpl", gender="neutral", device=0):
    """The HIP-resident model, created once per (type, gender, device)."""
    from .native import DeviceModel
    key = (model_type, gender, int(device))
    if key not in _DEVICE_MODELS:
        _DEVICE_MODELS[key] = DeviceModel(get_model(model_type, gender), get_gmm(), device=device)
    return _DEVICE_MODELS[key]
