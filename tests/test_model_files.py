"""The files the reference opens (smplify/smplify.py:46-80, models/smpl.py:56-66, config.py:1-6) through the product's loaders.

No licensed model exists here: the synthetic models are WRITTEN in the official layouts (pickled dict with a scipy-sparse joint
regressor, posedirs [NV,3,P], kintree_table, uint32 faces; SMPL-X npz with 45 hand components, hand means, 20- or 400-column
shapedirs) and read back; what reaches bf_model_create must equal the registered dict's descriptor arrays bit for bit."""
import os

import numpy as np
import pytest

from bodyfitting_amd import assets, layout, model_files, native as N, synthetic as S


def _desc_arrays(model, gmm):
    d, info, keep = N.model_desc(model, gmm)
    return info, keep


def _assert_same_descriptor(a, b, gmm, skip=()):
    ia, ka = _desc_arrays(a, gmm)
    ib, kb = _desc_arrays(b, gmm)
    assert set(ka) == set(kb)
    for k in ka:
        if k in skip:
            continue
        assert ka[k].dtype == kb[k].dtype and ka[k].shape == kb[k].shape, k
        assert np.array_equal(ka[k].view(np.uint8), kb[k].view(np.uint8)), "descriptor array %r differs" % k
    for k in ("n_verts", "n_joints", "n_betas", "n_selector", "n_joint_map", "n_loss_joints", "model_type"):
        assert ia[k] == ib[k], k


@pytest.fixture(scope="module")
def gmm():
    return S.make_gmm()


def test_smpl_pickle_in_the_official_layout_loads_to_the_same_descriptor(tmp_path, gmm):
    model = S.make_model("smpl", nv=690)
    path, vids = S.write_official_files(model, str(tmp_path), "male")
    assert path.endswith("smpl/SMPL_MALE.pkl")
    got = model_files.load("smpl", "male", str(tmp_path), vertex_ids=vids)
    _assert_same_descriptor(model, got, gmm)
    assert np.array_equal(got["J_regressor_h36m"], model["J_regressor_h36m"]) and got["parents"][0] == -1
    assert (got["joint_map"] == layout.SMPL_JOINT_MAP).all() and got["faces"].dtype == np.int32


def test_default_selector_vertices_are_smplxs_table(tmp_path, gmm):
    """without `vertex_ids` the loader takes the published table, as smplx.SMPL does (6890-vertex topology needed)"""
    model = S.make_model("smpl")
    S.write_official_files(model, str(tmp_path), "neutral")
    got = model_files.load("smpl", "neutral", str(tmp_path))
    assert got["selector_ids"].tolist() == [layout.VERTEX_IDS["smplh"][k] for k in layout.SELECTOR_ORDER]
    assert got["selector_ids"][:5].tolist() == [332, 6260, 2800, 4071, 583]           # nose, reye, leye, rear, lear [dep smplx]
    _assert_same_descriptor(model, got, gmm, skip=("selector_ids",))
    small = S.make_model("smpl", nv=690)
    S.write_official_files(small, str(tmp_path / "small"), "neutral")
    with pytest.raises(ValueError, match="outside the 690-vertex template"):
        model_files.load("smpl", "neutral", str(tmp_path / "small"))


@pytest.mark.parametrize("columns", [20, 400])
def test_smplx_npz_in_the_official_layout_loads_to_the_same_descriptor(tmp_path, gmm, columns):
    model = S.make_model("smplx", nv=1200)
    path, vids = S.write_official_files(model, str(tmp_path), "female", shape_columns=columns)
    assert path.endswith("smplx/SMPLX_FEMALE.npz")
    got = model_files.load("smplx", "female", str(tmp_path), vertex_ids=vids)
    _assert_same_descriptor(model, got, gmm)
    assert np.array_equal(got["shapedirs"], model["shapedirs"])                        # betas 0:10 + expression (10:20 or 300:310)
    assert got["neck_kin_chain"].tolist() == [12, 9, 6, 3, 0]
    assert got["left_hand_components"].shape == (6, 45) and len(got["joint_map"]) == 135
    assert np.array_equal(got["pose_mean"], model["pose_mean"])


def test_chumpy_arrays_in_a_pickle_are_read_without_chumpy(tmp_path):
    """the official SMPL pickles hold chumpy.ch.Ch objects: a stand-in module of that name pickles one, the loader's unpickler
    reads it with no chumpy importable"""
    import pickle
    import sys
    import types
    mod, sub = types.ModuleType("chumpy"), types.ModuleType("chumpy.ch")

    class Ch:
        def __init__(self, x):
            self.x = np.asarray(x)
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    sub.Ch = Ch
    mod.ch = sub
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod, sub
    try:
        payload = pickle.dumps({"v_template": Ch(np.arange(6.0).reshape(2, 3)), "plain": np.ones(2)}, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    p = tmp_path / "m.pkl"
    p.write_bytes(payload)
    d = model_files._read(str(p))
    assert np.array_equal(model_files._dense(d["v_template"]), np.arange(6, dtype=np.float32).reshape(2, 3))


def test_assets_resolves_the_references_paths(tmp_path, gmm, monkeypatch):
    """cwd-relative `data/smpl/SMPL_MALE.pkl` + `data/J_regressor_extra.npy`, as config.py:1-4 names them"""
    model = S.make_model("smpl")
    S.write_official_files(model, str(tmp_path / "data"), "male")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(assets, "_MODELS", {})
    got = assets.get_model("smpl", "male")
    _assert_same_descriptor(model, got, gmm, skip=("selector_ids",))
    with pytest.raises(FileNotFoundError, match="SMPLX_MALE.npz"):
        assets.get_model("smplx", "male")
    os.remove(tmp_path / "data" / "J_regressor_extra.npy")
    monkeypatch.setattr(assets, "_MODELS", {})
    with pytest.raises(FileNotFoundError, match="J_regressor_extra"):
        assets.get_model("smpl", "male")
