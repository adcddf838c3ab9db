"""The Python mirror of the reference interface: names, signatures, file outputs (no GPU needed)."""
import inspect
import json
import os
import sys

import numpy as np
import pytest

from bodyfitting_amd import io
from bodyfitting_amd.body_fitting import BodyFitting
from bodyfitting_amd.smplify import SMPLify
from bodyfitting_amd.smpl import SMPL, ModelOutput

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_signatures_match_the_reference():
    # reference smplify/smplify.py:21-31,84-86
    assert list(inspect.signature(SMPLify.__init__).parameters)[1:] == [
        "smpl_type", "age", "step_size", "batch_size", "num_iters", "gender", "use_mask", "device", "debug"]
    assert list(inspect.signature(SMPLify.__call__).parameters)[1:] == [
        "net_output", "c2ws", "Ks", "keypoints", "output_folder", "use_mask", "masks", "use_frames", "mask_frames",
        "keyframe", "imsize", "use_mesh", "meshfile", "displacement"]
    # reference smplify/body_fitting.py:78-80 (+ net_output, replacing the out-of-scope HMR)
    assert list(inspect.signature(BodyFitting.__call__).parameters)[1:16] == [
        "images", "c2ws", "Ks", "keypoints", "gender", "keyframe", "use_frames", "use_mask", "masks", "mask_frames",
        "render_skip", "output_folder", "use_mesh", "meshfile", "disp"]
    assert {"vertices", "joints", "full_pose", "betas", "global_orient", "body_pose", "joints_ori"} <= set(
        ModelOutput.__dataclass_fields__)
    assert ModelOutput(betas=np.zeros(3))["betas"].shape == (3,)


def test_dropin_import_lines():
    sys.path.insert(0, os.path.join(REPO, "bodyfitting_amd", "dropin"))
    try:
        for mod in ("smplify", "models"):
            sys.modules.pop(mod, None)
        from smplify.body_fitting import BodyFitting as B      # apps/genebody_fitting.py:9
        from smplify.smplify import SMPLify as S2
        from models.smpl import SMPL as M
        assert B is BodyFitting and S2 is SMPLify and M is SMPL
        from utils.mesh_grid_searcher import MeshGridSearcher as G1     # smplify/smplify.py:15
        from mesh_grid_searcher import MeshGridSearcher as G2           # thirdparty/mesh_grid/test_mesh_grid.py:2
        from bodyfitting_amd.mesh_grid_searcher import MeshGridSearcher as G0
        assert G1 is G0 and G2 is G0
        import utils
        assert os.path.abspath(utils.__path__[0]) == os.path.join(REPO, "bodyfitting_amd", "dropin", "utils")
    finally:
        sys.path.pop(0)
        for mod in [m for m in sys.modules if m.split(".")[0] in ("smplify", "models", "utils", "mesh_grid_searcher")]:
            sys.modules.pop(mod)


def test_dropin_utils_package_still_reaches_the_callers_own_utils(tmp_path):
    """apps/genebody_fitting.py:14 imports utils.io_utils next to the drop-in: a `utils` package further down sys.path keeps resolving"""
    (tmp_path / "utils").mkdir()
    (tmp_path / "utils" / "__init__.py").write_text("")
    (tmp_path / "utils" / "io_utils.py").write_text("MARK = 41\n")
    sys.path.insert(0, str(tmp_path))
    sys.path.insert(0, os.path.join(REPO, "bodyfitting_amd", "dropin"))
    try:
        for mod in [m for m in sys.modules if m.split(".")[0] == "utils"]:
            sys.modules.pop(mod)
        from utils.io_utils import MARK
        from utils.mesh_grid_searcher import MeshGridSearcher
        assert MARK == 41 and MeshGridSearcher.__module__ == "bodyfitting_amd.mesh_grid_searcher"
    finally:
        sys.path.pop(0)
        sys.path.pop(0)
        for mod in [m for m in sys.modules if m.split(".")[0] == "utils"]:
            sys.modules.pop(mod)


def test_obj_writer_format(tmp_path):
    verts = np.array([[0.123456, -1.0, 2.5], [1, 2, 3]], dtype=np.float32)
    faces = np.array([[0, 1, 0]], dtype=np.int32)
    p = tmp_path / "m.obj"
    io.save_obj_mesh(str(p), verts, faces)
    assert p.read_text() == "v 0.1235 -1.0000 2.5000\nv 1.0000 2.0000 3.0000\nf 1 2 1\n"


def test_openpose_reader(tmp_path):
    pose = np.random.default_rng(0).uniform(1, 500, (25, 3))
    weak = pose.copy()
    weak[:, 2] *= 0.1
    doc = {"version": 1.3, "people": [{"person_id": [-1], "pose_keypoints_2d": weak.reshape(-1).tolist()},
                                      {"person_id": [-1], "pose_keypoints_2d": pose.reshape(-1).tolist(),
                                       "hand_left_keypoints_2d": [0.0] * 63, "face_keypoints_2d": []}]}
    p = tmp_path / "kp.json"
    p.write_text(json.dumps(doc))
    got = io.load_openpose(str(p))
    assert set(got) == {"pose"}                    # all-zero / empty blocks are dropped
    np.testing.assert_allclose(got["pose"], pose)  # the higher-scoring person wins
    p.write_text(json.dumps({"people": []}))
    assert io.load_openpose(str(p)) is None


def test_genebody_camera_dict(tmp_path):
    """annots.npy -> Ks / c2ws with the crop adjustment of apps/genebody_fitting.py:134-140"""
    rng = np.random.default_rng(0)
    K = np.tile(np.array([[1200.0, 0, 640], [0, 1190.0, 360], [0, 0, 1]]), (3, 1, 1))
    RT = np.tile(np.eye(4), (3, 1, 1)); RT[:, :3, 3] = rng.normal(size=(3, 3))
    path = tmp_path / "annots.npy"
    np.save(path, {"cams": {"K": K, "RT": RT}}, allow_pickle=True)
    Ks, c2ws = io.load_genebody_cameras(str(path), views=[0, 5, 9], crops=[(100, 200, 612, 712)] * 3, load_size=256)
    assert Ks.dtype == np.float32 and Ks.shape == (3, 3, 3) and c2ws.shape == (3, 4, 4)
    np.testing.assert_allclose(Ks[0], [[600.0, 0, 220.0], [0, 595.0, 130.0], [0, 0, 1]], rtol=1e-6)
    np.testing.assert_allclose(c2ws, RT.astype(np.float32))
    Ks2, _ = io.load_genebody_cameras({"K": K, "RT": RT}, views=[0])
    np.testing.assert_allclose(Ks2[0], K[0])
