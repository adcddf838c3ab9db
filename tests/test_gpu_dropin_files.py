"""-m gpu: the two drop-in pieces round 5 added, through the C-ABI on the device:

* `utils.mesh_grid_searcher.MeshGridSearcher` (reference utils/mesh_grid_searcher.py:51-99) with the call sequence of the
  reference's own thirdparty/mesh_grid/test_mesh_grid.py:11-34 (build, 10+ random points in the bounding box, nearest_points,
  inside_mesh, signed distance) - checked against the oracle's brute force instead of trimesh (absent here);
* the body model read from files in the layouts the reference opens (smplify.py:46-80 -> data/smpl/SMPL_*.pkl, data/smplx/
  SMPLX_*.npz), then BASELINE config 2's fit against the reference's golden through that path."""
import numpy as np
import pytest

from conftest import load_golden
from bodyfitting_amd import assets, model_files, native as N, synthetic as S
from bodyfitting_amd.mesh_grid_searcher import MeshGridSearcher
from oracle import mesh_oracle as MO

pytestmark = pytest.mark.gpu


def test_mesh_grid_searcher_with_the_references_test_sequence():
    model = S.make_model("smpl", nv=690)
    _, sv, sf = S.make_scan_problem(model, 2)
    grid = MeshGridSearcher(sv, sf)                                    # test_mesh_grid.py:16
    # the attributes set_mesh leaves behind (mesh_grid_searcher.py:61-79), against the float32 restatement of those lines
    step, dims, origin = MO.grid_params(sv)
    assert grid.num.tolist() == [int(x) for x in dims] + [int(np.prod(dims))] and float(grid.step) == pytest.approx(step, rel=1e-6)
    np.testing.assert_allclose(grid.minmax, np.concatenate([origin, sv.max(0)]).astype(np.float32), atol=1e-6)
    assert grid.tri_num.shape == (int(np.prod(dims)),) and grid.tri_num[-1] == len(grid.tri_idx)
    lo, hi = sv.min(0), sv.max(0)
    pts = (np.random.default_rng(5).random((400, 3)) * (hi - lo) + lo).astype(np.float32)      # :18-21
    near, face_ids = grid.nearest_points(pts)                         # :23
    inside = grid.inside_mesh(pts)                                    # :24
    assert near.dtype == np.float32 and near.shape == (400, 3) and face_ids.dtype == np.int32 and inside.dtype == np.float32
    sdf = np.linalg.norm(near - pts, axis=1) * inside                 # :27
    ids_ref, near_ref, _ = MO.nearest_bruteforce(sv, sf, pts)       # the reference rule over ALL faces, float64
    np.testing.assert_allclose(np.abs(sdf), np.linalg.norm(near_ref - pts, axis=1), atol=2e-5)
    # ... and the reference's own float32 arithmetic (oracle/nearest_ref.c) on the grid the device built: where the faces agree
    # (everywhere but on exact ties along shared edges / corners) the points are the same bits
    from oracle import nearest_ref as NR
    f_o, p_o, _, _ = NR.search_nearest(grid.verts, grid.faces, pts, (float(grid.step), grid.num[:3], grid.minmax[:3], grid.tri_num, grid.tri_idx))
    same = face_ids == f_o
    assert same.mean() > 0.8 and (near[same].view(np.uint32) == p_o[same].view(np.uint32)).all()
    assert np.abs(near[~same] - p_o[~same]).max(initial=0.0) < 1e-6
    want_inside = MO.inside_mesh(sv, sf, pts, grid.step, grid.minmax[:3], grid.num[:3], grid.tri_num, grid.tri_idx)
    np.testing.assert_array_equal(inside, want_inside)
    assert set(np.unique(inside)) == {-1.0, 1.0}
    hit = grid.intersects_any(pts[:50], np.tile(np.float32([0, 0, 1]), (50, 1)))
    np.testing.assert_array_equal(hit, MO.intersects_any(sv, sf, pts[:50], np.tile(np.float32([0, 0, 1]), (50, 1))))
    # a second mesh on the same instance (set_mesh again), and the error before any mesh
    grid.set_mesh(sv * 2.0, sf)
    near2, _ = grid.nearest_points(pts * 2.0)
    np.testing.assert_allclose(near2, near * 2.0, atol=1e-4)
    grid.close()
    with pytest.raises(RuntimeError, match="set_mesh"):
        MeshGridSearcher().nearest_points(pts)


def test_cfg2_fit_through_the_official_model_files(tmp_path, monkeypatch):
    """write the synthetic SMPL the goldens were made with as data/smpl/SMPL_MALE.pkl + data/J_regressor_extra.npy, let the product
    load it the way smplify.py:50-56 does, and fit BASELINE config 2's frame 0: parameters after 1 / 10 / 100 iterations against
    the imported reference's golden at 1e-4.  Also: which fit-kernel instance such a model takes."""
    model = S.make_model("smpl")
    _, vids = S.write_official_files(model, str(tmp_path / "data"), "male")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(assets, "_MODELS", {})
    loaded = model_files.load("smpl", "male", "data", vertex_ids=vids)
    dev = N.DeviceModel(loaded, S.make_gmm())
    print("fit-kernel instance of a <= 4-bone, 24-joint model loaded from SMPL_MALE.pkl:", dev.fit_instance)
    assert dev.fit_instance == "sized"
    g = load_golden("cfg2_48view_100it_f0.npz")
    prob = S.make_problem(model, frame=0, n_views=48)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, 48)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    done = 0
    for k in (1, 10, 100):
        b.fit(k - done)
        done = k
        got = N.split_params(b.get_params()[0])
        for n in ("global_transl", "scale", "pose", "betas", "global_orient"):
            np.testing.assert_allclose(got[n], g[f"it{k}_{n}"], rtol=0, atol=1e-4, err_msg=f"it{k} {n}")
    b.close()
    dev.close()
    # a model with more than four bones on a loss selector vertex takes the table-driven instance
    wide = dict(loaded)
    w = loaded["lbs_weights"].copy()
    v = int(loaded["selector_ids"][0])
    w[v] = 0.0
    w[v, [12, 15, 9, 13, 14]] = 0.2
    wide["lbs_weights"] = w
    dev2 = N.DeviceModel(wide, S.make_gmm())
    assert dev2.fit_instance == "table-driven"
    dev2.close()
