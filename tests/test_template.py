"""The synthetic body is a SURFACE: one closed, consistently oriented genus-0 2-manifold of about a human's area with near-uniform
triangles (tools/make_template.py wrote bodyfitting_amd/data/template_*.npz).  The reference's own templates have the same
counts (smpl_uv/smpl_uv.obj: 6890 v / 13776 f; smplx_uv.obj: 10475 v / 20908 f - open at the eyes and mouth).  These statistics
are what the closest-point search, the silhouette loss and the SMPL+D stage are sensitive to."""
import numpy as np
import pytest

from bodyfitting_amd import synthetic as S


def stats(verts, faces):
    nv = len(verts)
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]).astype(np.int64)
    directed = e[:, 0] * nv + e[:, 1]
    es = np.sort(e, 1)
    uniq, counts = np.unique(es[:, 0] * nv + es[:, 1], return_counts=True)
    elen = np.linalg.norm(verts[uniq // nv] - verts[uniq % nv], axis=1)
    n = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
    area2 = np.linalg.norm(n, axis=1)
    e2 = sum(((verts[faces[:, (k + 1) % 3]] - verts[faces[:, k]]) ** 2).sum(1) for k in range(3))
    return {"closed": bool((counts == 2).all()), "oriented": len(np.unique(directed)) == len(directed),
            "euler": nv - len(uniq) + len(faces), "area": 0.5 * area2.sum(), "volume": (verts[faces[:, 0]] * n).sum() / 6.0,
            "edge": elen, "quality": 2.0 * np.sqrt(3.0) * area2 / e2}


@pytest.mark.parametrize("kind,nv,median_cm,p95_cm", [("smpl", 6890, (1.5, 2.0), 2.6), ("smplx", 10475, (1.15, 1.6), 2.2),
                                                      ("smpl", 690, (4.5, 6.0), 8.0), ("smplx", 1200, (3.0, 4.8), 6.5)])
def test_template_is_a_closed_genus0_surface_of_human_size(kind, nv, median_cm, p95_cm):
    verts, faces = S.load_template(kind, nv)
    assert verts.shape == (nv, 3) and faces.shape == (2 * nv - 4, 3)          # V - E + F = 2 with E = 3F/2
    assert faces.min() == 0 and faces.max() == nv - 1 and len(np.unique(faces)) == nv
    st = stats(verts, faces)
    assert st["closed"] and st["oriented"] and st["euler"] == 2 and st["volume"] > 0.02      # outward normals
    assert 1.5 < st["area"] < 2.1                                  # a human is ~1.8 m^2 (the tubes of rounds 1-2: 9.9 m^2)
    assert 1.7 < verts[:, 1].max() - verts[:, 1].min() < 1.85
    med, p5, p95 = np.median(st["edge"]) * 100, np.quantile(st["edge"], 0.05) * 100, np.quantile(st["edge"], 0.95) * 100
    assert median_cm[0] < med < median_cm[1], med
    assert p95 < p95_cm and p5 > 0.3 * med, (p5, p95)             # near-uniform: no tail of needle edges
    assert st["quality"].min() > 0.1, st["quality"].min()          # no sliver (1 = equilateral)


def test_every_fourth_vertex_samples_the_body_evenly():
    """the silhouette loss uses vertices [::4] (loss.py:99): with the template's Morton order that sample covers the surface"""
    verts, _ = S.load_template("smplx", 10475)
    sample = verts[::4]
    # every vertex has a sampled vertex within a few edge lengths
    d = np.sqrt(((verts[:, None, :] - sample[None, ::1, :]) ** 2).sum(-1).min(1)) if len(verts) < 3000 else None
    from scipy.spatial import cKDTree
    d = cKDTree(sample).query(verts)[0]
    assert d.max() < 0.05 and np.median(d) < 0.015


def test_model_weights_are_local_and_sparse():
    m = S.make_model("smplx", seed=0)
    w = m["lbs_weights"]
    assert ((w > 0).sum(1) <= 4).all() and np.allclose(w.sum(1), 1.0, atol=1e-6)
    # a finger joint moves finger vertices only: its weights live within a few centimetres of the joint
    rest, _ = S._smplx_rest()
    for j in (27, 39, 42, 54):
        v = m["v_template"][w[:, j] > 0.05]
        assert len(v) >= 3 and np.linalg.norm(v - rest[j], axis=1).max() < 0.06
    # and the inner thigh belongs to its own leg
    vt = m["v_template"]
    inner_left = (vt[:, 0] > 0.0) & (vt[:, 0] < 0.05) & (vt[:, 1] < -0.33) & (vt[:, 1] > -0.45)       # (below the crotch: the legs are apart)
    assert inner_left.sum() > 5 and (w[inner_left][:, [2, 5]].sum(1) < 0.25).all()
