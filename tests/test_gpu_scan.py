"""HIP scan path (closest-point grid search, point-cloud loss, dense reverse pass) through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden
from bodyfitting_amd import native as N
from bodyfitting_amd import synthetic as S
from oracle import mesh_oracle as MO

pytestmark = pytest.mark.gpu
PARAMS = ("global_transl", "scale", "pose", "betas", "global_orient")
THIN_TOL = 2e-2     # closest point on 400 : 1 needles in float32 (see test_nearest_on_adversarial_triangle_soup): 6.9e-3 .. 1.03e-2 observed


@pytest.fixture(scope="module")
def small():
    model = S.make_model("smpl", seed=0, nv=690)
    dev = N.DeviceModel(model, S.make_gmm(seed=0), device=0)
    yield model, dev
    dev.close()


def test_grid_matches_set_mesh(small):
    """cells / origin / step of MeshGridSearcher.set_mesh (utils/mesh_grid_searcher.py:63-71)"""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 0)
    scan = N.Scan(sv, sf)
    dims, origin, step = scan.grid_info()
    step_o, l_o, org_o = MO.grid_params(sv)
    assert step == pytest.approx(step_o, rel=1e-6)
    np.testing.assert_array_equal(dims, l_o)
    np.testing.assert_allclose(origin, org_o, atol=1e-6)
    assert scan.height == pytest.approx(float(sv[:, 1].max() - sv[:, 1].min()), rel=1e-6)
    scan.close()


@pytest.mark.parametrize("big", [False, True])
def test_device_grid_lists_match_insert_grid_surface(small, big):
    """the grid is built by HIP kernels (count, prefix sum, fill, sort + pack): its cell lists are exactly those of
    insert_grid_surface (mesh_grid_kernel.cu:110-157), and the same bytes on every build"""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 2)
    if big:                                                   # ~21k triangles, boundary cells on every side
        sv, sf = S.subdivide_mesh(sv, sf, 2)
        sv = sv.astype(np.float32)
    scan = N.Scan(sv, sf)
    dims, origin, step = scan.grid_info()
    tri_num, tri_idx = scan.grid_lists()
    tn_o, ti_o = MO.insert_grid_surface(sv, sf, step, origin, dims)
    np.testing.assert_array_equal(tri_num, tn_o)
    np.testing.assert_array_equal(tri_idx, ti_o)
    again = N.Scan(sv, sf)
    tn2, ti2 = again.grid_lists()
    np.testing.assert_array_equal(tri_idx, ti2)
    np.testing.assert_array_equal(tri_num, tn2)
    again.close(); scan.close()


def test_inside_mesh_matches_the_restated_rule(small):
    """MeshGridSearcher.inside_mesh on the device vs oracle/mesh_oracle.inside_mesh (same float32 tests, same cell walk)"""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 3)
    scan = N.Scan(sv, sf)
    dims, origin, step = scan.grid_info()
    tri_num, tri_idx = scan.grid_lists()
    rng = np.random.default_rng(9)
    q = np.concatenate([sv[rng.integers(0, len(sv), 150)] + rng.normal(0, 0.03, (150, 3)),
                        rng.uniform(sv.min(0) - 0.1, sv.max(0) + 0.1, (100, 3))]).astype(np.float32)
    got = scan.inside_mesh(q)
    want = MO.inside_mesh(sv, sf, q, step, origin, dims, tri_num, tri_idx)
    np.testing.assert_array_equal(got, want)
    assert (got > 0).sum() > 20 and (got < 0).sum() > 20
    scan.close()


def test_intersects_any_matches_the_bruteforce_rule(small):
    """MeshGridSearcher.intersects_any: the 3-D DDA over the grid finds a hit exactly when some triangle passes the
    reference's test (oracle: brute force over all triangles); rays from inside, outside and off the grid"""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 4)
    scan = N.Scan(sv, sf)
    rng = np.random.default_rng(11)
    lo, hi = sv.min(0), sv.max(0)
    o = np.concatenate([rng.uniform(lo - 0.3, hi + 0.3, (200, 3)), sv[rng.integers(0, len(sv), 100)] + rng.normal(0, 0.05, (100, 3)),
                        rng.uniform(lo - 3, lo - 1, (60, 3))]).astype(np.float32)
    d = rng.normal(size=o.shape).astype(np.float32)
    d[-30:] = sv[sf[rng.integers(0, len(sf), 30)]].mean(1) - o[-30:]        # far rays aimed at the middle of a triangle
    got = scan.intersects_any(o, d)
    want = MO.intersects_any(sv, sf, o, d)
    print("intersects_any: rays that differ from the brute force", int((got != want).sum()), "of", len(got))
    assert np.mean(got == want) > 0.995                                  # (a grazing ray may differ by a float32 rounding in the walk)
    assert want[-30:].all() and got[-30:].all() and 0.2 < want.mean() < 0.95
    assert not scan.intersects_any(o[:4], np.zeros((4, 3), np.float32)).any()
    scan.close()


def test_intersects_any_coplanar_and_degenerate_cases():
    """intersect_tri2's branches for rays in a triangle's plane and for degenerate triangles (mesh_grid_kernel.cu:781-1023), on integer
    coordinates so that every decision against 1e-9 is exact: the HIP walk agrees with the oracle's brute force on every ray"""
    v = np.array([[0, 0, 0], [4, 0, 0], [0, 4, 0],                       # a triangle in the plane z = 0
                  [0, 10, 0], [2, 10, 0], [4, 10, 0],                    # a triangle that is a segment
                  [8, 8, 3], [8, 8, 3], [8, 8, 3],                       # a triangle that is a point
                  [-6, -6, -5], [12, -6, -5], [-6, 14, -5], [0, 0, 9]], np.float32)   # a tetrahedron around them (a 3-D bounding box)
    f = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11], [9, 10, 12], [10, 11, 12], [11, 9, 12]], np.int32)
    rng = np.random.default_rng(3)
    n = 900
    o = np.stack([rng.integers(-5, 12, n), rng.integers(-5, 13, n), rng.choice([0, 0, 0, 1, 3], n)], 1).astype(np.float32)
    d = np.stack([rng.integers(-3, 4, n), rng.integers(-3, 4, n), rng.choice([0, 0, 0, 1, -1], n)], 1).astype(np.float32)
    d[:40] = np.array([8, 8, 3], np.float32) - o[:40]                    # aimed at the point triangle
    o[40:80, 1] = 10; d[40:80, 1] = 0; d[40:80, 2] = 0; d[40:80, 0] = rng.choice([-1, 1], 40)      # along the segment's line
    scan = N.Scan(v, f)
    got = scan.intersects_any(o, d)
    want = MO.intersects_any(v, f, o, d)
    np.testing.assert_array_equal(got, want)
    only_first = MO.intersects_any(v, f[:3], o, d)                       # hits that exist only through the degenerate branches
    assert only_first[:40].all() and 30 < only_first.sum() < n - 30
    planar = (o[:, 2] == 0) & (d[:, 2] == 0)
    assert only_first[planar].any() and (~only_first[planar]).any()
    scan.close()


@pytest.mark.parametrize("spread", [0.01, 0.2, 3.0])
def test_nearest_points_match_bruteforce_rule(small, spread):
    """near, far and way-outside-the-grid queries against the oracle's brute-force search"""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 1)
    rng = np.random.default_rng(7)
    q = (sv[rng.integers(0, len(sv), 400)] + rng.normal(0, spread, (400, 3))).astype(np.float32)
    scan = N.Scan(sv, sf)
    pts, ids, bary = scan.nearest_points(q)
    ids_o, pts_o, bary_o = MO.nearest_bruteforce(sv, sf, q)
    d = np.linalg.norm(pts - q, axis=1)
    d_o = np.linalg.norm(pts_o - q, axis=1)
    if spread < 1.0:
        np.testing.assert_allclose(d, d_o, rtol=2e-5, atol=2e-6)      # same distance (ties may pick another face)
    else:
        # metres outside the grid the walk keeps the reference's shell limit (mesh_grid_kernel.cu:254-257: one layer of cells short on
        # the far side, tests/test_nearest_ref_oracle.py): a per cent of such queries answer with a face ~1 % farther than the argmin
        short = ~np.isclose(d, d_o, rtol=2e-5, atol=2e-6)
        assert short.mean() < 0.02 and np.all(d[short] > d_o[short]) and np.all(d[short] < 1.03 * d_o[short])
        ids, pts, bary, ids_o, pts_o, bary_o, d, d_o, q = (x[~short] for x in (ids, pts, bary, ids_o, pts_o, bary_o, d, d_o, q))
    # (on a closed surface the closest point of a far query is often a mesh vertex or an edge, shared by several faces: they
    #  tie in exact arithmetic and the float32 distances through different triangles decide in the last bit)
    same = ids == ids_o
    assert same.mean() > 0.8
    # fp32 offsets relative to the query; metres away the reference's own arithmetic (Gram entries ~d^2 in float32 against edges of
    # centimetres) moves a coefficient more than the float64 rule's rounding does: the point to 1e-5 of the distance there
    tol = (5e-6 if spread < 1.0 else 1e-5) * max(1.0, float(d_o.max()))
    np.testing.assert_allclose(pts[same], pts_o[same], atol=tol)
    np.testing.assert_allclose(bary[same], bary_o[same], atol=(2e-4 if spread < 1.0 else 5e-4) * max(1.0, float(d_o.max())))
    np.testing.assert_allclose(bary.sum(1), 1.0, atol=1e-5)
    # the returned point really is that barycentric combination of that face
    np.testing.assert_allclose(np.einsum("qi,qik->qk", bary, sv[sf[ids]]), pts, atol=tol)
    scan.close()


def _against_reference_arithmetic(verts, faces, queries, allow_far_quirk=False):
    """bf_scan_nearest (default rule) against oracle/nearest_ref.c - the reference's search in its own float32 arithmetic.
    -> (n queries, n exact ties, n decisions that differ); asserts points / coefficients bit for bit wherever the faces agree"""
    from oracle import nearest_ref as NR
    verts = np.ascontiguousarray(verts, np.float32); faces = np.ascontiguousarray(faces, np.int32); queries = np.ascontiguousarray(queries, np.float32)
    scan = N.Scan(verts, faces)
    pts, ids, bary = scan.nearest_points(queries)
    dims, origin, step = scan.grid_info()
    tri_num, tri_idx = scan.grid_lists()
    scan.close()
    f_o, p_o, c_o, d_o = NR.search_nearest(verts, faces, queries, (step, dims, origin, tri_num, tri_idx))
    differ = np.nonzero(ids != f_o)[0]
    ties = 0
    if len(differ):
        _, d_mine, _ = NR.rule(verts, faces, ids[differ], queries[differ])          # the reference's rule on the face the kernel chose
        tie = d_mine.view(np.uint32) == d_o[differ].view(np.uint32)
        ties = int(tie.sum())
        assert tie.all(), "the kernel picked a face the reference's arithmetic ranks strictly behind its own, %d times (gap %.3g)" % (
            (~tie).sum(), float(((d_mine - d_o[differ]) / d_o[differ])[~tie].max()))
    same = ids == f_o
    assert (pts[same].view(np.uint32) == p_o[same].view(np.uint32)).all()
    assert (bary[same].view(np.uint32) == c_o[same].view(np.uint32)).all()
    return len(queries), ties, len(differ) - ties


@pytest.mark.parametrize("spread", [0.002, 0.02, 0.3, 3.0])
def test_nearest_equals_the_references_own_arithmetic(small, spread):
    """face ids equal to the reference's float32 arithmetic (oracle/nearest_ref.c = mesh_grid_kernel.cu:12-109, 239-353 + matrix.h,
    the latter pinned bit for bit to the reference's header) except where several faces return the SAME distance bit for bit - the
    reference's own answer then hangs on the order an atomicCAS race left in its lists; points and coefficients bit for bit.
    Near, far, and metres outside the grid (where the reference's shell limit stops one layer short, kept)."""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 1)
    rng = np.random.default_rng(21)
    q = (sv[rng.integers(0, len(sv), 3000)] + rng.normal(0, spread, (3000, 3))).astype(np.float32)
    n, ties, wrong = _against_reference_arithmetic(sv, sf, q)
    print("spread %g: %d queries, %d exact ties resolved the other way, %d decisions that differ" % (spread, n, ties, wrong))
    assert wrong == 0 and ties < 0.2 * n


@pytest.mark.parametrize("scale", [1.0, 0.1, 0.01, 0.001])
def test_nearest_equals_the_references_own_arithmetic_on_slivers(scale):
    """needles, slivers and obtuse triangles with edges from a metre down to a millimetre, queries in every Voronoi region and none
    dropped for being close to a branch decision: at the small scales the reference's absolute 1e-9 rank tests fire (solve4 drops a
    pivot and answers with a point on the long edge, tests/test_nearest_ref_oracle.py) - the kernel's general routines take those
    triangles and give the reference's answer, bit for bit"""
    from oracle import adversarial as ADV
    d = ADV.soup(seed=0, min_margin=0.0)
    v = (d["verts"].astype(np.float64) * scale).astype(np.float32)
    q = (d["queries"].astype(np.float64) * scale).astype(np.float32)
    n, ties, wrong = _against_reference_arithmetic(v, d["faces"], q)
    assert wrong == 0 and ties == 0                                      # (isolated triangles: nothing ties)
    # how many of these patches the straight-line paths decline (the second kernel's share)
    lib = N._lib.load()
    tri = v[d["faces"][d["owner"]]] - q[:, None, :]
    patches = np.ascontiguousarray(tri.reshape(-1, 9), np.float32)
    dist = np.empty(len(patches), np.float32); coeff = np.empty((len(patches), 3), np.float32)
    N._lib.check(lib.bf_nearest_selftest_rule(0, len(patches), N._lib.fptr(patches), 0, N._lib.fptr(dist), N._lib.fptr(coeff)), "selftest_rule")
    declined = float((dist < 0).mean())
    print("scale %g: the regular paths decline %.1f %% of the soup's patches" % (scale, 100 * declined))
    # (at scale 1 the queries are up to three units away: Gram entries above 1 put the border row's 1 out of the first pivot's place)
    assert declined > 0.2 if scale <= 0.01 else declined < 0.3


@pytest.mark.parametrize("spread", [0.0005, 0.005, 0.02])
def test_a_guess_bounds_the_search_and_never_changes_its_answer(small, spread):
    """bf_scan_nearest_hinted (what the fit loop does with its previous iteration's nearest points): the guess prunes cells and records
    (the screen of csrc/scan_kernels.hip) and is checked against what was found - for the exact answer as the guess, guesses a
    millimetre off the surface (half of them closer to the query than the surface is: a bound that is too small), half way to the
    query, on the query itself, NaN, infinite and a metre away, the result is the unguessed search's bit for bit; and that one is the
    reference's arithmetic (the tests above).  A degenerate triangle (coincident corners after the OBJ's 4 decimals) sits in the scan."""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 1)
    sv = sv.copy(); sv[sf[5, 1]] = sv[sf[5, 0]]                              # a repeated corner: its triangles are needles / points
    rng = np.random.default_rng(33)
    q = (sv[rng.integers(0, len(sv), 4000)] + rng.normal(0, spread, (4000, 3))).astype(np.float32)
    scan = N.Scan(sv, sf)
    pts, ids, bary = scan.nearest_points(q)
    guesses = {"exact": pts, "1 mm off": pts + rng.normal(0, 0.001, pts.shape).astype(np.float32), "half way": (pts + q) / 2,
               "the query": q.copy(), "nan": np.full_like(pts, np.nan), "inf": np.full_like(pts, np.inf), "a metre off": pts + np.float32(1.0),
               "mixed": np.where(rng.random((len(q), 1)) < 0.5, pts, q)}
    for name, g in guesses.items():
        p2, i2, b2 = scan.nearest_points_hinted(q, g.astype(np.float32))
        assert np.array_equal(i2, ids), name
        assert np.array_equal(p2.view(np.uint32), pts.view(np.uint32)) and np.array_equal(b2.view(np.uint32), bary.view(np.uint32)), name
    scan.close()
    n, ties, wrong = _against_reference_arithmetic(sv, sf, q)
    assert wrong == 0


@pytest.mark.parametrize("scale", [1.0, 0.01])
def test_guesses_on_the_adversarial_soup(scale):
    """the guessed search where the reference takes rank decisions (needles, slivers, obtuse triangles; at centimetre scale the absolute
    1e-9 tests fire): exact, half-way, NaN and mixed guesses give the unguessed search's bits - which are the reference arithmetic's
    (test_nearest_equals_the_references_own_arithmetic_on_slivers)"""
    from oracle import adversarial as ADV
    d = ADV.soup(seed=0, min_margin=0.0)
    v = (d["verts"].astype(np.float64) * scale).astype(np.float32)
    q = (d["queries"].astype(np.float64) * scale).astype(np.float32)
    scan = N.Scan(v, d["faces"])
    pts, ids, bary = scan.nearest_points(q)
    rng = np.random.default_rng(2)
    for name, g in (("exact", pts), ("half way", (pts + q) / 2), ("nan", np.full_like(pts, np.nan)),
                    ("mixed", np.where(rng.random((len(q), 1)) < 0.5, pts, pts + np.float32(0.3 * scale)))):
        p2, i2, b2 = scan.nearest_points_hinted(q, g.astype(np.float32))
        assert np.array_equal(i2, ids), name
        assert np.array_equal(p2.view(np.uint32), pts.view(np.uint32)) and np.array_equal(b2.view(np.uint32), bary.view(np.uint32)), name
    scan.close()


def test_rule_on_the_device_equals_the_oracle_patch_by_patch():
    """search_nearest_proj on explicit patches: regular, degenerate (repeated corners, collinear), far (Gram entries above 1) and
    tiny ones; distance and coefficients bit for bit against oracle/nearest_ref.c"""
    from oracle import nearest_ref as NR
    rng = np.random.default_rng(8)
    tri = rng.normal(size=(60000, 3, 3))
    tri[:10000] *= 0.01; tri[10000:20000] *= 1e-3; tri[20000:30000] *= 3.0; tri[30000:40000] *= 1e-4
    tri[40000:45000, 2] = tri[40000:45000, 1]                                          # a repeated corner
    t = rng.random((5000, 1)); tri[45000:50000, 2] = tri[45000:50000, 0] * t + tri[45000:50000, 1] * (1 - t)   # collinear
    tri[50000:55000] = tri[50000:55000] * 1e-2 + rng.normal(size=(5000, 1, 3)) * 2.0    # small triangle, far query
    tri[55000:] = np.round(tri[55000:] * 4) / 4                                       # dyadic: exact ties inside the rule
    q = np.zeros((len(tri), 3), np.float32)
    verts = tri.reshape(-1, 3).astype(np.float32)
    faces = np.arange(len(verts), dtype=np.int32).reshape(-1, 3)
    c_o, d_o, path = NR.rule(verts, faces, np.arange(len(tri)), q)
    lib = N._lib.load()
    patches = np.ascontiguousarray(verts.reshape(-1, 9))
    dist = np.empty(len(patches), np.float32); coeff = np.empty((len(patches), 3), np.float32)
    N._lib.check(lib.bf_nearest_selftest_rule(0, len(patches), N._lib.fptr(patches), 1, N._lib.fptr(dist), N._lib.fptr(coeff)), "selftest_rule")
    ok = ((dist.view(np.uint32) == d_o.view(np.uint32)) | (np.isnan(dist) & np.isnan(d_o))) & \
         ((coeff.view(np.uint32) == c_o.view(np.uint32)) | (np.isnan(coeff) & np.isnan(c_o))).all(1)
    assert ok.all(), "device rule differs from the oracle on %d patches (branches %s)" % ((~ok).sum(), np.unique(path[~ok]))
    assert len(np.unique(path)) >= 5                                                    # face, both edge kinds, clamped, the (.5, .5) fallback
    N._lib.check(lib.bf_nearest_selftest_rule(0, len(patches), N._lib.fptr(patches), 0, N._lib.fptr(dist), N._lib.fptr(coeff)), "selftest_rule")
    reg = dist >= 0
    assert 0.3 < reg.mean() < 0.9 and (dist[reg].view(np.uint32) == d_o[reg].view(np.uint32)).all()


def test_division_helper_is_ieee_division():
    """nearest_rule_ref.h's quot / recip (reciprocal refined once, quotient twice, no range scaling) against numpy's correctly rounded
    float32 division over the range the regular paths use it in: 1e-8 < |d| < 4, quotients up to 2^90, plus zeros and exact cases"""
    rng = np.random.default_rng(4)
    n = 1 << 24
    lib = N._lib.load()
    bad = 0
    for rep in range(4):
        den = (10.0 ** rng.uniform(-8, 0.6, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
        num = (10.0 ** rng.uniform(-20, 19, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
        num[::97] = 0.0
        num[1::101] = (den[1::101].astype(np.float64) * rng.integers(1, 1 << 12, len(den[1::101]))).astype(np.float32)      # near-exact quotients
        num[2::103] = np.nextafter(num[2::103], np.float32(0))
        keep = np.abs(num.astype(np.float64) / den) < 2.0 ** 90
        num, den = np.ascontiguousarray(num[keep]), np.ascontiguousarray(den[keep])
        out = np.empty(len(num), np.float32)
        N._lib.check(lib.bf_nearest_selftest_quot(0, len(num), N._lib.fptr(num), N._lib.fptr(den), N._lib.fptr(out)), "selftest_quot")
        bad += int((out.view(np.uint32) != (num / den).view(np.uint32)).sum())
    assert bad == 0


def test_fast_rule_is_still_there_and_close(small):
    """BF_NEAREST_FAST: the 2 x 2 normal equations with v_rcp_f32 - same distances to float32 noise, another face on a few per cent of
    shared edges (DESIGN 2.3 has the rates at config 5's size)"""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 1)
    rng = np.random.default_rng(22)
    q = (sv[rng.integers(0, len(sv), 3000)] + rng.normal(0, 0.02, (3000, 3))).astype(np.float32)
    scan = N.Scan(sv, sf)
    ref = scan.nearest_points(q)
    before = N.set_nearest_rule("fast")
    try:
        fast = scan.nearest_points(q)
    finally:
        N.set_nearest_rule(before)
    assert before == "reference"
    d_r, d_f = np.linalg.norm(ref[0] - q, axis=1), np.linalg.norm(fast[0] - q, axis=1)
    np.testing.assert_allclose(d_f, d_r, rtol=2e-5, atol=2e-6)
    assert 0.8 < np.mean(ref[1] == fast[1]) < 1.0
    again = scan.nearest_points(q)
    np.testing.assert_array_equal(again[1], ref[1])
    scan.close()


def test_nearest_on_adversarial_triangle_soup():
    """needle / sliver / obtuse triangles, queries in every Voronoi region (oracle/adversarial.py): the device search returns
    the face the restated rule picks - bit for bit, the soup has no ties - and its point, including where the rule's edge
    fallback (mesh_grid_kernel.cu:74-101) is NOT the exact closest point"""
    from oracle import adversarial as ADV
    d = ADV.soup(seed=0)
    scan = N.Scan(d["verts"], d["faces"])
    pts, ids, bary = scan.nearest_points(d["queries"])
    ids_o, pts_o, bary_o = MO.nearest_bruteforce(d["verts"], d["faces"], d["queries"])
    np.testing.assert_array_equal(ids_o, d["owner"])
    np.testing.assert_array_equal(ids, ids_o)
    # float32 vs the float64 rule: one ulp of a coordinate (1e-6 at |x| ~ 12) for well-shaped triangles; for the needles the
    # in-plane solve loses digits in proportion to the aspect ratio (400 : 1 here) whatever the formulation
    kinds = np.array(d["kind"])[d["owner"]]
    thin = np.isin(kinds, ("needle", "sliver", "obtuse175"))
    err = np.abs(pts - pts_o).max(1)
    print("adversarial soup: max |point - float64 rule| well-shaped %.2e, thin %.2e" % (err[~thin].max(), err[thin].max()))
    assert err[~thin].max() <= 2e-6 and err[thin].max() <= THIN_TOL
    np.testing.assert_allclose(np.einsum("qi,qik->qk", bary, d["verts"][d["faces"][ids]]), pts, atol=2e-6)
    assert bary.min() >= 0 and np.allclose(bary.sum(1), 1.0, atol=1e-5)
    d_rule, d_exact = ADV.rule_vs_exact(d)
    inexact = d_rule > d_exact * (1 + 1e-6) + 1e-12
    got = ((pts.astype(np.float64) - d["queries"]) ** 2).sum(1)
    np.testing.assert_allclose(got[inexact], d_rule[inexact], rtol=2e-5)          # the reference's answer, not the exact one
    assert inexact.sum() > 30 and np.all(got[inexact] > d_exact[inexact] * 1.0001)
    np.testing.assert_array_equal(bary[inexact].argmax(1), 2)                       # ... namely the obtuse corner
    scan.close()


def test_nearest_at_config5_size_against_bruteforce():
    """BASELINE config 5 at full size: the 10,475 vertices of an SMPL-X-shaped body against an 83,784-triangle scan, as the
    fit issues them.  Size-independent properties on every query (the point is that barycentric combination of that face;
    no query is farther from its answer than from the nearest scan VERTEX), and brute force over all faces on a
    512-query sample (same face except exact ties, same distance)."""
    model = S.make_model("smplx", seed=0)
    prob, sv, sf = S.make_scan_problem_smplx(model, 0, n_views=4, subdivide=1)
    assert len(sf) == 4 * (2 * 10475 - 4)            # the closed template once subdivided: 83,784 triangles
    rng = np.random.default_rng(5)
    verts = model["v_template"].astype(np.float64)
    # the body somewhere near the scan surface, as during the fit: scan vertices are the posed body + noise
    q = (sv[rng.integers(0, len(sv), 10475)] + rng.normal(0, 0.01, (10475, 3))).astype(np.float32)
    q[:2000] = (verts[:2000] * (sv[:, 1].max() - sv[:, 1].min()) / (verts[:, 1].max() - verts[:, 1].min())).astype(np.float32)   # and far ones
    scan = N.Scan(sv, sf)
    pts, ids, bary = scan.nearest_points(q)
    assert ids.min() >= 0 and ids.max() < len(sf)
    np.testing.assert_allclose(np.einsum("qi,qik->qk", bary, sv[sf[ids]]), pts, atol=5e-6)
    assert bary.min() >= 0 and np.allclose(bary.sum(1), 1.0, atol=1e-5)
    from scipy.spatial import cKDTree
    dv, _ = cKDTree(sv).query(q)
    d = np.linalg.norm(pts - q, axis=1)
    # the nearest scan vertex V bounds the answer: its triangles are candidates, and the rule answers with a point OF the
    # triangle (not always the closest one: obtuse triangles, oracle/adversarial.py), so d <= |q - V| + the triangle's diameter;
    # for all but a few queries the plain bound d <= |q - V| holds
    tri = sv[sf]
    longest = float(np.sqrt(max(((tri[:, i] - tri[:, (i + 1) % 3]) ** 2).sum(1).max() for i in range(3))))
    assert np.all(d <= dv + longest + 1e-6)
    assert np.mean(d <= dv * (1 + 1e-5) + 1e-6) > 0.99
    sample = rng.choice(len(q), 512, replace=False)
    ids_o, pts_o, _ = MO.nearest_bruteforce(sv, sf, q[sample], chunk=16)
    d_o = np.linalg.norm(pts_o - q[sample], axis=1)
    np.testing.assert_allclose(d[sample], d_o, rtol=2e-5, atol=2e-6)
    assert np.mean(ids[sample] == ids_o) > 0.8                      # (a closest point on a shared edge / vertex ties between the faces around it)
    again = scan.nearest_points(q)
    np.testing.assert_array_equal(again[1], ids)                    # deterministic
    # the same queries the way an iteration of the fit loop searches them: with the answers of the iteration before as guesses (here:
    # the answers for queries 0.5 mm / 5 mm away) - the guess only bounds the search, the result is the unguessed one, bit for bit
    for moved in (0.0005, 0.005):
        before = scan.nearest_points((q + rng.normal(0, moved, q.shape)).astype(np.float32))[0]
        p2, i2, b2 = scan.nearest_points_hinted(q, before)
        np.testing.assert_array_equal(i2, ids)
        assert np.array_equal(p2.view(np.uint32), pts.view(np.uint32)) and np.array_equal(b2.view(np.uint32), bary.view(np.uint32))
    scan.close()
    # and EVERY query against the reference's own float32 arithmetic (the grid walk of oracle/nearest_ref.c): faces equal outside
    # exact ties, points and coefficients bit for bit
    n, ties, wrong = _against_reference_arithmetic(sv, sf, q)
    print("config 5 size: %d queries, %d exact ties resolved the other way, %d decisions that differ" % (n, ties, wrong))
    assert wrong == 0 and ties < 0.1 * n


def test_nearest_points_backward_matches_autograd(small):
    """SurfaceNearest.backward w.r.t. the query points (unfinished in the reference): point-to-plane on faces, along the edge on
    edges, zero at corners - against torch.autograd of the same closed forms, and against central differences of the search"""
    model, _ = small
    _, sv, sf = S.make_scan_problem(model, 1)
    rng = np.random.default_rng(3)
    q = (sv[rng.integers(0, len(sv), 600)] + rng.normal(0, 0.03, (600, 3))).astype(np.float32)
    scan = N.Scan(sv, sf)
    pts, ids, bary = scan.nearest_points(q)
    g = rng.normal(size=q.shape).astype(np.float32)
    got = scan.nearest_points_backward(ids, bary, g)
    want = MO.nearest_backward(sv, sf, q, ids, bary, g)
    np.testing.assert_allclose(got, want, atol=2e-5 * np.abs(want).max())
    zeros = (bary == 0).sum(1)
    assert (zeros == 0).sum() > 50 and (zeros == 1).sum() > 50 and (zeros == 2).sum() > 5         # faces, edges and corners all occur
    assert np.all(got[zeros == 2] == 0)
    # the search itself, differenced: d(pts . g)/dq along a random direction, where the region does not change under the step
    h, v = 1e-3, rng.normal(size=q.shape)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    pp, ip, bp = scan.nearest_points((q + h * v).astype(np.float32))
    pm, im, bm = scan.nearest_points((q - h * v).astype(np.float32))
    same = (ip == ids) & (im == ids) & np.all((bp == 0) == (bary == 0), 1) & np.all((bm == 0) == (bary == 0), 1)
    fd = ((pp - pm).astype(np.float64) * g).sum(1) / (2 * h)
    an = (got.astype(np.float64) * v).sum(1)
    assert same.sum() > 200
    np.testing.assert_allclose(an[same], fd[same], atol=2e-2 * np.abs(fd[same]).max())
    scan.close()


def test_scan_fit_matches_reference_golden(small):
    """smplify.py loop with use_mesh=True: 11 keypoint-only iterations, then 19 with the point-cloud
    loss, against the imported reference (stand-in searcher)."""
    model, dev = small
    g = load_golden("scan_nv690_30it.npz")
    prob, sv, sf = S.make_scan_problem(model, frame=0, n_views=8)
    scan = N.Scan(sv, sf)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, 8)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    b.set_scans([scan])
    b.fit(30)            # the switch-on iteration is num_iters // 3 of THIS call, as in the reference
    got = N.split_params(b.get_params()[0])
    for n in PARAMS:
        np.testing.assert_allclose(got[n], g[f"it30_{n}"], rtol=0, atol=1e-4, err_msg=n)
    verts, joints, _, _ = b.get_result()
    np.testing.assert_allclose(verts[0], g["vertices"], atol=1e-4)
    np.testing.assert_allclose(joints[0], g["joints"], atol=1e-4)
    b.close()
    scan.close()


def _disp_metrics(model, sv, sf, base, disp):
    """end-state metrics of the SMPL+D stage (smplify.py:228-247) for `base + disp` against the scan, by the oracle's
    pieces: distribution of point-to-scan distances, the icp term, normal and laplacian energies"""
    import torch
    P = (base + disp).astype(np.float32)
    ids, cp, _ = MO.ReferenceSearcher(sv, sf).nearest(P)
    d = np.linalg.norm(P - cp, axis=1)
    faces_t = torch.as_tensor(np.asarray(model["faces"]), dtype=torch.long)
    norms = MO.compute_normal_torch(torch.tensor(P, dtype=torch.float64), faces_t)
    tris = sv.astype(np.float64)[sf]
    fn = torch.tensor(np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]).astype(np.float32), dtype=torch.float64)
    return {"mean": float(d.mean()), "median": float(np.median(d)), "p95": float(np.percentile(d, 95)), "icp": float(np.linalg.norm(P - cp)),
            "normal": float(MO.normal_loss(fn[torch.as_tensor(ids, dtype=torch.long)], norms)),
            "laplacian": float(MO.normal_laplacian_smoothness(norms, faces_t))}


def test_config5_iteration_counts_end_state_against_the_reference(small):
    """config 5's loop lengths on the reduced model: 300 iterations with the scan loss after 100, then 300 SMPL+D iterations,
    against the imported reference's run (scan_nv690_300it.npz).  First loop: parameters after 100 / 101 / 300 iterations.
    SMPL+D is chaotic under round-off (Adam's normalised 5 cm steps; fp32 and fp64 of one code diverge within 30 steps), so
    its 300 steps are held to the reference by their END STATE: the distribution of point-to-scan distances and the icp /
    normal / laplacian terms of the displaced mesh within a stated fraction of the reference's, and clearly better than the
    undisplaced mesh."""
    import ref_drift as RD
    model, dev = small
    g = load_golden("scan_nv690_300it.npz")
    # the reference's own drift over the same 300 + 300 iterations (8 threads instead of 1; initial pose moved by one ulp; the
    # closest-point search built with fused multiply-adds)
    sens = load_golden("sens_scan_nv690_300it.npz")
    band = {k: RD.band(g, sens, [f"it{k}_{n}" for n in PARAMS], variants=RD.SCAN_VARIANTS) for k in (100, 101, 300)}
    print("bands = K x the reference's largest drift under ten perturbations + the fused search (tests/ref_drift.py):", band)

    def rel_band(metric_of, ref_value, floor=0.05):
        """K x how far the reference's perturbed runs end from the reference in this metric (relative), at least `floor`"""
        return max(floor, RD.K * max(abs(metric_of(v) - ref_value) for v in RD.SCAN_VARIANTS) / abs(ref_value))
    prob, sv, sf = S.make_scan_problem(model, frame=0, n_views=8)
    scan = N.Scan(sv, sf)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, 8)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans([scan])
    done = 0
    for k in (100, 101, 300):
        b.fit(k - done, N.make_hyper(dense_after=100))               # 300 // 3: one reference loop cut at the snapshots
        done = k
        got = N.split_params(b.get_params()[0])
        drift = max(float(np.abs(got[n] - g[f"it{k}_{n}"]).max()) for n in PARAMS)
        print(f"scan loop, {k} iterations: max |param - reference| = {drift:.2e}")
        # the keypoint-only third and the first scan iteration hold the north-star tolerance (observed 7e-7); 199 more
        # iterations of the closest-point loss (the closest face of a vertex changes discretely, the reference's rule is not
        # even continuous on obtuse triangles) accumulate to 5e-3
        assert drift < band[k], (k, drift, band[k])
    assert band[100] < 2e-4 and band[101] < 1e-3                   # (the keypoint-only third is well conditioned in the reference too)
    verts, joints, _, _ = b.get_result()
    np.testing.assert_allclose(verts[0], g["vertices"], atol=RD.band(g, sens, ["vertices"], variants=RD.SCAN_VARIANTS))
    np.testing.assert_allclose(joints[0], g["joints"], atol=RD.band(g, sens, ["joints"], variants=RD.SCAN_VARIANTS))
    fit_ref = _disp_metrics(model, sv, sf, g["vertices"], 0 * g["vertices"])
    fit_got = _disp_metrics(model, sv, sf, verts[0], 0 * verts[0])
    print("scan loop end state  reference:", fit_ref, "\n                     HIP:      ", fit_got)
    # the closest-point term is ~10 % of the objective (keypoint terms ~2,400, 5 * imsize / height * icp ~270), so the distance
    # distribution is a soft quantity of the end state (held within a factor of two: rebuilds of the kernels that only changed an
    # fma contraction moved the mean between +16 % and +30 % of the reference's); the objective itself is held within 5 %
    fit_var = {v: _disp_metrics(model, sv, sf, sens[f"{v}_vertices"], 0 * sens[f"{v}_vertices"]) for v in RD.SCAN_VARIANTS}
    print("                     reference, perturbed:", fit_var)
    for key in ("mean", "median", "p95", "icp"):
        tol = rel_band(lambda v: fit_var[v][key], fit_ref[key])
        assert abs(fit_got[key] - fit_ref[key]) / fit_ref[key] < tol, (key, fit_got[key], fit_ref[key], tol)
    w_pc = 5.0 * 512.0 / float(sv[:, 1].max() - sv[:, 1].min())
    params_got = b.get_params()
    obj = {}
    cases = [("reference", N.pack_params({n: g[f"it300_{n}"] for n in PARAMS})[None], fit_ref["icp"]), ("HIP", params_got, fit_got["icp"])]
    cases += [(v, N.pack_params({n: sens[f"{v}_it300_{n}"] for n in PARAMS})[None], fit_var[v]["icp"]) for v in RD.SCAN_VARIANTS]
    for name, pk, icp in cases:
        probe = N.FrameBatch(dev, 1, 8)
        probe.set_cameras(c2w, K); probe.set_keypoints(kp, ndiv); probe.set_init(betas, pose); probe.set_scans([scan])   # (constant scale = height / 1.7)
        probe.set_params(pk)
        obj[name] = float(probe.loss_grad()[0].sum()) + w_pc * icp
        probe.close()
    print("objective after 300 iterations:", obj)
    # (Adam at lr 1e-2 does not settle: the reference's own objective moves between 2668 and 2768 over its last 12 iterations -
    #  oracle trace of the same loop - so two end states are "the same" within that band)
    assert obj["HIP"] == pytest.approx(obj["reference"], rel=rel_band(lambda v: obj[v], obj["reference"], floor=0.02))
    b.fit_displacement(300)
    disp = b.get_displacement()[0]
    want = _disp_metrics(model, sv, sf, g["vertices"], g["displacement"])
    before = _disp_metrics(model, sv, sf, g["vertices"], 0 * g["displacement"])
    got = _disp_metrics(model, sv, sf, verts[0], disp)
    print("SMPL+D end state  reference:", want, "\n                  HIP:      ", got, "\n                  before:   ", before)
    disp_var = {v: _disp_metrics(model, sv, sf, sens[f"{v}_vertices"], sens[f"{v}_displacement"]) for v in RD.SCAN_VARIANTS}
    print("                  reference, perturbed:", disp_var)
    for key in ("mean", "median", "p95", "icp", "laplacian", "normal"):
        tol = rel_band(lambda v: disp_var[v][key], want[key], floor=0.1)
        assert abs(got[key] - want[key]) / abs(want[key]) < tol, (key, got[key], want[key], tol)
    # (the stage is chaotic: eight HIP runs of it - both closest-point rules, the initial pose nudged by 0..3 ulp, tools/diag_disp.py -
    #  end with means of 2.08 .. 3.06 mm, icp 0.10 .. 0.21 and one to three vertices 31 .. 146 mm off; the median is the robust figure)
    assert got["mean"] < 0.75 * before["mean"] and got["median"] < 0.4 * before["median"]
    assert np.abs(disp).max() < 2 * np.abs(g["displacement"]).max()
    b.close()
    scan.close()


def test_two_frames_two_scans(small):
    """per-frame scans and per-frame constant scale inside one batch == each frame alone"""
    model, dev = small
    items = [S.make_scan_problem(model, frame=f, n_views=8, scan_scale=sc) for f, sc in ((0, 1.0), (1, 0.5))]
    scans = [N.Scan(sv, sf) for _, sv, sf in items]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([p for p, _, _ in items])
    b = N.FrameBatch(dev, 2, 8)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans(scans)
    b.fit(15)
    both = b.get_params()
    b.close()
    for i in range(2):
        b1 = N.FrameBatch(dev, 1, 8)
        b1.set_cameras(c2w[i:i + 1], K[i:i + 1]); b1.set_keypoints(kp[i:i + 1], ndiv[i:i + 1])
        b1.set_init(betas[i:i + 1], pose[i:i + 1]); b1.set_scans([scans[i]])
        b1.fit(15)
        np.testing.assert_array_equal(b1.get_params()[0], both[i])
        b1.close()
    for s in scans:
        s.close()


def test_displacement_stage_first_steps(small, gmm_bufs):
    """SMPL+D (smplify.py:228-247).  The stage is chaotic under round-off (see tests/test_scan_oracle.py), so
    parity is asserted where it is meaningful: the gradient of the first step (to fp32 round-off, against
    torch.autograd of the oracle), the first steps of the trajectory, and the reference's own first step."""
    import ctypes as C
    import torch
    from bodyfitting_amd import _lib
    from oracle import smplify_oracle as O
    model, dev = small
    g = load_golden("scan_nv690_30it.npz")
    prob, sv, sf = S.make_scan_problem(model, frame=0, n_views=8)
    scan = N.Scan(sv, sf)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, 8)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans([scan])
    b.fit(30)
    base = b.get_result()[0][0]
    b.fit_displacement(1)
    d1 = b.get_displacement()[0]
    np.testing.assert_allclose(d1, g["disp1"], atol=2e-5)                   # reference, first step
    m1 = np.empty((1, 690, 3), np.float32)
    _lib.check(_lib.load().bf_batch_debug_disp_moment(b._h, _lib.fptr(m1)))
    grad = m1[0] / 0.1
    # autograd gradient of the same objective at disp = 0 on the same base mesh (fp64 oracle pieces)
    bv = torch.tensor(base, dtype=torch.float64)
    disp = torch.zeros_like(bv, requires_grad=True)
    faces_t = torch.as_tensor(np.asarray(model["faces"]), dtype=torch.long)
    ids, cpts, _ = MO.ReferenceSearcher(sv, sf).nearest(base)              # (face ids feed the normal term: the reference's arithmetic)
    tris = sv.astype(np.float64)[sf]
    fnorm = torch.tensor(np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]).astype(np.float32), dtype=torch.float64)
    P = bv + disp
    norms = MO.compute_normal_torch(P, faces_t)
    c = float((sv[:, 1].max() - sv[:, 1].min()) / 1.7)
    loss = MO.point_cloud_loss(P, torch.tensor(cpts, dtype=torch.float64)) + (
        MO.normal_loss(fnorm[torch.as_tensor(ids, dtype=torch.long)], norms) + MO.normal_laplacian_smoothness(norms, faces_t)) * c * 0.1
    loss.backward()
    want = disp.grad.numpy()
    np.testing.assert_allclose(grad, want, atol=1e-4 * np.abs(want).max())   # (observed: 4e-6 abs on values up to 0.1)
    # a few more steps against the fp32 oracle trajectory from the golden's first step on
    b.fit_displacement(3)
    d3 = b.get_displacement()[0]
    res = O.fit(model, gmm_bufs, prob, 30, scan=(sv, sf), displacement=True, disp_snapshots=(3,))
    assert np.mean(np.abs(d3 - res["disp_snapshots"][3]) < 2e-4) > 0.97
    b.close()
    scan.close()


def test_scan_and_batch_may_be_destroyed_in_any_order(small):
    """bodyfit.h "ORDER OF DESTRUCTION": a scan destroyed while a batch still holds it detaches that batch's scans and the batch's
    next fit FAILS until bf_batch_set_scans is called again (with None: the fit then equals a fit that never had scans, bit for
    bit; the SMPL+D stage reports that nothing is attached); re-attaching and detaching afterwards never touch the freed scan, and
    a batch may go before its scans."""
    from bodyfitting_amd import _lib
    model, dev = small
    items = [S.make_scan_problem(model, frame=f, n_views=8) for f in (0, 1)]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([p for p, _, _ in items])

    def batch():
        b = N.FrameBatch(dev, 2, 8)
        b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
        return b
    plain = batch()
    plain.fit(12)
    want = plain.get_params()
    plain.close()
    scans = [N.Scan(sv, sf) for _, sv, sf in items]
    b = batch()
    b.set_scans(scans)
    b.fit(3)                                   # queued work that reads the scans
    scans[0].close()                           # destroy-then-detach: waits for the device, detaches BOTH frames' scans
    b.reset()
    with pytest.raises(_lib.BodyfitError, match="was destroyed"):      # ... and the batch says so instead of fitting without them
        b.fit(12)
    with pytest.raises(_lib.BodyfitError, match="was destroyed"):
        b.fit_displacement(2)
    b.set_scans(None)                          # detach after the destroy: nothing to touch; the batch goes on without scans
    b.reset()
    b.fit(12)
    np.testing.assert_array_equal(b.get_params(), want)
    with pytest.raises(_lib.BodyfitError, match="no scans attached"):
        b.fit_displacement(2)
    fresh = N.Scan(items[0][1], items[0][2])
    b.set_scans([fresh, scans[1]])             # replace after the destroy
    b.fit(2)
    b.close()                                  # the batch goes first ...
    fresh.close(); scans[1].close()            # ... its scans afterwards (no device-wide wait: nothing holds them)
    assert _lib.load().bf_device_cache_trim(dev.device) > 0      # the destroyed scans' blocks sat in the cache
    assert _lib.load().bf_device_cache_trim(dev.device) == 0


def test_building_a_scan_does_not_wait_for_a_fit_in_flight(small):
    """bf_scan_create runs on the NULL stream and waits for that stream only (the library's streams are non-blocking): with a long fit queued on a batch's stream, building a
    scan returns while the fit is still running (the fit alone takes several times as long as the build)"""
    import time
    model, dev = small
    p, sv, sf = S.make_scan_problem(model, frame=0, n_views=8)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([p])
    b = N.FrameBatch(dev, 1, 8)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    N.Scan(sv, sf).close()                     # warm: streams, block cache, code objects
    b.fit(50); b.sync()
    t0 = time.perf_counter(); N.Scan(sv, sf).close(); t_build = time.perf_counter() - t0
    t0 = time.perf_counter(); b.fit(60000); t_issue = time.perf_counter() - t0       # asynchronous: ~0.25 s of device time
    t0 = time.perf_counter(); s = N.Scan(sv, sf); t_under = time.perf_counter() - t0
    t0 = time.perf_counter(); b.sync(); t_rest = time.perf_counter() - t0
    print("scan build alone %.2f ms, under a fit in flight %.2f ms; the fit still ran %.1f ms after it (issue %.2f ms)" % (
        t_build * 1e3, t_under * 1e3, t_rest * 1e3, t_issue * 1e3))
    assert t_rest > 5 * t_under and t_under < 10 * t_build + 2e-3
    s.close(); b.close()
