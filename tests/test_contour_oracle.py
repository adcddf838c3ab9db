"""Contour extraction oracle (oracle/contour_oracle.py = Suzuki-Abe border following, what cv2.findContours with
RETR_EXTERNAL / CHAIN_APPROX_NONE returns; reference smplify/loss.py:73-83): known answers and an independent
set formulation."""
import numpy as np
import pytest

from bodyfitting_amd import synthetic as S
from oracle import contour_oracle as CO


def _img(rows):
    return np.array([[c == "#" for c in r] for r in rows], np.uint8)


def test_known_answers():
    assert CO.find_external_contours(np.zeros((5, 5), np.uint8)) == []
    c = CO.find_external_contours(_img(["....", ".#..", "...."]))
    assert len(c) == 1 and c[0].tolist() == [[1, 1]]                                     # single pixel: one point
    c = CO.find_external_contours(_img([".....", ".###.", "....."]))
    assert c[0].tolist() == [[1, 1], [2, 1], [3, 1], [2, 1]]                             # n-pixel line: 2n - 2 points
    c = CO.find_external_contours(_img(["....", ".##.", ".##.", "...."]))
    assert c[0].tolist() == [[1, 1], [1, 2], [2, 2], [2, 1]]                             # top-left, bottom-left, bottom-right, top-right: OpenCV's order for a box
    c = CO.find_external_contours(_img([".....", ".###.", ".###.", ".###.", "....."]))
    assert len(c[0]) == 8 and [2, 2] not in c[0].tolist()                                # filled 3x3: its 8 rim pixels once each
    # touching the image frame, diagonal (8-connected) pixels are one component
    c = CO.find_external_contours(_img(["#..", ".#.", "..#"]))
    assert len(c) == 1 and c[0].tolist() == [[0, 0], [1, 1], [2, 2], [1, 1]]


def test_hole_and_nested_component():
    ring = _img([".......",
                 ".#####.",
                 ".#...#.",
                 ".#.#.#.",
                 ".#...#.",
                 ".#####.",
                 "......."])
    c = CO.find_external_contours(ring)
    assert len(c) == 1 and len(c[0]) == 16                   # the ring's outer rim; neither the hole border nor the island
    assert [3, 3] not in c[0].tolist()
    two = _img(["..........",
                ".##....#..",
                ".##...###.",
                ".......#..",
                ".........."])
    c = CO.find_external_contours(two)
    assert [a[0].tolist() for a in c] == [[1, 1], [7, 1]]    # discovery order = raster order of the start pixels
    # which ONE the loss gets (loss.py:80 keeps OpenCV's first listed contour = the border met LAST by the scan)
    assert CO.extract_contour(two)[0].tolist() == [7, 1] and CO.extract_contour(two, "raster_first")[0].tolist() == [1, 1]
    assert CO.extract_contour(two, "longest").shape == (4, 2) and CO.extract_contour(two, "longest")[0].tolist() == [1, 1]   # a tie: the first


def test_concave_shape_in_opencv_order():
    """hand-computed: the point sequence cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) produces for a small concave shape.
    A "U" (two 3-pixel prongs joined by a 5-pixel base):

        . . . . . . .        Border following starts at the first foreground pixel of the raster scan, (1, 1), whose left
        . # . . . # .        neighbour is background.  OpenCV walks an outer border with the inside on its left hand: from
        . # . . . # .        (1,1) down the left prong to (1,3), along the base to (5,3), up the right prong to (5,1), back
        . # # # # # .        down it (a one-pixel-wide prong is passed twice) to (5,2), then along the top of the base -
        . . . . . . .        (4,3), (3,3), (2,3): the base is one pixel thick, its pixels are passed twice too - and up the
                             left prong through (1,2) back to the start."""
    u = _img([".......", ".#...#.", ".#...#.", ".#####.", "......."])
    want = [[1, 1], [1, 2], [1, 3], [2, 3], [3, 3], [4, 3], [5, 3], [5, 2], [5, 1], [5, 2], [4, 3], [3, 3], [2, 3], [1, 2]]
    c = CO.find_external_contours(u)
    assert len(c) == 1 and c[0].tolist() == want
    # the same shape two pixels thick: every rim pixel once, counter-clockwise on the screen (y down) from the top-left pixel
    v = _img(["..........", ".##....##.", ".##....##.", ".########.", ".########.", ".........."])
    want = ([[1, 1], [1, 2], [1, 3], [1, 4]] + [[x, 4] for x in range(2, 9)] + [[8, 3], [8, 2], [8, 1], [7, 1], [7, 2]] +
            [[6, 3], [5, 3], [4, 3], [3, 3]] + [[2, 2], [2, 1]])
    assert CO.find_external_contours(v)[0].tolist() == want


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_border_following_visits_exactly_the_outer_border_set(seed):
    """every external component: the set of visited pixels == foreground pixels 4-adjacent to the outside background"""
    rng = np.random.default_rng(seed)
    from scipy import ndimage
    m = ndimage.gaussian_filter(rng.normal(size=(48, 64)), 2.5) > 0.02
    m[20:23, 5:40] = True                                    # a bar, some thin spurs
    m[5, 10:30] = True
    cs = CO.find_external_contours(m)
    sets = CO.outer_border_set(m)
    assert len(cs) == len(sets) > 0
    for c in cs:
        start = (int(c[0, 0]), int(c[0, 1]))
        assert {(int(x), int(y)) for x, y in c} == sets[start]
        d = np.abs(np.diff(np.concatenate([c, c[:1]]), axis=0)).max(1)
        assert len(c) == 1 or np.all(d == 1)                 # consecutive points are 8-neighbours, the walk closes


def test_rendered_silhouette():
    model = S.make_model("smpl", seed=0, nv=690)
    prob = S.make_problem(model, frame=0, n_views=4, mask_frames=[0, 2]) if "mask_frames" in S.make_problem.__code__.co_varnames else None
    if prob is None or "masks" not in prob:
        pytest.skip("no mask renderer for this problem generator")
    for mk in prob["masks"]:
        c = CO.extract_contour(np.asarray(mk) > 128)
        assert len(c) > 100
        sets = CO.outer_border_set(np.asarray(mk) > 128)
        assert {(int(x), int(y)) for x, y in c} in sets.values()
