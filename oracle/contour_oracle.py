"""CPU oracle for the silhouette contours (reference smplify/loss.py:73-83).  TEST INFRASTRUCTURE ONLY.

The reference calls `cv2.findContours(mask * 255, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_NONE)` and keeps one contour.
OpenCV is a third-party dependency that is absent here (requirements.txt:7 pins opencv-python 4.1.2.30; the
3-value unpack at loss.py:80 is the OpenCV-3 API), so this module restates the published algorithm behind that
call - S. Suzuki, K. Abe, "Topological structural analysis of digitized binary images by border following",
CVGIP 30 (1985), Algorithm 1, which OpenCV's contour scanner implements - in plain Python:

  * raster scan of the zero-padded image; an OUTER border starts at an unmarked foreground pixel whose left
    neighbour is background (paper: step 1 (a));
  * border following from that pixel (steps 3.1 - 3.5): first non-zero neighbour clockwise from the left
    neighbour, then repeatedly the first non-zero neighbour counter-clockwise from the pixel just left; every
    pixel the walk stands on is one contour point (CHAIN_APPROX_NONE), so pixels of one-pixel-wide parts appear
    more than once;
  * marks: a border pixel whose right neighbour was examined and found empty gets a NEGATIVE mark, any other
    border pixel a positive one (step 3.4).  RETR_EXTERNAL keeps only borders that do not lie inside a hole of
    another component: a new border is skipped while the last marked pixel met on the current row is positive
    (we are between the left and the right edge of a traced border); hole borders are not followed at all.

Which contour the reference keeps: `contour[np.argmax([a.shape[1] for a in contour])]` - every OpenCV contour has
shape [C, 1, 2], so the argmax is over ones and the expression returns OpenCV's FIRST listed contour, although the
evident intent (and SURVEY.md's reading) is "the longest".  OpenCV's list runs against the raster order (a new contour
becomes the first child of the frame node), so that is the external border whose start pixel comes LAST in the scan:
`extract_contour(mask)` returns it by default; "raster_first" and "longest" are the alternatives the library also offers.
For a silhouette with one component - every GeneBody mask - all three coincide.

PARITY PINNING: unpinned against OpenCV itself (absent).  Pinned by known answers that are common knowledge of
what findContours returns (a single pixel -> 1 point; an n-pixel line -> 2n - 2 points; a filled rectangle -> its
perimeter pixels once each; a component inside a hole is not reported) and by an independent set formulation
(outer border pixels = foreground pixels of the component with a 4-neighbour in the outside background).
"""
from __future__ import annotations

import numpy as np

# neighbour directions, counter-clockwise on the screen (y down), starting east: code -> (dx, dy)
_DIRS = ((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))


def _follow(img, x0, y0):
    """Border following from the outer-border start pixel (x0, y0) of the padded int image `img` (0 background, 1
    unmarked foreground, 2 positive mark, -2 negative mark).  Marks the border, returns its points [(x, y), ...]."""
    def nz(x, y):
        return img[y, x] != 0
    # 3.1: first non-zero neighbour clockwise, starting after the left neighbour (code 4): codes 3, 2, 1, 0, 7, 6, 5
    s = 4
    first = None
    for _ in range(7):
        s = (s - 1) & 7
        if nz(x0 + _DIRS[s][0], y0 + _DIRS[s][1]):
            first = s
            break
    if first is None:
        img[y0, x0] = -2
        return [(x0, y0)]
    x1, y1 = x0 + _DIRS[first][0], y0 + _DIRS[first][1]
    pts = []
    x3, y3, s = x0, y0, first
    while True:
        s_end = s
        # 3.3: first non-zero neighbour counter-clockwise, starting after the direction we came from
        while True:
            s += 1
            x4, y4 = x3 + _DIRS[s & 7][0], y3 + _DIRS[s & 7][1]
            if nz(x4, y4):
                break
        # 3.4: the east neighbour (code 0 = 8) was examined and found empty <=> the search passed code 8
        if s >= 9:
            img[y3, x3] = -2
        elif img[y3, x3] == 1:
            img[y3, x3] = 2
        s &= 7
        pts.append((x3, y3))
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            return pts
        x3, y3 = x4, y4
        s = (s + 4) & 7


def find_external_contours(mask):
    """mask[H, W] (non-zero = foreground) -> list of int32[C, 2] arrays of (x, y) points, one per external outer
    border, in the order the raster scan meets them."""
    m = np.asarray(mask) != 0
    H, W = m.shape
    img = np.zeros((H + 2, W + 2), np.int32)
    img[1:-1, 1:-1] = m
    out = []
    for y in range(1, H + 1):
        inside = False                       # sign of the last marked pixel met on this row
        row = img[y]
        for x in range(1, W + 1):
            p = row[x]
            if p == 1 and row[x - 1] == 0 and not inside:
                pts = _follow(img, x, y)
                out.append(np.asarray(pts, np.int32) - 1)       # un-pad
                p = row[x]
            if p > 1:
                inside = True
            elif p < 0:
                inside = False
    return out


def extract_contour(mask, select="opencv_first"):
    """The ONE contour the silhouette loss gets, float32[C, 2] of (x, y).  select: "opencv_first" = what loss.py:80 keeps
    (`contour[argmax(ones)]` = OpenCV's first listed contour; OpenCV inserts every new contour as the FIRST child of the frame
    node - cvInsertNodeIntoTree - so its list runs against the raster order and contours[0] is the border met LAST),
    "raster_first" = the first border the raster scan meets, "longest" = the longest (first on ties)."""
    cs = find_external_contours(mask)
    if not cs:
        return np.zeros((0, 2), np.float32)
    if select == "opencv_first":
        return cs[-1].astype(np.float32)
    if select == "raster_first":
        return cs[0].astype(np.float32)
    assert select == "longest"
    return cs[int(np.argmax([len(c) for c in cs]))].astype(np.float32)


def outer_border_set(mask):
    """Independent formulation (no border following): per 8-connected component that is not enclosed by another one,
    the foreground pixels with a 4-neighbour in the OUTSIDE background (the background region, 4-connected, that
    contains the padded frame).  -> dict {the component's first pixel (x, y) in raster order: set of (x, y)}."""
    from scipy import ndimage
    fg = np.pad(np.asarray(mask) != 0, 1)
    lab, n = ndimage.label(fg, structure=np.ones((3, 3), int))
    frame_bg, _ = ndimage.label(~fg)                      # 4-connected background regions
    outside = frame_bg == frame_bg[0, 0]
    res = {}
    for k in range(1, n + 1):
        comp = lab == k
        nb = np.zeros_like(comp)
        nb[1:] |= outside[:-1]; nb[:-1] |= outside[1:]; nb[:, 1:] |= outside[:, :-1]; nb[:, :-1] |= outside[:, 1:]
        border = comp & nb
        if not border.any():
            continue                                       # enclosed in a hole of another component: not external
        ys, xs = np.nonzero(comp)
        first = np.lexsort((xs, ys))[0]
        res[(int(xs[first]) - 1, int(ys[first]) - 1)] = {(int(x) - 1, int(y) - 1) for y, x in zip(*np.nonzero(border))}
    return res


def border_pixels_rowmajor(mask):
    """The extractor the committed goldens were generated with (oracle/gen_golden.py stubs it in for cv2, which is
    absent): border points of the outer boundary - foreground pixels with a background pixel or the image edge in
    their 4-neighbourhood, holes ignored - of the largest 8-connected component, ONCE each, in row-major order.
    Same pixel set as border following on masks without one-pixel-wide parts; kept so that the goldens stay valid."""
    from scipy import ndimage
    fg = np.asarray(mask) > 0
    if not fg.any():
        return np.zeros((0, 2), np.float32)
    lab, n = ndimage.label(fg, structure=np.ones((3, 3), int))
    if n > 1:
        sizes = ndimage.sum(fg, lab, index=np.arange(1, n + 1))
        fg = lab == (1 + int(np.argmax(sizes)))
    fg = ndimage.binary_fill_holes(fg)                     # RETR_EXTERNAL: outer border only
    pad = np.pad(fg, 1, constant_values=False)
    all4 = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    ys, xs = np.nonzero(fg & ~all4)
    return np.stack([xs, ys], 1).astype(np.float32)


def border_pixels_rowmajor_all(masks):
    return [border_pixels_rowmajor(m) for m in masks]
