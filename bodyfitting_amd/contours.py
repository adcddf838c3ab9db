"""Silhouette contours without OpenCV.

The reference extracts them with `cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_NONE)` and keeps one
contour (smplify/loss.py:73-83).  cv2 is not available here, so this is a restatement of what that call
returns for a mask: the border points of the outer boundary in Suzuki-Abe's sense - foreground pixels with a
background pixel (or the image edge) in their 4-neighbourhood, holes ignored - of the largest 8-connected
component.  The silhouette loss only sums over the points, so their order does not matter.
Parity note: unpinned against cv2 itself (absent); pixel sets can differ from OpenCV's trace where it visits
a pixel twice (one-pixel-wide spurs).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def extract_contour(mask):
    """mask[H,W] (truthy = foreground) -> float32[C,2] contour points (x, y), row-major order."""
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


def extract_contours(masks):
    """list / array of masks -> list of float32[C,2] arrays (one per mask view), like loss.py:73-83."""
    return [extract_contour(m) for m in masks]
