"""Silhouette contours on the GPU (reference smplify/loss.py:73-83, `extract_countours`).

The reference extracts them with `cv2.findContours(mask * 255, RETR_EXTERNAL, CHAIN_APPROX_NONE)` and keeps one contour.
Here `libbodyfit.so` follows the borders itself (bf_contour_kernel: Suzuki-Abe border following, one wave per mask, the
image as bit planes in LDS) and keeps ONE external border per mask - the one `loss.py:80` keeps by default (OpenCV's first
listed contour = the last border the raster scan meets), or the first met, or the longest: include/bodyfit.h BF_CONTOUR_* - see
`bf_extract_contours` and
oracle/contour_oracle.py for the restatement it is tested against.  There is no CPU fallback: without the library or
a GPU these functions raise.  `FrameBatch.set_masks(masks, view_index, None)` runs the same kernel without the points
ever visiting the host.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def extract_contours(masks, device=0, select=_lib.CONTOUR_OPENCV_FIRST):
    """masks [M,H,W] (truthy = foreground) -> list of float32[C_m,2] arrays of (x, y) contour points in border-following
    order (every pixel the walk stands on: pixels of one-pixel-wide parts appear more than once), like loss.py:73-83."""
    m = np.ascontiguousarray(np.asarray(masks) > 0, dtype=np.uint8)
    if m.ndim == 2:
        m = m[None]
    n, H, W = m.shape
    lib = _lib.load()
    counts = np.zeros(n, np.int32)
    mp = m.ctypes.data_as(C.POINTER(C.c_uint8))
    _lib.check(lib.bf_extract_contours(int(device), n, H, W, mp, _lib.iptr(counts), None, int(select)), "bf_extract_contours")
    xy = np.zeros((max(int(counts.sum()), 1), 2), np.float32)
    _lib.check(lib.bf_extract_contours(int(device), n, H, W, mp, _lib.iptr(counts), _lib.fptr(xy), int(select)), "bf_extract_contours")
    ends = np.cumsum(counts)
    return [xy[e - c:e].copy() for c, e in zip(counts, ends)]


def extract_contour(mask, device=0, select=_lib.CONTOUR_OPENCV_FIRST):
    """mask[H,W] -> float32[C,2]"""
    return extract_contours(np.asarray(mask)[None], device, select)[0]
