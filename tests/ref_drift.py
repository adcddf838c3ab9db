"""How far the REFERENCE moves from itself (tests/golden/sens_*.npz, cfg3_*_{threads8,ulp}.npz: the imported reference re-run with 8
intra-op threads instead of 1, and with the initial pose moved by one float32 ulp - oracle/gen_golden.py: sensitivity_goldens,
cfg3_goldens).  The parity bands of the round-off-amplifying loops (silhouette, scan, SMPL+D) are K x this measured drift, never
below the north-star tolerance: a HIP result is "the reference's" when it is no further from the reference than the reference is
from itself under perturbations that change no mathematics."""
import numpy as np

from conftest import load_golden

VARIANTS = ("threads8", "ulp")
# the scan loops have a third perturbation (tests/golden/sens_scan_*.npz, round 4): the closest-point search - whose face ids are decided by
# last bits - built with the compiler free to fuse multiply-adds, as nvcc's default is for the reference's own build
SCAN_VARIANTS = VARIANTS + ("fused",)
K = 3.0                 # band = K x the reference's own drift (the larger of the two perturbations)
FLOOR = 1e-4            # north star: 1e-4 abs


def drift(base, sens, keys, variants=VARIANTS, prefix_base=""):
    """max over the variants and `keys` of max |variant - base|; base / sens: npz files (sens holds '<variant>_<key>')"""
    worst = 0.0
    for v in variants:
        for k in keys:
            worst = max(worst, float(np.abs(np.asarray(sens[f"{v}_{k}"], np.float64) - np.asarray(base[prefix_base + k], np.float64)).max()))
    return worst


def band(base, sens, keys, floor=FLOOR, k=K, **kw):
    return max(floor, k * drift(base, sens, keys, **kw))


def cfg3_variants():
    base = load_golden("cfg3_smplx_48view_8mask_200it_base.npz")
    return base, {v: load_golden(f"cfg3_smplx_48view_8mask_200it_{v}.npz") for v in VARIANTS}


def cfg3_drift(keys):
    base, var = cfg3_variants()
    return max(float(np.abs(np.asarray(var[v][k], np.float64) - np.asarray(base[k], np.float64)).max()) for v in VARIANTS for k in keys)
