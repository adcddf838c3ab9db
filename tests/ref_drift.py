"""How far the REFERENCE moves from itself (tests/golden/sens_*.npz, cfg3_*_<variant>.npz: the imported reference re-run under TEN
perturbations that change no mathematics - 2 / 4 / 8 intra-op threads instead of 1, the initial pose / the keypoints' pixel coordinates /
the intrinsics / the camera-to-world matrices moved by one float32 ulp, and combinations: oracle/gen_golden.py PERTURBATIONS).  The
parity bands of the round-off-amplifying loops (silhouette, scan, SMPL+D) are K x the LARGEST of those drifts, never below the
north-star tolerance: a HIP result is "the reference's" when it is no further from the reference than the reference is from itself.
Rounds 3-5 had two perturbed runs and K = 3; round 6 has ten and K = 2 (a band on two draws of a chaotic system is
weak in both directions).  `position()` says where a HIP error sits among the reference's own drifts."""
import numpy as np

from conftest import load_golden

VARIANTS = ("threads8", "ulp", "threads2", "threads4", "ulp_down", "ulp_kp", "ulp_cam", "ulp_ext", "t4_ulp_down", "t2_ulp_kp")
# the scan loops have a third perturbation (tests/golden/sens_scan_*.npz, round 4): the closest-point search - whose face ids are decided by
# last bits - built with the compiler free to fuse multiply-adds, as nvcc's default is for the reference's own build
SCAN_VARIANTS = VARIANTS + ("fused",)
K = 2.0                 # band = K x the reference's own drift (the largest of the ten perturbations)
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


def drifts(base, sens, keys, variants=VARIANTS, prefix_base=""):
    """the drift of every variant apart (max over `keys`), ascending"""
    return sorted(max(float(np.abs(np.asarray(sens[f"{v}_{k}"], np.float64) - np.asarray(base[prefix_base + k], np.float64)).max()) for k in keys)
                  for v in variants)


def position(err, own):
    """'above n of N reference drifts (median m, max M)': where a HIP error sits in the reference's own distribution"""
    own = sorted(own)
    return "HIP %.3g is above %d of the reference's %d own drifts (median %.3g, max %.3g)" % (
        err, sum(1 for d in own if d < err), len(own), own[len(own) // 2], own[-1])


def cfg3_drifts(keys):
    base, var = cfg3_variants()
    return sorted(max(float(np.abs(np.asarray(var[v][k], np.float64) - np.asarray(base[k], np.float64)).max()) for k in keys) for v in VARIANTS)
