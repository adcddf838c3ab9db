"""ctypes binding of libbodyfit.so (include/bodyfit.h).  No fallback: without the HIP library the
product path raises - there is deliberately no CPU implementation behind this ABI."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BODYFIT_LIB selects a diagnostic build (e.g. libbodyfit_stamp.so) for bring-up tools; never a different backend
LIB_PATH = os.environ.get("BODYFIT_LIB") or os.path.join(_HERE, "libbodyfit.so")


class BodyfitError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_verts", C.c_int32), ("n_joints", C.c_int32), ("n_betas", C.c_int32),
        ("v_template", C.POINTER(C.c_float)), ("shapedirs", C.POINTER(C.c_float)),
        ("posedirs", C.POINTER(C.c_float)), ("j_regressor", C.POINTER(C.c_float)),
        ("lbs_weights", C.POINTER(C.c_float)), ("parents", C.POINTER(C.c_int32)),
        ("n_selector", C.c_int32), ("selector_ids", C.POINTER(C.c_int32)),
        ("n_extra", C.c_int32), ("j_regressor_extra", C.POINTER(C.c_float)),
        ("n_joint_map", C.c_int32), ("joint_map", C.POINTER(C.c_int32)),
        ("n_loss_joints", C.c_int32),
        ("gmm_components", C.c_int32), ("gmm_dim", C.c_int32),
        ("gmm_means", C.POINTER(C.c_float)), ("gmm_precisions", C.POINTER(C.c_float)),
        ("gmm_nll_weights", C.POINTER(C.c_float)),
        ("n_faces", C.c_int32), ("faces", C.POINTER(C.c_int32)),
        ("model_kind", C.c_int32), ("pose_mean", C.POINTER(C.c_float)), ("n_hand_pca", C.c_int32),
        ("left_hand_components", C.POINTER(C.c_float)), ("right_hand_components", C.POINTER(C.c_float)),
        ("n_lmk_static", C.c_int32), ("lmk_faces_idx", C.POINTER(C.c_int32)), ("lmk_bary_coords", C.POINTER(C.c_float)),
        ("n_lmk_dynamic", C.c_int32), ("n_dyn_rows", C.c_int32), ("dynamic_lmk_faces_idx", C.POINTER(C.c_int32)),
        ("dynamic_lmk_bary_coords", C.POINTER(C.c_float)), ("neck_joint", C.c_int32),
    ]


class Hyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "sigma", "pose_prior_weight", "angle_prior_weight", "shape_prior_weight", "constant_scale",
        "imsize", "lr", "lr_transl_scale", "adam_beta1", "adam_beta2", "adam_eps", "lr_displacement", "mask_cdist_form", "dense_after")]


CONTOUR_OPENCV_FIRST, CONTOUR_RASTER_FIRST, CONTOUR_LONGEST = 0, 1, 2
NEAREST_REFERENCE, NEAREST_FAST = 0, 1
FIT_DEFAULT, FIT_DENSE, FIT_NO_VERTICES, FIT_FETCH, FIT_RESET, FIT_GRAPH, FIT_NOTIME = 0, 1, 2, 4, 8, 16, 32

# every entry point include/bodyfit.h declares: name -> (restype, argtypes)
_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)
_VP = C.c_void_p
SIGNATURES = {
    "bf_last_error": (C.c_char_p, []),
    "bf_version": (C.c_char_p, []),
    "bf_device_count": (C.c_int, []),
    "bf_hyper_default": (None, [C.POINTER(Hyper)]),
    "bf_model_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(_VP)]),
    "bf_model_destroy": (None, [_VP]),
    "bf_model_n_params": (C.c_int, [_VP]),
    "bf_model_fit_instance": (C.c_int, [_VP]),
    "bf_device_cache_trim": (C.c_int64, [C.c_int]),
    "bf_batch_dense_timing": (C.c_int, [_VP, C.c_int, _FP]),
    "bf_batch_dense_resident": (C.c_int, [_VP]),
    "bf_smpl_forward": (C.c_int, [_VP, C.c_int, _FP, _FP, _FP, _FP, _FP, _FP]),
    "bf_model_forward": (C.c_int, [_VP, C.c_int, _FP, _FP, _FP]),
    "bf_batch_create": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "bf_batch_destroy": (None, [_VP]),
    "bf_batch_set_cameras": (C.c_int, [_VP, _FP, _FP]),
    "bf_batch_set_keypoints": (C.c_int, [_VP, _FP, _IP]),
    "bf_batch_set_init": (C.c_int, [_VP, _FP, _FP]),
    "bf_batch_stage_inputs": (C.c_int, [_VP, _FP, _IP, _FP, _FP]),
    "bf_batch_get_previous": (C.c_int, [_VP, _FP, _FP, _FP, _FP, _FP]),
    "bf_batch_reset": (C.c_int, [_VP]),
    "bf_batch_set_params": (C.c_int, [_VP, _FP]),
    "bf_batch_get_params": (C.c_int, [_VP, _FP]),
    "bf_fit": (C.c_int, [_VP, C.c_int, C.POINTER(Hyper), C.c_uint32]),
    "bf_loss_grad": (C.c_int, [_VP, C.POINTER(Hyper), _FP, _FP]),
    "bf_batch_sync": (C.c_int, [_VP]),
    "bf_batch_get_result": (C.c_int, [_VP, _FP, _FP, _FP, _FP]),
    "bf_batch_export_params_dev": (C.c_int, [_VP, _VP]),
    "bf_scan_create": (C.c_int, [C.c_int, C.c_int, _FP, C.c_int, _IP, C.POINTER(_VP)]),
    "bf_scan_destroy": (None, [_VP]),
    "bf_scan_height": (C.c_float, [_VP]),
    "bf_scan_grid_info": (C.c_int, [_VP, _IP, _FP]),
    "bf_scan_grid_lists": (C.c_int, [_VP, _IP, _IP, _IP]),
    "bf_scan_inside": (C.c_int, [_VP, C.c_int, _FP, _FP]),
    "bf_scan_intersects": (C.c_int, [_VP, C.c_int, _FP, _FP, C.POINTER(C.c_uint8)]),
    "bf_scan_nearest": (C.c_int, [_VP, C.c_int, _FP, _IP, _FP, _FP]),
    "bf_scan_nearest_hinted": (C.c_int, [_VP, C.c_int, _FP, _FP, _IP, _FP, _FP, C.c_int, _FP]),
    "bf_scan_nearest_backward": (C.c_int, [_VP, C.c_int, _IP, _FP, _FP, _FP]),
    "bf_batch_mesh_span": (C.c_int, [_VP, C.c_int, _FP]),
    "bf_nearest_rule_set": (C.c_int, [C.c_int]),
    "bf_nearest_rule_get": (C.c_int, []),
    "bf_mask_fold_set": (C.c_int, [C.c_int]),
    "bf_mask_fold_get": (C.c_int, []),
    "bf_nearest_selftest_quot": (C.c_int, [C.c_int, C.c_int, _FP, _FP, _FP]),
    "bf_nearest_selftest_rule": (C.c_int, [C.c_int, C.c_int, _FP, C.c_int, _FP, _FP]),
    "bf_batch_set_scans": (C.c_int, [_VP, C.POINTER(_VP)]),
    "bf_batch_set_masks": (C.c_int, [_VP, C.c_int, _IP, C.c_int, C.c_int, C.POINTER(C.c_uint8), _IP, _FP, C.c_int]),
    "bf_batch_stage_masks": (C.c_int, [_VP, C.c_int, _IP, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int]),
    "bf_extract_contours": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), _IP, _FP, C.c_int]),
    "bf_batch_mask_loss": (C.c_int, [_VP, C.POINTER(Hyper), _FP, _FP]),
    "bf_fit_displacement": (C.c_int, [_VP, C.c_int, C.POINTER(Hyper)]),
    "bf_batch_get_displacement": (C.c_int, [_VP, _FP]),
    "bf_batch_last_timing": (C.c_int, [_VP, _FP]),
    "bf_batch_timing_reset": (C.c_int, [_VP]),
    "bf_batch_timing_sum": (C.c_int, [_VP, _FP, _IP]),
    "bf_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, _IP, _IP]),
    "bf_shard_capacity": (C.c_int, [C.c_int, C.c_int]),
    "bf_shard_unpack": (C.c_int, [_FP, C.c_int, C.c_int, C.c_int, _FP]),
    "bf_shard_contour_offsets": (C.c_int, [C.c_int, C.c_int, C.c_int, _IP, C.POINTER(C.c_int64)]),
    "bf_group_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int, _IP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "bf_group_destroy": (None, [_VP]),
    "bf_group_n_devices": (C.c_int, [_VP]),
    "bf_group_n_params": (C.c_int, [_VP]),
    "bf_group_shard": (C.c_int, [_VP, C.c_int, _IP, _IP, _IP]),
    "bf_group_batch": (_VP, [_VP, C.c_int]),
    "bf_group_model": (_VP, [_VP, C.c_int]),
    "bf_group_set_cameras": (C.c_int, [_VP, _FP, _FP]),
    "bf_group_set_keypoints": (C.c_int, [_VP, _FP, _IP]),
    "bf_group_set_init": (C.c_int, [_VP, _FP, _FP]),
    "bf_group_stage_inputs": (C.c_int, [_VP, _FP, _IP, _FP, _FP]),
    "bf_group_set_masks": (C.c_int, [_VP, C.c_int, _IP, C.c_int, C.c_int, C.POINTER(C.c_uint8), _IP, _FP, C.c_int]),
    "bf_group_stage_masks": (C.c_int, [_VP, C.c_int, _IP, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int]),
    "bf_group_set_scans": (C.c_int, [_VP, C.POINTER(_VP)]),
    "bf_group_fit": (C.c_int, [_VP, C.c_int, C.POINTER(Hyper), C.c_uint32]),
    "bf_group_fit_displacement": (C.c_int, [_VP, C.c_int, C.POINTER(Hyper)]),
    "bf_group_sync": (C.c_int, [_VP]),
    "bf_group_comm_size": (C.c_int, [_VP]),
    "bf_group_gather_params": (C.c_int, [_VP, _FP, C.c_int]),
    "bf_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "bf_comm_create": (C.c_int, [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "bf_comm_destroy": (None, [_VP]),
    "bf_comm_size": (C.c_int, [_VP]),
    "bf_comm_barrier": (C.c_int, [_VP]),
    "bf_comm_allreduce": (C.c_int, [_VP, C.POINTER(C.c_double), C.c_int]),
    "bf_comm_gather_params": (C.c_int, [_VP, _VP, C.c_int, _FP]),
    "bf_texfit_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _FP, C.c_int, C.POINTER(_VP)]),
    "bf_texfit_destroy": (None, [_VP]),
    "bf_texfit_set_mesh": (C.c_int, [_VP, C.c_int, C.c_int, _FP, C.c_int, _IP, _FP]),
    "bf_texfit_render": (C.c_int, [_VP, C.c_int, _FP, _FP, _FP, C.c_float, _FP]),
    "bf_texfit_render_ndc": (C.c_int, [_VP, C.c_int, _FP, C.c_int, _IP, _FP, _FP, _FP]),
    "bf_texfit_step": (C.c_int, [_VP, _FP, _FP, _FP, C.c_float, C.c_float, C.POINTER(C.c_double)]),
    "bf_texfit_loss_grad": (C.c_int, [_VP, _FP, _FP, _FP, C.c_float, C.POINTER(C.c_double), _FP]),
    "bf_texfit_get_textures": (C.c_int, [_VP, _FP]),
    "bf_batch_debug_dump": (C.c_int, [_VP, _FP, C.c_int]),
    "bf_batch_debug_disp_moment": (C.c_int, [_VP, _FP]),
}

_lib = None


def load():
    """Return the loaded library; raise BodyfitError (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BodyfitError(
            f"{LIB_PATH} is missing - build it with `make -C bodyfitting_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")
    # (see csrc/api.hip: bf_more_hw_queues - also here, for the case that something in this process touches HIP before the library loads)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise BodyfitError(f"could not load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().bf_last_error().decode("utf-8", "replace")
        raise BodyfitError(f"{what or 'libbodyfit'} failed ({rc}): {msg}")


def fptr(a):
    return a.ctypes.data_as(_FP) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(_IP) if a is not None else None
