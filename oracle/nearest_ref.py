"""ctypes front of oracle/nearest_ref.c - the reference's closest-point search in its OWN float32 arithmetic.
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

  elim(n, A, b, eps)                 restated solve3 / solve4                     matrix.h:13-316
  matrix_ref(n, A, b, eps)           the REFERENCE's solve3 / solve4 themselves (oracle/_ref/libmatrix_ref.so, built from
                                     /root/reference/thirdparty/mesh_grid/matrix.h where it lies; None when absent)
  rule(verts, faces, face, queries)  search_nearest_proj per (face, query) pair   mesh_grid_kernel.cu:12-109
  nearest_allfaces(...)              the rule over all faces, first strictly closer in face order
  search_nearest(...)                the grid walk of search_nearest_point_kenerel mesh_grid_kernel.cu:239-353

`fused=True` selects the build in which the compiler may fuse multiply-adds (nvcc's default for the reference's own
build): the two builds bracket what "the reference's bits" can be.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(quiet=True):
    """make -C oracle (the restatement, and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL if quiet else None)


def lib(fused=False):
    name = "libnearest_oracle_fma.so" if fused else "libnearest_oracle.so"
    if name not in _libs:
        path = os.path.join(HERE, name)
        if not fused and os.environ.get("BF_NEAREST_ORACLE_LIB"):          # (`make -C oracle sanitize`: the strict build under ASan / UBSan)
            path = os.path.abspath(os.environ["BF_NEAREST_ORACLE_LIB"])
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        assert L.ref_real_bytes() == 4
        L.ref_elim_batch.argtypes = [C.c_int, f32p, f32p, C.c_int32, C.c_float, i32p]
        L.ref_rule_pairs.argtypes = [f32p, i32p, i32p, f32p, C.c_int32, f32p, f32p, i32p]
        L.ref_nearest_allfaces.argtypes = [f32p, i32p, C.c_int32, f32p, C.c_int32, f32p, f32p, i32p, f32p, i32p]
        L.ref_search_nearest.argtypes = [i32p, i32p, i32p, f32p, C.c_float, f32p, i32p, f32p, C.c_int32, f32p, f32p, i32p, f32p,
                                         C.c_void_p]
        for fn in (L.ref_elim_batch, L.ref_rule_pairs, L.ref_nearest_allfaces, L.ref_search_nearest):
            fn.restype = None
        _libs[name] = L
    return _libs[name]


def matrix_ref_lib():
    """The reference's matrix.h behind C entry points, or None where /root/reference never was (the GPU box gets the built file)."""
    if "mref" not in _libs:
        path = os.path.join(HERE, "_ref", "libmatrix_ref.so")
        if not os.path.exists(path) and os.path.exists("/root/reference/thirdparty/mesh_grid/matrix.h"):
            build()
        if not os.path.exists(path):
            _libs["mref"] = None
        else:
            L = C.CDLL(path)
            L.mref_solve_batch.argtypes = [C.c_int, f32p, f32p, C.c_int32, C.c_float, i32p]
            L.mref_solve_batch.restype = None
            _libs["mref"] = L
    return _libs["mref"]


def _elim(fn, n, A, b, eps):
    A = np.ascontiguousarray(A, np.float32).reshape(-1, n * n).copy()
    b = np.ascontiguousarray(b, np.float32).reshape(-1, n).copy()
    valid = np.zeros(len(A), np.int32)
    fn(n, A, b, len(A), np.float32(eps), valid)
    return A, b, valid


def elim(n, A, b, eps=1e-9, fused=False):
    """-> (A after the elimination, x, valid) for a batch of n x n systems (column-major like the reference: A[r + n c])."""
    return _elim(lib(fused).ref_elim_batch, n, A, b, eps)


def matrix_ref(n, A, b, eps=1e-9):
    L = matrix_ref_lib()
    return None if L is None else _elim(L.mref_solve_batch, n, A, b, eps)


def _mesh(verts, faces):
    return np.ascontiguousarray(verts, np.float32).reshape(-1, 3), np.ascontiguousarray(faces, np.int32).reshape(-1, 3)


def rule(verts, faces, face_of_pair, queries, fused=False):
    """search_nearest_proj for pairs (face_of_pair[p], queries[p]) -> (coeff[P,3], dist2[P], path[P])."""
    v, f = _mesh(verts, faces)
    q = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
    fp = np.ascontiguousarray(face_of_pair, np.int32)
    coeff, dist, path = np.zeros((len(q), 3), np.float32), np.zeros(len(q), np.float32), np.zeros(len(q), np.int32)
    lib(fused).ref_rule_pairs(v, f, fp, q, len(q), coeff, dist, path)
    return coeff, dist, path


def nearest_allfaces(verts, faces, queries, fused=False):
    """-> (face[Q] int32, point[Q,3], coeff[Q,3], dist2[Q], ties[Q]): the rule over ALL faces, first strictly closer in face
    order; ties = how many other faces return the winner's distance bit for bit."""
    v, f = _mesh(verts, faces)
    q = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
    n = len(q)
    coeff, proj = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    face, dist, ties = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    lib(fused).ref_nearest_allfaces(v, f, len(f), q, n, coeff, proj, face, dist, ties)
    return face, proj, coeff, dist, ties


def search_nearest(verts, faces, queries, grid, fused=False, stats=False):
    """search_nearest_point_kenerel.  grid = (step, num[3], origin[3], tri_num, tri_idx) as MeshGridSearcher.set_mesh /
    insert_grid_surface leave them (oracle.mesh_oracle.grid_params / insert_grid_surface).
    -> (face[Q], point[Q,3], coeff[Q,3], dist2[Q]) (+ [evaluations, cells, shells] when stats)."""
    v, f = _mesh(verts, faces)
    q = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
    step, num, origin, tri_num, tri_idx = grid
    num = np.asarray(num, np.int64)
    size = np.array([num[0], num[1], num[2], num[0] * num[1] * num[2]], np.int32)
    n = len(q)
    coeff, proj = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    face, dist = np.zeros(n, np.int32), np.zeros(n, np.float32)
    st = np.zeros(3, np.int64)
    lib(fused).ref_search_nearest(np.ascontiguousarray(tri_num, np.int32), np.ascontiguousarray(tri_idx, np.int32), size,
                                  np.ascontiguousarray(origin, np.float32), np.float32(step), v, f, q, n, coeff, proj, face, dist,
                                  st.ctypes.data if stats else None)
    return (face, proj, coeff, dist, st) if stats else (face, proj, coeff, dist)
