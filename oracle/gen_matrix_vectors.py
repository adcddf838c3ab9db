#!/usr/bin/env python3
"""Known-answer vectors of the REFERENCE's solve3 / solve4 (thirdparty/mesh_grid/matrix.h:13-316) -> tests/golden/matrix_ref_vectors.npz.

Build-container tool: the outputs come from oracle/_ref/libmatrix_ref.so, i.e. matrix.h itself compiled where it lies under
/root/reference (oracle/Makefile; no reference source enters this repository - the file holds numbers only).  The systems: random
dense at six magnitudes, rank-deficient ones (zero rows / columns, repeated columns, entries scattered around the 1e-9 threshold) and
the bordered Gram systems search_nearest_proj builds (mesh_grid_kernel.cu:31-38, 46-51) for regular, sliver and repeated-corner
triangles from 0.1 mm to 3 m.  tests/test_nearest_ref_oracle.py holds oracle/nearest_ref.c to these bit for bit wherever
/root/reference is absent (and to the library itself, on 10^6 systems, where it is present)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import nearest_ref as NR   # noqa: E402


def systems(n, rng, count):
    """-> (A[count*K, n*n], b[count*K, n]) float32, column-major systems A[r + n c]"""
    out_A, out_b = [], []

    def add(A, b):
        out_A.append(np.asarray(A, np.float32).reshape(-1, n * n)); out_b.append(np.asarray(b, np.float32).reshape(-1, n))
    for s in (1.0, 1e-3, 1e-6, 1e-9, 1e-10, 1e3):
        add(rng.normal(size=(count, n * n)) * s, rng.normal(size=(count, n)) * s)
    A = rng.normal(size=(count, n, n)); b = rng.normal(size=(count, n))
    for i in range(count):
        for _ in range(rng.integers(0, 4)):
            c = rng.integers(0, n)
            if rng.random() < 0.5:
                A[i, c, :] = 0
            else:
                A[i, :, c] = 0
        if rng.random() < 0.3:
            c1, c2 = rng.integers(0, n, 2); A[i, c1] = A[i, c2]
        if rng.random() < 0.3:
            b[i, rng.integers(0, n)] = 0
    add(A.reshape(count, -1), b)
    add(rng.choice([0.0, 1e-9, 9.9e-10, 1.1e-9, 1.0, -1e-9, 2e-9], size=(count, n * n)), rng.choice([0.0, 1e-9, 1.1e-9, 1.0], size=(count, n)))
    add(rng.choice([0.0, 0.0, 0.0, 1e-10, 1.0, -2.0], size=(2 * count, n * n)), rng.choice([0.0, 1e-10, 1.0], size=(2 * count, n)))   # mostly zeros
    for s in (1.0, 1e-1, 1e-2, 1e-3, 1e-4, 3.0):
        P = rng.normal(size=(count, n - 1, 3)) * s
        if n == 4:
            t = rng.random((count, 1))
            P[::3, 2] = P[::3, 0] * t[::3] + P[::3, 1] * (1 - t[::3]) + rng.normal(size=(len(P[::3]), 3)) * s * 1e-4
            P[1::7, 1] = P[1::7, 0]
        P = P.astype(np.float32)
        G = np.einsum("nik,njk->nij", P, P).astype(np.float32)
        K = np.zeros((count, n, n), np.float32); K[:, :n - 1, :n - 1] = G; K[:, n - 1, :n - 1] = 1; K[:, :n - 1, n - 1] = 1
        rhs = np.zeros((count, n), np.float32); rhs[:, n - 1] = 1
        add(K.reshape(count, -1), rhs)
    return np.concatenate(out_A), np.concatenate(out_b)


def main():
    if NR.matrix_ref_lib() is None:
        sys.exit("oracle/_ref/libmatrix_ref.so is not built (needs /root/reference): nothing to generate from")
    rng = np.random.default_rng(20251003)
    out = {}
    for n in (3, 4):
        A, b = systems(n, rng, 45)
        with np.errstate(all="ignore"):
            Ao, x, valid = NR.matrix_ref(n, A, b, 1e-9)
        out.update({"A%d" % n: A, "b%d" % n: b, "A%d_after" % n: Ao, "x%d" % n: x, "valid%d" % n: valid.astype(np.uint8)})
        print("n = %d: %d systems, %d reported inconsistent" % (n, len(A), int((valid == 0).sum())))
    out["eps"] = np.float32(1e-9)
    path = os.path.join(REPO, "tests", "golden", "matrix_ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
