"""`utils.mesh_grid_searcher.MeshGridSearcher` of the reference (utils/mesh_grid_searcher.py:51-99; the same file sits in
thirdparty/mesh_grid/) over the HIP closest-point grid (`native.Scan`, the `bf_scan_*` entry points that replace the `mesh_grid`
pybind module, mesh_grid.cpp:129-136).

Same constructor and methods: `MeshGridSearcher(verts, faces)`, `set_mesh`, `nearest_points(points) -> (nearest points, face ids)`,
`inside_mesh(points) -> +1 / -1 per point`, `intersects_any(origins, directions) -> bool per ray`.  The attributes the reference
leaves on the instance after `set_mesh` are there too (`verts`, `faces`, `step`, `num`, `minmax`, `tri_num`, `tri_idx` - the
grid the device built, read back lazily).  numpy in -> numpy out; torch tensors in -> torch tensors out on the tensors' device
(torch is imported only in that case).
"""
from __future__ import annotations

import numpy as np

from .native import Scan


def _is_tensor(x):
    return hasattr(x, "detach") and hasattr(x, "cpu")


def _to_np(x, dtype):
    if _is_tensor(x):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))


class MeshGridSearcher:
    def __init__(self, verts=None, faces=None, device=0):
        self._scan = None
        self._device = device
        if verts is not None and faces is not None:
            self.set_mesh(verts, faces, device)

    @staticmethod
    def _device_index(device):
        if isinstance(device, int):
            return device
        if isinstance(device, str):                       # 'cuda:0' (the reference's default argument)
            return int(device.split(":")[1]) if ":" in device else 0
        idx = getattr(device, "index", None)
        return 0 if idx is None else int(idx)

    def set_mesh(self, verts, faces, device=0):
        """utils/mesh_grid_searcher.py:56-79: cell edge = (bounding-box volume / n_verts)^(1/3), the grid centred on the box;
        `insert_grid_surface` = bf_scan_create's build (count, scan, fill: the lists equal the reference's as sets per cell)"""
        self.close()
        self.verts = _to_np(verts, np.float32).reshape(-1, 3)
        self.faces = _to_np(faces, np.int32).reshape(-1, 3)
        self._scan = Scan(self.verts, self.faces, device=self._device_index(device))
        dims, origin, step = self._scan.grid_info()
        self.step = np.float32(step)
        self.num = np.concatenate([dims, [int(np.prod(dims))]]).astype(np.int32)          # [nx, ny, nz, nx ny nz]
        self.minmax = np.concatenate([origin, self.verts.max(0)]).astype(np.float32)      # [grid origin | max corner]
        self._lists = None

    def _grid_lists(self):
        if self._lists is None:
            self._lists = self._scan.grid_lists()
        return self._lists

    @property
    def tri_num(self):
        return self._grid_lists()[0]

    @property
    def tri_idx(self):
        return self._grid_lists()[1]

    def close(self):
        if getattr(self, "_scan", None) is not None:
            self._scan.close()
            self._scan = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _need_mesh(self):
        if self._scan is None:
            raise RuntimeError("MeshGridSearcher: set_mesh() first")

    @staticmethod
    def _back(like, *arrays):
        if not _is_tensor(like):
            return arrays
        import torch
        return tuple(torch.from_numpy(np.ascontiguousarray(a)).to(like.device) for a in arrays)

    def nearest_points(self, points):
        """:81-84 -> (nearest_pts[n,3] float32, nearest_faces[n] int32); no gradient, like the reference's SurfaceNearest
        (its backward is commented out, :17-49)"""
        self._need_mesh()
        pts, ids, _ = self._scan.nearest_points(_to_np(points, np.float32).reshape(-1, 3))
        return self._back(points, pts, ids)

    def nearest_points_with_coefficients(self, points):
        """the third output of the kernel (`coeff`, :10-14), which the reference's wrapper drops"""
        self._need_mesh()
        pts, ids, bary = self._scan.nearest_points(_to_np(points, np.float32).reshape(-1, 3))
        return self._back(points, pts, ids, bary)

    def inside_mesh(self, points):
        """:86-91 -> float32[n]: +1 inside, -1 outside"""
        self._need_mesh()
        return self._back(points, self._scan.inside_mesh(_to_np(points, np.float32).reshape(-1, 3)))[0]

    def intersects_any(self, origins, directions):
        """:93-99 -> bool[n]"""
        self._need_mesh()
        o = _to_np(origins, np.float32).reshape(-1, 3)
        return self._back(origins, self._scan.intersects_any(o, _to_np(directions, np.float32).reshape(-1, 3)))[0]
