"""Texture fitting on the GPU: host-side mirror of the reference's `TextureFitting` loop (smplify/texture_fitting.py:173-301).

`TextureFitting.fit(smpld_mesh, scan_mesh)` is the loop of `TextureFitting.__call__` (:240-275) with the meshes handed over
as arrays - (vertices[NV,3], faces[NF,3], textures[NF,ts,ts,ts,3]), what `nr.load_obj(..., load_texture=True)` returns - in
place of OBJ directories: ring views first (`gen_cam_views`, 18 views x 5 rounds), random views on the sphere after, one
render of each mesh + L1 loss + Adam step on the SMPL+D textures per iteration, all inside libbodyfit (bf_texfit_*).
OBJ / MTL / image files, the UV texture image and the inpainting CNN (:276-289) stay with the caller.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

ROUND_VIEWS = 18          # texture_fitting.py:254
ROUND_VIEW_ITERS = 5      # texture_fitting.py:246


def _unit(v):
    return v / np.linalg.norm(v)


def gen_cam_views(center, viewnum, dist, gl=False):
    """World-to-camera 4x4 poses of `viewnum` cameras on a horizontal ring of radius `dist` around `center`, all looking
    along their +z away from the centre (utils/renderer.py:7-25); gl=True flips y and z (OpenCV -> OpenGL axes)."""
    center = np.asarray(center, np.float64).reshape(3)
    flip = np.diag([1.0, -1.0, -1.0, 1.0]) if gl else np.eye(4)
    poses = []
    for k in range(viewnum):
        theta = 2.0 * np.pi * k / viewnum
        offset = dist * np.array([np.cos(theta), 0.0, -np.sin(theta)])
        c2w = np.eye(4)
        axis_z = _unit(offset)
        axis_x = np.cross(np.array([0.0, 1.0, 0.0]), axis_z)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = axis_x, np.cross(axis_z, axis_x), axis_z, offset + center
        poses.append(flip @ np.linalg.inv(c2w))
    return poses


def sphere2rot(rad, theta, phi, t=(0.0, 0.0, 0.0)):
    """Camera-to-world pose at spherical coordinates (rad, theta from +y, phi around y) about `t`, z axis towards `t`
    (texture_fitting.py:63-83)."""
    st, ct, sp, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
    eye = rad * np.array([st * sp, ct, st * cp])
    axis_z = _unit(-eye)
    side = np.array([np.sin(phi + np.pi / 2), 0.0, np.cos(phi + np.pi / 2)])
    axis_y = _unit(np.cross(axis_z, side))
    axis_x = _unit(np.cross(axis_y, axis_z))
    pose = np.eye(4)
    pose[:3, :3] = np.column_stack([axis_x, axis_y, axis_z])
    pose[:3, 3] = eye + np.asarray(t, np.float64).reshape(3)
    return pose


def scene_bound(scan_verts):
    """centre of the scan's bounding box and the camera distance bound_y / 0.8 (texture_fitting.py:235-239)"""
    v = np.asarray(scan_verts, np.float32).reshape(-1, 3)
    center = (v.max(0) + v.min(0)) / np.float32(2.0)
    return center, float((v.max(0) - v.min(0))[1] / np.float32(0.8))


def _mesh(mesh):
    v, f, t = mesh
    v = np.ascontiguousarray(np.asarray(v, np.float32).reshape(-1, 3))
    f = np.ascontiguousarray(np.asarray(f).reshape(-1, 3), dtype=np.int32)
    t = np.ascontiguousarray(t, dtype=np.float32)
    if t.ndim != 5 or t.shape[0] != len(f) or t.shape[4] != 3 or not (t.shape[1] == t.shape[2] == t.shape[3]):
        raise ValueError("textures must be [n_faces, ts, ts, ts, 3]")
    return v, f, t


class Renderer:
    """The renderer configuration the loop builds (texture_fitting.py:250-252: projection camera, K from the image size,
    ambient light only, white background, near 0, far 2 x dist, 2 x 2 anti-aliasing) with two mesh slots on the device."""

    TARGET, FITTED = 0, 1

    def __init__(self, image_size, texture_size, near, far, background=(1.0, 1.0, 1.0), anti_aliasing=True, K=None,
                 orig_size=None, device=0):
        self._lib = _lib.load()
        self.image_size, self.texture_size = int(image_size), int(texture_size)
        self.K = np.ascontiguousarray(K if K is not None else
                                      [[image_size, 0.0, image_size // 2], [0.0, image_size, image_size // 2], [0.0, 0.0, 1.0]],
                                      dtype=np.float32).reshape(3, 3)
        self.orig_size = float(image_size if orig_size is None else orig_size)
        bg = np.ascontiguousarray(background, dtype=np.float32).reshape(3)
        self._h = C.c_void_p()
        _lib.check(self._lib.bf_texfit_create(int(device), self.image_size, self.texture_size, float(near), float(far),
                                              _lib.fptr(bg), int(bool(anti_aliasing)), C.byref(self._h)), "bf_texfit_create")
        self._n_faces = [0, 0]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bf_texfit_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_mesh(self, which, mesh):
        v, f, t = _mesh(mesh)
        if t.shape[1] != self.texture_size:
            raise ValueError(f"texture size {t.shape[1]} != {self.texture_size}")
        _lib.check(self._lib.bf_texfit_set_mesh(self._h, int(which), len(v), _lib.fptr(v), len(f), _lib.iptr(f), _lib.fptr(t)),
                   "bf_texfit_set_mesh")
        self._n_faces[which] = len(f)

    @staticmethod
    def _view(pose):
        pose = np.asarray(pose, np.float64)
        return (np.ascontiguousarray(pose[:3, :3], dtype=np.float32), np.ascontiguousarray(pose[:3, 3], dtype=np.float32))

    def render_rgb(self, which, pose):
        """`Renderer.render_rgb(vertices, faces, textures, R=R, t=t)` of neural_renderer for the mesh in slot `which`;
        pose = world-to-camera 4x4 -> rgb[3, image_size, image_size]"""
        R, t = self._view(pose)
        out = np.empty((3, self.image_size, self.image_size), np.float32)
        _lib.check(self._lib.bf_texfit_render(self._h, int(which), _lib.fptr(R), _lib.fptr(t), _lib.fptr(self.K),
                                              self.orig_size, _lib.fptr(out)), "bf_texfit_render")
        return out

    def render_texture(self, uv, uv_faces, textures=None):
        """`Renderer.render_texture(filename_obj, textures)` of neural_renderer (renderer.py:294-346) with the OBJ's contents as
        arrays: uv[n,2] = its `vt` lines, uv_faces[NF,3] = the 0-based vt indices of its faces (same face order as the fitted
        mesh), textures (default: the fitted mesh's current ones) -> (rgb[3, image_size, image_size], depth[image_size, image_size]).
        The UV-space image of the fitted textures: every face is drawn at its UV triangle, front and back."""
        t = self.textures() if textures is None else np.ascontiguousarray(textures, dtype=np.float32)
        uv = np.asarray(uv, np.float64).reshape(-1, 2)
        verts = np.ascontiguousarray(np.concatenate([uv * 2.0 - 1.0, np.ones((len(uv), 1))], 1), dtype=np.float32)      # :303-304
        f = np.ascontiguousarray(np.asarray(uv_faces).reshape(-1, 3), dtype=np.int32)
        if len(f) != len(t):
            raise ValueError("one UV triangle per textured face")
        faces = np.ascontiguousarray(np.concatenate([f, f[:, ::-1]], 0))                      # fill back (:338-339)
        tex = np.ascontiguousarray(np.concatenate([t, t.transpose(0, 3, 2, 1, 4)], 0))       # textures.permute((0, 1, 4, 3, 2, 5)) (:340)
        rgb = np.empty((3, self.image_size, self.image_size), np.float32)
        depth = np.empty((self.image_size, self.image_size), np.float32)
        _lib.check(self._lib.bf_texfit_render_ndc(self._h, len(verts), _lib.fptr(verts), len(faces), _lib.iptr(faces), _lib.fptr(tex),
                                                  _lib.fptr(rgb), _lib.fptr(depth)), "bf_texfit_render_ndc")
        return rgb, depth

    def step(self, pose, lr):
        """one iteration (:262-270) from this view -> the L1 loss before the step"""
        R, t = self._view(pose)
        loss = C.c_double()
        _lib.check(self._lib.bf_texfit_step(self._h, _lib.fptr(R), _lib.fptr(t), _lib.fptr(self.K), self.orig_size,
                                            float(lr), C.byref(loss)), "bf_texfit_step")
        return loss.value

    def loss_grad(self, pose):
        """-> (L1 loss, d loss / d textures of the fitted mesh) from this view, no step"""
        R, t = self._view(pose)
        ts, loss = self.texture_size, C.c_double()
        grad = np.empty((self._n_faces[self.FITTED], ts, ts, ts, 3), np.float32)
        _lib.check(self._lib.bf_texfit_loss_grad(self._h, _lib.fptr(R), _lib.fptr(t), _lib.fptr(self.K), self.orig_size,
                                                 C.byref(loss), _lib.fptr(grad)), "bf_texfit_loss_grad")
        return loss.value, grad

    def textures(self):
        n = self._n_faces[self.FITTED]
        ts = self.texture_size
        out = np.empty((n, ts, ts, ts, 3), np.float32)
        _lib.check(self._lib.bf_texfit_get_textures(self._h, _lib.fptr(out)), "bf_texfit_get_textures")
        return out


def to8b(x):
    """texture_fitting.py:41"""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def render_texture_map(renderer, uv, uv_faces, textures=None):
    """`render_texture_map(renderer, objdir, textures)` (smplify/texture_fitting.py:149-171, morph=False as the loop calls it, :298):
    the UV-space texture image as uint8 [H, W, 3] in the channel order the reference writes to smpl.png (BGR: its [:, :, ::-1])."""
    rgb, _ = renderer.render_texture(uv, uv_faces, textures)
    return to8b(rgb.transpose(1, 2, 0)[:, :, ::-1])


class TextureFitting:
    """`TextureFitting(render_img_size, lrate, iter_num)` of the reference (:173-190); `fit` = the loop of `__call__`."""

    def __init__(self, render_img_size=512, lrate=1e-2, iter_num=200, logging=False, device=0, seed=None):
        self.img_size, self.lrate, self.iter_num, self.logging, self.device = render_img_size, lrate, iter_num, logging, device
        self.rng = np.random.default_rng(seed)

    def view(self, i, round_poses, center, dist):
        """the view of iteration i (:257-263): the ring first, then uniformly random spherical coordinates"""
        if i < ROUND_VIEW_ITERS * len(round_poses):
            return round_poses[i % len(round_poses)]
        return np.linalg.inv(sphere2rot(dist, self.rng.uniform(0, np.pi), self.rng.uniform(0, np.pi * 2), t=center))

    def fit(self, smpld_mesh, scan_mesh, poses=None):
        """-> (fitted textures [NF,ts,ts,ts,3], losses[iter_num]); `poses` overrides the view schedule (tests)"""
        smpld_mesh, scan_mesh = _mesh(smpld_mesh), _mesh(scan_mesh)
        center, dist = scene_bound(scan_mesh[0])
        r = Renderer(self.img_size, smpld_mesh[2].shape[1], near=0.0, far=2.0 * dist, device=self.device)
        try:
            r.set_mesh(Renderer.TARGET, scan_mesh)
            r.set_mesh(Renderer.FITTED, smpld_mesh)
            ring = gen_cam_views(center, ROUND_VIEWS, dist, gl=True)
            losses = []
            for i in range(self.iter_num):
                pose = poses[i] if poses is not None else self.view(i, ring, center, dist)
                losses.append(r.step(pose, self.lrate))
                if self.logging:
                    print(f"texture fitting iter {i}, loss {losses[-1]}")
            return r.textures(), np.asarray(losses)
        finally:
            r.close()
