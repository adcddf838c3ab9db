"""CPU oracle for the texture-fitting loop (reference smplify/texture_fitting.py:220-275).  TEST INFRASTRUCTURE ONLY.

The loop optimises the per-face textures of the SMPL+D mesh with Adam so that its renderings match renderings of the
textured scan: every iteration renders both meshes from one view with `neural_renderer` (thirdparty/neural_renderer, a CUDA
extension: absent here and unbuildable without the CUDA toolkit + ATen - PARITY UNPINNED), takes
`loss = sum |scan_img - smpl_img|` and steps the textures.  Only the textures are differentiated, so of neural_renderer the
loop exercises exactly this, restated below in numpy (float32, the kernels' operation order):

  * `projection`                      neural_renderer/projection.py:6-42 (zero distortion: `Renderer.__init__`, renderer.py:39-41)
  * `lighting` with ambient 1, directional 0: textures x 1 (lighting.py:33-37; renderer.py:180-190)
  * `forward_face_index_map` 1 + 2    cuda/rasterize_cuda_kernel.cu:24-174: back-face cull, the three edge tests at the pixel centre,
                                      barycentric weights from the inverted pixel-space triangle (clamped to [0,1], renormalised),
                                      perspective-correct depth, near/far, z-buffer with strict `<` in face order
  * `forward_texture_sampling`        kernel.cu:177-252: texture index = weight x (ts - 1) x depth / z_vertex, 8-corner blend
  * `forward_background`, flip, 2x2 average pooling (anti-aliasing)   rasterize.py:181-190,300-318
  * `backward_textures`               kernel.cu:498-540 (atomicAdd of sampling weight x dL/drgb), through the pooling / flip / mask
  * L1 loss, torch.optim.Adam on the textures   texture_fitting.py:242-244,266-270

View schedule: `gen_cam_views` (utils/renderer.py:7-25) and `sphere2rot` (texture_fitting.py:63-83).

Pinned by known answers only (tests/test_texfit_oracle.py): a constant-colour triangle renders that colour inside and the
background outside with half-tones on the anti-aliased edge; a texture that is 1 at one corner's end of the barycentric
cube renders the barycentric coordinate; the nearer of two overlapping triangles wins; the gradient equals finite
differences of the loss with respect to the textures.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def gen_cam_views(center, viewnum, dist, gl=False):
    """utils/renderer.py:7-25: world-to-camera poses on a horizontal ring around `center`"""
    def viewmatrix(z, up, translation):
        vec3 = z / np.linalg.norm(z)
        up = up / np.linalg.norm(up)
        vec1 = np.cross(up, vec3)
        vec2 = np.cross(vec3, vec1)
        view = np.stack([vec1, vec2, vec3, translation], axis=1)
        return np.concatenate([view, np.array([[0, 0, 0, 1]])], axis=0)
    cv2gl = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]]) if gl else np.eye(4)
    poses = []
    for theta in np.linspace(0, 2 * np.pi, viewnum + 1)[:-1]:
        z = np.array([np.cos(theta), 0, -np.sin(theta)]) * dist
        poses.append(cv2gl @ np.linalg.inv(viewmatrix(z, np.array([0, 1, 0]), z + center)))
    return poses


def sphere2rot(rad, theta, phi, t=(0, 0, 0)):
    """texture_fitting.py:63-83: camera-to-world pose looking at `t` from spherical coordinates"""
    def normalize(x):
        return x / np.linalg.norm(x)
    transl = np.array([rad * np.sin(theta) * np.sin(phi), rad * np.cos(theta), rad * np.sin(theta) * np.cos(phi)])
    z = normalize(-transl)
    right = np.array([np.sin(phi + np.pi / 2), 0, np.cos(phi + np.pi / 2)])
    y = normalize(np.cross(z, right))
    x = normalize(np.cross(y, z))
    R = np.eye(4)
    R[:3, :3] = np.stack([x, y, z], axis=1)
    R[:3, 3] = transl + np.array(t)
    return R


def project(vertices, K, R, t, orig_size, eps=F32(1e-9)):
    """projection.py:6-42 with zero distortion -> [NV,3] (u, v in [-1,1], z)"""
    # vertices @ R^T + t, written out (k ascending, every product and sum rounded: what the device kernel does; the order
    # cuBLAS uses for the reference's 3-term dot products is not knowable here)
    a, b, c = (np.asarray(vertices, F32)[:, k] for k in range(3))
    Rm, tv = np.asarray(R, F32).reshape(3, 3), np.asarray(t, F32).reshape(3)
    x, y, z = (((a * Rm[r, 0] + b * Rm[r, 1]).astype(F32) + c * Rm[r, 2]).astype(F32) + tv[r] for r in range(3))
    x, y, z = x.astype(F32), y.astype(F32), z.astype(F32)
    x_ = (x / (z + eps)).astype(F32)
    y_ = (y / (z + eps)).astype(F32)
    K = np.asarray(K, F32).reshape(3, 3)
    u = ((x_ * K[0, 0] + y_ * K[0, 1]).astype(F32) + K[0, 2]).astype(F32)
    w = ((x_ * K[1, 0] + y_ * K[1, 1]).astype(F32) + K[1, 2]).astype(F32)
    os_ = F32(orig_size)
    w = (os_ - w).astype(F32)
    u = (F32(2) * (u - os_ / F32(2)) / os_).astype(F32)
    w = (F32(2) * (w - os_ / F32(2)) / os_).astype(F32)
    return np.stack([u, w, z], 1).astype(F32)


def rasterize(face_verts, image_size, near, far):
    """forward_face_index_map (kernel.cu:24-174).  face_verts[NF,3,3] = projected (x, y, z) per corner.
    -> face_index[is,is] int32 (-1 = none), weight[is,is,3], depth[is,is] (far where empty); row yi = pixel row of the kernel."""
    f = np.asarray(face_verts, F32).reshape(-1, 9)
    is_ = int(image_size)
    nf = len(f)
    front = ~((f[:, 7] - f[:, 1]) * (f[:, 3] - f[:, 0]) < (f[:, 4] - f[:, 1]) * (f[:, 6] - f[:, 0]))
    p = (F32(0.5) * (f.reshape(nf, 3, 3)[:, :, :2] * F32(is_) + F32(is_) - F32(1))).astype(F32)        # [nf,3,2]
    inv = np.zeros((nf, 9), F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv[:, 0] = p[:, 1, 1] - p[:, 2, 1]; inv[:, 1] = p[:, 2, 0] - p[:, 1, 0]; inv[:, 2] = p[:, 1, 0] * p[:, 2, 1] - p[:, 2, 0] * p[:, 1, 1]
        inv[:, 3] = p[:, 2, 1] - p[:, 0, 1]; inv[:, 4] = p[:, 0, 0] - p[:, 2, 0]; inv[:, 5] = p[:, 2, 0] * p[:, 0, 1] - p[:, 0, 0] * p[:, 2, 1]
        inv[:, 6] = p[:, 0, 1] - p[:, 1, 1]; inv[:, 7] = p[:, 1, 0] - p[:, 0, 0]; inv[:, 8] = p[:, 0, 0] * p[:, 1, 1] - p[:, 1, 0] * p[:, 0, 1]
        den = (p[:, 2, 0] * (p[:, 0, 1] - p[:, 1, 1]) + p[:, 0, 0] * (p[:, 1, 1] - p[:, 2, 1]) + p[:, 1, 0] * (p[:, 2, 1] - p[:, 0, 1])).astype(F32)
        inv = (inv / den[:, None]).astype(F32)
    face_index = np.full((is_, is_), -1, np.int32)
    weight = np.zeros((is_, is_, 3), F32)
    depth = np.full((is_, is_), F32(far), F32)
    for yi in range(is_):
        yp = F32((2.0 * yi + 1 - is_) / is_)
        for xi in range(is_):
            xp = F32((2.0 * xi + 1 - is_) / is_)
            out = (((yp - f[:, 1]) * (f[:, 3] - f[:, 0]) < (xp - f[:, 0]) * (f[:, 4] - f[:, 1])) |
                   ((yp - f[:, 4]) * (f[:, 6] - f[:, 3]) < (xp - f[:, 3]) * (f[:, 7] - f[:, 4])) |
                   ((yp - f[:, 7]) * (f[:, 0] - f[:, 6]) < (xp - f[:, 6]) * (f[:, 1] - f[:, 7])))
            cand = np.nonzero(front & ~out)[0]
            if len(cand) == 0:
                continue
            w = (inv[cand].reshape(-1, 3, 3)[:, :, 0] * F32(xi) + inv[cand].reshape(-1, 3, 3)[:, :, 1] * F32(yi) + inv[cand].reshape(-1, 3, 3)[:, :, 2]).astype(F32)
            w = np.minimum(np.maximum(w, F32(0)), F32(1))
            w = (w / w.sum(1, keepdims=True, dtype=F32)).astype(F32)
            with np.errstate(divide="ignore"):
                zp = (F32(1) / (w[:, 0] / f[cand, 2] + w[:, 1] / f[cand, 5] + w[:, 2] / f[cand, 8])).astype(F32)
            ok = ~((zp <= F32(near)) | (F32(far) <= zp))
            if not ok.any():
                continue
            zz = np.where(ok, zp, np.inf)
            k = int(np.argmin(zz))                     # first minimum = strict `<` in face order
            if zz[k] < depth[yi, xi]:
                depth[yi, xi] = zp[k]; face_index[yi, xi] = cand[k]; weight[yi, xi] = w[k]
    return face_index, weight, depth


def sample_textures(face_verts, textures, face_index, weight, depth, eps=F32(1e-4)):
    """forward_texture_sampling (kernel.cu:177-252) -> rgb[is,is,3], sampling_index[is,is,8], sampling_weight[is,is,8]"""
    f = np.asarray(face_verts, F32).reshape(-1, 9)
    ts = textures.shape[1]
    is_ = face_index.shape[0]
    rgb = np.zeros((is_, is_, 3), F32)
    sidx = np.zeros((is_, is_, 8), np.int32)
    sw = np.zeros((is_, is_, 8), F32)
    tex = np.asarray(textures, F32).reshape(len(f), ts * ts * ts, 3)
    for yi, xi in zip(*np.nonzero(face_index >= 0)):
        fi = face_index[yi, xi]
        tif = np.empty(3, F32)
        for k in range(3):
            v = F32(weight[yi, xi, k] * F32(ts - 1) * (depth[yi, xi] / f[fi, 3 * k + 2]))
            tif[k] = min(max(v, F32(0)), F32(ts - 1) - eps)
        px = np.zeros(3, F32)
        for pn in range(8):
            w = F32(1)
            ti = [0, 0, 0]
            for k in range(3):
                fl = int(tif[k])
                if (pn >> k) % 2 == 0:
                    w = F32(w * (F32(1) - (tif[k] - F32(fl)))); ti[k] = fl
                else:
                    w = F32(w * (tif[k] - F32(fl))); ti[k] = fl + 1
            isc = ti[0] * ts * ts + ti[1] * ts + ti[2]
            px = (px + w * tex[fi, isc]).astype(F32)
            sidx[yi, xi, pn] = isc; sw[yi, xi, pn] = w
        rgb[yi, xi] = px
    return rgb, sidx, sw


def render(verts, faces, textures, K, R, t, orig_size, image_size, near, far, background=(1, 1, 1), anti_aliasing=True, keep=None):
    """Renderer.render_rgb (renderer.py:174-232) for camera_mode='projection', ambient light 1: -> rgb[3, image_size, image_size]"""
    pv = project(verts, K, R, t, orig_size)
    fv = pv[np.asarray(faces, np.int64)]                                 # vertices_to_faces
    is2 = image_size * 2 if anti_aliasing else image_size
    fi, w, d = rasterize(fv, is2, near, far)
    tex = (np.asarray(textures, F32) * F32(1.0)).astype(F32)             # lighting: ambient 1 x (1,1,1)
    rgb, sidx, sw = sample_textures(fv, tex, fi, w, d)
    mask = (fi >= 0).astype(F32)[:, :, None]
    rgb = (rgb * mask + (F32(1) - mask) * np.asarray(background, F32)[None, None, :]).astype(F32)
    img = rgb.transpose(2, 0, 1)[:, ::-1, :]                             # permute + vertical flip
    if anti_aliasing:
        img = img.reshape(3, image_size, 2, image_size, 2).astype(F32)
        img = ((img[:, :, 0, :, 0] + img[:, :, 0, :, 1] + img[:, :, 1, :, 0] + img[:, :, 1, :, 1]) * F32(0.25)).astype(F32)
    if keep is not None:
        keep.update(face_index=fi, sampling_index=sidx, sampling_weight=sw, is2=is2)
    return np.ascontiguousarray(img, F32)


def uv_mesh(uv, uv_faces, textures):
    """What Renderer.render_texture (renderer.py:294-340) builds from an OBJ's `vt` lines and `f v/vt` indices: vertices
    (2u - 1, 2v - 1, 1), the faces followed by the same faces with their corners reversed ("fill back"), the textures followed by
    the same cubes with axes 0 and 2 swapped (textures.permute((0, 1, 4, 3, 2, 5)))."""
    uv = np.asarray(uv, np.float64).reshape(-1, 2)
    verts = np.concatenate([uv * 2.0 - 1.0, np.ones((len(uv), 1))], 1).astype(F32)          # (python floats, then astype(float32): :303-304)
    f = np.asarray(uv_faces, np.int32).reshape(-1, 3)
    t = np.asarray(textures, F32)
    return verts, np.concatenate([f, f[:, ::-1]], 0), np.concatenate([t, t.transpose(0, 3, 2, 1, 4)], 0)


def render_texture(uv, uv_faces, textures, image_size, near, far, background=(1, 1, 1), anti_aliasing=True):
    """Renderer.render_texture -> nr.rasterize_rgbad (rasterize.py:254-330) on vertices that are already normalised device
    coordinates: -> (rgb[3, is, is], depth[is, is]); render_texture_map (texture_fitting.py:149-151) keeps the rgb part."""
    verts, faces, tex = uv_mesh(uv, uv_faces, textures)
    fv = verts[faces.astype(np.int64)]
    is2 = image_size * 2 if anti_aliasing else image_size
    fi, w, d = rasterize(fv, is2, near, far)
    rgb, _, _ = sample_textures(fv, tex, fi, w, d)
    mask = (fi >= 0).astype(F32)[:, :, None]
    rgb = (rgb * mask + (F32(1) - mask) * np.asarray(background, F32)[None, None, :]).astype(F32)
    img = rgb.transpose(2, 0, 1)[:, ::-1, :]
    dep = d[::-1, :]
    if anti_aliasing:
        img = img.reshape(3, image_size, 2, image_size, 2).astype(F32)
        img = ((img[:, :, 0, :, 0] + img[:, :, 0, :, 1] + img[:, :, 1, :, 0] + img[:, :, 1, :, 1]) * F32(0.25)).astype(F32)
        dep = dep.reshape(image_size, 2, image_size, 2).astype(F32)
        dep = ((dep[:, 0, :, 0] + dep[:, 0, :, 1] + dep[:, 1, :, 0] + dep[:, 1, :, 1]) * F32(0.25)).astype(F32)
    return np.ascontiguousarray(img, F32), np.ascontiguousarray(dep, F32)


def texture_grad(grad_img, keep, n_faces, ts, image_size, anti_aliasing=True):
    """dL/dtextures from dL/d(rendered image)[3,is,is]: pooling, flip, background mask, backward_textures (kernel.cu:498-540)"""
    is2 = keep["is2"]
    g = np.asarray(grad_img, F32)
    if anti_aliasing:
        g = np.repeat(np.repeat(g, 2, axis=1), 2, axis=2) * F32(0.25)
    g = g[:, ::-1, :].transpose(1, 2, 0)                                 # un-flip, to [is2,is2,3]
    out = np.zeros((n_faces, ts * ts * ts, 3), F32)
    fi = keep["face_index"]
    for yi, xi in zip(*np.nonzero(fi >= 0)):
        for pn in range(8):
            out[fi[yi, xi], keep["sampling_index"][yi, xi, pn]] += keep["sampling_weight"][yi, xi, pn] * g[yi, xi]
    return out.reshape(n_faces, ts, ts, ts, 3)


class TextureFit:
    """texture_fitting.py:240-275 without the file formats: Adam (torch defaults, lr) on the fitted mesh's textures"""

    def __init__(self, target, mesh, image_size, near, far, lr=1e-2, background=(1, 1, 1), anti_aliasing=True):
        self.target, self.mesh = target, (mesh[0], mesh[1], np.array(mesh[2], F32))
        self.cfg = dict(image_size=image_size, near=near, far=far, background=background, anti_aliasing=anti_aliasing)
        self.lr, self.step_no = lr, 0
        self.m = np.zeros_like(self.mesh[2])
        self.v = np.zeros_like(self.mesh[2])

    def adam(self, g):
        """torch.optim.Adam, single-tensor form (betas 0.9 / 0.999, eps 1e-8), bias corrections in double like torch's python floats"""
        self.step_no += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        self.m = (self.m + (g - self.m) * F32(1 - b1)).astype(F32)
        self.v = (self.v * F32(b2) + g * g * F32(1 - b2)).astype(F32)
        bc1, bc2 = 1 - b1 ** self.step_no, 1 - b2 ** self.step_no
        denom = (np.sqrt(self.v) / F32(np.sqrt(bc2)) + F32(eps)).astype(F32)
        tex = (self.mesh[2] - F32(self.lr / bc1) * (self.m / denom)).astype(F32)
        self.mesh = (self.mesh[0], self.mesh[1], tex)

    def step(self, K, R, t, orig_size):
        keep = {}
        a = render(*self.target, K, R, t, orig_size, **self.cfg)
        b = render(*self.mesh, K, R, t, orig_size, keep=keep, **self.cfg)
        loss = float(np.abs(a - b).astype(np.float64).sum())
        g = texture_grad(np.sign(b - a).astype(F32), keep, len(self.mesh[1]), self.mesh[2].shape[1], self.cfg["image_size"],
                         self.cfg["anti_aliasing"])
        self.adam(g)
        return loss, a, b
