"""Bring-up helper (not a test): compare the kernel's first-iteration intermediates with the
analytic numpy restatement.  Usage on the GPU box: python tests/gpu_debug.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N          # noqa: E402
from bodyfitting_amd import synthetic as S       # noqa: E402
from oracle import analytic as A                 # noqa: E402

NAMES = ["R", "J", "G=[GR|Gt]", "vp", "vsel", "g_t,g_s", "dGR", "dGt", "dR", "gth", "q", "dfeat", "dJ", "drel"]


def main():
    model, gmm = S.make_model("smpl"), S.make_gmm()
    gb = S.gmm_buffers(gmm)
    prob = S.make_problem(model, 0, 48)
    dev = N.DeviceModel(model, gmm)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, 48)
    b.set_cameras(c2w, K)
    b.set_keypoints(kp, ndiv)
    b.set_init(betas, pose)
    params = {"global_transl": np.array([0.02, -0.01, 0.03]), "scale": np.array([1.1]),
              "pose": prob["init_pose"][0, 3:].astype(np.float64), "betas": np.linspace(-0.5, 0.5, 10),
              "global_orient": prob["init_pose"][0, :3].astype(np.float64)}
    b.set_params(N.pack_params(params)[None])
    terms, grads = b.loss_grad()
    tab, views = A.build_fit_tables(model), A.build_views(prob)
    loss, t64, g64, aux = A.loss_grad(tab, gb, views, params)
    print("terms gpu", terms[0], "oracle", list(t64.values()))
    d = aux["dump"]
    aux["dump"] = d[:2] + [np.concatenate([d[2], d[3][:, :, None]], 2)] + d[4:]      # kernel keeps rows [GR_r | Gt_r]
    sizes = [a.size for a in aux["dump"]]
    dump = b.debug_dump(int(sum(sizes)))
    o = 0
    for name, ref in zip(NAMES, aux["dump"]):
        got = dump[o:o + ref.size].reshape(ref.shape)
        o += ref.size
        err = np.abs(got - ref).max()
        print(f"{name:12s} max|ref|={np.abs(ref).max():.4e} maxerr={err:.3e} rel={err / (np.abs(ref).max() + 1e-30):.2e}")
    got = N.split_params(grads[0])
    for k, v in g64.items():
        err = np.abs(got[k] - v).max()
        print(f"grad {k:14s} max|ref|={np.abs(v).max():.4e} maxerr={err:.3e} rel={err / np.abs(v).max():.2e}")
    b.fit(100)
    print("timing", b.last_timing())
    p = N.split_params(b.get_params()[0])
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg2_48view_100it_f0.npz"))
    b2 = N.FrameBatch(dev, 1, 48)
    b2.set_cameras(c2w, K); b2.set_keypoints(kp, ndiv); b2.set_init(betas, pose)
    b2.fit(100)
    print("timing (from init)", b2.last_timing())
    p = N.split_params(b2.get_params()[0])
    for k in ("global_transl", "scale", "pose", "betas", "global_orient"):
        print("fit100", k, np.abs(p[k] - g[f"it100_{k}"]).max())
    for _ in range(3):
        b2.set_init(betas, pose)
        b2.fit(100)
        print("timing again", b2.last_timing())


if __name__ == "__main__":
    main()
