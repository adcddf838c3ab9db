"""Hand-derived gradients of the SMPLify objective, in numpy.  TEST INFRASTRUCTURE ONLY.

This is the *derivation check* for the HIP kernels: the same forward / reverse sweeps the device
code performs (csrc/fit_kernels.hip), written in vectorised numpy at selectable precision and held
against torch.autograd of oracle/smplify_oracle.py in tests/test_analytic_vs_autograd.py (float64,
agreement ~1e-12).  Only tests/ may import it.

Structure exploited: with the keypoint-only objective only the first 25 of
the 49 joints enter the loss (reference smplify/loss.py:163), i.e. 14 chain joints and 11 selector
vertices; everything else in the mesh carries no gradient.  So one iteration needs
  J(beta)      = J_regressor v_template + (J_regressor shapedirs) beta        (pre-contracted)
  v_k(beta,th) for the 11 selector vertices only (their 33 posedirs columns)
and the full 6890-vertex mesh is evaluated once, for the returned `vertices`.

Reference sites restated: smplx 0.1.13 lbs, smplify/smplify.py:189-190 (similarity),
smplify/loss.py:22-61,132-136,197-216 (projection, GMoF, priors), smplify/prior.py:181-196 (merged
GMM), torch Adam single-tensor path.
"""
from __future__ import annotations

import numpy as np

SIGMA = 100.0
W_POSE, W_ANGLE, W_SHAPE = 4.78 ** 2, 15.2 ** 2, 5.0 ** 2
ANGLE_IDX = np.array([52, 55, 9, 12])
ANGLE_SIGN = np.array([1.0, -1.0, -1.0, -1.0])
N_LOSS_JOINTS = 25


def build_fit_tables(model, dtype=np.float64):
    """Model-level constants of the sparse fit (computed once per model, in float64)."""
    jr = model["J_regressor"].astype(np.float64)
    vt = model["v_template"].astype(np.float64)
    sd = model["shapedirs"].astype(np.float64)
    nj = jr.shape[0]
    src = np.asarray(model["joint_map"][:N_LOSS_JOINTS])
    sel_vertex = []            # unique selector vertices, in first-use order
    kind, index = [], []       # per loss joint: 0 = chain joint `index`, 1 = selected vertex slot `index`
    for s in src:
        if s < nj:
            kind.append(0)
            index.append(int(s))
        else:
            vid = int(model["selector_ids"][s - nj])
            if vid not in sel_vertex:
                sel_vertex.append(vid)
            kind.append(1)
            index.append(sel_vertex.index(vid))
    sel = np.array(sel_vertex, dtype=np.int64)
    cols = (3 * sel[:, None] + np.arange(3)[None]).reshape(-1)
    return {
        "parents": np.asarray(model["parents"], dtype=np.int64),
        "J_template": (jr @ vt).astype(dtype),                              # [NJ,3]
        "J_dirs": np.einsum("jv,vkl->jkl", jr, sd).astype(dtype),          # [NJ,3,NB]
        "sel": sel,
        "sel_template": model["v_template"][sel].astype(dtype),             # [S,3]
        "sel_shapedirs": model["shapedirs"][sel].astype(dtype),             # [S,3,NB]
        "sel_posedirs": model["posedirs"][:, cols].astype(dtype).reshape(-1, len(sel), 3),  # [P,S,3]
        "sel_weights": model["lbs_weights"][sel].astype(dtype),             # [S,NJ]
        "kind": np.array(kind), "index": np.array(index),
    }


def build_views(problem, dtype=np.float64):
    """Per-view 3x4 pixel projection K [R|t] with w2c = inverse(c2w) (smplify.py:131-135) and the
    keypoint table; a view without a detection keeps its slot with confidence 0 (loss.py:157) and the
    divisor stays len(use_frames) (loss.py:197)."""
    c2w = np.asarray(problem["c2ws"], dtype=np.float32).astype(np.float64)
    w2c = np.linalg.inv(c2w)
    K = np.asarray(problem["Ks"], dtype=np.float32).astype(np.float64)
    P = np.einsum("vij,vjk->vik", K, w2c[:, :3, :])
    kp = np.zeros((len(c2w), N_LOSS_JOINTS, 3))
    for v, k in enumerate(problem["keypoints"]):
        if k is not None:
            kp[v] = np.asarray(k["pose"], dtype=np.float32)
    return P.astype(dtype), kp.astype(dtype), len(problem["use_frames"])


# ----------------------------------------------------------------------------------------------
def rodrigues_fwd(theta):
    """theta[N,3] -> R[N,3,3] with the smplx quirk angle = ||theta + 1e-8||."""
    u = theta + theta.dtype.type(1e-8)
    a = np.sqrt((u * u).sum(1))
    n = theta / a[:, None]
    K = np.zeros((len(theta), 3, 3), dtype=theta.dtype)
    K[:, 0, 1], K[:, 0, 2] = -n[:, 2], n[:, 1]
    K[:, 1, 0], K[:, 1, 2] = n[:, 2], -n[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -n[:, 1], n[:, 0]
    KK = K @ K
    s, c = np.sin(a), np.cos(a)
    R = np.eye(3, dtype=theta.dtype)[None] + s[:, None, None] * K + (1 - c)[:, None, None] * KK
    return R, (u, a, K, KK, s, c)


def rodrigues_bwd(theta, cache, dR):
    u, a, K, KK, s, c = cache
    Kt = np.swapaxes(K, 1, 2)
    da = c * (dR * K).sum((1, 2)) + s * (dR * KK).sum((1, 2))
    H = s[:, None, None] * dR + (1 - c)[:, None, None] * (dR @ Kt + Kt @ dR)
    dn = np.stack([H[:, 2, 1] - H[:, 1, 2], H[:, 0, 2] - H[:, 2, 0], H[:, 1, 0] - H[:, 0, 1]], 1)
    da = da - (dn * theta).sum(1) / (a * a)
    return dn / a[:, None] + (da / a)[:, None] * u


def loss_grad(tab, gmm_bufs, views, params, c=0.3, imsize=512):
    """Objective value, its four terms and the analytic gradient w.r.t. the 86 optimised scalars."""
    dt = tab["J_template"].dtype
    P, kp, ndiv = views
    means, prec, nllw = (np.asarray(x, dtype=dt) for x in gmm_bufs)
    t = np.asarray(params["global_transl"], dtype=dt)
    s = dt.type(np.asarray(params["scale"]).reshape(-1)[0])
    pose = np.asarray(params["pose"], dtype=dt)
    beta = np.asarray(params["betas"], dtype=dt)
    theta = np.concatenate([np.asarray(params["global_orient"], dtype=dt), pose]).reshape(-1, 3)
    par = tab["parents"]
    nj = len(par)

    # ---- forward --------------------------------------------------------------------------
    R, rcache = rodrigues_fwd(theta)
    J = tab["J_template"] + tab["J_dirs"] @ beta
    feat = (R[1:] - np.eye(3, dtype=dt)).reshape(-1)
    GR = np.zeros((nj, 3, 3), dtype=dt)
    Gt = np.zeros((nj, 3), dtype=dt)
    rel = J.copy()
    rel[1:] -= J[par[1:]]
    GR[0], Gt[0] = R[0], J[0]
    for i in range(1, nj):
        GR[i] = GR[par[i]] @ R[i]
        Gt[i] = GR[par[i]] @ rel[i] + Gt[par[i]]
    At = Gt - np.einsum("jab,jb->ja", GR, J)
    vs = tab["sel_template"] + tab["sel_shapedirs"] @ beta
    vp = vs + np.einsum("p,psk->sk", feat, tab["sel_posedirs"])
    W = tab["sel_weights"]
    TR = np.einsum("sj,jab->sab", W, GR)
    Tt = W @ At
    vsel = np.einsum("sab,sb->sa", TR, vp) + Tt
    X = np.where(tab["kind"][:, None] == 0, Gt[np.where(tab["kind"] == 0, tab["index"], 0)],
                 vsel[np.where(tab["kind"] == 1, tab["index"], 0)])
    sc = s * dt.type(c)
    Y = X + t
    Xw = Y * sc
    pix = np.einsum("vab,jb->vja", P[:, :, :3], Xw) + P[:, None, :, 3]
    uv = pix[:, :, :2] / pix[:, :, 2:3]
    coeff = dt.type(imsize / 1024.0)
    r = (kp[:, :, :2] - uv) / coeff
    s2 = dt.type(SIGMA * SIGMA)
    rho = s2 * r * r / (s2 + r * r)
    conf2 = kp[:, :, 2] ** 2
    loss_2d = (conf2 * rho.sum(-1)).sum() / ndiv

    d = pose[None] - means
    pd = np.einsum("mij,mj->mi", prec, d)
    q = 0.5 * (pd * d).sum(1) - np.log(nllw)
    mstar = int(np.argmin(q))
    e = np.exp(pose[ANGLE_IDX] * ANGLE_SIGN.astype(dt)) ** 2
    terms = {"reprojection_loss": loss_2d, "pose_prior_loss": W_POSE * q[mstar],
             "angle_prior_loss": W_ANGLE * e.sum(), "shape_prior_loss": W_SHAPE * (beta * beta).sum()}
    loss = sum(terms.values())

    # ---- reverse ----------------------------------------------------------------------------
    drho = 2 * s2 * s2 * r / (s2 + r * r) ** 2
    duv = conf2[:, :, None] * drho * (-1.0 / coeff) / ndiv
    dpix = np.concatenate([duv / pix[:, :, 2:3], -(duv * uv).sum(-1, keepdims=True) / pix[:, :, 2:3]], -1)
    dXw = np.einsum("vab,vja->jb", P[:, :, :3], dpix)
    g_t = dXw.sum(0) * sc
    g_s = (dXw * Y).sum() * dt.type(c)
    dX = dXw * sc
    dGt = np.zeros_like(Gt)
    dvsel = np.zeros_like(vsel)
    for k in range(N_LOSS_JOINTS):
        if tab["kind"][k] == 0:
            dGt[tab["index"][k]] += dX[k]
        else:
            dvsel[tab["index"][k]] += dX[k]
    # skinning of the selector vertices
    dTR = dvsel[:, :, None] * vp[:, None, :]
    dvp = np.einsum("sab,sa->sb", TR, dvsel)
    dGR = np.einsum("sj,sab->jab", W, dTR)
    dAt = W.T @ dvsel
    dfeat = np.einsum("psk,sk->p", tab["sel_posedirs"], dvp)
    g_beta = np.einsum("skl,sk->l", tab["sel_shapedirs"], dvp)
    # A_i.t = G_i.t - G_i.R J_i
    dGt += dAt
    dGR -= dAt[:, :, None] * J[:, None, :]
    dJ = -np.einsum("jab,ja->jb", GR, dAt)
    # kinematic chain, leaves to root
    dR = np.zeros_like(R)
    dJ_direct = dJ.copy()
    drel_all = np.zeros_like(J)
    for i in range(nj - 1, 0, -1):
        p = par[i]
        dR[i] = GR[p].T @ dGR[i]
        dGR[p] += dGR[i] @ R[i].T + np.outer(dGt[i], rel[i])
        drel = GR[p].T @ dGt[i]
        drel_all[i] = drel
        dGt[p] += dGt[i]
        dJ[i] += drel
        dJ[p] -= drel
    dR[0] = dGR[0]
    dJ[0] += dGt[0]
    drel_all[0] = dGt[0]
    dR[1:] += dfeat.reshape(nj - 1, 3, 3)
    g_theta = rodrigues_bwd(theta, rcache, dR)
    g_beta = g_beta + np.einsum("jkl,jk->l", tab["J_dirs"], dJ)
    # priors
    psym = 0.5 * (prec[mstar] + prec[mstar].T)
    g_pose = g_theta[1:].reshape(-1) + W_POSE * (psym @ d[mstar])
    g_pose[ANGLE_IDX] += W_ANGLE * 2.0 * e * ANGLE_SIGN.astype(dt)
    g_beta = g_beta + 2.0 * W_SHAPE * beta
    grads = {"global_transl": g_t, "scale": np.array([g_s]), "pose": g_pose, "betas": g_beta,
             "global_orient": g_theta[0]}
    aux = {"joints25_world": Xw, "gmm_component": mstar, "q": q,
           # intermediates in the order csrc/fit_kernels.hip dumps them (debug hook)
           "dump": [R, J, GR, Gt, vp, vsel, np.concatenate([g_t, [g_s]]), dGR, dGt, dR,
                    g_theta, q, dfeat, dJ_direct, drel_all]}
    return float(loss), {k: float(v) for k, v in terms.items()}, grads, aux


PARAM_ORDER = ("global_transl", "scale", "pose", "betas", "global_orient")   # smplify.py:167-171
PARAM_LR = {"global_transl": 0.1, "scale": 0.1, "pose": 1e-2, "betas": 1e-2, "global_orient": 1e-2}


def adam_step(params, grads, state, step, dtype, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update, `step` counts from 1."""
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
    for k in PARAM_ORDER:
        g = grads[k].astype(dtype)
        m, v = state.setdefault(k, (np.zeros_like(g), np.zeros_like(g)))
        m = m + (g - m) * dtype(1.0 - beta1)
        v = v * dtype(beta2) + dtype(1.0 - beta2) * g * g
        denom = np.sqrt(v) / dtype(bc2_sqrt) + dtype(eps)
        params[k] = params[k] - dtype(PARAM_LR[k] / bc1) * (m / denom)
        state[k] = (m, v)


def fit(model, gmm_bufs, problem, num_iters=100, dtype=np.float64, snapshots=()):
    """The loop of smplify.py:177-213 driven by the analytic gradient; returns stepped parameters."""
    tab = build_fit_tables(model, dtype)
    views = build_views(problem, dtype)
    params = {"global_transl": np.zeros(3, dtype), "scale": np.ones(1, dtype),
              "pose": np.asarray(problem["init_pose"][0, 3:], dtype=dtype),
              "betas": np.asarray(problem["init_betas"][0], dtype=dtype),
              "global_orient": np.asarray(problem["init_pose"][0, :3], dtype=dtype)}
    state, snaps, losses = {}, {}, []
    for i in range(num_iters):
        loss, _, grads, _ = loss_grad(tab, gmm_bufs, views, params, problem.get("constant_scale", 0.3),
                                      problem["imsize"])
        losses.append(loss)
        adam_step(params, grads, state, i + 1, dtype)
        if (i + 1) in snapshots:
            snaps[i + 1] = {k: v.copy() for k, v in params.items()}
    return params, snaps, losses
