"""CPU oracle for the multi-view SMPLify hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module; the product path (bodyfitting_amd/) never does and fails loudly without its HIP
library.

This is a torch-CPU restatement (forward math + torch.autograd + torch.optim.Adam, i.e.
the same machinery the reference executes) of:

  * the optimisation loop          reference smplify/smplify.py:103-226
  * the keypoint / prior losses    reference smplify/loss.py:22-61,132-230
  * the merged GMM prior           reference smplify/prior.py:143-160,181-196
  * the 49-joint SMPL wrapper      reference models/smpl.py:56-83
  * the SMPL forward pass          smplx==0.1.13 `lbs()` and `SMPL.forward`
                                   (requirements.txt:5 - NOT vendored in /root/reference;
                                   restated from its published semantics)

PARITY PINNING
  * loop, losses, priors, wrapper: pinned.  tests/golden/*.npz were produced by importing the
    unmodified reference modules in the build container (oracle/gen_golden.py) and
    tests/test_oracle_golden.py holds this restatement to them.
  * smplx LBS arithmetic: **parity unpinned** - smplx is an un-vendored dependency and the
    reference holds no test or golden vector for it; the imported reference loop above ran
    on oracle/smplx_standin, which shares `lbs()` below.  Known-answer tests (identity pose,
    single rotation, finite differences) are the only independent pins on it.
"""
from __future__ import annotations

import numpy as np
import torch

SKELETON_LENGTH = 25          # reference smplify/loss.py:17
SIGMA = 100.0                 # loss.py:139
SHAPE_PRIOR_WEIGHT = 5.0      # loss.py:140
ANGLE_PRIOR_WEIGHT = 15.2     # loss.py:140
POSE_PRIOR_WEIGHT = 4.78      # loss.py:141
ANGLE_IDX = (52, 55, 9, 12)   # loss.py:60  (55-3, 58-3, 12-3, 15-3)
ANGLE_SIGN = (1.0, -1.0, -1.0, -1.0)


# ----------------------------------------------------------------------------------------------
# smplx 0.1.13 forward (restated)
# ----------------------------------------------------------------------------------------------

def batch_rodrigues(rot_vecs):
    """[N,3] axis-angle -> [N,3,3].  angle = ||theta + 1e-8|| (added per component *before* the
    norm), direction = theta / angle, R = I + sin K + (1-cos) K K."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    axis = rot_vecs / angle
    s = torch.sin(angle)[:, :, None]
    c = torch.cos(angle)[:, :, None]
    x, y, z = axis[:, 0:1], axis[:, 1:2], axis[:, 2:3]
    o = torch.zeros_like(x)
    K = torch.cat([o, -z, y, z, o, -x, -y, x, o], dim=1).view(n, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype).unsqueeze(0)
    return eye + s * K + (1.0 - c) * torch.bmm(K, K)


def rigid_chain(rot_mats, joints, parents):
    """smplx `batch_rigid_transform`: returns (posed_joints[B,NJ,3], A[B,NJ,4,4])."""
    B, NJ = joints.shape[:2]
    rel = joints.clone()
    rel[:, 1:] = joints[:, 1:] - joints[:, parents[1:]]
    top = torch.cat([rot_mats, rel.unsqueeze(-1)], dim=-1)                       # [B,NJ,3,4]
    bottom = torch.zeros(B, NJ, 1, 4, dtype=joints.dtype)
    bottom[..., 3] = 1.0
    local = torch.cat([top, bottom], dim=-2)                                     # [B,NJ,4,4]
    chain = [local[:, 0]]
    for i in range(1, NJ):
        chain.append(torch.matmul(chain[int(parents[i])], local[:, i]))
    G = torch.stack(chain, dim=1)
    posed = G[:, :, :3, 3]
    j_h = torch.cat([joints, torch.zeros(B, NJ, 1, dtype=joints.dtype)], dim=2).unsqueeze(-1)
    corr = torch.matmul(G, j_h)                                                  # [B,NJ,4,1]
    A = G - torch.nn.functional.pad(corr, [3, 0])
    return posed, A


def lbs(betas, full_pose, m):
    """smplx `lbs()`: (vertices[B,NV,3], posed chain joints[B,NJ,3]).  `m` holds torch tensors
    v_template[NV,3], shapedirs[NV,3,NB], posedirs[P,3NV], J_regressor[NJ,NV], lbs_weights[NV,NJ],
    parents (python list / int array)."""
    B = betas.shape[0]
    v_shaped = m["v_template"].unsqueeze(0) + torch.einsum("bl,mkl->bmk", betas, m["shapedirs"])
    J = torch.einsum("bik,ji->bjk", v_shaped, m["J_regressor"])
    NJ = J.shape[1]
    R = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, NJ, 3, 3)
    eye = torch.eye(3, dtype=betas.dtype)
    feat = (R[:, 1:] - eye).reshape(B, -1)
    v_posed = v_shaped + torch.matmul(feat, m["posedirs"]).view(B, -1, 3)
    posed_j, A = rigid_chain(R, J, m["parents"])
    T = torch.matmul(m["lbs_weights"].unsqueeze(0).expand(B, -1, -1), A.view(B, NJ, 16)).view(B, -1, 4, 4)
    ones = torch.ones(B, v_posed.shape[1], 1, dtype=betas.dtype)
    v_h = torch.matmul(T, torch.cat([v_posed, ones], dim=2).unsqueeze(-1))
    return v_h[:, :, :3, 0], posed_j


def smpl_forward(m, betas, global_orient, body_pose):
    """smplx `SMPL.forward` + the reference wrapper models/smpl.py:69-83.

    Returns dict(vertices[B,NV,3], joints[B,49,3], joints_ori[B,45,3], full_pose[B,72])."""
    full_pose = torch.cat([global_orient, body_pose], dim=1)
    verts, chain_j = lbs(betas, full_pose, m)
    picked = verts[:, m["selector_ids"]]                         # VertexJointSelector: 21 vertices
    joints45 = torch.cat([chain_j, picked], dim=1)
    extra = torch.einsum("bik,ji->bjk", verts, m["J_regressor_extra"])           # smpl.py:72
    joints = torch.cat([joints45, extra], dim=1)[:, m["joint_map"]]              # smpl.py:74-75
    return {"vertices": verts, "joints": joints, "joints_ori": joints45, "full_pose": full_pose}


def smplx_forward(m, betas, global_orient, body_pose, leye_pose, reye_pose, left_hand_pose, right_hand_pose,
                  jaw_pose=None, expression=None, mapped=True):
    """smplx 0.1.13 `SMPLX.forward` (use_pca with 6 components, flat_hand_mean=False, use_face_contour=True) + the
    JointMapper of reference models/utils.py:16-29,75-94.  Returns dict(vertices[B,NV,3],
    joints[B,135,3], full_pose[B,165])."""
    B = betas.shape[0]
    dt = betas.dtype
    z3 = torch.zeros(B, 3, dtype=dt)
    jaw = z3 if jaw_pose is None else jaw_pose.reshape(B, 3)
    lh = torch.einsum("bi,ij->bj", left_hand_pose, m["left_hand_components"])
    rh = torch.einsum("bi,ij->bj", right_hand_pose, m["right_hand_components"])
    full_pose = torch.cat([global_orient, body_pose, jaw, leye_pose.reshape(B, 3), reye_pose.reshape(B, 3), lh, rh], dim=1)
    full_pose = full_pose + m["pose_mean"]
    expr = torch.zeros(B, m["shapedirs"].shape[2] - betas.shape[1], dtype=dt) if expression is None else expression
    verts, chain_j = lbs(torch.cat([betas, expr], dim=1), full_pose, m)
    # dynamic face-contour landmarks: yaw of the neck's global rotation -> row of a 79-entry table
    R = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, -1, 3, 3)
    rel = torch.eye(3, dtype=dt).unsqueeze(0).expand(B, -1, -1)
    for j in m["neck_kin_chain"]:
        rel = torch.bmm(R[:, int(j)], rel)
    sy = torch.sqrt(rel[:, 0, 0] ** 2 + rel[:, 1, 0] ** 2)
    yaw = torch.atan2(-rel[:, 2, 0], sy)
    y = torch.round(torch.clamp(-yaw * 180.0 / np.pi, max=39)).to(torch.long)
    neg = y.lt(0).to(torch.long)
    far = y.lt(-39).to(torch.long)
    y = neg * (far * 78 + (1 - far) * (39 - y)) + (1 - neg) * y
    lmk_faces = torch.cat([m["lmk_faces_idx"].unsqueeze(0).expand(B, -1), m["dynamic_lmk_faces_idx"][y]], 1)        # [B,68]
    lmk_bary = torch.cat([m["lmk_bary_coords"].unsqueeze(0).expand(B, -1, -1), m["dynamic_lmk_bary_coords"][y]], 1)
    tri = m["faces"][lmk_faces]                                                                               # [B,68,3]
    lv = torch.stack([verts[b][tri[b]] for b in range(B)])                                                    # [B,68,3,3]
    landmarks = torch.einsum("blfi,blf->bli", lv, lmk_bary)
    parts = [chain_j, verts[:, m["selector_ids"]]]                                                              # 55 + 21
    if "J_regressor_extra" in m and m["J_regressor_extra"].shape[0] > 0:
        # (not a reference configuration: smplx has no extra regressor; the ABI's all-joints layout puts such rows here, as
        #  models/smpl.py:72-75 does for SMPL)
        parts.append(torch.einsum("bik,ji->bjk", verts, m["J_regressor_extra"]))
    joints = torch.cat(parts + [landmarks], dim=1)                                                              # ... + 68
    return {"vertices": verts, "joints": joints[:, m["joint_map"]] if mapped else joints, "full_pose": full_pose, "dyn_row": y}


def to_torch_model(model, dtype=torch.float32):
    """numpy model dict (bodyfitting_amd.synthetic.make_model) -> torch tensors."""
    out = {}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "J_regressor_extra"):
        if k in model:
            out[k] = torch.as_tensor(np.asarray(model[k]), dtype=dtype)
    out["parents"] = [int(p) for p in model["parents"]]
    out["selector_ids"] = torch.as_tensor(np.asarray(model["selector_ids"]), dtype=torch.long)
    out["joint_map"] = torch.as_tensor(np.asarray(model["joint_map"]), dtype=torch.long)
    for k in ("left_hand_components", "right_hand_components", "pose_mean", "lmk_bary_coords", "dynamic_lmk_bary_coords"):
        if k in model:
            out[k] = torch.as_tensor(np.asarray(model[k]), dtype=dtype)
    for k in ("lmk_faces_idx", "dynamic_lmk_faces_idx", "faces"):
        if k in model and model.get("model_type") == "smplx":
            out[k] = torch.as_tensor(np.asarray(model[k]), dtype=torch.long)
    if "neck_kin_chain" in model:
        out["neck_kin_chain"] = [int(j) for j in model["neck_kin_chain"]]
    return out


# ----------------------------------------------------------------------------------------------
# losses (reference smplify/loss.py, smplify/prior.py)
# ----------------------------------------------------------------------------------------------

def perspective_projection(points, rotation, translation, K):
    """loss.py:22-43.  points[B,N,3], rotation[B,3,3], translation[B,3], K[3,3] -> [B,N,2].
    No epsilon on z and no behind-camera guard, as in the reference."""
    cam = torch.einsum("bij,bkj->bki", rotation, points) + translation.unsqueeze(1)
    pix = torch.einsum("ij,bkj->bki", K, cam)
    return pix[:, :, :2] / pix[:, :, 2:3]


def gmof(x, sigma):
    """loss.py:45-51 Geman-McClure."""
    x2 = x * x
    s2 = sigma * sigma
    return (s2 * x2) / (s2 + x2)


def reprojection_loss(cord, cord_gt, conf, scale_coeff, sigma):
    """loss.py:132-136."""
    err = gmof((cord_gt - cord) / scale_coeff, sigma)
    return ((conf ** 2) * err.sum(dim=-1)).sum(dim=-1)


def angle_prior(pose):
    """loss.py:54-61: exp(theta_k * sign_k) ** 2 on four body-pose dofs."""
    idx = torch.tensor(ANGLE_IDX, dtype=torch.long)
    sign = torch.tensor(ANGLE_SIGN, dtype=pose.dtype)
    return torch.exp(pose[:, idx] * sign) ** 2


def gmm_merged_nll(pose, means, precisions, nll_weights):
    """prior.py:181-196: min over components of 0.5 d'Pd - log(w~).  pose[B,69]."""
    d = pose.unsqueeze(1) - means                                                # [B,M,69]
    pd = torch.einsum("mij,bmj->bmi", precisions, d)
    quad = (pd * d).sum(dim=-1)
    ll = 0.5 * quad - torch.log(nll_weights)
    return torch.min(ll, dim=1)[0]


HANDS_LENGTH, FACE_LENGTH = 42, 68                                  # loss.py:18-19


def multiview_keypoint_loss(w2cs, Ks, keypoints, model_joints, poses, betas, n_use_frames, gmm,
                            imsize=512, sigma=SIGMA, use_hand_face=False):
    """loss.py:139-230 for smpl_type='smpl' (use_hand_face False).

    w2cs[V,4,4], Ks[V,3,3] tensors; keypoints: list of None | float tensor [25,3].
    Returns (total scalar, dict of the four terms) - the dict mirrors loss.py:219-224."""
    scale_coeff = imsize / 1024.0
    per_view, hand, face = [], [], []
    for i in range(len(keypoints)):
        if keypoints[i] is None:
            continue                                                             # loss.py:157
        w2c = w2cs[i]
        uv = perspective_projection(model_joints, w2c[:3, :3].unsqueeze(0), w2c[:3, 3].unsqueeze(0), Ks[i])
        gt, conf = keypoints[i][:, :2], keypoints[i][:, 2]
        per_view.append(reprojection_loss(uv[0, :SKELETON_LENGTH], gt[:SKELETON_LENGTH], conf[:SKELETON_LENGTH], scale_coeff, sigma))
        if use_hand_face:       # keypoints[i] rows: body 25 | hand_left 21 | hand_right 21 | face 68 (already FACE_MAPPING-ordered)
            a, b2, c2 = SKELETON_LENGTH, SKELETON_LENGTH + HANDS_LENGTH // 2, SKELETON_LENGTH + HANDS_LENGTH
            hand.append(reprojection_loss(uv[0, a:b2], gt[a:b2], conf[a:b2], scale_coeff, sigma))
            hand.append(reprojection_loss(uv[0, b2:c2], gt[b2:c2], conf[b2:c2], scale_coeff, sigma))
            face.append(reprojection_loss(uv[0, c2:], gt[c2:], conf[c2:], scale_coeff, sigma))
    loss_2d = torch.sum(torch.stack(per_view, dim=0)) / n_use_frames             # loss.py:197
    if use_hand_face:                                                            # loss.py:199-203
        loss_2d = loss_2d + torch.sum(torch.stack(hand, dim=0)) / n_use_frames
        loss_2d = loss_2d + torch.sum(torch.stack(face, dim=0)) / n_use_frames
        poses = torch.cat([poses, torch.zeros_like(poses[:, :6])], dim=-1)       # loss.py:206-207
    pose_prior = (POSE_PRIOR_WEIGHT ** 2) * gmm_merged_nll(poses, *gmm)
    ang = (ANGLE_PRIOR_WEIGHT ** 2) * angle_prior(poses).sum(dim=-1)
    shape = (SHAPE_PRIOR_WEIGHT ** 2) * (betas ** 2).sum(dim=-1)
    total = loss_2d + pose_prior + ang + shape
    terms = {"reprojection_loss": loss_2d, "pose_prior_loss": pose_prior,
             "angle_prior_loss": ang, "shape_prior_loss": shape}
    return total.sum(), terms


def multview_mask_loss(contours, masks, verts, w2cs, Ks, imsize=512, epsilon=10.0, pairwise="exact"):
    """loss.py:85-130.  contours: list of [C,2] tensors, masks [M,H,W] (0/1), verts [NV,3], w2cs [M,4,4],
    Ks [M,3,3].  pairwise="torch" evaluates the distances literally as the reference does (torch.cdist, which
    for these sizes takes the |a|^2+|b|^2-2ab matmul form: ~1e-2 px of fp32 noise) - used to pin this
    restatement to the reference goldens bit for bit; pairwise="exact" sums (a-b)^2 directly, which is what
    the HIP kernel does."""
    v4 = verts[::4]
    total = 0.0
    uvs = []
    for i in range(len(contours)):
        uv = perspective_projection(v4.unsqueeze(0), w2cs[i][None, :3, :3], w2cs[i][None, :3, 3], Ks[i]).squeeze(0)
        inside = ((uv < imsize) & (uv >= 0)).all(dim=1)
        uvs.append(uv)
        pin = uv[inside]
        if len(pin) == 0 or len(contours[i]) == 0:
            continue
        if pairwise == "torch":
            dist = torch.cdist(pin.unsqueeze(0), contours[i][:, None, :].unsqueeze(0)).squeeze(0)   # [C,Ni,1]
            mind, index = torch.min(dist, 1)
            mind, idx = mind[:, 0], index[:, 0]
        else:
            diff = contours[i][:, None, :] - pin[None, :, :]                   # [C,Ni,2]
            d2 = (diff * diff).sum(-1)
            idx = torch.argmin(d2, dim=1)                                     # first minimum, like torch.min
            mind = torch.sqrt(d2[torch.arange(len(idx)), idx])
        cl = pin[idx].long()
        outside = (masks[i][cl[:, 1], cl[:, 0]] < 0.1).to(verts.dtype)
        total = total + torch.sum(mind * (outside * (epsilon - 1) + 1))
    grid = torch.stack(uvs, 0).view(len(contours), -1, 1, 2) / imsize * 2 - 1
    binary = torch.nn.functional.grid_sample((1 - masks[:, None]).to(verts.dtype), grid, align_corners=False)
    return total + binary.sum() * epsilon


def to_torch_gmm(gmm_bufs, dtype=torch.float32):
    means, precisions, nll_w = gmm_bufs
    return (torch.as_tensor(means, dtype=dtype), torch.as_tensor(precisions, dtype=dtype),
            torch.as_tensor(nll_w, dtype=dtype).unsqueeze(0))


# ----------------------------------------------------------------------------------------------
# the loop (reference smplify/smplify.py:84-226, smpl_type='smpl', keypoint-only)
# ----------------------------------------------------------------------------------------------

def prepare_views(c2ws, Ks, keypoints, dtype=torch.float32):
    """smplify.py:131-135: stack c2w, invert with torch.inverse in `dtype`."""
    c2w = torch.as_tensor(np.asarray(c2ws), dtype=torch.float32).to(dtype)
    w2cs = torch.inverse(c2w)
    Kt = torch.as_tensor(np.asarray(Ks), dtype=torch.float32).to(dtype)
    kps = [None if k is None else torch.as_tensor(np.asarray(k["pose"]), dtype=torch.float32).to(dtype)
           for k in keypoints]
    return w2cs, Kt, kps


def fit(model, gmm_bufs, problem, num_iters=100, dtype=torch.float32, snapshots=(), trace=None, scan=None,
        displacement=False, disp_snapshots=(), mask_pairwise="exact"):
    """Run the reference optimisation loop; returns the rtn_dict of smplify.py:216-226 as numpy.

    `snapshots`: iteration counts k at which the optimised parameters *after k steps* are
    recorded under result['snapshots'][k].  `trace`: optional list receiving per-iteration
    (loss, terms) floats."""
    m = to_torch_model(model, dtype)
    gmm = to_torch_gmm(gmm_bufs, dtype)
    w2cs, Kt, kps = prepare_views(problem["c2ws"], problem["Ks"], problem["keypoints"], dtype)
    n_use = len(problem["use_frames"])
    c = float(problem.get("constant_scale", 0.3))                                # smplify.py:160
    scan_height = None
    if scan is not None:                                                         # smplify.py:146-156
        from oracle import mesh_oracle as MO
        scan_v, scan_f = np.asarray(scan[0], np.float64), np.asarray(scan[1])
        searcher = MO.ReferenceSearcher(scan[0], scan[1])      # MeshGridSearcher in the reference's own float32 arithmetic (smplify.py:146-148)
        scan_height = float((scan_v.max(0) - scan_v.min(0))[1])
        c = scan_height / 1.7
    mask_in = None
    if problem.get("masks") is not None and problem.get("use_mask", True):     # smplify.py:138-144
        from oracle.contour_oracle import border_pixels_rowmajor_all as extract_contours     # (what the goldens were made with)
        mk = (np.array(problem["masks"]) > 128).astype(np.float32)
        idx = [problem["use_frames"].index(f) for f in problem["mask_frames"]]
        mask_in = ([torch.as_tensor(c, dtype=dtype) for c in extract_contours(mk)], torch.as_tensor(mk, dtype=dtype),
                   w2cs[idx], Kt[idx])
    init_pose = torch.as_tensor(problem["init_pose"], dtype=torch.float32).to(dtype)
    init_betas = torch.as_tensor(problem["init_betas"], dtype=torch.float32).to(dtype)

    body_pose = init_pose[:, 3:].detach().clone().requires_grad_(True)
    betas = init_betas.detach().clone().requires_grad_(True)
    global_orient = init_pose[:, :3].detach().clone().requires_grad_(True)
    global_transl = torch.zeros(1, 3, dtype=dtype, requires_grad=True)
    body_scale = torch.ones(1, 1, dtype=dtype, requires_grad=True)
    opt = torch.optim.Adam([{"params": global_transl, "lr": 0.1}, {"params": body_scale, "lr": 0.1},
                            {"params": body_pose}, {"params": betas}, {"params": global_orient}],
                           lr=1e-2, betas=(0.9, 0.999))                          # smplify.py:167-174
    snaps = {}

    def pack():
        return {"global_transl": global_transl.detach().numpy().copy()[0],
                "scale": body_scale.detach().numpy().copy()[0],
                "pose": body_pose.detach().numpy().copy()[0],
                "betas": betas.detach().numpy().copy()[0],
                "global_orient": global_orient.detach().numpy().copy()[0]}

    out = None
    for i in range(num_iters):
        out = smpl_forward(m, betas, global_orient, body_pose)
        model_joints = (out["joints"] + global_transl) * body_scale * c         # smplify.py:189
        body_vertices = (out["vertices"] + global_transl) * body_scale * c      # smplify.py:190
        loss, terms = multiview_keypoint_loss(w2cs, Kt, kps, model_joints, body_pose, betas, n_use, gmm,
                                              imsize=problem["imsize"])
        if mask_in is not None and i > (num_iters // 3):                          # smplify.py:197-199,210
            loss = loss + 5 * multview_mask_loss(mask_in[0], mask_in[1], body_vertices[0], mask_in[2], mask_in[3],
                                                 imsize=problem["imsize"], pairwise=mask_pairwise)
        if scan is not None and i > (num_iters // 3):                             # smplify.py:205-210
            _, cpts, _ = searcher.nearest(body_vertices.detach().numpy()[0])
            pc = MO.point_cloud_loss(body_vertices, torch.as_tensor(cpts, dtype=dtype)) / scan_height * problem["imsize"]
            loss = loss + 5 * pc
        if trace is not None:
            trace.append((float(loss), {k: float(v) for k, v in terms.items()}))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if (i + 1) in snapshots:
            snaps[i + 1] = pack()

    # smplify.py:216-226.  vertices/joints/full_pose come from the LAST forward (parameters before
    # the final step); pose/betas/orient/transl/scale are the stepped parameters.
    res = {
        "vertices": body_vertices.detach().numpy()[0],
        "joints": model_joints.detach().numpy()[0],
        "pose": body_pose.detach().numpy()[0].copy(),
        "betas": betas.detach().numpy()[0].copy(),
        "global_orient": global_orient.detach().numpy()[0].copy(),
        "faces": np.asarray(model["faces"], dtype=np.int32),
        "global_transl": (global_transl * body_scale).detach().numpy()[0],
        "scale": body_scale.detach().numpy()[0].copy(),
        "full_pose": out["full_pose"].detach().numpy()[0],
        "raw_transl": global_transl.detach().numpy()[0].copy(),
        "snapshots": snaps,
    }
    if displacement and scan is not None:
        # SMPL+D stage, smplify.py:228-247: Adam(lr=5e-2) on a per-vertex displacement
        bv = body_vertices.detach()
        disp = torch.zeros_like(bv, requires_grad=True)
        opt_d = torch.optim.Adam([disp], lr=5e-2, betas=(0.9, 0.999))
        faces_t = torch.as_tensor(np.asarray(model["faces"]), dtype=torch.long)
        tris = scan_v[scan_f]
        face_norms = torch.as_tensor(np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]), dtype=torch.float32).to(dtype)
        dsn = {}
        for i in range(num_iters):
            deformed = bv + disp
            norms = MO.compute_normal_torch(deformed[0], faces_t)
            ids, cpts, _ = searcher.nearest(deformed.detach().numpy()[0])
            icp = MO.point_cloud_loss(deformed, torch.as_tensor(cpts, dtype=dtype))
            nl = MO.normal_loss(face_norms[torch.as_tensor(ids, dtype=torch.long)], norms)
            sm = MO.normal_laplacian_smoothness(norms, faces_t)
            loss_d = icp + (nl + sm) * c * 0.1
            opt_d.zero_grad()
            loss_d.backward()
            opt_d.step()
            if (i + 1) in disp_snapshots:
                dsn[i + 1] = disp.detach().numpy()[0].copy()
        res["displacement"] = disp.detach().numpy()[0].copy()
        res["disp_snapshots"] = dsn
    return res


def loss_and_grad(model, gmm_bufs, problem, params, dtype=torch.float64):
    """One evaluation of the objective and its autograd gradient at `params`
    (dict global_transl[3], scale[1], pose[69], betas[10], global_orient[3]).  Used to pin the
    hand-derived gradients of the HIP kernels.  Returns (loss, terms, grads, joints49, vertices)."""
    m = to_torch_model(model, dtype)
    gmm = to_torch_gmm(gmm_bufs, dtype)
    w2cs, Kt, kps = prepare_views(problem["c2ws"], problem["Ks"], problem["keypoints"], dtype)
    c = float(problem.get("constant_scale", 0.3))
    p = {k: torch.tensor(np.asarray(v, dtype=np.float64).reshape(1, -1), dtype=dtype, requires_grad=True)
         for k, v in params.items()}
    out = smpl_forward(m, p["betas"], p["global_orient"], p["pose"])
    mj = (out["joints"] + p["global_transl"]) * p["scale"] * c
    bv = (out["vertices"] + p["global_transl"]) * p["scale"] * c
    loss, terms = multiview_keypoint_loss(w2cs, Kt, kps, mj, p["pose"], p["betas"],
                                          len(problem["use_frames"]), gmm, imsize=problem["imsize"])
    loss.backward()
    grads = {k: v.grad.numpy()[0].copy() for k, v in p.items()}
    return (float(loss), {k: float(v) for k, v in terms.items()}, grads,
            mj.detach().numpy()[0], bv.detach().numpy()[0])


SMPLX_PARAMS = ("global_transl", "scale", "pose", "betas", "global_orient", "leye_pose", "reye_pose",
                "left_hand_pose", "right_hand_pose")                                   # smplify.py:167-173


def fit_smplx(model, gmm_bufs, problem, num_iters=100, dtype=torch.float32, snapshots=(), mask_pairwise="exact", scan=None):
    """The reference loop for smpl_type='smplx' (smplify.py:103-226): body_pose = init[:, 3:66], zero eyes / hand
    PCA, jaw and expression never optimised, hands + face keypoints in the loss (use_hand_face).  scan = (verts,
    faces): use_mesh=True - constant scale scan_height / 1.7 and the point-cloud loss after num_iters // 3
    (smplify.py:146-156,205-210), as in fit()."""
    from bodyfitting_amd.keypoints import pack_keypoints_smplx
    m = to_torch_model(model, dtype)
    gmm = to_torch_gmm(gmm_bufs, dtype)
    c2w = torch.as_tensor(np.asarray(problem["c2ws"]), dtype=torch.float32).to(dtype)
    w2cs = torch.inverse(c2w)
    Kt = torch.as_tensor(np.asarray(problem["Ks"]), dtype=torch.float32).to(dtype)
    kps = [None if k is None else torch.as_tensor(pack_keypoints_smplx(k)).to(dtype) for k in problem["keypoints"]]
    n_use = len(problem["use_frames"])
    c = float(problem.get("constant_scale", 0.3))
    scan_height = None
    if scan is not None:                                                         # smplify.py:146-156
        from oracle import mesh_oracle as MO
        scan_v, scan_f = np.asarray(scan[0], np.float64), np.asarray(scan[1])
        searcher = MO.ReferenceSearcher(scan[0], scan[1])      # MeshGridSearcher in the reference's own float32 arithmetic (smplify.py:146-148)
        scan_height = float((scan_v.max(0) - scan_v.min(0))[1])
        c = scan_height / 1.7
    init_pose = torch.as_tensor(problem["init_pose"], dtype=torch.float32).to(dtype)
    P = {"global_transl": torch.zeros(1, 3, dtype=dtype), "scale": torch.ones(1, 1, dtype=dtype),
         "pose": init_pose[:, 3:66].clone(), "betas": torch.as_tensor(problem["init_betas"], dtype=torch.float32).to(dtype).clone(),
         "global_orient": init_pose[:, :3].clone(), "leye_pose": torch.zeros(1, 3, dtype=dtype), "reye_pose": torch.zeros(1, 3, dtype=dtype),
         "left_hand_pose": torch.zeros(1, 6, dtype=dtype), "right_hand_pose": torch.zeros(1, 6, dtype=dtype)}
    for v in P.values():
        v.requires_grad_(True)
    groups = [{"params": P["global_transl"], "lr": 0.1}, {"params": P["scale"], "lr": 0.1}] + [{"params": P[k]} for k in SMPLX_PARAMS[2:]]
    opt = torch.optim.Adam(groups, lr=1e-2, betas=(0.9, 0.999))
    mask_in = None
    if problem.get("masks") is not None:
        from oracle.contour_oracle import border_pixels_rowmajor_all as extract_contours     # (what the goldens were made with)
        mk = (np.array(problem["masks"]) > 128).astype(np.float32)
        idx = [problem["use_frames"].index(f) for f in problem["mask_frames"]]
        mask_in = ([torch.as_tensor(cc, dtype=dtype) for cc in extract_contours(mk)], torch.as_tensor(mk, dtype=dtype), w2cs[idx], Kt[idx])
    snaps = {}
    for i in range(num_iters):
        out = smplx_forward(m, P["betas"], P["global_orient"], P["pose"], P["leye_pose"], P["reye_pose"],
                            P["left_hand_pose"], P["right_hand_pose"])
        mj = (out["joints"] + P["global_transl"]) * P["scale"] * c
        bv = (out["vertices"] + P["global_transl"]) * P["scale"] * c
        loss, terms = multiview_keypoint_loss(w2cs, Kt, kps, mj, P["pose"], P["betas"], n_use, gmm, imsize=problem["imsize"],
                                              use_hand_face=True)
        if mask_in is not None and i > (num_iters // 3):
            loss = loss + 5 * multview_mask_loss(mask_in[0], mask_in[1], bv[0], mask_in[2], mask_in[3], imsize=problem["imsize"],
                                                 pairwise=mask_pairwise)
        if scan is not None and i > (num_iters // 3):                             # smplify.py:205-210
            _, cpts, _ = searcher.nearest(bv.detach().numpy()[0])
            loss = loss + 5 * (MO.point_cloud_loss(bv, torch.as_tensor(cpts, dtype=dtype)) / scan_height * problem["imsize"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        if (i + 1) in snapshots:
            snaps[i + 1] = {k: v.detach().numpy().copy()[0] for k, v in P.items()}
    return {"vertices": bv.detach().numpy()[0], "joints": mj.detach().numpy()[0], "full_pose": out["full_pose"].detach().numpy()[0],
            "params": {k: v.detach().numpy().copy()[0] for k, v in P.items()}, "snapshots": snaps,
            "global_transl": (P["global_transl"] * P["scale"]).detach().numpy()[0], "loss": float(loss),
            "terms": {k: float(v) for k, v in terms.items()}}


def smplx_loss_and_grad(model, gmm_bufs, problem, params, dtype=torch.float64):
    """objective + autograd gradient at `params` (dict over SMPLX_PARAMS) for smpl_type='smplx'"""
    from bodyfitting_amd.keypoints import pack_keypoints_smplx
    m = to_torch_model(model, dtype)
    gmm = to_torch_gmm(gmm_bufs, dtype)
    w2cs = torch.inverse(torch.as_tensor(np.asarray(problem["c2ws"]), dtype=torch.float32).to(dtype))
    Kt = torch.as_tensor(np.asarray(problem["Ks"]), dtype=torch.float32).to(dtype)
    kps = [None if k is None else torch.as_tensor(pack_keypoints_smplx(k)).to(dtype) for k in problem["keypoints"]]
    P = {k: torch.tensor(np.asarray(params[k], np.float64).reshape(1, -1), dtype=dtype, requires_grad=True) for k in SMPLX_PARAMS}
    c = float(problem.get("constant_scale", 0.3))
    out = smplx_forward(m, P["betas"], P["global_orient"], P["pose"], P["leye_pose"], P["reye_pose"], P["left_hand_pose"], P["right_hand_pose"])
    mj = (out["joints"] + P["global_transl"]) * P["scale"] * c
    bv = (out["vertices"] + P["global_transl"]) * P["scale"] * c
    loss, terms = multiview_keypoint_loss(w2cs, Kt, kps, mj, P["pose"], P["betas"], len(problem["use_frames"]), gmm,
                                          imsize=problem["imsize"], use_hand_face=True)
    loss.backward()
    return (float(loss), {k: float(v) for k, v in terms.items()}, {k: v.grad.numpy()[0].copy() for k, v in P.items()},
            mj.detach().numpy()[0], bv.detach().numpy()[0], int(out["dyn_row"][0]))
