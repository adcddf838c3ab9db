#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_smplx.py tests/test_gpu_mask.py tests/test_gpu_configs_full.py -m gpu -q -k "not config5" 2>&1 | tail -3
for s in 0 1 0 1; do BF_KP_SPLIT=$s python tools/bench_configs.py --cfg3 --reps 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split $s', d['ms_per_iteration'], d['ms_per_fit'])"; done
python - <<'PY'
import os, numpy as np, sys
sys.path.insert(0, os.getcwd())
from bodyfitting_amd import native as N, synthetic as S
model, gmm = S.make_model("smplx", seed=0), S.make_gmm(seed=0)
dev = N.DeviceModel(model, gmm, device=0)
mf = list(range(0, 48, 6))
prob = S.make_problem_smplx(model, frame=0, n_views=48, mask_frames=mf)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
out = {}
for s in ("0", "1"):
    os.environ["BF_KP_SPLIT"] = s
    b = N.FrameBatch(dev, 1, 48)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_masks(np.array(prob["masks"])[None], mf, None)
    b.fit(200)
    out[s] = (b.get_params().copy(), b.get_result()[0].copy())
    b.close()
print("split == unsplit, bit for bit:", np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1]))
PY
