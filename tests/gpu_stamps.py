"""Bring-up helper (not a test): per-wave, per-phase cycle stamps of the LAST fit iteration from the -DBF_STAMP build.
    BODYFIT_LIB=bodyfitting_amd/libbodyfit_stamp.so python tests/gpu_stamps.py
Every wave stores the raw shader clock when it ARRIVES at a barrier and when it LEAVES it (fit_kernels.hip: BF_T / BF_SYNC).  Printed per
phase: when each wave arrived, in cycles after the previous barrier released - the largest number of a row is the phase's pole."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyfitting_amd import native as N, synthetic as S   # noqa: E402

model, gmm = S.make_model("smpl"), S.make_gmm()
dev = N.DeviceModel(model, gmm)
c2w, K, kp, ndiv, betas, pose = N.pack_problem([S.make_problem(model, 0, 48)])
b = N.FrameBatch(dev, 1, 48)
b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
NAMES = os.environ.get("BF_PHASES", "A BD F I").split()
for rep in range(3):
    b.reset(); b.fit(100); b.sync()
    raw = b.debug_dump(4352 + 192)[4352:].view(np.int32).astype(np.int64).reshape(-1, 8)      # [2 b + {arrive, leave}][wave]
    nb = len(NAMES)
    arrive, leave = raw[0:2 * nb:2], raw[1:2 * nb:2]
    print("rep", rep, "timing", b.last_timing())
    prev = leave[nb - 1]                       # the previous iteration's last barrier released (same slots, one iteration earlier: the
    total = 0                                  # stamps of the LAST iteration overwrite them, so phase A's start is taken from the span)
    for p in range(nb):
        start = leave[p - 1] if p > 0 else None
        if start is None:
            print("  phase %s: (arrival clock, raw)  %s" % (NAMES[p], " ".join("%6d" % ((x - arrive[p].min()) & 0xFFFFFFFF) for x in arrive[p])))
            continue
        rel = (arrive[p] - start.max()) & 0xFFFFFFFF
        rel = np.where(rel > 1 << 31, rel - (1 << 32), rel)
        span = int(((leave[p].max() - leave[p - 1].max()) & 0xFFFFFFFF))
        total += span
        print("  phase %s: arrive after previous release, waves 0-7: %s | phase span %5d" % (NAMES[p], " ".join("%6d" % x for x in rel), span))
    t0 = raw[23, 0]
    marks = {k: [int((x - t0) & 0xFFFFFFFF) for x in raw[k]] for k in range(12, 22) if raw[k].any()}
    for k, v in marks.items():
        print("  mark %d (cycles after the last iteration's top, waves 0-7; 0 = not stamped by that wave): %s" % (k, [x if r else 0 for x, r in zip(v, raw[k])]))
    it_len = int((raw[23, 0] - raw[22, 0]) & 0xFFFFFFFF)
    a_span = int((leave[0].max() - raw[23, :4].min()) & 0xFFFFFFFF)
    print("  one iteration (top to top, wave 0): %d cycles; phase A span %d; sum of spans B..I %d" % (it_len, a_span, total))
