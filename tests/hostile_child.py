"""Dense fits in FRESH processes whose HOST is not the tidy one the library hopes for (tests/test_gpu_hostile_host.py):
  `hip_first`      the process initialises HIP, allocates and creates streams BEFORE libbodyfit is loaded - so the library's request for
                   8 hardware queues comes too late - optionally with GPU_MAX_HW_QUEUES pinned to a small value by the 'user';
  `busy_neighbour` other streams of the same process keep the device busy (memsets over 256 MB, back to back, from a thread)
                   while the dense loops run.
Each child fits the same two small jobs - a silhouette loop and a scan loop + SMPL+D, both with the resident fit launch if the
self-tests allow it - and stores the parameters and which schedule ran; the parent holds them against its own run, bit for bit."""
import ctypes
import os
import sys
import threading
import traceback

import numpy as np


def _hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise OSError("libamdhip64 not found")


def jobs():
    """(the two jobs; importing bodyfitting_amd loads libbodyfit - call this only when the scenario allows it)"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from bodyfitting_amd import native as N, synthetic as S
    out = {}
    model = S.make_model("smpl", seed=0, nv=690)
    dev = N.DeviceModel(model, S.make_gmm(seed=0), device=0)
    # silhouette loop: 8 views, 4 mask views, 24 iterations (the silhouette is on for i > 8)
    mask_frames = [1, 3, 5, 7]
    prob = S.make_problem(model, frame=0, n_views=8, mask_frames=mask_frames)
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([prob])
    b = N.FrameBatch(dev, 1, 8)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose)
    b.set_masks(np.array(prob["masks"])[None], mask_frames, None)
    b.fit(24)
    out["mask_params"] = b.get_params()
    out["mask_resident"] = b.dense_resident()
    b.close()
    # scan loop + SMPL+D: 2 frames, 8 views, 18 + 6 iterations
    items = [S.make_scan_problem(model, frame=f, n_views=8) for f in (0, 1)]
    scans = [N.Scan(sv, sf) for _, sv, sf in items]
    c2w, K, kp, ndiv, betas, pose = N.pack_problem([p for p, _, _ in items])
    b = N.FrameBatch(dev, 2, 8)
    b.set_cameras(c2w, K); b.set_keypoints(kp, ndiv); b.set_init(betas, pose); b.set_scans(scans)
    b.fit(18)
    out["scan_params"] = b.get_params()
    out["scan_resident"] = b.dense_resident()
    b.fit_displacement(6)
    out["scan_disp"] = b.get_displacement()
    b.set_scans(None); b.close()
    for s in scans:
        s.close()
    dev.close()
    return out


def _run(out_path, body):
    try:
        np.savez(out_path, **body())
    except BaseException:
        with open(out_path + ".err", "w") as fh:
            fh.write(traceback.format_exc())
        raise


def hip_first(out_path, max_queues):
    def body():
        if max_queues:
            os.environ["GPU_MAX_HW_QUEUES"] = str(max_queues)      # the USER's value: the library never overrides it
        else:
            os.environ.pop("GPU_MAX_HW_QUEUES", None)
        assert "bodyfitting_amd._lib" not in sys.modules
        hip = _hip()
        assert hip.hipInit(0) == 0 and hip.hipSetDevice(0) == 0
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)) == 0
        streams = [ctypes.c_void_p() for _ in range(3)]
        for s in streams:
            assert hip.hipStreamCreate(ctypes.byref(s)) == 0
            assert hip.hipMemsetAsync(p, 0, ctypes.c_size_t(1 << 20), s) == 0
        assert hip.hipDeviceSynchronize() == 0                     # HIP is up, queues exist: only now the library comes in
        res = jobs()
        res["queues_env"] = np.array(os.environ.get("GPU_MAX_HW_QUEUES", ""))
        return res
    _run(out_path, body)


def busy_neighbour(out_path):
    def body():
        hip = _hip()
        assert hip.hipInit(0) == 0 and hip.hipSetDevice(0) == 0
        n = 256 << 20
        bufs, streams = [], []
        for _ in range(2):
            p, s = ctypes.c_void_p(), ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)) == 0 and hip.hipStreamCreate(ctypes.byref(s)) == 0
            bufs.append(p); streams.append(s)
        stop = threading.Event()
        count = [0]

        def hammer():
            hip.hipSetDevice(0)
            while not stop.is_set():
                for p, s in zip(bufs, streams):
                    hip.hipMemsetAsync(p, count[0] & 0xff, ctypes.c_size_t(n), s)
                for s in streams:
                    hip.hipStreamSynchronize(s)
                count[0] += 1
        t = threading.Thread(target=hammer, daemon=True)
        t.start()
        try:
            res = jobs()
        finally:
            stop.set(); t.join(30)
        res["neighbour_rounds"] = np.array(count[0])
        return res
    _run(out_path, body)
