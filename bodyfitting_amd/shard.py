"""Frame sharding over the GPUs of one node - numpy + ctypes over libbodyfit, no PyTorch.

Frames are independent (the reference fits them in a serial Python loop, apps/genebody_fitting.py:183-192), so frame f
goes to shard f // ceil(F / n) in contiguous blocks, model and cameras are replicated, and nothing is exchanged during
the fit.  The single collective is the final all-gather of the packed parameters [frames_per_gpu, n_params], done by
RCCL inside libbodyfit (csrc/group.hip).  Two ways to drive it:

* `Group`  - ONE process, n devices (bf_group_*: ncclCommInitAll, grouped all-gather).  `python bench.py --gpus N`.
* `Comm`   - one process PER device, started by a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (the benchmark
  contract's `python -m torch.distributed.run ... bench.py`): rank 0 creates the RCCL unique id and the ranks exchange
  it through `FileRendezvous` (all ranks are on one node), then ncclCommInitRank.

The partition itself (`shard_range`, `shard_capacity`, `unpack`) is host arithmetic inside the library and needs no GPU.
"""
from __future__ import annotations

import contextlib
import sys
import ctypes as C
import os
import tempfile
import time

import numpy as np

from . import _lib
from .native import FrameBatch, _f32, _i32


# ---- the partition (host arithmetic in libbodyfit) -----------------------------------------------------------------
def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of frames owned by `rank`; blocks differ by at most one frame."""
    first, count = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.load().bf_shard_range(int(n_frames), int(world), int(rank), C.byref(first), C.byref(count)), "bf_shard_range")
    return first.value, first.value + count.value


def shard_sizes(n_frames, world):
    return [hi - lo for lo, hi in (shard_range(n_frames, r, world) for r in range(world))]


def shard_capacity(n_frames, world):
    """frames per shard the all-gather is padded to"""
    return int(_lib.load().bf_shard_capacity(int(n_frames), int(world)))


def pad_block(local, n_frames, world):
    """this rank's [frames_here, width] block zero-padded to the [capacity, width] send buffer of the all-gather"""
    local = _f32(local)
    out = np.zeros((shard_capacity(n_frames, world), local.shape[1]), np.float32)
    out[: len(local)] = local
    return out


def unpack(gathered, n_frames, world):
    """gathered[world, capacity, width] (what the all-gather leaves on every rank) -> [n_frames, width] in frame order"""
    g = _f32(gathered)
    width = g.shape[-1]
    assert g.size == world * shard_capacity(n_frames, world) * width
    out = np.empty((n_frames, width), np.float32)
    _lib.check(_lib.load().bf_shard_unpack(_lib.fptr(g), int(n_frames), int(world), int(width), _lib.fptr(out)), "bf_shard_unpack")
    return out


# ---- one process, n devices -----------------------------------------------------------------------------------------

@contextlib.contextmanager
def _quiet_stdout():
    """RCCL may print a version banner to STDOUT when a communicator comes up; callers that print a machine-read line on
    stdout (bench.py) must not have it mixed in, so file descriptor 1 points at stderr for the duration of the call."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        os.dup2(saved, 1)
        os.close(saved)


class _BorrowedBatch(FrameBatch):
    """a device's block of a Group, seen through the ordinary FrameBatch interface (the group owns the handle)"""

    def __init__(self, group, i):                      # noqa: super().__init__ would create a batch
        self._lib = group._lib
        self.model = group.models[i]
        self._h = C.c_void_p(self._lib.bf_group_batch(group._h, i))
        self.V = group.V
        self.F = group.shards[i][2]

    def close(self):
        self._h = None


class _BorrowedModel:
    def __init__(self, template, handle, device):
        self.__dict__.update({k: v for k, v in template.__dict__.items() if k not in ("_h", "device")})
        self._h, self.device = handle, device

    def close(self):
        self._h = None


class Group:
    """F frames sharded over n GPUs driven by this process (bf_group).  `model_desc` comes from native.model_desc()."""

    def __init__(self, model, gmm, n_frames, n_views, n_devices=None, devices=None):
        from .native import model_desc
        lib = self._lib = _lib.load()
        if n_devices is None:
            n_devices = len(devices) if devices is not None else lib.bf_device_count()
        desc, info, keep = model_desc(model, gmm)
        devs = None if devices is None else _i32(devices)
        self._h = C.c_void_p()
        with _quiet_stdout():
            rc = lib.bf_group_create(C.byref(desc), int(n_devices), _lib.iptr(devs), int(n_frames), int(n_views), C.byref(self._h))
        _lib.check(rc, "bf_group_create")
        del keep
        self.n, self.F, self.V = int(n_devices), int(n_frames), int(n_views)
        self.n_params = lib.bf_group_n_params(self._h)
        self.info = info
        self.shards = []
        for i in range(self.n):
            d, f, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            _lib.check(lib.bf_group_shard(self._h, i, C.byref(d), C.byref(f), C.byref(c)), "bf_group_shard")
            self.shards.append((d.value, f.value, c.value))
        template = type("Info", (), {})()
        template.__dict__.update(info, _lib=lib, n_params=self.n_params)
        self.models = [_BorrowedModel(template, C.c_void_p(lib.bf_group_model(self._h, i)), self.shards[i][0]) for i in range(self.n)]
        self.batches = [_BorrowedBatch(self, i) for i in range(self.n)]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bf_group_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_cameras(self, c2ws, Ks):
        c2w, K = _f32(c2ws, (self.F, self.V, 4, 4)), _f32(Ks, (self.F, self.V, 3, 3))
        _lib.check(self._lib.bf_group_set_cameras(self._h, _lib.fptr(c2w), _lib.fptr(K)), "bf_group_set_cameras")

    def set_keypoints(self, keypoints, n_use_frames=None):
        kp = _f32(keypoints, (self.F, self.V, self.info["n_loss_joints"], 3))
        nd = None if n_use_frames is None else _i32(np.broadcast_to(np.asarray(n_use_frames), (self.F,)))
        _lib.check(self._lib.bf_group_set_keypoints(self._h, _lib.fptr(kp), _lib.iptr(nd)), "bf_group_set_keypoints")

    def set_init(self, init_betas, init_pose):
        b = _f32(init_betas, (self.F, self.info["n_betas"]))
        p = _f32(np.asarray(init_pose).reshape(self.F, -1)[:, :72], (self.F, 72))
        _lib.check(self._lib.bf_group_set_init(self._h, _lib.fptr(b), _lib.fptr(p)), "bf_group_set_init")

    def stage_inputs(self, keypoints, n_use_frames, init_betas, init_pose):
        """the whole job's NEXT frames, without draining the devices (bf_group_stage_inputs); the next fit() needs FIT_RESET"""
        kp = _f32(keypoints, (self.F, self.V, self.info["n_loss_joints"], 3))
        nd = None if n_use_frames is None else _i32(np.broadcast_to(np.asarray(n_use_frames), (self.F,)))
        b = _f32(init_betas, (self.F, self.info["n_betas"]))
        p = _f32(np.asarray(init_pose).reshape(self.F, -1)[:, :72], (self.F, 72))
        _lib.check(self._lib.bf_group_stage_inputs(self._h, _lib.fptr(kp), _lib.iptr(nd), _lib.fptr(b), _lib.fptr(p)), "bf_group_stage_inputs")

    def stage_masks(self, masks, view_index, contour_select=_lib.CONTOUR_OPENCV_FIRST):
        """the next step's masks[F,M,H,W] (views and shape of the attached ones), every device its block, under the fits in flight"""
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        F, M, H, W = masks.shape
        assert F == self.F
        vi = _i32(view_index)
        _lib.check(self._lib.bf_group_stage_masks(self._h, int(M), _lib.iptr(vi), int(H), int(W), masks.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                  int(contour_select)), "bf_group_stage_masks")

    def set_masks(self, masks, view_index, contours=None, contour_select=_lib.CONTOUR_OPENCV_FIRST):
        """masks[F,M,H,W] uint8 of the whole job; contours = per (frame, mask view) arrays of (x, y) points, or None = extracted
        on the devices (bf_group_set_masks hands every device its block; the devices work side by side)"""
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        F, M, H, W = masks.shape
        assert F == self.F
        vi = _i32(view_index)
        cnt = xy = None
        if contours is not None:
            flat = [np.asarray(c, np.float32).reshape(-1, 2) for fr in contours for c in fr]
            cnt = _i32([len(c) for c in flat])
            xy = _f32(np.concatenate(flat) if flat else np.zeros((0, 2)))
        _lib.check(self._lib.bf_group_set_masks(self._h, int(M), _lib.iptr(vi), int(H), int(W), masks.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                _lib.iptr(cnt), _lib.fptr(xy), int(contour_select)), "bf_group_set_masks")

    def device_of_frame(self, f):
        for d, first, count in self.shards:
            if first <= f < first + count:
                return d
        raise IndexError(f)

    def set_scans(self, scans):
        """one native.Scan per frame of the job, each created on device_of_frame(f); None detaches"""
        if scans is None:
            _lib.check(self._lib.bf_group_set_scans(self._h, None), "bf_group_set_scans")
            self._scans = None
            return
        assert len(scans) == self.F
        arr = (C.c_void_p * self.F)(*[s._h for s in scans])
        _lib.check(self._lib.bf_group_set_scans(self._h, arr), "bf_group_set_scans")
        self._scans = list(scans)          # (keep them alive)

    def fit(self, n_iters, hyper=None, flags=_lib.FIT_DEFAULT):
        hp = C.byref(hyper) if hyper is not None else None
        _lib.check(self._lib.bf_group_fit(self._h, int(n_iters), hp, int(flags)), "bf_group_fit")

    def fit_displacement(self, n_iters, hyper=None):
        hp = C.byref(hyper) if hyper is not None else None
        _lib.check(self._lib.bf_group_fit_displacement(self._h, int(n_iters), hp), "bf_group_fit_displacement")

    def sync(self):
        _lib.check(self._lib.bf_group_sync(self._h), "bf_group_sync")

    def comm_size(self):
        with _quiet_stdout():              # (the communicator - and RCCL's banner on stdout - comes up on first use)
            n = self._lib.bf_group_comm_size(self._h)
        if n < 0:
            _lib.check(n, "bf_group_comm_size")
        return n

    def gather_params(self, from_peer=0):
        """the one collective of the path: -> [F, n_params] as it arrived on device `from_peer`"""
        out = np.empty((self.F, self.n_params), np.float32)
        with _quiet_stdout():
            rc = self._lib.bf_group_gather_params(self._h, _lib.fptr(out), int(from_peer))
        _lib.check(rc, "bf_group_gather_params")
        return out


# ---- one process per device -----------------------------------------------------------------------------------------
def _process_start_ticks(pid):
    """start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat), or 0 where /proc is not available"""
    try:
        with open(f"/proc/{pid}/stat") as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return 0


class FileRendezvous:
    """Ranks of ONE node hand each other small byte strings through the file system (atomic rename, polling).  Stands in
    for the store a launcher would offer, without importing it.

    The directory is named by what the ranks of one `torch.distributed.run --nnodes=1` launch share and no other launch
    does: MASTER_PORT, TORCHELASTIC_RUN_ID, TORCHELASTIC_RESTART_COUNT (a restarted worker group must not read the previous
    group's RCCL id), and the launcher process itself - its pid AND its start time, so a recycled pid of a crashed job does
    not lead to that job's leftovers.  It is created with mode 0700 and must belong to this user; the last barrier of
    `cleanup` removes it."""

    def __init__(self, rank, world, key=None, root=None, timeout=300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if key is None:
            ppid = os.getppid()
            key = "-".join(str(x) for x in (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), ppid, _process_start_ticks(ppid)))
        self.dir = os.path.join(root or tempfile.gettempdir(), f"bodyfit-rdzv-{os.getuid()}-{key}")
        try:
            os.mkdir(self.dir, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(self.dir)
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError(f"rendezvous directory {self.dir} is not a private directory of this user")
        self._n_barrier = 0

    def _path(self, name, rank):
        return os.path.join(self.dir, f"{name}.{rank}")

    def put(self, name, data: bytes):
        tmp = self._path(name, self.rank) + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, self._path(name, self.rank))

    def get(self, name, rank) -> bytes:
        path, t0 = self._path(name, rank), time.monotonic()
        while not os.path.exists(path):
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f"rendezvous: rank {rank} never published {name!r} in {self.dir}")
            time.sleep(0.005)
        with open(path, "rb") as f:
            return f.read()

    def broadcast(self, name, data=None, root=0) -> bytes:
        if self.rank == root:
            self.put(name, data)
            return data
        return self.get(name, root)

    def all_gather(self, name, data: bytes):
        self.put(name, data)
        return [self.get(name, r) for r in range(self.world)]

    def barrier(self, name):
        """every use of a barrier gets files of its own (a name used twice must not pass on the first use's files)"""
        self._n_barrier += 1
        self.all_gather(f"barrier-{name}-{self._n_barrier}", b"1")

    def cleanup(self):
        """After a final barrier nobody reads anything but the `left` marks: every rank leaves one, rank 0 waits for all of
        them and removes the directory."""
        import shutil
        self.barrier("cleanup")
        self.put("left", b"1")
        if self.rank == 0:
            for r in range(self.world):
                self.get("left", r)
            shutil.rmtree(self.dir, ignore_errors=True)


class Comm:
    """This rank's end of the RCCL communicator (bf_comm): barrier, max over ranks, final gather of the parameters."""

    def __init__(self, rank, world, device, rendezvous=None):
        lib = self._lib = _lib.load()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        rdzv = rendezvous or FileRendezvous(rank, world)
        uid = None
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            _lib.check(lib.bf_comm_unique_id(buf), "bf_comm_unique_id")
            uid = bytes(buf)
        uid = rdzv.broadcast("rccl-unique-id", uid)
        self._h = C.c_void_p()
        with _quiet_stdout():
            rc = lib.bf_comm_create((C.c_uint8 * 128).from_buffer_copy(uid), self.rank, self.world, self.device, C.byref(self._h))
        _lib.check(rc, "bf_comm_create")
        self.rendezvous = rdzv

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bf_comm_destroy(self._h)
            self._h = None

    def size(self):
        return int(self._lib.bf_comm_size(self._h))

    def barrier(self):
        _lib.check(self._lib.bf_comm_barrier(self._h), "bf_comm_barrier")

    def max(self, value):
        v = C.c_double(float(value))
        _lib.check(self._lib.bf_comm_allreduce(self._h, C.byref(v), 1), "bf_comm_allreduce")
        return v.value

    def gather_params(self, batch, n_frames):
        out = np.empty((int(n_frames), batch.model.n_params), np.float32)
        _lib.check(self._lib.bf_comm_gather_params(self._h, batch._h, int(n_frames), _lib.fptr(out)), "bf_comm_gather_params")
        return out
