"""The C-ABI library loads on a box without a GPU and exports every symbol include/bodyfit.h declares."""
import ctypes
import os
import re

import pytest

from bodyfitting_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "bodyfit.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} not built - run __graft_entry__.build()")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_no_compute_without_gpu_is_an_error_not_a_fallback(smpl_model, gmm):
    """without a device the product path must fail loudly (no CPU fallback exists)."""
    lib = _lib.load()
    assert lib.bf_version().startswith(b"bodyfit-mi355x")
    if lib.bf_device_count() > 0:
        pytest.skip("a GPU is present")
    from bodyfitting_amd.native import DeviceModel
    with pytest.raises(_lib.BodyfitError):
        DeviceModel(smpl_model, gmm)


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libbodyfit.so"))
    with pytest.raises(_lib.BodyfitError):
        _lib.load()


def test_loading_the_library_asks_for_eight_hardware_queues():
    """csrc/api.hip: bf_more_hw_queues - with HIP's default of four hardware queues an RCCL communicator in the process pushes the
    resident fit launch onto the batch stream's queue (dense fits 3x slower, measured); the library raises the limit when it is
    loaded, unless the user set one"""
    import ctypes
    import subprocess
    import sys
    code = ("import os, ctypes; os.environ.pop('GPU_MAX_HW_QUEUES', None); "
            "from bodyfitting_amd import _lib; _lib.load(); "
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; print(libc.getenv(b'GPU_MAX_HW_QUEUES').decode())")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=REPO, env={k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"})
    assert out.returncode == 0 and out.stdout.strip() == "8", out.stderr[-500:]
    out = subprocess.run([sys.executable, "-c", code.replace("os.environ.pop('GPU_MAX_HW_QUEUES', None)", "os.environ['GPU_MAX_HW_QUEUES'] = '6'")],
                         capture_output=True, text=True, cwd=REPO)
    assert out.stdout.strip() == "6"
