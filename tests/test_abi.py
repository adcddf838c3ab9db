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
