import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


FRESH = None      # multiprocessing context whose children are forked from a server started BEFORE this process touched a GPU


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A process that has initialised HIP must not exec another program on the GPU boxes, and a fork of it inherits that state.  The
    # multi-process GPU test (one rank per process, tests/test_gpu_group.py) therefore gets its children from a fork server that is
    # started here, at configuration time - before any test has made a HIP call.
    global FRESH
    try:
        import multiprocessing
        from multiprocessing import forkserver
        FRESH = multiprocessing.get_context("forkserver")
        forkserver.ensure_running()
    except Exception:                                      # (no fork server: the test that needs it skips)
        FRESH = None


@pytest.fixture(scope="session")
def smpl_model():
    from bodyfitting_amd import synthetic as S
    return S.make_model("smpl", seed=0)


@pytest.fixture(scope="session")
def gmm():
    from bodyfitting_amd import synthetic as S
    return S.make_gmm(seed=0)


@pytest.fixture(scope="session")
def gmm_bufs(gmm):
    from bodyfitting_amd import synthetic as S
    return S.gmm_buffers(gmm)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def dev_model(smpl_model, gmm):
    """The HIP-resident model; only -m gpu tests request it."""
    from bodyfitting_amd.native import DeviceModel
    m = DeviceModel(smpl_model, gmm, device=0)
    yield m
    m.close()
