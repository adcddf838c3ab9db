import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
