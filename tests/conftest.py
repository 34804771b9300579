import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def kats():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "block_kats.npz"))


@pytest.fixture(scope="session")
def bamd():
    """The product library on a GPU box: must be built in-tree and must see a device (no fallback)."""
    import booster_amd
    from booster_amd import build
    build.build()
    n = booster_amd.device_count()
    assert n >= 1, "no HIP device visible: the gpu tests must run on the MI355X box"
    return booster_amd
