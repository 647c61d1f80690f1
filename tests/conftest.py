import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle as O
    O.load()
    return O


@pytest.fixture(scope="session")
def hiplib():
    """librdst_hip.so; built with hipcc if missing (cross-compiles without a GPU)."""
    from rdst_amd import build as B
    from rdst_amd import _lib
    B.build()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu():
    """torch + a visible HIP device; gpu tests fail (not skip) when the device route is unusable."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a HIP device"
    from rdst_amd import build as B
    B.build()
    import rdst_amd
    return rdst_amd
