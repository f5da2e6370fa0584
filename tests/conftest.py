import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """Session fixture for -m gpu tests: the HIP extension must be loaded and a GPU visible.
    No fallback: a missing extension or device is a hard failure, not a skip."""
    import torch
    from ctranslate2_amd import _lib
    L = _lib.lib()
    assert torch.cuda.is_available(), "gpu-marked test without a visible GPU"
    assert L.ct2x_device_count() >= 1
    _lib.check(L.ct2x_set_device(0))
    return L
