import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if needed) and load libcloudsc2_hip.so; GPU tests go through it, never around it."""
    import __graft_entry__ as ge

    ge.build()
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib

    return _lib.load()


@pytest.fixture(scope="session")
def gpu(hip_lib):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (these tests never fall back to the CPU)")
    return torch.device("cuda:0")
