import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ndt_lib():
    """The C-ABI library; built on demand here (hipcc cross-compiles without a GPU)."""
    from gtsam_ndt_amd import build, _lib
    build.build_hip()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu_lib(ndt_lib):
    if ndt_lib.ndt_device_count() < 1:
        pytest.fail("gpu-marked test started without a HIP device: there is no CPU fallback to test")
    return ndt_lib
