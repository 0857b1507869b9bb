import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def free_port() -> int:
    """A TCP port nobody is listening on (for torch.distributed rendezvous on 127.0.0.1): taken from a socket
    bound to port 0 rather than derived from the pid, so two tests - or a test and bench.py's default port - on a
    shared box do not collide."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


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
