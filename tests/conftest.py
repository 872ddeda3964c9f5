import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts (like bench.py): replicas sharing the device in test_gpu_peer.py need 4 concurrent learner streams

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """A libcruxhip context on device 0. GPU tests FAIL (not skip) when the HIP library or device is missing."""
    import crux_jl_amd as crux
    return crux.default_context()
