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


def _reload_switches():
    """the library reads its CRUX_* switches when a context is created (csrc/switches.h); a test that changes one mid-process has them re-read"""
    try:
        import crux_jl_amd as crux
        crux.reload_switches()
    except Exception:       # noqa: BLE001  (CPU-only run: no library to tell)
        pass


@pytest.fixture
def monkeypatch():
    """pytest's monkeypatch, with setenv / delenv of a CRUX_* variable followed by crux_reload_switches() (and once more after the undo)."""
    from _pytest.monkeypatch import MonkeyPatch
    mp = MonkeyPatch()

    class _MP:
        def setenv(self, k, v, *a, **kw):
            mp.setenv(k, v, *a, **kw)
            if k.startswith("CRUX_"):
                _reload_switches()

        def delenv(self, k, *a, **kw):
            mp.delenv(k, *a, **kw)
            if k.startswith("CRUX_"):
                _reload_switches()

        def __getattr__(self, n):
            return getattr(mp, n)
    yield _MP()
    mp.undo()
    _reload_switches()
