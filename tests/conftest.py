import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts (like bench.py): replicas sharing the device in test_gpu_peer.py need 4 concurrent learner streams

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "watchdog(seconds): this test's own watchdog limit (default CRUX_TEST_WATCHDOG_S = 120): tests that start `python bench.py` pay a first `import torch` of 1-2 min on a fresh box")


# ---- the suite must not be taken down by one stuck test (one same-device replica group that never returned cost 161 tests and the smoke run) ---------------
# (1) order: single-process parity first; everything that puts several replicas on ONE device last (cross-process pairs, then in-process groups, the 3- and 4-replica ones at
#     the very end) -- under the driver's `-x` a failure there costs nothing that comes before it;
# (2) a watchdog per GPU test: after WATCHDOG_S (120) seconds it dumps every thread's stack, calls crux_abort_all() (raises the host abort word of every context: replica-group
#     kernels waiting for a peer return CRUX_EHIP within ~100 us, csrc/peer_wait.h) and the test is reported FAILED; if the test is still stuck GRACE_S later the process
#     ends with one verdict line (exit code 70) instead of the driver's rc 124 after 20 minutes.
WATCHDOG_S = float(os.environ.get("CRUX_TEST_WATCHDOG_S", "120"))      # 4 x the slowest test of the suite (29 s: an oracle-heavy window test) on the boxes seen so far
GRACE_S = float(os.environ.get("CRUX_TEST_WATCHDOG_GRACE_S", "20"))
_LAST = {"test_gpu_peer_xproc.py": 1, "test_gpu_peer.py": 2}


def _order_key(item):
    mod = os.path.basename(str(item.fspath)); k = _LAST.get(mod, 0)
    many = 0
    if k == 2:
        cs = getattr(item, "callspec", None); R = cs.params.get("R", 2) if cs is not None else 2
        many = 1 if (R >= 3 or "three_replicas" in item.name or "four_replicas" in item.name) else 0
    return (k, many)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_order_key)      # stable: the collection order inside every class of tests stays


def _abort_all():
    try:
        import crux_jl_amd as crux
        return crux.abort_all()
    except Exception:       # noqa: BLE001
        return -1


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    if item.get_closest_marker("gpu") is None or WATCHDOG_S <= 0:
        yield; return
    import faulthandler
    import threading
    done, fired = threading.Event(), threading.Event()
    wm = item.get_closest_marker("watchdog"); limit = float(wm.args[0]) if wm is not None and wm.args else WATCHDOG_S

    def watch():
        if done.wait(limit):
            return
        fired.set()
        sys.stderr.write("\n[watchdog] %s has not finished after %.0f s: stacks below, then crux_abort_all()\n" % (item.nodeid, limit)); sys.stderr.flush()
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        n = _abort_all()
        sys.stderr.write("[watchdog] crux_abort_all() told %d context(s)\n" % n); sys.stderr.flush()
        if done.wait(GRACE_S):
            return
        sys.stderr.write("[watchdog] FAILED %s: still stuck %.0f s after the abort (not a replica-group wait); ending the run with exit code 70\n" % (item.nodeid, GRACE_S)); sys.stderr.flush()
        sys.stdout.write("\nFAILED %s - watchdog: stuck for more than %.0f s\n" % (item.nodeid, limit + GRACE_S)); sys.stdout.flush()
        os._exit(70)
    item._crux_watchdog_fired = fired
    t = threading.Thread(target=watch, name="crux-test-watchdog", daemon=True); t.start()
    try:
        yield
    finally:
        done.set()


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    outcome = yield
    fired = getattr(item, "_crux_watchdog_fired", None)
    if fired is not None and fired.is_set() and outcome.excinfo is None:
        outcome.force_exception(AssertionError("watchdog: the test ran past its limit (crux_abort_all() was called; see stderr for the stacks)"))


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
