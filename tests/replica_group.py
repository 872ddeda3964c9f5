"""Helpers of the tests that put SEVERAL replicas of a group on ONE device (tests/test_gpu_peer.py, the group tests of tests/test_gpu_fs2.py): the attach that skips when the
device cannot run the replicas side by side, the bounds on the in-kernel waits, and the thread runner that calls a stuck group off from the host instead of hanging in a
teardown. SURVEY 8(e); the reference itself is single-process (src/model_free/on_policy.jl:80-109)."""
import sys
import threading
import time

import pytest

from parity import crux


def bound(ctxs):
    """the waits of a group under test stay well inside the suite's watchdog (conftest.py): 5 s for one exchange, 10 s for all exchanges of one launch (csrc/peer_wait.h)"""
    for c in ctxs:
        c.peer_set_timeout_ms(5000); c.peer_set_budget_ms(10000)


def attach_or_skip(ctxs, owned=()):
    """crux_peer_attach_local; a device that cannot run the replicas' kernels side by side (the library's handshake / rendezvous probe says so) skips the test with its message.
    owned: contexts to close on the way out."""
    try:
        crux.peer_attach_local(ctxs)
    except crux.CruxError as e:
        for c in owned:
            c.close()
        if "hardware queue" in str(e):
            pytest.skip("no side-by-side placement of %d replicas on this device: %s" % (len(ctxs), e))
        raise
    bound(ctxs)


def run_threads(fns, seconds=60.0):
    """every replica's training call on its own (daemon) thread. A group that has not returned after `seconds` is called off FROM THE HOST (crux_abort_all: the kernels' flag
    waits poll a pinned word) before anything synchronises a stream -- the test then fails with the replicas' errors instead of hanging in its teardown."""
    errs = [None] * len(fns)

    def wrap(i):
        try:
            fns[i]()
        except Exception as e:      # noqa: BLE001
            errs[i] = e
    ts = [threading.Thread(target=wrap, args=(i,), daemon=True) for i in range(len(fns))]
    [t.start() for t in ts]
    deadline = time.time() + seconds
    [t.join(max(0.0, deadline - time.time())) for t in ts]
    stuck = [i for i, t in enumerate(ts) if t.is_alive()]
    if stuck:
        print("replicas %r have not returned after %.0f s: crux_abort_all()" % (stuck, seconds), file=sys.stderr)
        crux.abort_all(); [t.join(15.0) for t in ts]
    if any(e is not None for e in errs):
        print("replica errors:", [repr(e)[:300] if e is not None else None for e in errs], file=sys.stderr)
    assert not any(t.is_alive() for t in ts), "a replica did not return even after the host's abort"
    assert not stuck, "replicas %r needed the host's abort after %.0f s: %r" % (stuck, seconds, [repr(e)[:200] for e in errs])
    return errs


def run_threads_raise(fns, seconds=60.0):
    for e in run_threads(fns, seconds):
        if e is not None:
            raise e
