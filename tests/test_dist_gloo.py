"""World-size-2 gloo test of the multi-GPU protocol on CPU: each rank computes the loss gradient of ITS shard,
the flat gradients are SUM all-reduced, Adam is applied to the mean -> identical parameters on every rank, equal (up to
summation order) to single-process training on the concatenated batch. The per-rank compute stand-in is the oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(seed_data, n):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from crux_jl_amd import _lib as L
    rng = np.random.default_rng(seed_data)
    net = O.OMlp([4, 64, 64, 2], ["relu", "relu", "identity"]).init_glorot(1).adam_init(float(np.float32(3e-4)))
    buf = O.OBuffer(4, 2, L.ACTION_DISCRETE, n, ["return", "logprob", "advantage"])
    a = np.zeros((2, n), np.bool_); a[rng.integers(0, 2, n), np.arange(n)] = True
    buf.push({"s": rng.standard_normal((4, n)).astype(np.float32), "a": a, "sp": rng.standard_normal((4, n)).astype(np.float32),
              "r": np.ones((1, n), np.float32), "done": np.zeros((1, n), np.bool_), "return": rng.standard_normal((1, n)).astype(np.float32),
              "logprob": np.full((1, n), -0.69, np.float32), "advantage": rng.standard_normal((1, n)).astype(np.float32)})
    cfg = L.TrainCfg(); cfg.loss, cfg.head, cfg.batch_size, cfg.epochs = 0, 0, n, 1
    cfg.eps_clip, cfg.lambda_p, cfg.lambda_e, cfg.target_kl = 0.2, 1.0, 0.1, -1.0
    return O, L, net, buf, cfg


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crux_jl_amd import dist as cdist
    n = 64
    O, L, net, buf, cfg = _make(100 + rank, n)              # independent shard per rank
    assert cdist.partition_envs(64, world, rank) == (32 * rank, 32 * rank + 32)
    for step in range(3):
        ids = np.arange(n, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32)
        O.chk(O.lib().orc_loss_grad(net.h, buf.h, C.byref(cfg), O.vpz(ids), n, O.vpz(info)))
        g = torch.from_numpy(net.grads)                     # aliases the oracle's gradient buffer
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        O.chk(O.lib().orc_adam_apply(net.h, 1.0 / world))
    out[rank] = net.params.copy()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_concatenated_batch():
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager(); out = mgr.dict()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    [p.start() for p in procs]; [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(out[0], out[1])                   # replicas stay bit-identical
    # single process on the concatenated batch (mean over 128 == mean of the two 64-sample means)
    O, L, net, buf, cfg = _make(100, 64)
    O2, _, net2, buf2, _ = _make(101, 64)
    big = O.OBuffer(4, 2, L.ACTION_DISCRETE, 128, ["return", "logprob", "advantage"]); big.push_buffer(buf); big.push_buffer(buf2)
    cfg.batch_size = 128
    for step in range(3):
        ids = np.arange(128, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32)
        O.chk(O.lib().orc_train_step(net.h, big.h, C.byref(cfg), O.vpz(ids), 128, O.vpz(info)))
    assert np.abs(net.params - out[0]).max() < 1e-6


def _worker_periodic(rank, world, port, out, k, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crux_jl_amd import dist as cdist
    n = 64
    O, L, net, buf, cfg = _make(100 + rank, n)
    ids = np.arange(n, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32)
    mv = {}

    def local_step(i):
        O.chk(O.lib().orc_train_step(net.h, buf.h, C.byref(cfg), O.vpz(ids), n, O.vpz(info)))

    def state():
        mv["m"], mv["v"], _ = net.adam_state(); return [net.params, mv["m"], mv["v"]]        # net.params aliases the oracle's buffer; m / v are copies, written back below

    class Avg(cdist.StateAverager):
        def __call__(self, arrays):
            super().__call__(arrays); net.set_adam_state(arrays[1], arrays[2])
    n_sync = cdist.train_periodic(local_step, state, steps, k, Avg())
    m, v, bp = net.adam_state()
    out[rank] = (net.params.copy(), m, v, n_sync)
    dist.destroy_process_group()


@pytest.mark.parametrize("k", [4])
def test_two_rank_periodic_form_matches_the_local_sgd_twin(k):
    """k = 4 (SURVEY 8e's k > 1 row; the in-kernel form is tests/test_gpu_peer.py::test_periodic_form_equals_the_local_sgd_twin): two gloo ranks take local oracle steps on their own
    shards and average theta, m, v after every 4th through dist.StateAverager; a single process running both learners with numpy's (a + b) * 0.5 must give the same bits."""
    steps = 8
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager(); out = mgr.dict()
    port = 30500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker_periodic, args=(r, 2, port, out, k, steps)) for r in range(2)]
    [p.start() for p in procs]; [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for j in range(3):
        assert np.array_equal(out[0][j], out[1][j])          # replicas leave every exchange bit-identical
    assert out[0][3] == out[1][3] == steps // k
    twins = [_make(100 + r, 64) for r in range(2)]
    ids = np.arange(64, dtype=np.int64)
    for i in range(steps):
        for O, L, net, buf, cfg in twins:
            info = np.zeros(L.INFO_N, np.float32)
            O.chk(O.lib().orc_train_step(net.h, buf.h, C.byref(cfg), O.vpz(ids), 64, O.vpz(info)))
        if (i + 1) % k == 0:
            nets = [t[2] for t in twins]; st = [n.adam_state() for n in nets]
            th = (nets[0].params + nets[1].params) * np.float32(0.5); m = (st[0][0] + st[1][0]) * np.float32(0.5); v = (st[0][1] + st[1][1]) * np.float32(0.5)
            for n in nets:
                n.params[:] = th; n.set_adam_state(m, v)
    tm, tv, _ = twins[0][2].adam_state()
    assert np.array_equal(twins[0][2].params, out[0][0]) and np.array_equal(tm, out[0][1]) and np.array_equal(tv, out[0][2])
    # and local steps really diverge between exchanges: the per-step form gives different parameters
    from crux_jl_amd import dist as cdist
    assert not cdist.sync_due(4, 1) and cdist.sync_due(4, 4) and not cdist.sync_due(5, 4) and not cdist.sync_due(0, 4)
