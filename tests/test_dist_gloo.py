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
