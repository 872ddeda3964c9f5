"""The replica-group exchange ACROSS PROCESSES (BASELINE configs[4]): every real multi-GPU run enters through
crux_peer_export -> handle exchange -> crux_peer_attach (hipIpc), which tests/test_gpu_peer.py (crux_peer_attach_local inside one process) never touches. Here two PROCESSES on
device 0 (tests/peer_xproc_worker.py, one rank each) form the group at C5's shapes (17 -> 64 -> 64 -> 6 tanh Gaussian actor, 17 -> 64 -> 64 -> 1 critic) and at C2's:

  * k = 1: distinct shards leave the replicas bit-identical and equal the ORACLE's single learner on the concatenated batch (the tolerances of test_gpu_peer.py);
  * identical shards reproduce, BIT FOR BIT, a group of one running the same kernel instantiation (g + g and the halving are exact; ADVICE r4: the un-grouped kernel is another
    compilation and only agrees to 1e-6);
  * k = 8 (the in-kernel periodic form) equals the oracle's local-SGD twin;
  * a rank that dies with the group attached: the survivor's training call returns CRUX_EHIP after the in-kernel timeout -- no hang;
  * the launcher form: `bench.py --gpus 2 --same-device --selftest --workload c5` (torch.distributed rendezvous, relaunch under torchrun, selftest, probe iteration), and the
    fallback ladder of bench.py when crux_peer_attach is made to fail on every rank.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import parity
from parity import L, O, crux

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "peer_xproc_worker.py")
EXTRAS = ["return", "logprob", "advantage"]


def _shard(family, seed, E=8, T=128):
    """one rank's buffer after rollout + GAE + returns + whiten, made by the ORACLE (the parent can then rebuild the concatenated batch exactly)"""
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES[family]
    _, oa = parity.make_pair(adims, acts, 50, 0, kind, n_extra=0 if disc else ad, extra_init=-0.5); _, oc = parity.make_pair(cdims, acts, 50, 1)
    ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, E * T, EXTRAS)
    env = O.OEnv("cartpole", E, 60, 0.99, seed) if family == "cartpole" else O.OEnv(okind, E, 60, 0.99, seed, so=od, sa=ad)
    env.rollout(oa, parity.rollout_cfg(head=head), ob, T)
    O.chk(O.lib().orc_fill_gae(ob.h, oc.h, 0.95, 0.99)); O.chk(O.lib().orc_fill_returns(ob.h, 0.99)); O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"]))
    return {k: ob[k] for k in ob.keys()}


def _interleave(shards, bs):
    """row (rank r, position j) of the global buffer: minibatch j // bs holds [bs rows of rank 0 | bs rows of rank 1 | ...]."""
    R, N = len(shards), shards[0]["s"].shape[1]
    pos = np.empty((R, N), np.int64)
    for r in range(R):
        j = np.arange(N); pos[r] = (j // bs) * (R * bs) + r * bs + (j % bs)
    out = {}
    for k in shards[0]:
        rows = shards[0][k].shape[0]; a = np.empty((rows, R * N), shards[0][k].dtype, order="F")
        for r in range(R):
            a[:, pos[r]] = shards[r][k]
        out[k] = a
    return out, pos


def _run_group(d, world, cfg, shards, perms, seconds=75):
    """write the inputs, start one worker process per rank, wait (killing exactly the PIDs started here on a timeout), return the exit codes"""
    json.dump(cfg, open(os.path.join(d, "cfg.json"), "w"))
    for r in range(world):
        np.savez(os.path.join(d, "shard_%d.npz" % r), **shards[r]); np.save(os.path.join(d, "perms_%d.npy" % r), perms[r])
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, "--dir", d, "--rank", str(r), "--world", str(world)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs, rcs = [], []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=seconds)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, _ = p.communicate()
            o = (o or b"") + b"\n[test] killed: no result within %d s" % seconds
        outs.append(o.decode(errors="replace")); rcs.append(p.returncode)
    return rcs, outs


def _results(d, world):
    return [dict(np.load(os.path.join(d, "out_%d.npz" % r))) for r in range(world)]


def _assert_replicas_identical(res, names):
    for n in names:
        for f in ("_params", "_m", "_v", "_bp"):
            assert all(np.array_equal(res[0][n + f], x[n + f]) for x in res[1:]), "replicas differ in %s%s" % (n, f)


def _family_cfg(family, which, bs=128, epochs=2, k=1, **kw):
    le = 0.1 if parity.FAMILIES[family][2] else 0.0
    cfg = {"family": family, "which": which, "bs": bs, "epochs": epochs, "k": k, "seed": 77, "stream": 3, "P": {"eps": 0.2, "lambda_p": 1.0, "lambda_e": le}}
    cfg.update(kw); return cfg


def _oracle_nets(family):
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES[family]
    return {"actor": (adims, acts, 0 if disc else ad, "ppo", head, 3), "critic": (cdims, parity.CRITIC_ACTS.get(family, acts), 0, "value_mse", "deterministic", 4)}


@pytest.mark.parametrize("family", ["synth_c5", "cartpole"])
def test_two_processes_equal_the_single_learner_on_the_concatenated_batch(gpu_ctx, family):
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES[family]
    bs, epochs, world = 128, 2, 2
    shards = [_shard(family, 200 + r) for r in range(world)]; N = shards[0]["s"].shape[1]
    rng = np.random.default_rng(5)
    perms = [np.stack([np.stack([rng.permutation(N) for _ in range(epochs)]) for _ in range(2)]) for _ in range(world)]      # [rank][actor | critic][epoch][N]
    cfg = _family_cfg(family, ["actor", "critic"], bs, epochs)
    with tempfile.TemporaryDirectory(prefix="crux_xproc_") as d:
        rcs, outs = _run_group(d, world, cfg, shards, perms)
        assert rcs == [0] * world, "\n".join(outs)
        res = _results(d, world)
    _assert_replicas_identical(res, ["actor", "critic"])
    assert all(np.array_equal(res[0][w + "_info"], res[1][w + "_info"]) for w in ("actor", "critic"))           # the statistics are global
    glob, pos = _interleave(shards, bs)
    kindc = L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS
    for i, (which, (dims, ac, nx, loss, hd, strm)) in enumerate(_oracle_nets(family).items()):
        ob = O.OBuffer(od, ad, kindc, world * N, EXTRAS); ob.push(glob)
        o = O.OMlp(dims, ac, nx).init_glorot(77, strm, -0.5).adam_init(float(np.float32(3e-4)))
        gperm = np.empty((epochs, world * N), np.int64)
        for e in range(epochs):
            for r in range(world):
                gperm[e, pos[r]] = pos[r][perms[r][i][e]]
        tc = parity.train_cfg(loss, hd, world * bs, epochs, -1.0, 0, le=cfg["P"]["lambda_e"]); oi = np.zeros(L.INFO_N, np.float32)
        O.chk(O.lib().orc_batch_train(o.h, ob.h, C.byref(tc), O.vpz(gperm), O.vpz(oi), None))
        dth = float(np.abs(res[0][which + "_params"] - o.params).max())
        print("%s %s: two PROCESSES vs the concatenated-batch oracle after %d steps: max |dtheta| = %.3g; loss %.6g / %.6g" % (family, which, epochs * (N // bs), dth, res[0][which + "_info"][0], float(oi[0])))
        assert dth < parity.window_tol(0)
        assert res[0][which + "_info"][2] == epochs * (N // bs)
        assert abs(res[0][which + "_info"][0] - float(oi[0])) < 2e-5 * max(1.0, abs(float(oi[0])))
        assert abs(res[0][which + "_info"][1] - float(oi[1])) < 2e-5 * max(1.0, abs(float(oi[1])))


def _solo_reference(family, shard, perms, cfg):
    """the same learners in a GROUP OF ONE inside this process (crux_peer_attach with nranks = 1): the replica-group instantiation of the kernel, no peer"""
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES[family]
    ctx = crux.Context(0)
    try:
        ctx.peer_attach(0, 1, ctx.peer_export()[None, :])
        if cfg["k"] > 1:
            ctx.peer_set_sync_every(cfg["k"])
        N = shard["s"].shape[1]
        S, A = crux.ContinuousSpace(od), (crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad))
        buf = crux.ExperienceBuffer(S, A, N, EXTRAS, ctx=ctx)
        if disc:
            actor = crux.DiscreteNetwork(parity.chain(adims, acts), list(range(1, ad + 1)), ctx=ctx, seed=77, stream=3)
        else:
            actor = crux.GaussianPolicy(parity.chain(adims, acts), np.full(ad, -0.5, np.float32), ctx=ctx, seed=77, stream=3)
        critic = crux.ContinuousNetwork(parity.chain(cdims, parity.CRITIC_ACTS.get(family, acts)), ctx=ctx, seed=77, stream=4)
        out = {}
        for i, which in enumerate(("actor", "critic")):
            net = actor if which == "actor" else critic
            buf.clear_(); buf.push_(shard)
            opt = crux.TrainingParams(loss=crux.ppo_loss if which == "actor" else crux.value_mse_loss, batch_size=cfg["bs"], epochs=cfg["epochs"], name="n_")
            crux.batch_train_(net, opt, cfg["P"], buf, perms=perms[i] + 1)
            m, v, bp = net.adam_state()
            out[which + "_params"], out[which + "_m"], out[which + "_v"], out[which + "_bp"] = net.get_params(), m, v, bp
        ctx.peer_detach()
        return out
    finally:
        ctx.close()


@pytest.mark.parametrize("family,k", [("synth_c5", 1), ("synth_c5", 8), ("cartpole", 1), ("cartpole", 8)])
def test_identical_shards_across_processes_are_bit_identical_to_a_group_of_one(gpu_ctx, family, k):
    """both ranks hold the SAME rows and shuffles. k = 1: the exchanged sum is g + g, the mean (g + g) x 0.5 = g exactly; k = 8: the averaged theta / m / v are those of either
    replica. A group of one runs the same instantiation of the kernel, so the two-process group must reproduce it BIT FOR BIT; a lost, torn or stale slot read cannot."""
    bs, epochs, world = 128, 2, 2
    shard = _shard(family, 210); N = shard["s"].shape[1]
    rng = np.random.default_rng(6); p1 = np.stack([np.stack([rng.permutation(N) for _ in range(epochs)]) for _ in range(2)])
    cfg = _family_cfg(family, ["actor", "critic"], bs, epochs, k=k)
    with tempfile.TemporaryDirectory(prefix="crux_xproc_") as d:
        rcs, outs = _run_group(d, world, cfg, [shard, shard], [p1, p1])
        assert rcs == [0] * world, "\n".join(outs)
        res = _results(d, world)
    _assert_replicas_identical(res, ["actor", "critic"])
    ref = _solo_reference(family, shard, p1, cfg)
    for n in ("actor", "critic"):
        for f in ("_params", "_m", "_v", "_bp"):
            dd = float(np.abs(np.asarray(res[0][n + f], np.float64) - np.asarray(ref[n + f], np.float64)).max())
            assert np.array_equal(res[0][n + f], ref[n + f]), "%s%s differs from the group of one: max |d| = %.3g" % (n, f, dd)


def _local_sgd_twin(family, which, shards, perms, bs, epochs, k, le):
    """the oracle twin of the in-kernel periodic form (tests/test_gpu_peer.py: _local_sgd_twin, for any family): R oracle learners on their shards and shuffles, k local steps,
    then theta, m, v <- float32 sum in rank order x float32(1 / R)."""
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES[family]
    dims, ac, nx, loss, hd, strm = _oracle_nets(family)[which]
    R, N = len(shards), shards[0]["s"].shape[1]; nmb = N // bs
    os_, obs = [], []
    for r in range(R):
        ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, N, EXTRAS); ob.push(shards[r]); obs.append(ob)
        os_.append(O.OMlp(dims, ac, nx).init_glorot(77, strm, -0.5).adam_init(float(np.float32(3e-4))))
    inv = np.float32(1.0) / np.float32(R)

    def mean(xs):
        acc = xs[0].astype(np.float32).copy()
        for x in xs[1:]:
            acc = acc + x
        return acc * inv
    oi = np.zeros(L.INFO_N, np.float32); tc = parity.train_cfg(loss, hd, bs, 1, -1.0, 0, le=le)
    order = [np.arange(N, dtype=np.int64) for _ in range(R)]
    for e in range(epochs):
        for r in range(R):
            order[r] = order[r][perms[r][e]]
        for c0 in range(0, nmb, k):
            for r in range(R):
                for j in range(c0, c0 + k):
                    ids = np.ascontiguousarray(order[r][j * bs:(j + 1) * bs])
                    O.chk(O.lib().orc_train_step(os_[r].h, obs[r].h, C.byref(tc), O.vpz(ids), bs, O.vpz(oi)))
            th = mean([o.params for o in os_]); st = [o.adam_state() for o in os_]
            m = mean([x[0] for x in st]); v = mean([x[1] for x in st])
            for o in os_:
                o.params[:] = th; o.set_adam_state(m, v)
    return os_[0]


def test_periodic_form_across_processes_equals_the_local_sgd_twin(gpu_ctx):
    """crux_peer_set_sync_every(8) between two processes at C5's shapes: local Adam steps, theta / m / v averaged through the hipIpc-mapped slots after every 8th step"""
    family, bs, epochs, world, k = "synth_c5", 128, 2, 2, 8
    shards = [_shard(family, 700 + r) for r in range(world)]; N = shards[0]["s"].shape[1]
    rng = np.random.default_rng(9)
    perms = [np.stack([np.stack([rng.permutation(N) for _ in range(epochs)]) for _ in range(2)]) for _ in range(world)]
    cfg = _family_cfg(family, ["actor", "critic"], bs, epochs, k=k)
    with tempfile.TemporaryDirectory(prefix="crux_xproc_") as d:
        rcs, outs = _run_group(d, world, cfg, shards, perms)
        assert rcs == [0] * world, "\n".join(outs)
        res = _results(d, world)
    _assert_replicas_identical(res, ["actor", "critic"])
    for i, which in enumerate(("actor", "critic")):
        o = _local_sgd_twin(family, which, shards, [p[i] for p in perms], bs, epochs, k, cfg["P"]["lambda_e"])
        om, ov, _ = o.adam_state()
        dth = float(np.abs(res[0][which + "_params"] - o.params).max()); dm = float(np.abs(res[0][which + "_m"] - om).max()); dv = float(np.abs(res[0][which + "_v"] - ov).max())
        print("periodic form across processes, c5 %s, k = %d, %d steps: |dtheta| = %.3g |dm| = %.3g |dv| = %.3g vs the local-SGD oracle twin" % (which, k, epochs * (N // bs), dth, dm, dv))
        assert dth < 1e-6 and dm < 1e-6 * max(1.0, float(np.abs(om).max())) and dv < 1e-6 * max(1.0, float(np.abs(ov).max()))


def test_pair_call_across_processes_with_kl_early_stopping(gpu_ctx):
    """crux_policy_gradient_training on both ranks (two persistent kernels and two exchange streams per rank); the KL statistic is all-reduced with the gradient, so both
    processes stop on the same minibatch and stay bit-identical"""
    family, bs, world = "synth_c5", 128, 2
    shards = [_shard(family, 300 + r) for r in range(world)]; N = shards[0]["s"].shape[1]
    perms = [np.zeros((2, 1, N), np.int64) for _ in range(world)]      # (unused by the pair call: it shuffles with its seeds)
    for tk, ep in ((None, 3), (1e-6, 6)):
        cfg = _family_cfg(family, "pair", bs, ep, target_kl=tk)
        with tempfile.TemporaryDirectory(prefix="crux_xproc_") as d:
            rcs, outs = _run_group(d, world, cfg, shards, perms)
            assert rcs == [0] * world, "\n".join(outs)
            res = _results(d, world)
        _assert_replicas_identical(res, ["actor", "critic"])
        infos = [json.loads(bytes(x["info_json"]).decode()) for x in res]
        assert infos[0]["actor_batches_trained"] == infos[1]["actor_batches_trained"] and infos[0]["kl"] == infos[1]["kl"]
        if tk is None:
            assert infos[0]["actor_batches_trained"] == ep * (N // bs) and infos[0]["critic_batches_trained"] == ep * (N // bs)
        else:
            assert infos[0]["actor_batches_trained"] < ep * (N // bs), infos[0]


def test_a_dead_peer_yields_EHIP_on_the_survivor_not_a_hang(gpu_ctx):
    """rank 1 attaches and dies; rank 0 trains: its learner workgroups wait for rank 1's flag, give up after the in-kernel timeout (crux_peer_set_timeout_ms: 2 s here, 30 s by
    default) and the call returns CRUX_EHIP ("a replica of the group did not answer") -- the GPU is not left spinning."""
    family, bs, world = "synth_c5", 128, 2
    shards = [_shard(family, 400 + r, E=4) for r in range(world)]; N = shards[0]["s"].shape[1]
    perms = [np.stack([np.arange(N)[None, :]] * 2) for _ in range(world)]
    cfg = _family_cfg(family, ["actor"], bs, 1, timeout_ms=2000, die_rank=1)
    with tempfile.TemporaryDirectory(prefix="crux_xproc_") as d:
        rcs, outs = _run_group(d, world, cfg, shards, perms, seconds=60)
        assert rcs == [0, 0], "\n".join(outs)
        assert not os.path.exists(os.path.join(d, "out_0.npz")), "the survivor trained to the end without its peer"
        err = json.load(open(os.path.join(d, "err_0.json")))
    print("survivor:", err)
    assert err["code"] == L.EHIP and 1.5 < err["seconds"] < 30.0
    assert "did not answer" in err["message"] or "replica" in err["message"]
    # the device is still usable from this process afterwards
    g = crux.ContinuousNetwork(parity.chain([4, 64, 64, 1], parity.ACTS), ctx=gpu_ctx, seed=1, stream=0)
    assert np.isfinite(g.get_params()).all(); gpu_ctx.sync()


def _bench(args, seconds=270):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(29600 + os.getpid() % 300))
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=seconds)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, "rc %d\nstdout: %s\nstderr: %s" % (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    return json.loads(lines[0]), p.stderr


@pytest.mark.watchdog(300)
def test_bench_launcher_two_ranks_same_device_selftest_c5():
    """the command the driver's SCALE run issues, shrunk to one GPU: relaunch under torch.distributed.run, gloo rendezvous, hipIpc handle exchange, probe iteration, selftest,
    one timed C5-shard iteration per rank through the in-kernel exchange"""
    out, err = _bench(["--gpus", "2", "--same-device", "--selftest", "--workload", "c5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra", "--replicas", "0"])
    assert out["n_gpus"] == 2 and out["exchange"]["kind"] == "peer_slots_grad_every_step" and out["exchange"]["fell_back"] is False
    assert out["replicas_bit_identical_after_run"] is True
    st = out["selftest"]
    assert st["passed"] and st["identical_shards_replicas_equal"] and st["distinct_shards_replicas_bit_identical"]
    assert st["identical_shards_bit_identical_to_group_of_one"] is True, st
    assert out["value"] > 0 and out["exchange"]["flag_wait_per_rank"][0]["n"] > 0


@pytest.mark.watchdog(300)
def test_bench_fallback_ladder_when_peer_attach_fails():
    """crux_peer_attach made to fail on every rank (corrupted handles): all ranks agree to leave the in-kernel exchange and land on periodic parameter averaging -- the library's
    RCCL communicator when every rank has its own device, torch.distributed (gloo here: the ranks share the device, where RCCL cannot run) otherwise -- and still finish with
    identical replicas"""
    out, err = _bench(["--gpus", "2", "--same-device", "--workload", "c5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra", "--replicas", "0", "--inject-attach-failure"])
    ex = out["exchange"]
    assert ex["requested"] == "grad" and ex["fell_back"] is True and ex["kind"].startswith(("torch_allreduce_params", "rccl_params")), ex
    assert "cannot attach the peer regions" in err
    assert out["replicas_bit_identical_after_run"] is True and out["value"] > 0
