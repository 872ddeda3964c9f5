"""GPU parity: one PPO iteration through the C ABI vs the CPU oracle (same Philox-defined randomness)."""
import ctypes as C

import numpy as np
import pytest

import parity
from parity import L, O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_envs,T,bs,epochs", [(4, 64, 32, 2), (8, 128, 128, 2), (3, 50, 64, 1)])
def test_ppo_iteration_matches_oracle(gpu_ctx, n_envs, T, bs, epochs):
    res = parity.ppo_iteration_parity(n_envs=n_envs, T=T, batch_size=bs, epochs=epochs, seed=5 + n_envs)
    assert res["ok"], res


@pytest.mark.parametrize("target_kl", [-1.0, 0.002])
def test_policy_gradient_training_pair_matches_sequential_oracle(gpu_ctx, target_kl):
    """actor || critic concurrent learners (exact when no early stopping) and the sequential fallback (KL stop) both equal the oracle."""
    res = parity.ppo_iteration_parity(n_envs=8, T=64, batch_size=64, epochs=3, seed=21, target_kl=target_kl, pair=True)
    assert res["ok"], res


@pytest.mark.parametrize("algo", ["a2c", "reinforce"])
def test_a2c_and_reinforce_solve_match_oracle_loop(gpu_ctx, algo):
    """solve(::OnPolicySolver) for A2C (a2c.jl:32-52) and REINFORCE (reinforce.jl:30-42): two iterations vs the same loop on the oracle."""
    import ctypes as C
    from parity import crux, L, O
    E, T, bs, seed, iters = 4, 32, 32, 9, 2
    N = E * T
    ga, oa = parity.make_pair(parity.ACTOR_DIMS, parity.ACTS, seed, 0, "discrete")
    gc, oc = parity.make_pair(parity.CRITIC_DIMS, parity.ACTS, seed, 1)
    S = crux.ContinuousSpace(4)
    mdp = crux.CartPoleMDP(n_envs=E, seed=seed)
    opt = {"batch_size": bs, "epochs": 2, "shuffle_seed": 31}
    if algo == "a2c":
        solver = crux.A2C(crux.ActorCritic(ga, gc), S, N=iters * N, dN=N, max_steps=25, a_opt=dict(opt), c_opt=dict(opt, shuffle_seed=32), lambda_e=0.05)
        extras = ["return", "logprob", "advantage"]
    else:
        solver = crux.REINFORCE(ga, S, N=iters * N, dN=N, max_steps=25, a_opt=dict(opt))
        extras = ["return", "logprob"]
    crux.solve(solver, mdp)
    # ---- oracle loop (on_policy.jl:80-109)
    ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, N, extras); oe = O.OEnv("cartpole", E, 25, 0.99, seed)
    oa.adam_init(float(np.float32(3e-4))); oc.adam_init(float(np.float32(3e-4)))
    info = np.zeros(L.INFO_N, np.float32); ol = O.lib(); ctr_a = ctr_c = 0
    for it in range(iters):
        cfg = parity.rollout_cfg(True, True, "categorical", i0=it * N)
        oe.rollout(oa, cfg, ob, T)
        if algo == "a2c":
            O.chk(ol.orc_fill_gae(ob.h, oc.h, 0.95, 0.99))
        O.chk(ol.orc_fill_returns(ob.h, 0.99))
        if algo == "a2c":
            O.chk(ol.orc_whiten(ob.h, L.COL["advantage"]))
        ca = parity.train_cfg(algo, "categorical", bs, 2, 0.015, 31, counter=ctr_a, le=0.05 if algo == "a2c" else 0.1)
        O.chk(ol.orc_batch_train(oa.h, ob.h, C.byref(ca), None, O.vpz(info), None))
        ctr_a += int(info[L.INFO["epochs_run"]])
        if algo == "a2c":
            cc = parity.train_cfg("value_mse", "deterministic", bs, 2, -1.0, 32, counter=ctr_c)
            O.chk(ol.orc_batch_train(oc.h, ob.h, C.byref(cc), None, O.vpz(info), None)); ctr_c += int(info[L.INFO["epochs_run"]])
    assert np.abs(ga.get_params() - oa.params).max() < 2e-6
    if algo == "a2c":
        assert np.abs(gc.get_params() - oc.params).max() < 2e-6
    assert np.isfinite(solver.history[-1]["actor_loss"]) and "kl" in solver.history[-1]


def test_multi_seed_batched_learners_match_single_calls(gpu_ctx, monkeypatch):
    """crux_policy_gradient_training_multi: n independent learners in two batched launches == n single calls with the per-replica seeds (bit for bit:
    same kernel, same arithmetic; only the launch geometry differs). The population launch runs the sample-split two-CU kernel, so the single calls are kept on
    it too (CRUX_FS=0: the feature-split form sums a minibatch gradient in another order)."""
    monkeypatch.setenv("CRUX_FS", "0")
    import os
    from parity import crux
    if os.environ.get("CRUX_MFMA_X2") == "0" or os.environ.get("CRUX_MFMA_WAVES4") or os.environ.get("CRUX_FORCE_GENERIC"):
        pytest.skip("a debug switch routes the single calls to a different kernel than the batched launch: bitwise equality is not expected")
    rng = np.random.default_rng(4); n_rep, n, bs = 5, 512, 128
    extras = ["return", "logprob", "advantage"]
    def make(seed):
        a = crux.DiscreteNetwork(parity.chain(parity.ACTOR_DIMS, parity.ACTS), [1, 2], seed=seed, stream=0)
        c = crux.ContinuousNetwork(parity.chain(parity.CRITIC_DIMS, parity.ACTS), seed=seed, stream=1)
        return crux.ActorCritic(a, c)
    datas = []
    for r in range(n_rep):
        ai = rng.integers(0, 2, n)
        datas.append({"s": rng.normal(0, 1, (4, n)).astype(np.float32), "a": np.eye(2, dtype=bool)[:, ai], "sp": rng.normal(0, 1, (4, n)).astype(np.float32), "r": np.ones((1, n), np.float32),
                      "done": np.zeros((1, n), bool), "episode_end": np.zeros((1, n), bool), "return": rng.normal(0, 1, (1, n)).astype(np.float32),
                      "logprob": rng.normal(-0.7, 0.05, (1, n)).astype(np.float32), "advantage": rng.normal(0, 1, (1, n)).astype(np.float32)})
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    def run(multi):
        pis = [make(50 + r) for r in range(n_rep)]
        bufs = [crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), n, extras) for _ in range(n_rep)]
        for b, d in zip(bufs, datas):
            b.push_(d)
        if multi:
            a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=3, name="actor_", shuffle_seed=900)
            c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=3, name="critic_", shuffle_seed=950)
            infos = crux.policy_gradient_training_multi(pis, a_opt, c_opt, P, bufs)
        else:
            infos = []
            for r in range(n_rep):
                class _S:
                    pass
                sv = _S(); sv.agent = crux.PolicyParams(pis[r]); sv.P = P
                sv.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=3, name="actor_", shuffle_seed=900 + r)
                sv.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=3, name="critic_", shuffle_seed=950 + r)
                infos.append(crux.policy_gradient_training(sv, bufs[r]))
        return [(pi.A.get_params(), pi.C.get_params()) for pi in pis], [b["s"] for b in bufs], infos
    pm, sm_, im = run(True); ps, ss, is_ = run(False)
    for r in range(n_rep):
        assert np.array_equal(pm[r][0], ps[r][0]) and np.array_equal(pm[r][1], ps[r][1]), r
        assert np.array_equal(sm_[r], ss[r])                                              # the buffers end up in the same (reference) row order
        assert im[r]["actor_loss"] == is_[r]["actor_loss"] and im[r]["critic_loss"] == is_[r]["critic_loss"]
    assert not np.array_equal(pm[0][0], pm[1][0])


def test_batched_rollouts_match_single_rollouts(gpu_ctx):
    """crux_rollout_multi == one crux_rollout per problem (same kernel; only the launch geometry differs): every column bit for bit."""
    import os
    from parity import crux
    if os.environ.get("CRUX_FORCE_GENERIC"):
        pytest.skip("the debug switch routes the single rollouts to the generic kernel, the batched launch exists for the register-resident one only")
    n_rep, E, T = 3, 4, 48
    def build():
        out = []
        for r in range(n_rep):
            a = crux.DiscreteNetwork(parity.chain(parity.ACTOR_DIMS, parity.ACTS), [1, 2], seed=70 + r, stream=0)
            c = crux.ContinuousNetwork(parity.chain(parity.CRITIC_DIMS, parity.ACTS), seed=70 + r, stream=1)
            pi = crux.ActorCritic(a, c); extras = ["return", "logprob", "advantage"]
            buf = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), E * T, extras)
            smp = crux.Sampler(crux.CartPoleMDP(n_envs=E, seed=300 + r), pi, max_steps=30, required_columns=extras, lam=0.95)
            out.append((pi, buf, smp))
        return out
    pa, pb = build(), build()
    ia = crux.steps_multi_([q[2] for q in pa], [q[1] for q in pa], Nsteps=E * T, explore=True, i=0, reset=True)
    ib = [crux.steps_(q[2], q[1], Nsteps=E * T, explore=True, i=0, reset=True) for q in pb]
    for r in range(n_rep):
        for k in pa[r][1].keys():
            assert np.array_equal(pa[r][1][k], pb[r][1][k], equal_nan=True) if pa[r][1][k].dtype.kind == "f" else np.array_equal(pa[r][1][k], pb[r][1][k]), (r, k)
        assert ia[r]["n_episode_end"] == ib[r]["n_episode_end"] and ia[r]["sum_r"] == ib[r]["sum_r"]
    assert not np.array_equal(pa[0][1]["s"], pa[1][1]["s"])
    crux.whiten_multi_([q[1] for q in pa], "advantage")                 # one launch for all problems == whiten_ per buffer
    for q in pb:
        crux.whiten_(q[1], "advantage")
    for r in range(n_rep):
        assert np.array_equal(pa[r][1]["advantage"], pb[r][1]["advantage"]), r


@pytest.mark.gpu
def test_synced_training_equals_plain_call_without_and_with_a_group_of_one(gpu_ctx):
    """crux_policy_gradient_training_synced: the learner launched per chunk of `sync_every` epochs (+ the stream-ordered RCCL all-reduce of
    parameters and Adam moments) must equal the single-launch call bit for bit when there is nothing to average: (1) no communicator,
    (2) a real RCCL communicator of size 1 (exercises dlopen, ncclCommInitRank, the grouped all-reduce and the 1/n scaling on the GPU)."""
    from parity import crux
    rng = np.random.default_rng(11); n, bs, E = 640, 128, 7
    extras = ["return", "logprob", "advantage"]
    ai = rng.integers(0, 2, n)
    data = {"s": rng.normal(0, 1, (4, n)).astype(np.float32), "a": np.eye(2, dtype=bool)[:, ai], "sp": rng.normal(0, 1, (4, n)).astype(np.float32), "r": np.ones((1, n), np.float32),
            "done": np.zeros((1, n), bool), "episode_end": np.zeros((1, n), bool), "return": rng.normal(0, 1, (1, n)).astype(np.float32),
            "logprob": rng.normal(-0.7, 0.05, (1, n)).astype(np.float32), "advantage": rng.normal(0, 1, (1, n)).astype(np.float32)}
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}

    def run(mode, ctx):
        a = crux.DiscreteNetwork(parity.chain(parity.ACTOR_DIMS, parity.ACTS), [1, 2], seed=3, stream=0, ctx=ctx)
        c = crux.ContinuousNetwork(parity.chain(parity.CRITIC_DIMS, parity.ACTS), seed=3, stream=1, ctx=ctx)
        pi = crux.ActorCritic(a, c)
        b = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), n, extras, ctx=ctx); b.push_(data)

        class _S:
            pass
        sv = _S(); sv.agent = crux.PolicyParams(pi); sv.P = P
        sv.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=E, name="actor_", shuffle_seed=31)
        sv.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=E, name="critic_", shuffle_seed=32)
        outs = []
        for _ in range(2):   # twice: Adam state and shuffle counters carry over
            info = crux.policy_gradient_training(sv, b) if mode == "plain" else crux.policy_gradient_training_synced(sv, b, sync_every=3)
            outs.append(info)
        m, v = a.get_adam_state() if hasattr(a, "get_adam_state") else (None, None)
        return pi.A.get_params(), pi.C.get_params(), b["s"], b["advantage"], outs, m, v

    ref = run("plain", gpu_ctx)
    got = run("synced", gpu_ctx)
    ctx1 = crux.Context(0)
    try:
        ctx1.comm_init(0, 1, ctx1.comm_unique_id())
        assert ctx1.comm_size() == 1
        grp = run("synced", ctx1)
        # crux_allreduce_grads on the group of one: a SUM over one rank leaves the gradient of crux_loss_grad as it is
        net = crux.DiscreteNetwork(parity.chain(parity.ACTOR_DIMS, parity.ACTS), [1, 2], seed=3, stream=0, ctx=ctx1)
        bb = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), n, extras, ctx=ctx1); bb.push_(data)
        pp = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=1, name="actor_"); crux.api._ensure_opt(net, pp)
        cfg = crux.api._train_cfg(net, pp, P); ids0 = np.arange(bs, dtype=np.int64); raw = np.zeros(L.INFO_N, np.float32)
        ctx1.check(ctx1.lib.crux_loss_grad(net.h, bb.h, C.byref(cfg), ids0.ctypes.data_as(C.c_void_p), bs, raw.ctypes.data_as(C.c_void_p)))
        g0 = np.empty(net.n_params, np.float32); ctx1.d2h(ctx1.lib.crux_mlp_grads_ptr(net.h), g0)
        ctx1.check(ctx1.lib.crux_allreduce_grads(net.h)); ctx1.sync()
        g1 = np.empty(net.n_params, np.float32); ctx1.d2h(ctx1.lib.crux_mlp_grads_ptr(net.h), g1)
        assert np.array_equal(g0, g1) and np.abs(g0).max() > 0
    finally:
        ctx1.comm_destroy()
    for other in (got, grp):
        for x, y in zip(ref[:4], other[:4]):
            assert np.array_equal(x, y)
        if ref[5] is not None:
            assert np.array_equal(ref[5], other[5]) and np.array_equal(ref[6], other[6])
        for i0, i1 in zip(ref[4], other[4]):
            assert i0["actor_batches_trained"] == i1["actor_batches_trained"] == E * (n // bs)
            for k in ("actor_loss", "critic_loss", "kl"):
                if k in i0:
                    assert np.isclose(i0[k], i1[k], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_one_cu_population_launch_matches_single_calls_and_the_two_cu_form(gpu_ctx):
    """Populations above 64 learners train with ONE CU per learner (the one-CU form of k_train_mfma, batched): bit-identical to single calls under
    Context.set_learner_cus(1), and equal to the two-CU form (different summation order of the minibatch gradient) to fp32 tolerance."""
    import os
    from parity import crux
    if os.environ.get("CRUX_MFMA_WAVES4") or os.environ.get("CRUX_FORCE_GENERIC"):
        pytest.skip("a debug switch routes the single calls to a different kernel than the batched launch")
    rng = np.random.default_rng(9); n_rep, n, bs, E = 66, 256, 128, 2
    extras = ["return", "logprob", "advantage"]
    datas = []
    for r in range(n_rep):
        ai = rng.integers(0, 2, n)
        datas.append({"s": rng.normal(0, 1, (4, n)).astype(np.float32), "a": np.eye(2, dtype=bool)[:, ai], "sp": rng.normal(0, 1, (4, n)).astype(np.float32), "r": np.ones((1, n), np.float32),
                      "done": np.zeros((1, n), bool), "episode_end": np.zeros((1, n), bool), "return": rng.normal(0, 1, (1, n)).astype(np.float32),
                      "logprob": rng.normal(-0.7, 0.05, (1, n)).astype(np.float32), "advantage": rng.normal(0, 1, (1, n)).astype(np.float32)})
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}

    def run(multi, cus):
        nonlocal n_rep
        ctx = crux.Context(0); ctx.set_learner_cus(cus)
        pis = [crux.ActorCritic(crux.DiscreteNetwork(parity.chain(parity.ACTOR_DIMS, parity.ACTS), [1, 2], seed=80 + r, stream=0, ctx=ctx),
                                crux.ContinuousNetwork(parity.chain(parity.CRITIC_DIMS, parity.ACTS), seed=80 + r, stream=1, ctx=ctx)) for r in range(n_rep)]
        bufs = [crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), n, extras, ctx=ctx) for _ in range(n_rep)]
        for b, d in zip(bufs, datas):
            b.push_(d)
        if multi:
            a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=E, name="actor_", shuffle_seed=700)
            c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=E, name="critic_", shuffle_seed=750)
            crux.policy_gradient_training_multi(pis, a_opt, c_opt, P, bufs)
        else:
            for r in range(n_rep):
                class _S:
                    pass
                sv = _S(); sv.agent = crux.PolicyParams(pis[r]); sv.P = P
                sv.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=E, name="actor_", shuffle_seed=700 + r)
                sv.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=E, name="critic_", shuffle_seed=750 + r)
                crux.policy_gradient_training(sv, bufs[r])
        return [(pi.A.get_params(), pi.C.get_params()) for pi in pis], [b["s"] for b in bufs]

    pm, sm_ = run(True, 0)        # automatic: 66 > 64 learners -> one CU each
    ps, ss = run(False, 1)        # single calls on the one-CU kernel
    for r in range(n_rep):
        assert np.array_equal(pm[r][0], ps[r][0]) and np.array_equal(pm[r][1], ps[r][1]), r
        assert np.array_equal(sm_[r], ss[r])
    n_rep = 20                    # a population that fits the two-CU form: both forms on the same 20 problems
    p1, s1 = run(True, 1); p2, s2 = run(True, 2)
    for r in range(n_rep):
        assert np.array_equal(p1[r][0], pm[r][0]) and np.array_equal(s1[r], s2[r])
        assert not np.array_equal(p1[r][0], p2[r][0])          # different kernels, different summation order ...
        assert np.allclose(p1[r][0], p2[r][0], rtol=0, atol=2e-5) and np.allclose(p1[r][1], p2[r][1], rtol=0, atol=2e-5), r      # ... same result to fp32 tolerance


@pytest.mark.gpu
def test_solve_writes_tensorboard_events_at_the_logger_period(gpu_ctx, tmp_path):
    """solve(PPO(...; log=LoggerParams(...))) (on_policy.jl:105, logging.jl:29-57): every `period` steps the training info, the evaluation
    closures (log_undiscounted_return over fresh episodes) and log_episode_averages land in a TensorBoard event file readtb can read."""
    from parity import crux
    from crux_jl_amd import logging as lg
    a = crux.DiscreteNetwork(parity.chain(parity.ACTOR_DIMS, parity.ACTS), [1, 2], seed=5, stream=0)
    c = crux.ContinuousNetwork(parity.chain(parity.CRITIC_DIMS, parity.ACTS), seed=5, stream=1)
    N, dN = 4 * 256, 256
    logp = lg.LoggerParams(dir=str(tmp_path / "log" / "ppo"), period=512, fns=[lg.log_undiscounted_return(4), lg.log_episode_averages(["r"], 512)])
    sv = crux.PPO(crux.ActorCritic(a, c), crux.ContinuousSpace(4), N=N, dN=dN, max_steps=40, a_opt={"epochs": 1}, c_opt={"epochs": 1}, log=logp)
    crux.solve(sv, crux.CartPoleMDP(n_envs=4, seed=1))
    h = lg.readtb(logp.logger.logdir)
    assert h["actor_loss"][0] == [512, 1024] and h["undiscounted_return"][0] == [0, 512, 1024] and sorted(set(h["avg_r"][0]) - {0}) == [512, 1024]   # step 0: the pre-train log(S.log, S.i, S=S) of on_policy.jl:88 (evaluation closures only);   # avg_r: record_avgr (ppo.jl:46) and log_episode_averages both write it, as in the reference
    assert np.isclose(h["actor_loss"][1][-1], sv.history[-1]["actor_loss"]) and all(1.0 <= v <= 40.0 for v in h["undiscounted_return"][1])
    assert all(np.isfinite(v) and 1.0 <= v <= 40.0 for i_, v in zip(h["avg_r"][0], h["avg_r"][1]) if i_ > 0)          # CartPole: reward 1 per step, episodes of 1..max_steps steps
