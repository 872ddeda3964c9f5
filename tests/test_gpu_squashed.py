"""GPU parity: SquashedGaussianPolicy (src/policies.jl:353-400) in the rollout kernels and the policy-gradient learner kernels vs the oracle.
The oracle side is pinned by torch float64 autograd and the reference's exploration/logpdf self-consistency test (tests/test_oracle_golden.py)."""
import ctypes as C

import numpy as np
import pytest

import parity
from parity import crux, L, O

pytestmark = pytest.mark.gpu
ASC = 2.0


def _pair(dims, acts, seed, ctx=None, ls=-0.4):
    ad = dims[-1]
    g = crux.SquashedGaussianPolicy(parity.chain(dims, acts), np.full(ad, ls, np.float32), ASC, seed=seed, stream=0, **({"ctx": ctx} if ctx else {}))
    o = O.OMlp(dims, acts, ad).init_glorot(seed, 0, ls); O.chk(O.lib().orc_mlp_set_squash(o.h, ASC))
    assert np.array_equal(g.get_params(), o.params)
    return g, o


@pytest.mark.parametrize("hidden", [64, 32])          # 64 -> k_rollout_h64, 32 -> generic k_rollout
@pytest.mark.parametrize("explore", [True, False])
def test_squashed_rollout_matches_oracle(gpu_ctx, hidden, explore):
    E, T = 4, 50
    g, o = _pair([3, hidden, hidden, 1], ["relu", "relu", "identity"], 6)
    extras = ["logprob"]
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(3), crux.ContinuousSpace(1), E * T, extras); ob = O.OBuffer(3, 1, L.ACTION_CONTINUOUS, E * T, extras)
    gs = crux.Sampler(crux.PendulumMDP(n_envs=E, seed=8), crux.PolicyParams(g), max_steps=30, required_columns=extras)
    oe = O.OEnv("pendulum", E, 30, 0.99, 8)
    info = crux.steps_(gs, gb, Nsteps=E * T, explore=explore, i=0, reset=True)
    osr, one = oe.rollout(o, parity.rollout_cfg(explore, True, "gaussian"), ob, T)
    assert info["n_episode_end"] == one and abs(info["sum_r"] - osr) < 1e-5 * max(1, abs(osr))
    a = gb["a"]; assert np.abs(a).max() <= ASC and np.abs(a - ob["a"]).max() < 2e-5
    for k in ("s", "sp", "r"):
        assert np.abs(gb[k] - ob[k]).max() < 1e-4 * max(1, np.abs(ob[k]).max()), k
    assert np.array_equal(gb["done"], ob["done"]) and np.array_equal(gb["episode_end"], ob["episode_end"])
    if explore:
        assert np.abs(gb["logprob"] - ob["logprob"]).max() < 1e-4
    else:
        assert np.isnan(gb["logprob"]).all()                      # no logprob without exploration (sampler.jl:73-76)


@pytest.mark.parametrize("cus", [0, 1])                  # 0: two-CU kernel where the shape allows, 1: one-CU 8-wave kernel
@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("dims,act,n,bs,loss", [([3, 64, 64, 1], "relu", 256, 128, "ppo"), ([17, 64, 64, 6], "tanh", 256, 128, "ppo"), ([3, 64, 64, 1], "relu", 200, 128, "a2c"),
                                                ([17, 64, 64, 6], "relu", 128, 64, "ppo")])
def test_squashed_learner_matches_oracle(gpu_ctx, monkeypatch, cus, force_generic, dims, act, n, bs, loss):
    if force_generic:
        if cus:
            pytest.skip("one switch at a time")
        monkeypatch.setenv("CRUX_FORCE_GENERIC", "1")
    ctx = crux.Context(0); ctx.set_learner_cus(cus)
    rng = np.random.default_rng(12); od, ad = dims[0], dims[-1]; acts = [act, act, "identity"]
    g, o = _pair(dims, acts, 15, ctx=ctx)
    p0 = o.params.copy(); p0[-ad:] = np.linspace(-0.6, 0.1, ad).astype(np.float32)
    if ad > 1:
        p0[-1] = 2.3                                          # one log-std beyond LOG_STD_MAX: sigma clamps, its gradient through clamp is 0
    g.set_params(p0); o.params[:] = p0
    extras = ["return", "logprob", "advantage"]
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.ContinuousSpace(ad), n, extras, ctx=ctx); ob = O.OBuffer(od, ad, L.ACTION_CONTINUOUS, n, extras)
    d = {"s": rng.normal(0, 1, (od, n)).astype(np.float32), "a": (ASC * np.tanh(rng.normal(0, 1.0, (ad, n)))).astype(np.float32), "sp": rng.normal(0, 1, (od, n)).astype(np.float32),
         "r": rng.normal(0, 1, (1, n)).astype(np.float32), "done": np.zeros((1, n), bool), "episode_end": np.zeros((1, n), bool),
         "return": rng.normal(0, 1, (1, n)).astype(np.float32), "advantage": rng.normal(0, 1, (1, n)).astype(np.float32)}
    d["a"][0, 0] = ASC                                         # exactly on the bound: clamped before atanh
    gb.push_(d); ob.push(d)
    # old logprob = the policy's own logpdf of the stored actions (+ noise), so that ratios are O(1): evaluate with the oracle
    cfg = parity.train_cfg(loss, "gaussian", bs, 2, -1.0, 7, 0, le=0.05)
    ids = np.arange(n, dtype=np.int64); oi = np.zeros(L.INFO_N, np.float32)
    ob.col("logprob")[...] = 0.0; ob.col("advantage")[...] = 1.0
    O.chk(O.lib().orc_loss_grad(o.h, ob.h, C.byref(parity.train_cfg("logpdf_bc", "gaussian", n, 1)), O.vpz(ids), n, O.vpz(oi)))
    # per-sample logpdf via n single-row calls would be slow; a noisy constant around the mean logpdf is enough for O(1) ratios
    lp = (-oi[L.INFO["kl"]] + 0.3 * rng.standard_normal((1, n))).astype(np.float32)
    d["logprob"] = lp; gb.clear_(); ob2 = O.OBuffer(od, ad, L.ACTION_CONTINUOUS, n, extras); gb.push_(d); ob2.push(d); ob = ob2
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.05}
    lossobj = crux.ppo_loss if loss == "ppo" else crux.a2c_loss
    p = crux.TrainingParams(loss=lossobj, optimizer=crux.Adam(np.float32(1e-3)), batch_size=bs, epochs=2, name="actor_", shuffle_seed=7)
    o.adam_init(float(np.float32(1e-3)))
    # gradient of one minibatch
    g.attach_optimizer(p.optimizer); raw = np.zeros(L.INFO_N, np.float32); tc = crux.api._train_cfg(g, p, P); mb = rng.permutation(n)[:bs].astype(np.int64)
    ctx.check(ctx.lib.crux_loss_grad(g.h, gb.h, C.byref(tc), O.vpz(mb), bs, O.vpz(raw)))
    O.chk(O.lib().orc_loss_grad(o.h, ob.h, C.byref(cfg), O.vpz(mb), bs, O.vpz(oi)))
    gg = np.empty(o.n, np.float32); ctx.d2h(ctx.lib.crux_mlp_grads_ptr(g.h), gg)
    assert np.abs(gg - o.grads).max() < 5e-6 * max(1.0, np.abs(o.grads).max())        # measured 4e-7 of the gradient's scale
    for k in ("loss", "grad_norm", "kl", "entropy", "clip_fraction"):
        assert abs(raw[L.INFO[k]] - oi[L.INFO[k]]) < 2e-6 * max(1.0, abs(oi[L.INFO[k]])), k
    # batch_train!: 2 epochs of the persistent learner
    info = crux.batch_train_(g, p, P, gb); oinfo = np.zeros(L.INFO_N, np.float32)
    O.chk(O.lib().orc_batch_train(o.h, ob.h, C.byref(cfg), None, O.vpz(oinfo), None))
    assert info["actor_batches_trained"] == int(oinfo[L.INFO["batches_trained"]])
    dp = np.abs(g.get_params() - o.params)
    assert dp.max() < 5e-6 and np.mean(dp > 3e-5) <= 2e-3, (dp.max(), np.mean(dp > 3e-5))
    assert abs(info["actor_loss"] - oinfo[0]) < 2e-3 * max(1, abs(oinfo[0]))


def test_ppo_with_squashed_gaussian_policy_solves(gpu_ctx):
    """examples/rl/pendulum.jl:20,32: PPO(pi=ActorCritic(SG(), V()), ...) with SG() = SquashedGaussianPolicy(mu, zeros(1), 2f0): the solve loop runs, actions are
    bounded by ascale and the greedy action is ascale*tanh(mu)."""
    acts = ["relu", "relu", "identity"]
    sg = crux.SquashedGaussianPolicy(parity.chain([3, 64, 64, 1], acts), np.zeros(1, np.float32), 2.0, seed=4)
    v = crux.ContinuousNetwork(parity.chain([3, 64, 64, 1], acts), seed=5)
    sv = crux.PPO(crux.ActorCritic(sg, v), crux.ContinuousSpace(3), N=3 * 512, dN=512, max_steps=100, lambda_e=0.0, a_opt={"epochs": 3, "batch_size": 128}, c_opt={"epochs": 3, "batch_size": 128})
    p0 = sg.get_params().copy()
    crux.solve(sv, crux.PendulumMDP(n_envs=8, seed=2))
    assert len(sv.history) == 3 and np.isfinite(sg.get_params()).all() and not np.array_equal(sg.get_params(), p0)
    assert np.abs(sv.buffer["a"]).max() <= 2.0 and np.isfinite(sv.buffer["logprob"]).all()
    assert all(np.isfinite(h["actor_loss"]) and abs(h["kl"]) < 1.0 for h in sv.history)
