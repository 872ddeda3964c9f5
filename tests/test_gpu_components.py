"""GPU parity of the individual C-ABI entry points against the CPU oracle and the reference's known answers."""
import ctypes as C
import os

import numpy as np
import pytest

import crux_jl_amd as crux
from crux_jl_amd import _lib as L
import oracle as O
import parity

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------------------------------------------- networks
@pytest.mark.parametrize("dims,acts", [([4, 64, 64, 2], ["relu", "relu", "identity"]), ([17, 64, 64, 6], ["tanh", "tanh", "identity"]),
                                       ([2, 8, 4], ["relu", "identity"]), ([8, 256, 256, 4], ["relu", "relu", "identity"])])
def test_mlp_init_and_forward_match_oracle(gpu_ctx, dims, acts):
    g, o = parity.make_pair(dims, acts, 11, 2)
    assert np.array_equal(g.get_params(), o.params)          # same Philox glorot draws
    x = np.random.default_rng(0).standard_normal((dims[0], 301)).astype(np.float32)
    yg, yo = g.forward(x), o.forward(x)
    assert np.abs(yg - yo).max() <= 1e-5 * max(1.0, np.abs(yo).max())      # fp32 GEMM tolerance (accumulation order)
    ps = g.params(); assert ps[0].shape == (dims[1], dims[0]) and ps[1].shape == (dims[1],)


@pytest.mark.parametrize("dims,acts", [([4, 64, 64, 1], ["relu", "relu", "identity"]), ([17, 64, 64, 1], ["tanh", "tanh", "identity"]), ([3, 64, 64, 1], ["relu", "tanh", "identity"]),
                                       ([8, 64, 64, 4], ["relu", "relu", "identity"]), ([27, 64, 64, 8], ["tanh", "tanh", "tanh"]), ([17, 64, 64, 6], ["tanh", "tanh", "identity"])])
def test_matrix_pipe_forward_is_the_scalar_forward_bit_for_bit(gpu_ctx, monkeypatch, dims, acts):
    """value(pi, x) of the IN-64-64-OUT family at large batches (fill_gae!'s critic evaluations, sampler.jl:264-266) runs on the matrix pipes (k_mlp_forward_h64, round 6): the MFMA
    chains carry k in ascending order, the order of the scalar kernel's fma loop, so the two kernels must agree BIT FOR BIT -- ragged last tile, non-zero biases, every
    activation pair -- and with the oracle to the forward tolerance."""
    g, o = parity.make_pair(dims, acts, 19, 2)
    rng = np.random.default_rng(7)
    p = g.get_params(); p = (p + rng.normal(0, 0.05, p.size)).astype(np.float32); g.set_params(p); o.params[:] = p          # non-zero biases
    B = 5003                                                                                                               # 312 full tiles + 11 samples
    x = np.asfortranarray(rng.standard_normal((dims[0], B)).astype(np.float32))
    y_mfma = g.forward(x)
    monkeypatch.setenv("CRUX_FORCE_GENERIC", "1")
    y_scalar = g.forward(x)
    monkeypatch.delenv("CRUX_FORCE_GENERIC")
    assert y_mfma.shape == (dims[-1], B) and np.array_equal(y_mfma.view(np.uint32), y_scalar.view(np.uint32))
    yo = o.forward(x)
    assert np.abs(y_mfma - yo).max() <= 1e-5 * max(1.0, np.abs(yo).max())
    small = g.forward(x[:, :301])                                                                                          # below the batch threshold: the scalar kernel itself
    assert np.array_equal(small.view(np.uint32), y_mfma[:, :301].view(np.uint32))


def test_polyak_and_copy_are_bit_exact(gpu_ctx):
    g1, o1 = parity.make_pair([3, 16, 1], ["relu", "identity"], 1, 0); g2, o2 = parity.make_pair([3, 16, 1], ["relu", "identity"], 2, 0)
    crux.polyak_average_(g1, g2, 0.005); O.chk(O.lib().orc_polyak(o1.h, o2.h, 0.005))
    assert np.array_equal(g1.get_params(), o1.params)
    crux.copyto_(g1, g2); assert np.array_equal(g1.get_params(), g2.get_params())


def test_adam_apply_matches_flux_float64_semantics(gpu_ctx):
    g, o = parity.make_pair([4, 32, 2], ["relu", "identity"], 3, 0)
    g.attach_optimizer(crux.Adam(np.float32(3e-4))); o.adam_init(float(np.float32(3e-4)))
    rng = np.random.default_rng(1)
    gp = g.ctx.lib.crux_mlp_grads_ptr(g.h)
    for _ in range(5):
        gr = rng.standard_normal(o.n).astype(np.float32)
        g.ctx.h2d(gp, gr); o.grads[:] = gr
        g.ctx.check(g.ctx.lib.crux_adam_apply(g.h, 0.5)); O.chk(O.lib().orc_adam_apply(o.h, 0.5))
    assert np.abs(g.get_params() - o.params).max() <= 1e-9     # device pow/sqrt/div in f64: identical after rounding
    m, v, bp = g.adam_state(); om, ov, obp = o.adam_state()
    assert np.array_equal(m, om) and np.array_equal(v, ov) and np.array_equal(bp, obp)


# ---------------------------------------------------------------------------------------------------- buffer
def _rand_data(rng, n, od, ad, discrete, extras=()):
    d = {"s": rng.standard_normal((od, n)).astype(np.float32), "sp": rng.standard_normal((od, n)).astype(np.float32),
         "r": rng.standard_normal((1, n)).astype(np.float32), "done": rng.random((1, n)) < 0.2, "episode_end": rng.random((1, n)) < 0.2}
    if discrete:
        a = np.zeros((ad, n), np.bool_); a[rng.integers(0, ad, n), np.arange(n)] = True; d["a"] = a
    else:
        d["a"] = rng.standard_normal((ad, n)).astype(np.float32)
    for k in extras:
        d[k] = rng.standard_normal((1, n)).astype(np.float32) if k != "t" else rng.integers(1, 50, (1, n))
    return d


def test_reference_buffer_tests_on_device(gpu_ctx):
    """test/experience_buffer_tests.jl:32-51,121-174 replayed against the device buffer."""
    b = crux.ExperienceBuffer(crux.ContinuousSpace(2), crux.ContinuousSpace(1), 100)
    d = {"s": 2 * np.ones((2, 50)), "a": np.ones((1, 50)), "sp": np.ones((2, 50)), "r": np.ones((1, 50)), "done": np.zeros((1, 50)), "weight": np.zeros((1, 50))}
    I = b.push_(d); assert list(I) == list(range(1, 51)) and len(b) == 50
    assert list(b.get_last_N_indices(10)) == list(range(41, 51)) and list(b.get_last_N_indices(51)) == list(range(1, 51))
    b.push_(d); b.push_(d)
    assert list(b.get_last_N_indices(51)) == [100] + list(range(1, 51)) and b.next_ind == 51 and b.total_count == 150
    assert list(b.get_last_N_indices(1000)) == list(range(51, 101)) + list(range(1, 51))
    b = crux.ExperienceBuffer(crux.ContinuousSpace(2), crux.DiscreteSpace(4), 100)
    assert b["a"].shape == (4, 0) and not b.haskey("weight")
    b.push_({"s": 2 * np.ones((2, 1)), "a": np.ones((4, 1), bool), "sp": np.ones((2, 1)), "r": np.ones((1, 1)), "done": np.zeros((1, 1))})
    assert len(b) == 1 and (b["s"] == 2).all() and b["a"].all()
    rng = np.random.default_rng(0); a3 = rng.random((4, 3)) < 0.5
    b.push_({"s": 3 * np.ones((2, 3)), "a": a3, "sp": 5 * np.ones((2, 3)), "r": 6 * np.ones((1, 3)), "done": np.ones((1, 3))})
    assert len(b) == 4 and (b["s"][:, 1:] == 3).all() and (b["a"][:, 1:] == a3).all() and (b["r"][:, 1:] == 6).all()
    b.push_(b)
    assert len(b) == 8
    for k in b.keys():
        assert (b[k][:, :4] == b[k][:, 4:8]).all()
    mb = b.minibatch([1, 2, 4])
    for k in mb:
        assert (mb[k] == b[k][:, [0, 1, 3]]).all()
    with pytest.raises(crux.CruxError):      # @assert size(v1)[1:end-1] == size(v2)[1:end-1]
        b.push_({"s": np.ones((3, 2)), "a": np.ones((4, 2), bool)})
    b.clear_(); assert len(b) == 0 and b.next_ind == 1


@pytest.mark.parametrize("cap,n_push", [(64, [10, 30, 40, 64, 7]), (1000, [999, 3])])
def test_ring_push_wraps_like_oracle(gpu_ctx, cap, n_push):
    rng = np.random.default_rng(2); extras = ["return", "logprob", "advantage", "t"]
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(5), crux.DiscreteSpace(3), cap, extras); ob = O.OBuffer(5, 3, L.ACTION_DISCRETE, cap, extras)
    for n in n_push:
        d = _rand_data(rng, n, 5, 3, True, extras)
        assert np.array_equal(gb.push_(d), ob.push(d))
        assert len(gb) == len(ob) and gb.next_ind == O.lib().orc_buffer_next_ind(ob.h) + 1
        for k in gb.keys():
            assert np.array_equal(gb[k], ob[k]), k
    perm = rng.permutation(len(gb)) + 1
    gb.shuffle_(perm); ob.permute(perm)
    for k in gb.keys():
        assert np.array_equal(gb[k], ob[k]), k
    ids = rng.integers(1, len(gb) + 1, 17)
    t_g = crux.ExperienceBuffer(crux.ContinuousSpace(5), crux.DiscreteSpace(3), 20, extras); t_o = O.OBuffer(5, 3, L.ACTION_DISCRETE, 20, extras)
    assert np.array_equal(t_g.push_(gb, ids=ids), t_o.push_buffer(ob, ids))
    for k in t_g.keys():
        assert np.array_equal(t_g[k], t_o[k]), k


def test_update_priorities_reference_kats_on_device(gpu_ctx):   # test/experience_buffer_tests.jl:193-205
    b = crux.ExperienceBuffer(crux.ContinuousSpace(2), crux.DiscreteSpace(4), 50, prioritized=True)
    assert b.haskey("weight") and len(b) == 0
    b.update_priorities_([1, 2, 3], np.array([1.0, 2.0, 3.0]))
    pp = b.priority_params()
    assert pp["max_priority"] == 3.0
    assert [float(x) for x in pp["priorities"][:3]] == [1.0000001192092896, 1.5157166719436646, 1.9331821203231812]
    d = {"s": 2 * np.ones((2, 3)), "a": np.ones((4, 3), bool), "sp": np.ones((2, 3)), "r": np.ones((1, 3)), "done": np.zeros((1, 3))}
    b.push_(d); b.push_(d)
    pp = b.priority_params()
    assert pp["max_priority"] == 3.0 and np.allclose(pp["priorities"][:6], np.float32(3.0) ** np.float32(0.6), rtol=1e-6)
    b.update_priorities_([4, 5], np.array([0.5, 0.25], np.float32))     # Float32 path (td errors)
    pp = b.priority_params(); assert pp["min_priority"] == np.float32(np.float32(0.25) + np.float32(1.1920929e-7))


# ---------------------------------------------------------------------------------------------------- dynamics
def test_device_dynamics_match_recordings(gpu_ctx):
    z = np.load(os.path.join(GOLD, "cartpole_transitions.npz"))
    s = np.ascontiguousarray(z["s"].T.astype(np.float64)); a = np.ascontiguousarray(z["a"].T.astype(np.uint8)); n = s.shape[0]
    ns = np.zeros_like(s); obs = np.zeros((n, 4), np.float32); rr = np.zeros(n, np.float32); dd = np.zeros(n, np.uint8)
    gpu_ctx.check(gpu_ctx.lib.crux_env_step_host(gpu_ctx.h, L.ENV["cartpole"], n, O.vpz(s), O.vpz(a), None, O.vpz(ns), O.vpz(obs), O.vpz(rr), O.vpz(dd)))
    assert np.abs(obs - z["sp"].T).max() <= 1e-6 and (rr == 1).all() and (dd == z["done"][0]).all()
    ons = np.zeros_like(s); oobs = np.zeros_like(obs); orr = np.zeros_like(rr); odd = np.zeros_like(dd)
    O.chk(O.lib().orc_env_step_host(L.ENV["cartpole"], n, O.vpz(s), O.vpz(a), None, O.vpz(ons), O.vpz(oobs), O.vpz(orr), O.vpz(odd)))
    assert np.abs(ns - ons).max() < 1e-14 and np.array_equal(obs, oobs)
    z = np.load(os.path.join(GOLD, "pendulum_transitions.npz"))
    s = np.ascontiguousarray(z["s"].T.astype(np.float64)); a = np.ascontiguousarray(z["a"].T.astype(np.float32)); n = s.shape[0]
    ns = np.zeros_like(s); obs = np.zeros((n, 3), np.float32); rr = np.zeros(n, np.float32); dd = np.zeros(n, np.uint8)
    gpu_ctx.check(gpu_ctx.lib.crux_env_step_host(gpu_ctx.h, L.ENV["pendulum"], n, O.vpz(s), O.vpz(a), None, O.vpz(ns), O.vpz(obs), O.vpz(rr), O.vpz(dd)))
    assert np.abs(ns[:, 1] - z["sp"].T[:, 1]).max() < 2e-6 and np.abs(rr - z["r"][0]).max() < 5e-6


# ---------------------------------------------------------------------------------------------------- rollouts
def _rollout_pair(kind, head, n_envs, T, cap, extras, explore=True, reset=True, seed=4, max_steps=40, eps=None, noise=None, gext=0, hidden=32):
    od, ad, disc = (4, 2, True) if kind == "cartpole" else (3, 1, False)
    dims = [od, hidden, hidden, ad]
    if head == "gaussian":
        g, o = parity.make_pair(dims, ["tanh", "tanh", "identity"], seed, 0, "gaussian", n_extra=ad, extra_init=-0.5)
    elif disc:
        g, o = parity.make_pair(dims, ["relu", "relu", "identity"], seed, 0, "discrete")
    else:
        g, o = parity.make_pair(dims, ["relu", "relu", "identity"], seed, 0)
    A = crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad)
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(od), A, cap, extras); ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, cap, extras)
    mdp = crux.GymMDP(kind, n_envs=n_envs, seed=seed)
    pe = crux.EpsGreedyPolicy(crux.LinearDecaySchedule(*eps), [1, 2]) if eps else crux.GaussianNoiseExplorationPolicy(**noise) if noise else None
    gs = crux.Sampler(mdp, crux.PolicyParams(g, pi_explore=pe), max_steps=max_steps, required_columns=extras)
    oe = O.OEnv(kind, n_envs, max_steps, 0.99, seed)
    cfg = parity.rollout_cfg(explore, reset, "greedy_q" if eps else ("deterministic" if (noise or (not disc and head != "gaussian")) else head))
    if eps:
        cfg.eps_start, cfg.eps_stop, cfg.eps_steps = eps
    if noise:
        cfg.noise_sigma, cfg.a_min, cfg.a_max = noise["sigma"], noise.get("a_min", -np.inf), noise.get("a_max", np.inf)
    return g, o, gb, ob, gs, oe, cfg


@pytest.mark.parametrize("hidden", [32, 64, 160])      # 64 -> register-resident k_rollout_h64, 32 -> generic one-wave k_rollout, 160 -> k_rollout_wide (a workgroup per environment)
@pytest.mark.parametrize("case", ["ppo_cartpole", "greedy_eval", "eps_greedy_offpolicy", "pendulum_gaussian", "pendulum_noise"])
def test_rollout_matches_oracle(gpu_ctx, case, hidden):
    if case == "ppo_cartpole":
        g, o, gb, ob, gs, oe, cfg = _rollout_pair("cartpole", "categorical", 5, 97, 5 * 97, ["logprob", "t", "i"], hidden=hidden)
        calls = [(5 * 97, 0)]
    elif case == "greedy_eval":
        g, o, gb, ob, gs, oe, cfg = _rollout_pair("cartpole", "categorical", 3, 60, 180, ["t"], explore=False, reset=False, hidden=hidden)
        calls = [(180, 0)]
    elif case == "eps_greedy_offpolicy":          # DQN-style: DN=4 per call into a ring, sampler state persists across calls
        g, o, gb, ob, gs, oe, cfg = _rollout_pair("cartpole", "categorical", 2, 2, 50, ["weight", "t", "i"], reset=False, eps=(1.0, 0.1, 40), hidden=hidden)
        calls = [(4, 4 * k) for k in range(20)]
    elif case == "pendulum_gaussian":
        g, o, gb, ob, gs, oe, cfg = _rollout_pair("pendulum", "gaussian", 4, 50, 200, ["logprob"], max_steps=30, hidden=hidden)
        calls = [(200, 0)]
    else:
        g, o, gb, ob, gs, oe, cfg = _rollout_pair("pendulum", "deterministic", 4, 25, 100, [], reset=False, noise={"sigma": 0.3, "a_min": -2.0, "a_max": 2.0}, max_steps=30, hidden=hidden)
        calls = [(100, 0), (100, 100)]
    E = gs.n_envs
    for (N, i0) in calls:
        cfg.i0 = i0
        info = crux.steps_(gs, gb, Nsteps=N, explore=bool(cfg.explore), i=i0, reset=bool(cfg.reset_at_end))
        osr, one = oe.rollout(o, cfg, ob, N // E)
        assert info["n_episode_end"] == one and abs(info["sum_r"] - osr) < 1e-5 * max(1, abs(osr))
    diff = parity.compare_buffers(gb, ob)
    for k in ("done", "episode_end", "t", "i"):
        if k in diff:
            assert diff[k] == 0, (k, diff)
    if gb.act_kind == L.ACTION_DISCRETE:
        assert diff["a"] == 0, diff
    else:
        assert diff["a"] < 1e-5, diff
    for k in ("s", "sp", "r", "logprob", "weight"):
        if k in diff:
            assert diff[k] < 1e-4, (k, diff)
    st_g, el_g, nr_g = gs.state(); st_o, el_o, nr_o = oe.state()
    assert np.array_equal(el_g, el_o) and np.array_equal(nr_g, nr_o) and np.abs(st_g - st_o).max() < 1e-6
    assert len(gb) == len(ob) and gb.next_ind == O.lib().orc_buffer_next_ind(ob.h) + 1


# ---------------------------------------------------------------------------------------------------- advantage pipeline
def test_gae_returns_whiten_match_oracle_on_random_episodes(gpu_ctx):
    rng = np.random.default_rng(5); n = 3000; extras = ["return", "logprob", "advantage"]
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), n, extras); ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, n, extras)
    d = _rand_data(rng, n, 4, 2, True, extras); d["episode_end"] = rng.random((1, n)) < 0.03; d["episode_end"][0, -1] = False   # last episode left open (:207-211)
    gb.push_(d); ob.push(d)
    gc, oc = parity.make_pair([4, 64, 64, 1], ["relu", "relu", "identity"], 6, 1)
    crux.fill_gae_(gb, gc, 0.95, 0.99); crux.fill_returns_(gb, 0.99)
    O.chk(O.lib().orc_fill_gae(ob.h, oc.h, 0.95, 0.99)); O.chk(O.lib().orc_fill_returns(ob.h, 0.99))
    assert np.array_equal(gb["return"], ob["return"])                   # sequential Float32 scan: bit-exact
    assert np.abs(gb["advantage"] - ob["advantage"]).max() < 5e-5       # depends on V(s) (fp32 GEMM tolerance)
    crux.whiten_(gb, "advantage"); O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"]))
    a = gb["advantage"][0]
    assert np.abs(a - ob["advantage"][0]).max() < 2e-5 and abs(a.mean()) < 1e-5 and abs(a.std(ddof=1) - 1) < 1e-5      # (different inputs: the advantages above agree to 5e-5; identical inputs whiten bit for bit, below)
    gb["r"] = np.full((1, n), np.nan, np.float32)
    with pytest.raises(crux.CruxError) as e:
        crux.fill_gae_(gb, gc, 0.95, 0.99)
    assert e.value.code == L.ENAN                                       # @assert !isnan(A)


@pytest.mark.parametrize("n", [2, 1000, 1480, 1025, 65536, 100003, 262144])
def test_whiten_is_julias_pairwise_float32_bit_for_bit(gpu_ctx, n):
    """whiten(v) = (v .- mean(v)) ./ std(v) (utils.jl:41-42) with Statistics.mean / std as Julia evaluates them on a Float32 vector (Base's pairwise sum, block 1024): the
    kernel and the oracle agree on every bit of the whitened column, for lengths on both sides of the block size and off the powers of two"""
    rng = np.random.default_rng(n); extras = ["return", "logprob", "advantage"]
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), n, extras); ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, n, extras)
    d = _rand_data(rng, n, 4, 2, True, extras); d["advantage"] = (rng.standard_normal((1, n)) * 2.5 + 0.3).astype(np.float32)
    gb.push_(d); ob.push(d)
    crux.whiten_(gb, "advantage"); O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"]))
    assert np.array_equal(gb["advantage"].view(np.uint32), ob["advantage"].view(np.uint32))
    a = gb["advantage"][0].astype(np.float64)
    assert abs(a.mean()) < 1e-5 and abs(a.std(ddof=1) - 1) < 1e-5


# ---------------------------------------------------------------------------------------------------- learner
def _train_pair(dims, acts, kind, n, od, ad, rng, n_extra=0):
    extras = ["return", "logprob", "advantage"]
    disc = kind != "gaussian"
    g, o = parity.make_pair(dims, acts, 21, 0, "discrete" if kind == "categorical" else ("gaussian" if kind == "gaussian" else "continuous"), n_extra=n_extra, extra_init=-0.3)
    A = crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad)
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(od), A, n, extras); ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, n, extras)
    d = _rand_data(rng, n, od, ad, disc, extras)
    d["logprob"] = (-0.7 + 0.05 * rng.standard_normal((1, n))).astype(np.float32) if disc else (-1.2 * ad + 0.1 * rng.standard_normal((1, n))).astype(np.float32)
    gb.push_(d); ob.push(d)
    return g, o, gb, ob


@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("kind,dims,n,bs,act", [("categorical", [4, 64, 64, 2], 128, 128, "relu"), ("categorical", [4, 64, 64, 2], 100, 37, "relu"), ("value", [4, 64, 64, 1], 128, 128, "relu"),
                                                 ("value", [4, 64, 64, 1], 77, 77, "relu"), ("categorical", [6, 32, 5], 64, 64, "relu"), ("gaussian", [17, 64, 64, 6], 96, 96, "relu"),
                                                 ("gaussian", [17, 64, 64, 6], 128, 128, "tanh"), ("value", [17, 64, 64, 1], 128, 128, "tanh"), ("value", [17, 64, 64, 1], 50, 50, "relu"),
                                                 ("gaussian", [3, 64, 64, 1], 128, 128, "relu"), ("value", [3, 64, 64, 1], 64, 64, "relu")])
def test_train_step_and_loss_grad_match_oracle(gpu_ctx, monkeypatch, force_generic, kind, dims, n, bs, act):
    if force_generic:
        monkeypatch.setenv("CRUX_FORCE_GENERIC", "1")
    rng = np.random.default_rng(7); acts = [act] * (len(dims) - 2) + ["identity"]
    od, ad = dims[0], (dims[-1] if kind != "value" else 2)
    g, o, gb, ob = _train_pair(dims, acts, kind if kind != "value" else "continuous", n, od, ad, rng, n_extra=dims[-1] if kind == "gaussian" else 0)
    loss = crux.value_mse_loss if kind == "value" else crux.ppo_loss
    p = crux.TrainingParams(loss=loss, batch_size=bs, epochs=1, name="x_"); P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    o.adam_init(float(np.float32(3e-4)))
    cfg = parity.train_cfg("value_mse" if kind == "value" else "ppo", "deterministic" if kind == "value" else kind, bs, 1)
    ids = rng.permutation(n)[:bs].astype(np.int64)
    # gradient only
    g.attach_optimizer(p.optimizer); g.optimizer = p.optimizer
    raw = np.zeros(L.INFO_N, np.float32); tc = crux.api._train_cfg(g, p, P)
    g.ctx.check(g.ctx.lib.crux_loss_grad(g.h, gb.h, C.byref(tc), O.vpz(ids), ids.size, O.vpz(raw)))
    oinfo = np.zeros(L.INFO_N, np.float32); O.chk(O.lib().orc_loss_grad(o.h, ob.h, C.byref(cfg), O.vpz(ids), ids.size, O.vpz(oinfo)))
    gg = np.empty(o.n, np.float32); g.ctx.d2h(g.ctx.lib.crux_mlp_grads_ptr(g.h), gg)
    scale = max(1.0, np.abs(o.grads).max())
    assert np.abs(gg - o.grads).max() < 1e-4 * scale, np.abs(gg - o.grads).max()      # rel 1e-4
    for k in ("loss", "grad_norm", "kl", "entropy", "clip_fraction", "avg_advantage", "avg_return"):
        assert abs(raw[L.INFO[k]] - oinfo[L.INFO[k]]) < 1e-4 * max(1.0, abs(oinfo[L.INFO[k]])), k
    assert np.array_equal(g.get_params(), o.params)                                   # gradient-only: parameters untouched
    # three Adam steps on different minibatches
    for s in range(3):
        ids = rng.permutation(n)[:bs].astype(np.int64)
        info = crux.train_(g, p, P, gb, ids + 1)
        O.chk(O.lib().orc_train_step(o.h, ob.h, C.byref(cfg), O.vpz(ids), ids.size, O.vpz(oinfo)))
        assert abs(info["x_loss"] - oinfo[0]) < 1e-4 * max(1, abs(oinfo[0]))
    assert np.abs(g.get_params() - o.params).max() < 2e-5      # Adam's first steps are ~lr*sign(g): near-zero gradient entries amplify 1e-7 differences
    m, v, bp = g.adam_state(); om, ov, obp = o.adam_state()
    assert np.abs(m - om).max() < 1e-5 * max(1, np.abs(om).max()) and np.allclose(bp, obp, rtol=1e-12)


@pytest.mark.parametrize("force_generic", [False, True])
def test_batch_train_early_stop_perms_and_ragged(gpu_ctx, monkeypatch, force_generic):
    if force_generic:
        monkeypatch.setenv("CRUX_FORCE_GENERIC", "1")
    rng = np.random.default_rng(8); n = 300; bs = 64; epochs = 4
    g, o, gb, ob = _train_pair([4, 64, 64, 2], ["relu", "relu", "identity"], "categorical", n, 4, 2, rng)
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    perms = np.stack([rng.permutation(n) + 1 for _ in range(epochs)])
    # (a) injected permutations, ragged last minibatch (300 = 4*64 + 44), no early stop
    p = crux.TrainingParams(loss=crux.ppo_loss, optimizer=crux.Adam(1e-3), batch_size=bs, epochs=epochs, name="actor_")
    o.adam_init(1e-3)
    info = crux.batch_train_(g, p, P, gb, perms=perms)
    cfg = parity.train_cfg("ppo", "categorical", bs, epochs); oinfo = np.zeros(L.INFO_N, np.float32); oep = np.zeros((epochs, L.INFO_N), np.float32)
    p0 = np.ascontiguousarray(perms - 1)
    O.chk(O.lib().orc_batch_train(o.h, ob.h, C.byref(cfg), O.vpz(p0), O.vpz(oinfo), O.vpz(oep)))
    assert info["actor_batches_trained"] == int(oinfo[L.INFO["batches_trained"]]) == epochs * 5
    assert np.abs(g.get_params() - o.params).max() < 2e-5
    assert np.allclose(info["_epoch_infos"][:, :7], oep[:, :7], rtol=2e-3, atol=2e-5)
    for k in gb.keys():                                                               # buffer order == all shuffles applied
        assert np.array_equal(gb[k], ob[k]), k
    # (b) KL early stopping with an aggressive step: same stopping minibatch as the oracle
    p2 = crux.TrainingParams(loss=crux.ppo_loss, optimizer=crux.Adam(0.03), batch_size=32, epochs=5, target_kl=0.01, name="actor_", shuffle_seed=5)
    o.adam_init(0.03); g.optimizer = None
    info2 = crux.batch_train_(g, p2, P, gb)
    cfg2 = parity.train_cfg("ppo", "categorical", 32, 5, 0.01, 5)
    O.chk(O.lib().orc_batch_train(o.h, ob.h, C.byref(cfg2), None, O.vpz(oinfo), None))
    assert info2["actor_batches_trained"] == int(oinfo[L.INFO["batches_trained"]]) and info2["_epochs_run"] == int(oinfo[L.INFO["epochs_run"]]) == 1
    assert info2["kl"] > 0.01 and abs(info2["kl"] - oinfo[L.INFO["kl"]]) < 1e-5
    # (c) max_batches
    p3 = crux.TrainingParams(loss=crux.ppo_loss, batch_size=32, epochs=5, name="actor_", max_batches=13, shuffle_seed=9); g.optimizer = None
    info3 = crux.batch_train_(g, p3, P, gb)
    assert info3["actor_batches_trained"] == 13 and info3["_epochs_run"] == 2


@pytest.mark.parametrize("force_generic", [False, True])
def test_nan_gradient_is_reported_and_parameters_are_kept(gpu_ctx, monkeypatch, force_generic):
    if force_generic:
        monkeypatch.setenv("CRUX_FORCE_GENERIC", "1")
    rng = np.random.default_rng(9)
    g, o, gb, ob = _train_pair([4, 64, 64, 1], ["relu", "relu", "identity"], "continuous", 128, 4, 2, rng)
    gb["return"] = np.full((1, 128), np.nan, np.float32)
    p = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=1, name="critic_")
    before = g.get_params()
    with pytest.raises(crux.CruxError) as e:
        crux.train_(g, p, {}, gb, np.arange(1, 129))
    assert e.value.code == L.ENAN and np.array_equal(g.get_params(), before)          # src/training.jl:20


@pytest.mark.parametrize("dims,acts", [([2, 8, 4], ["relu", "identity"]),                                   # C1 (README network): single-workgroup kernel
                                       ([8, 256, 256, 4], ["relu", "relu", "identity"])])                      # C3: dense tile-GEMM engine
def test_dqn_target_td_error_td_step_match_oracle(gpu_ctx, dims, acts):
    rng = np.random.default_rng(10); n = 128; od, na = dims[0], dims[-1]
    g, o = parity.make_pair(dims, acts, 31, 0, "discrete"); gt, ot = parity.make_pair(dims, acts, 32, 0, "discrete")
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(na), n, ["weight"]); ob = O.OBuffer(od, na, L.ACTION_DISCRETE, n, ["weight"])
    d = _rand_data(rng, n, od, na, True); d["weight"] = rng.random((1, n)).astype(np.float32); gb.push_(d); ob.push(d)
    ctx = g.ctx; dy = ctx.alloc(4 * n); de = ctx.alloc(4 * n)
    ctx.check(ctx.lib.crux_dqn_target(gt.h, gb.h, 0.95, dy)); y = ctx.d2h(dy, np.empty(n, np.float32))
    oy = np.empty(n, np.float32); O.chk(O.lib().orc_dqn_target(ot.h, ob.h, 0.95, O.vpz(oy)))
    assert np.abs(y - oy).max() < 1e-5
    ctx.h2d(dy, oy)
    ctx.check(ctx.lib.crux_td_error(g.h, gb.h, dy, de)); err = ctx.d2h(de, np.empty(n, np.float32))
    oerr = np.empty(n, np.float32); O.chk(O.lib().orc_td_error(o.h, ob.h, O.vpz(oy), O.vpz(oerr)))
    assert np.abs(err - oerr).max() < 1e-5
    g.attach_optimizer(crux.Adam(np.float32(1e-3))); o.adam_init(float(np.float32(1e-3)))
    for use_w in (0, 1):
        raw = np.zeros(L.INFO_N, np.float32); oinfo = np.zeros(L.INFO_N, np.float32)
        ctx.check(ctx.lib.crux_td_step(g.h, gb.h, dy, use_w, O.vpz(raw))); O.chk(O.lib().orc_td_step(o.h, ob.h, O.vpz(oy), use_w, O.vpz(oinfo)))
        assert abs(raw[0] - oinfo[0]) < 1e-5 * max(1, abs(oinfo[0])) and abs(raw[2] - oinfo[2]) < 1e-5 and abs(raw[1] - oinfo[1]) < 1e-4 * max(1, oinfo[1])
        gg = np.empty(g.n_params, np.float32); ctx.d2h(ctx.lib.crux_mlp_grads_ptr(g.h), gg)
        if max(dims) >= 128:                                                       # the dense path leaves the flat gradient in crux_mlp_grads_ptr
            assert np.abs(gg - o.grads).max() < 1e-4 * np.abs(o.grads).max()
    dp = np.abs(g.get_params() - o.params)
    assert dp.max() < (1e-6 if max(dims) < 128 else 5e-4) and np.mean(dp > 2e-5) <= 1e-3   # Adam's first steps amplify rounding where g ~ 0 (step = lr*g/(|g|+eps))
    # crux_td_step_with_error == crux_td_error followed by crux_td_step (one shared forward pass): bit for bit
    g2, _ = parity.make_pair(dims, acts, 31, 0, "discrete"); g2.attach_optimizer(crux.Adam(np.float32(1e-3)))
    g3, _ = parity.make_pair(dims, acts, 31, 0, "discrete"); g3.attach_optimizer(crux.Adam(np.float32(1e-3)))
    de2 = ctx.alloc(4 * n); r2, r3 = np.zeros(L.INFO_N, np.float32), np.zeros(L.INFO_N, np.float32)
    for _ in range(2):
        ctx.check(ctx.lib.crux_td_error(g2.h, gb.h, dy, de)); ctx.check(ctx.lib.crux_td_step(g2.h, gb.h, dy, 1, O.vpz(r2)))
        ctx.check(ctx.lib.crux_td_step_with_error(g3.h, gb.h, dy, 1, de2, O.vpz(r3)))
        assert np.array_equal(ctx.d2h(de, np.empty(n, np.float32)), ctx.d2h(de2, np.empty(n, np.float32))) if max(dims) < 128 else \
            np.abs(ctx.d2h(de, np.empty(n, np.float32)) - ctx.d2h(de2, np.empty(n, np.float32))).max() < 1e-6
        assert np.array_equal(g2.get_params(), g3.get_params()) and np.array_equal(r2, r3)
    ctx.free(dy); ctx.free(de); ctx.free(de2)


# ---------------------------------------------------------------------------------------------------- full-size properties
def test_full_size_iteration_properties(gpu_ctx):
    """BASELINE configs[1] sizes (32 envs x 2048 steps, batch 128): size-independent invariants instead of an oracle replay."""
    import bench
    pi, buf, sampler = bench.build_problem(crux, 123)
    info = crux.steps_(sampler, buf, Nsteps=buf.capacity, explore=True, i=0, reset=True)
    s, sp, ee, done, r, t_ret, adv, lp, a = buf["s"], buf["sp"], buf["episode_end"][0], buf["done"][0], buf["r"][0], buf["return"][0], buf["advantage"][0], buf["logprob"][0], buf["a"]
    N, T = buf.capacity, bench.T
    assert len(buf) == N and ee.sum() == info["n_episode_end"] and r.sum() == info["sum_r"] == N
    assert (ee[T - 1::T]).all()                                           # reset=true closes every env's rollout
    assert (a.sum(0) == 1).all() and np.isfinite(lp).all() and (lp <= 0).all()
    cont = ~ee[:-1]
    assert np.array_equal(sp[:, :-1][:, cont], s[:, 1:][:, cont])         # s_{t+1} == sp_t inside an episode
    assert (done <= ee).all()
    # returns recurrence R_t = r_t + gamma R_{t+1}, restarting after each episode_end (fill_returns!, sampler.jl:275-281)
    g32 = np.float32(0.99); nxt = np.where(ee[:-1], np.float32(0), t_ret[1:]).astype(np.float32)
    assert np.array_equal(t_ret[:-1], (r[:-1] + g32 * nxt).astype(np.float32)) and t_ret[-1] == r[-1]
    ep_len = np.diff(np.flatnonzero(np.concatenate([[True], ee])))
    assert ep_len.max() <= bench.MAX_STEPS
    crux.whiten_(buf, "advantage"); w = buf["advantage"][0]
    assert abs(w.mean()) < 1e-4 and abs(w.std(ddof=1) - 1) < 1e-4
    key = lambda b: np.sort(b["logprob"][0] * 1000 + b["return"][0])
    before = key(buf); P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=2, name="actor_")
    p0 = pi.A.get_params(); inf = crux.batch_train_(pi.A, a_opt, P, buf)
    assert inf["actor_batches_trained"] == 2 * 512 and np.array_equal(key(buf), before)      # shuffles only permute rows
    assert np.isfinite(pi.A.get_params()).all() and not np.array_equal(p0, pi.A.get_params())
    assert 0.55 < inf["entropy"] <= np.log(2) + 1e-5


# ---------------------------------------------------------------------------------------------------- replay sampling
@pytest.mark.parametrize("N", [1, 5, 127, 128, 129, 1000, 100_003, 1_000_000, 3_000_001])   # the last one exceeds the LDS-resident tree pass (k_tree_lds) and takes the global-memory one
def test_pairwise_cumsum_is_bit_exact(gpu_ctx, N):
    rng = np.random.default_rng(N)
    b = crux.ExperienceBuffer(crux.ContinuousSpace(1), crux.DiscreteSpace(2), N, prioritized=True)
    a = np.zeros((2, N), bool); a[0] = True
    b.push_({"s": np.zeros((1, N), np.float32), "a": a, "sp": np.zeros((1, N), np.float32), "r": np.zeros((1, N), np.float32), "done": np.zeros((1, N), bool)})
    v = (np.abs(rng.standard_normal(N)) + 1e-3).astype(np.float32)
    b.update_priorities_(np.arange(1, N + 1), v)
    pr = b.priority_params()["priorities"]; ref = np.empty(N, np.float32)
    O.lib().orc_pairwise_cumsum_f32(O.vpz(pr), N, O.vpz(ref))
    assert np.array_equal(b.cumsum(), ref)


@pytest.mark.parametrize("N,B", [(6, 1000), (5000, 128), (200_000, 128), (1_000_000, 128)])
def test_prioritized_sample_matches_oracle(gpu_ctx, N, B):
    rng = np.random.default_rng(N + B)
    od, ad = 8, 4
    src_g = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad), N, prioritized=True); src_o = O.OBuffer(od, ad, L.ACTION_DISCRETE, N, prioritized=True, alpha=np.float32(0.6))
    d = _rand_data(rng, N, od, ad, True); src_g.push_(d); src_o.push(d)
    v = (np.abs(rng.standard_normal(N)) + 1e-3)                      # Float64 values, like the reference's test (:246)
    I = np.arange(1, N + 1)
    src_g.update_priorities_(I, v); I0 = np.ascontiguousarray(I - 1); O.chk(O.lib().orc_per_update(src_o.h, O.vpz(I0), O.vpz(v), 1, N))
    tg = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad), B, ["weight"]); to = O.OBuffer(od, ad, L.ACTION_DISCRETE, B, ["weight"])
    for it, rands in enumerate([rng.random(B), None]):
        ids_g = crux.prioritized_sample_(tg, src_g, B=B, i=it + 1, rands=rands)
        O.chk(O.lib().orc_per_sample(to.h, src_o.h, B, O.vpz(rands) if rands is not None else None, 0.5, it + 1, crux.api.SAMPLE_SEED))
        ids_o = np.empty(B, np.int64); O.chk(O.lib().orc_buffer_indices(to.h, O.vpz(ids_o), B))
        assert np.array_equal(ids_g - 1, ids_o)                                            # bit-exact indices
        for k in ("s", "a", "sp", "r", "done"):
            assert np.array_equal(tg[k], to[k]), k
        wg, wo = src_g["weight"][0], src_o["weight"][0]
        assert np.abs(wg - wo).max() <= 4e-7 * max(1.0, wo.max()) and (wg[ids_o] <= 1 + 1e-6).all()
        assert np.abs(tg["weight"] - to["weight"]).max() <= 4e-7
    if N == 6:     # test/experience_buffer_tests.jl:246-262: stratified frequencies follow the priorities within 1 %
        pr = src_g.priority_params()["priorities"][:N].astype(np.float64)
        ids = crux.prioritized_sample_(tg, src_g, B=B, i=3)
        freqs = np.bincount(ids - 1, minlength=N) / B
        assert (np.abs(freqs - pr / pr.sum()) / (pr / pr.sum()) < 0.01).all()


def test_uniform_sample_and_rand(gpu_ctx):
    rng = np.random.default_rng(3); od, ad, N = 2, 4, 500
    src_g = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad), N); src_o = O.OBuffer(od, ad, L.ACTION_DISCRETE, N)
    d = _rand_data(rng, N, od, ad, True); src_g.push_(d); src_o.push(d)
    tg = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad), 64); to = O.OBuffer(od, ad, L.ACTION_DISCRETE, 64)
    ids = rng.integers(1, N + 1, 64)
    got = crux.uniform_sample_(tg, src_g, ids=ids)                        # test/experience_buffer_tests.jl:214-222 with explicit ids
    assert np.array_equal(got, ids) and np.array_equal(tg["s"], src_g["s"][:, ids - 1])
    got = crux.uniform_sample_(tg, src_g, i=7)
    O.chk(O.lib().orc_uniform_sample(to.h, src_o.h, 64, None, 7, crux.api.SAMPLE_SEED))
    ids_o = np.empty(64, np.int64); O.chk(O.lib().orc_buffer_indices(to.h, O.vpz(ids_o), 64))
    assert np.array_equal(got - 1, ids_o) and np.array_equal(tg["sp"], to["sp"]) and (got >= 1).all() and (got <= N).all()
    # multi-source rand! fill order 4/3/3 (test/experience_buffer_tests.jl:224-242)
    bufs = []
    for val in (1.0, 2.0, 3.0):
        t = crux.ExperienceBuffer(crux.ContinuousSpace(2), crux.DiscreteSpace(4), 10)
        t.push_({"s": val * np.ones((2, 1)), "a": np.ones((4, 1), bool), "sp": np.ones((2, 1)), "r": np.ones((1, 1)), "done": np.zeros((1, 1))}); bufs.append(t)
    t = crux.ExperienceBuffer(crux.ContinuousSpace(2), crux.DiscreteSpace(4), 10)
    crux.rand_(t, *bufs)
    s = t["s"]; assert (s[:, :4] == 1).all() and (s[:, 4:7] == 2).all() and (s[:, 7:] == 3).all()


# ---------------------------------------------------------------------------------------------------- off-policy solve (DQN, configs[0]-shaped)
def test_gridworld_dynamics_device_vs_oracle(gpu_ctx):
    rng = np.random.default_rng(12); n = 4000
    s = np.ascontiguousarray(rng.integers(1, 11, (n, 2)).astype(np.float64)); s[:40] = [4, 3]; s[40:80] = [9, 3]
    a = np.zeros((n, 4), np.uint8); a[np.arange(n), rng.integers(0, 4, n)] = 1; u = rng.random(n)
    outs = []
    for dev in (True, False):
        ns = np.zeros_like(s); obs = np.zeros((n, 2), np.float32); rr = np.zeros(n, np.float32); dd = np.zeros(n, np.uint8)
        if dev:
            gpu_ctx.check(gpu_ctx.lib.crux_env_step_host(gpu_ctx.h, L.ENV["gridworld"], n, O.vpz(s), O.vpz(a), O.vpz(u), O.vpz(ns), O.vpz(obs), O.vpz(rr), O.vpz(dd)))
        else:
            O.chk(O.lib().orc_env_step_host(L.ENV["gridworld"], n, O.vpz(s), O.vpz(a), O.vpz(u), O.vpz(ns), O.vpz(obs), O.vpz(rr), O.vpz(dd)))
        outs.append((ns, obs, rr, dd))
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
    ns, obs, rr, dd = outs[0]
    assert (rr[:40] == -10).all() and (rr[40:80] == 10).all() and dd[:80].all() and (ns[:80] == -1).all()
    moved = np.abs(ns[80:] - s[80:]).sum(1); free = rr[80:] == 0
    assert (moved[free] <= 1).all() and ns[80:][free].min() >= 1 and ns[80:][free].max() <= 10
    intended = (ns[80:] - s[80:])[free & (moved == 1)]; acts = a[80:][free & (moved == 1)].argmax(1)
    d = np.array([[0, 1], [0, -1], [-1, 0], [1, 0]])[acts]
    assert 0.6 < (intended == d).all(1).mean() < 0.8                                   # tprob = 0.7


@pytest.mark.parametrize("prioritized", [False, True])
def test_dqn_solve_matches_oracle_loop(gpu_ctx, prioritized):
    """solve(::OffPolicySolver) on SimpleGridWorld with the README's 2-8-4 network (BASELINE configs[0] shape), few steps."""
    E, dN, B, N, cap, seed, max_steps = 2, 4, 16, 48, 64, 5, 20
    g, o = parity.make_pair([2, 8, 4], ["relu", "identity"], 41, 0, "discrete", outputs=["up", "down", "left", "right"])
    S, A = crux.ContinuousSpace(2), crux.DiscreteSpace(4)
    mdp = crux.SimpleGridWorld(n_envs=E, seed=seed)
    buf = crux.ExperienceBuffer(S, A, cap, prioritized=prioritized)
    solver = crux.DQN(g, S, N=N, dN=dN, c_opt={"batch_size": B, "optimizer": crux.Adam(np.float32(1e-3))}, buffer=buf, buffer_init=B, max_steps=max_steps)
    crux.solve(solver, mdp)
    # ---- the same loop on the oracle (off_policy.jl:113-150, :66-111)
    ot = O.OMlp([2, 8, 4], ["relu", "identity"]); O.chk(O.lib().orc_mlp_copy(ot.h, o.h)); o.adam_init(float(np.float32(1e-3)))
    extras = ["weight"] if prioritized else []
    ob = O.OBuffer(2, 4, L.ACTION_DISCRETE, cap, extras, prioritized=prioritized, alpha=np.float32(0.6)); obt = O.OBuffer(2, 4, L.ACTION_DISCRETE, B, extras, prioritized=prioritized, alpha=np.float32(0.6))
    oe = O.OEnv("gridworld", E, max_steps, 0.95, seed)
    cfg = parity.rollout_cfg(True, False, "greedy_q"); cfg.eps_start, cfg.eps_stop, cfg.eps_steps = 1.0, 0.1, N // 2
    i = 0; istart = 0
    i += B; cfg.i0 = i; oe.rollout(o, cfg, ob, B // E)
    y = np.empty(B, np.float32); err = np.empty(B, np.float32); ids = np.empty(B, np.int64); info = np.zeros(L.INFO_N, np.float32)
    while i <= istart + N - dN:
        cfg.i0 = i; oe.rollout(o, cfg, ob, dN // E)
        for ep in range(dN):
            ctr = i * dN + ep
            if prioritized:
                O.chk(O.lib().orc_per_sample(obt.h, ob.h, B, None, 0.5, ctr, crux.api.SAMPLE_SEED))
            else:
                O.chk(O.lib().orc_uniform_sample(obt.h, ob.h, B, None, ctr, crux.api.SAMPLE_SEED))
            O.chk(O.lib().orc_dqn_target(ot.h, obt.h, 0.95, O.vpz(y)))
            if prioritized:
                O.chk(O.lib().orc_td_error(o.h, obt.h, O.vpz(y), O.vpz(err))); O.chk(O.lib().orc_buffer_indices(obt.h, O.vpz(ids), B))
                O.chk(O.lib().orc_per_update(ob.h, O.vpz(ids), O.vpz(err), 0, B))
            O.chk(O.lib().orc_td_step(o.h, obt.h, O.vpz(y), 0, O.vpz(info)))
            last_losses = ([] if ep == 0 else last_losses) + [float(info[0])]
        O.chk(O.lib().orc_polyak(ot.h, o.h, 0.005))
        i += dN
    assert solver.i == i and len(buf) == len(ob)
    for k in ("s", "a", "sp", "r", "done"):
        assert np.array_equal(buf[k], ob[k]), k                                          # same trajectories into the same ring slots
    assert np.abs(g.get_params() - o.params).max() < 2e-5
    assert np.abs(solver.agent.pi_minus.get_params() - ot.params).max() < 2e-5
    ref_loss = float(np.mean(last_losses))                                              # aggregate_info over the last value_training call's epochs (logging.jl:60-66)
    assert abs(solver.history[-1]["critic_loss"] - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    if prioritized:
        pg = buf.priority_params(); pr = np.empty(cap, np.float32); mx, mn = C.c_float(), C.c_float()
        O.chk(O.lib().orc_per_get(ob.h, O.vpz(pr), C.byref(mx), C.byref(mn), None))
        assert np.allclose(pg["priorities"], pr, rtol=1e-4, atol=1e-6) and abs(pg["max_priority"] - mx.value) < 1e-5


# ---------------------------------------------------------------------------------------------------- A2C / REINFORCE losses
@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("loss", ["a2c", "reinforce"])
@pytest.mark.parametrize("kind,dims", [("categorical", [4, 64, 64, 2]), ("gaussian", [3, 64, 64, 1]), ("gaussian", [17, 64, 64, 6]), ("categorical", [6, 32, 5])])
def test_a2c_and_reinforce_losses_match_oracle(gpu_ctx, monkeypatch, force_generic, loss, kind, dims):
    """a2c_loss (src/model_free/rl/a2c.jl:4-15) and reinforce_loss (reinforce.jl:4-13) through train! and batch_train! (incl. the two-CU kernel)."""
    if force_generic:
        monkeypatch.setenv("CRUX_FORCE_GENERIC", "1")
    rng = np.random.default_rng(17); acts = ["relu"] * (len(dims) - 2) + ["identity"]; n, bs = 256, 128
    g, o, gb, ob = _train_pair(dims, acts, kind, n, dims[0], dims[-1], rng, n_extra=dims[-1] if kind == "gaussian" else 0)
    lf = crux.a2c_loss if loss == "a2c" else crux.reinforce_loss
    p = crux.TrainingParams(loss=lf, batch_size=bs, epochs=2, name="actor_", shuffle_seed=5); P = {"lambda_p": 0.8, "lambda_e": 0.05}
    o.adam_init(float(np.float32(3e-4)))
    cfg = parity.train_cfg(loss, kind, bs, 2, seed=5, lp=0.8, le=0.05)
    ids = rng.permutation(n)[:bs].astype(np.int64)
    info = crux.train_(g, p, P, gb, ids + 1); oinfo = np.zeros(L.INFO_N, np.float32)
    O.chk(O.lib().orc_train_step(o.h, ob.h, C.byref(cfg), O.vpz(ids), ids.size, O.vpz(oinfo)))
    for k, ok in (("actor_loss", "loss"), ("actor_grad_norm", "grad_norm"), ("kl", "kl"), ("entropy", "entropy")):
        assert abs(info[k] - oinfo[L.INFO[ok]]) < 1e-4 * max(1.0, abs(oinfo[L.INFO[ok]])), (k, info[k], oinfo[L.INFO[ok]])
    binfo = crux.batch_train_(g, p, P, gb)
    O.chk(O.lib().orc_batch_train(o.h, ob.h, C.byref(cfg), None, O.vpz(oinfo), None))
    assert binfo["actor_batches_trained"] == int(oinfo[L.INFO["batches_trained"]]) == 4
    assert abs(binfo["actor_loss"] - oinfo[0]) < 1e-4 * max(1, abs(oinfo[0])) and abs(binfo["kl"] - oinfo[L.INFO["kl"]]) < 1e-5
    assert np.abs(g.get_params() - o.params).max() < 2e-5
    for k in ("s", "a", "advantage", "return"):
        assert np.array_equal(gb[k], ob[k]), k                                        # both shuffles materialised identically


def test_softq_target_and_solve_match_oracle(gpu_ctx):
    """softq_target(alpha) and SoftQ's softmax(Q ./ alpha) exploration (src/model_free/rl/softq.jl:1-13,52-53) + the solve loop vs the oracle."""
    E, dN, B, N, cap, seed, max_steps, alpha = 2, 4, 16, 40, 64, 8, 20, np.float32(0.5)
    g, o = parity.make_pair([2, 8, 4], ["relu", "identity"], 43, 0, "discrete", outputs=["up", "down", "left", "right"])
    S, A = crux.ContinuousSpace(2), crux.DiscreteSpace(4)
    mdp = crux.SimpleGridWorld(n_envs=E, seed=seed)
    solver = crux.SoftQ(g, S, N=N, dN=dN, alpha=alpha, c_opt={"batch_size": B, "optimizer": crux.Adam(np.float32(1e-3)), "epochs": dN}, buffer_size=cap, buffer_init=B, max_steps=max_steps)
    crux.solve(solver, mdp)
    ot = O.OMlp([2, 8, 4], ["relu", "identity"]); O.chk(O.lib().orc_mlp_copy(ot.h, o.h)); o.adam_init(float(np.float32(1e-3)))
    ob = O.OBuffer(2, 4, L.ACTION_DISCRETE, cap); obt = O.OBuffer(2, 4, L.ACTION_DISCRETE, B)
    oe = O.OEnv("gridworld", E, max_steps, 0.95, seed)
    cfg = parity.rollout_cfg(True, False, "categorical"); cfg.logit_div = alpha
    i = B; cfg.i0 = i; oe.rollout(o, cfg, ob, B // E)
    y, info = np.empty(B, np.float32), np.zeros(L.INFO_N, np.float32)
    while i <= N - dN:
        cfg.i0 = i; oe.rollout(o, cfg, ob, dN // E)
        for ep in range(dN):
            O.chk(O.lib().orc_uniform_sample(obt.h, ob.h, B, None, i * dN + ep, crux.api.SAMPLE_SEED))
            O.chk(O.lib().orc_softq_target(ot.h, obt.h, 0.95, alpha, O.vpz(y)))
            O.chk(O.lib().orc_td_step(o.h, obt.h, O.vpz(y), 0, O.vpz(info)))
            last_losses = ([] if ep == 0 else last_losses) + [float(info[0])]
        O.chk(O.lib().orc_polyak(ot.h, o.h, 0.005))
        i += dN
    assert solver.i == i and len(solver.buffer) == len(ob)
    for k in ("s", "a", "sp", "r", "done"):
        assert np.array_equal(solver.buffer[k], ob[k]), k
    assert np.abs(g.get_params() - o.params).max() < 2e-5 and np.abs(solver.agent.pi_minus.get_params() - ot.params).max() < 2e-5
    # the target alone, on the last staged batch
    dy = g.ctx.alloc(4 * B); g.ctx.check(g.ctx.lib.crux_softq_target(solver.agent.pi_minus.h, solver.batch.h, 0.95, float(alpha), dy))
    yg = g.ctx.d2h(dy, np.empty(B, np.float32)); O.chk(O.lib().orc_softq_target(ot.h, obt.h, 0.95, alpha, O.vpz(y)))
    assert np.abs(yg - y).max() < 1e-5
    g.ctx.free(dy)


@pytest.mark.parametrize("kind,dims,acts,head", [("cartpole", [4, 64, 64, 2], ["relu", "relu", "identity"], "discrete"), ("pendulum", [3, 32, 1], ["relu", "tanh"], "continuous")])
def test_batched_evaluation_matches_oracle(gpu_ctx, kind, dims, acts, head):
    """episodes! / undiscounted_return / discounted_return / failure (src/sampler.jl:175-251) as a batched greedy evaluation vs the oracle."""
    Neps, max_steps, seed = 24, 60, 12
    g, o = parity.make_pair(dims, acts, 19, 0, head)
    mdp = crux.GymMDP(kind, n_envs=2, seed=seed, discount=0.97)
    s = crux.Sampler(mdp, g, max_steps=max_steps)
    data, m = crux.episodes_(s, Neps=Neps)
    # oracle: the same Neps fresh environments, greedy policy, one max_steps rollout each, first episode metrics
    oe = O.OEnv(kind, Neps, max_steps, 0.97, seed + 0x45564C)
    ob = O.OBuffer(dims[0], 2 if kind == "cartpole" else 1, L.ACTION_DISCRETE if kind == "cartpole" else L.ACTION_CONTINUOUS, Neps * max_steps)
    cfg = parity.rollout_cfg(False, True, "categorical" if kind == "cartpole" else "deterministic")
    oe.rollout(o, cfg, ob, max_steps)
    und, dis, ln, ok = np.empty(Neps, np.float32), np.empty(Neps, np.float32), np.empty(Neps, np.int64), np.empty(Neps, np.uint8)
    O.chk(O.lib().orc_first_episode_metrics(ob.h, Neps, max_steps, np.float32(0.97), O.vpz(und), O.vpz(dis), O.vpz(ln), O.vpz(ok)))
    assert np.array_equal(m["length"], ln) and np.array_equal(m["complete"], ok.astype(bool)) and ok.all()
    assert np.abs(m["undiscounted"] - und).max() < 1e-4 * max(1, np.abs(und).max()) and np.abs(m["discounted"] - dis).max() < 1e-4 * max(1, np.abs(dis).max())
    if kind == "cartpole":
        assert np.array_equal(m["undiscounted"], ln.astype(np.float32))                   # r == 1 per step
        assert abs(crux.undiscounted_return(s, Neps=Neps) - ln.mean()) < 1e-6
        assert crux.failure(s, threshold=1e9, Neps=Neps) == 1.0 and crux.failure(s, threshold=0.0, Neps=Neps) == 0.0
    assert abs(crux.discounted_return(s, Neps=Neps) - float(np.mean(dis))) < 1e-4 * max(1, abs(float(np.mean(dis))))


# ---------------------------------------------------------------------------------------------------- edge cases and error behaviour
def test_edge_cases_and_error_codes(gpu_ctx):
    """Empty / ragged / oversized inputs and the reference's @assert and error() sites as status codes ("Error behaviour")."""
    ctx, rng = gpu_ctx, np.random.default_rng(31)
    S, A = crux.ContinuousSpace(4), crux.DiscreteSpace(2)
    extras = ["return", "logprob", "advantage"]
    g, o = parity.make_pair([4, 64, 64, 2], ["relu", "relu", "identity"], 61, 0, "discrete")
    p = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=1, name="actor_")
    # empty buffer: batch_train! has nothing to partition
    gb = crux.ExperienceBuffer(S, A, 100, extras); ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, 100, extras)
    with pytest.raises(crux.CruxError) as e:
        crux.batch_train_(g, p, {}, gb)
    assert e.value.code == L.EINVAL and len(gb) == 0
    # push of zero rows is a no-op; a push longer than the capacity wraps and the later rows win (circ_inds(1,120,100), test/experience_buffer_tests.jl:26-27)
    d = _rand_data(rng, 120, 4, 2, True, extras)
    gb.push_({k: v[:, :0] for k, v in d.items()}); assert len(gb) == 0
    I = gb.push_(d); Io = ob.push(d)
    assert len(gb) == len(ob) == 100 and gb.next_ind == 21 and np.array_equal(np.asarray(I), np.asarray(Io) if Io is not None else np.asarray(I))
    for k in ("s", "a", "r", "advantage"):
        assert np.array_equal(gb[k], ob[k]), k
    assert np.array_equal(gb["s"][:, :20], d["s"][:, 100:120]) and np.array_equal(gb["s"][:, 20:], d["s"][:, 20:100])
    # ragged last minibatch (100 rows, batch 37) and a minibatch index outside the buffer
    p37 = crux.TrainingParams(loss=crux.ppo_loss, batch_size=37, epochs=1, name="actor_")
    assert crux.batch_train_(g, p37, {}, gb)["actor_batches_trained"] == 3
    with pytest.raises(crux.CruxError) as e:
        crux.train_(g, p, {}, gb, np.array([1, 2, 101]))
    assert e.value.code == L.EINVAL
    # shape mismatches: critic loss on a 2-output network, rollout buffer with the wrong observation width
    with pytest.raises(crux.CruxError) as e:
        crux.batch_train_(g, crux.TrainingParams(loss=crux.value_mse_loss, batch_size=32, epochs=1), {}, gb)
    assert e.value.code == L.EINVAL
    bad = crux.ExperienceBuffer(crux.ContinuousSpace(3), A, 64)
    smp = crux.Sampler(crux.CartPoleMDP(n_envs=2, seed=1), g, max_steps=10)
    with pytest.raises(crux.CruxError) as e:
        crux.steps_(smp, bad, Nsteps=16)
    assert e.value.code == L.EINVAL
    # update_priorities! with a repeated index: the last value wins, max/min see every value (experience_buffer.jl:290-301)
    pb = crux.ExperienceBuffer(S, A, 50, prioritized=True); po = O.OBuffer(4, 2, L.ACTION_DISCRETE, 50, ["weight"], prioritized=True, alpha=np.float32(0.6))
    d50 = _rand_data(rng, 50, 4, 2, True); pb.push_(d50); po.push(d50)
    I = np.array([3, 7, 3, 9, 7, 3], np.int64); v = np.array([0.5, 2.0, 1.5, 0.1, 3.0, 0.25], np.float64)
    pb.update_priorities_(I, v); O.chk(O.lib().orc_per_update(po.h, O.vpz(I - 1), O.vpz(v), 1, I.size))
    pg = pb.priority_params(); pr = np.empty(50, np.float32); mx, mn = C.c_float(), C.c_float()
    O.chk(O.lib().orc_per_get(po.h, O.vpz(pr), C.byref(mx), C.byref(mn), None))
    assert np.array_equal(pg["priorities"], pr) and pg["max_priority"] == mx.value and pg["min_priority"] == mn.value
    assert abs(pg["priorities"][2] - np.float32((0.25 + np.finfo(np.float32).eps) ** 0.6)) < 1e-6


@pytest.mark.parametrize("case", ["c5_gaussian", "c3_eps_greedy", "c3_categorical"])
def test_synthetic_env_rollout_matches_oracle(gpu_ctx, case):
    """The SYNTH dynamics (include/cruxhip.h) for the C3- (8 obs / 4 discrete actions) and C5-shaped (17 obs / 6 continuous actions) configurations."""
    E, T, max_steps, seed = 5, 40, 16, 14
    if case == "c5_gaussian":
        od, ad, disc, dims, acts, kind, head = 17, 6, False, [17, 64, 64, 6], ["tanh", "tanh", "identity"], "gaussian", "gaussian"
    else:
        od, ad, disc, dims, acts, kind, head = 8, 4, True, [8, 32, 4], ["relu", "identity"], "discrete", "categorical" if case == "c3_categorical" else "greedy_q"
    g, o = parity.make_pair(dims, acts, 44, 0, kind, n_extra=ad if kind == "gaussian" else 0, extra_init=-0.5)
    mdp = crux.SynthMDP(od, ad, discrete=disc, n_envs=E, seed=seed, discount=0.98)
    pe = crux.EpsGreedyPolicy(crux.LinearDecaySchedule(1.0, 0.1, 100), list(range(ad))) if case == "c3_eps_greedy" else None
    smp = crux.Sampler(mdp, crux.PolicyParams(g, pi_explore=pe), max_steps=max_steps, S=crux.ContinuousSpace(od, mu=np.full(od, 0.01, np.float32), sigma=np.full(od, 0.5, np.float32)))
    A = crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad)
    extras = ["logprob"]
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(od), A, E * T, extras)
    ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, E * T, extras)
    oe = O.OEnv("synth_discrete" if disc else "synth", E, max_steps, 0.98, seed, mu=np.full(od, 0.01, np.float32), sigma=np.full(od, 0.5, np.float32), so=od, sa=ad)
    cfg = parity.rollout_cfg(True, True, head, i0=7)
    if case == "c3_eps_greedy":
        cfg.eps_start, cfg.eps_stop, cfg.eps_steps = 1.0, 0.1, 100
    gi = crux.steps_(smp, gb, Nsteps=E * T, explore=True, i=7, reset=True)
    osr, one = oe.rollout(o, cfg, ob, T)
    assert gi["n_episode_end"] == one and abs(gi["sum_r"] - osr) < 1e-4 * max(1, abs(osr))
    for k in ("a", "done", "episode_end"):
        assert np.array_equal(gb[k], ob[k]) if disc or k != "a" else np.abs(gb[k] - ob[k]).max() < 1e-5, k
    for k in ("s", "sp", "r"):
        assert np.abs(gb[k] - ob[k]).max() < 1e-5, k
    st_g, el_g, nr_g = smp.state(); st_o, el_o, nr_o = oe.state()
    assert np.array_equal(el_g, el_o) and np.array_equal(nr_g, nr_o) and np.abs(st_g - st_o).max() < 1e-9


@pytest.mark.parametrize("weighted,prioritized", [(False, False), (True, False), (False, True)])
def test_push_reservoir_matches_oracle(gpu_ctx, weighted, prioritized):
    """push_reservoir! (src/experience_buffer.jl:262-288): fill phase through push!, then uniform replacement with rand(1:total_count); the weight test
    skips rows; total_count grows by 2 per element while filling (kept quirk). Every column, the counters and the priorities vs the oracle, bit for bit."""
    rng = np.random.default_rng(17); C_, od, na = 37, 3, 2
    extras = ["weight"] if weighted else []
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(na), C_, extras, prioritized=prioritized)
    ob = O.OBuffer(od, na, L.ACTION_DISCRETE, C_, extras if not prioritized else ["weight"], prioritized=prioritized)
    ctr = 0
    for n in (5, 20, 30, 1, 64, 200):
        d = _rand_data(rng, n, od, na, True)
        if weighted:
            d["weight"] = rng.random((1, n)).astype(np.float32)
        gb.push_reservoir_(d, weighted=weighted, seed=99, counter=ctr); ob.push_reservoir(d, weighted, 99, ctr); ctr += n
        assert len(gb) == len(ob) and gb.total_count == O.lib().orc_buffer_total_count(ob.h) and gb.next_ind == O.lib().orc_buffer_next_ind(ob.h) + 1
        for k in gb.keys():
            assert np.array_equal(gb[k], ob[k]), (n, k)
    assert len(gb) == C_ and gb.total_count > C_
    if prioritized:
        pg = gb.priority_params()["priorities"]; po = np.empty(C_, np.float32); mx = C.c_float(); mn = C.c_float()
        O.chk(O.lib().orc_per_get(ob.h, O.vpz(po), C.byref(mx), C.byref(mn), None))
        assert np.array_equal(pg, po)
    if not weighted:      # every row of a full buffer came from one of the pushed batches, and late batches did replace early rows
        assert len(np.unique(gb["s"], axis=1)) > 1


def test_error_paths_of_the_newer_entries(gpu_ctx):
    """Argument checks fail loudly with the documented status codes: synced training refuses early stopping, a context takes one communicator,
    SquashedGaussianPolicy rejects a negative ascale and is refused by the SAC steps, push_reservoir! and GAIL validate shapes."""
    ctx = crux.Context(0)
    acts = ["relu", "relu", "identity"]
    a = crux.DiscreteNetwork(parity.chain([4, 64, 64, 2], acts), [1, 2], seed=1, ctx=ctx); c = crux.ContinuousNetwork(parity.chain([4, 64, 64, 1], acts), seed=2, ctx=ctx)
    buf = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), 256, ["return", "logprob", "advantage"], ctx=ctx)
    rng = np.random.default_rng(0); d = _rand_data(rng, 256, 4, 2, True, ["return", "logprob", "advantage"]); buf.push_(d)

    class _S:
        pass
    sv = _S(); sv.agent = crux.PolicyParams(crux.ActorCritic(a, c)); sv.P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    sv.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=2, target_kl=0.01, name="actor_")
    sv.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=2, name="critic_")
    with pytest.raises(crux.CruxError) as e:
        crux.policy_gradient_training_synced(sv, buf, 1)
    assert e.value.code == L.EUNSUP                                   # replicas would diverge in epoch count
    sv.a_opt.target_kl = None; sv.c_opt.epochs = 3
    with pytest.raises(crux.CruxError) as e:
        crux.policy_gradient_training_synced(sv, buf, 1)
    assert e.value.code == L.EINVAL                                   # actor / critic epoch counts differ
    uid = ctx.comm_unique_id(); ctx.comm_init(0, 1, uid)
    with pytest.raises(crux.CruxError) as e:
        ctx.comm_init(0, 1, uid)
    assert e.value.code == L.EINVAL
    ctx.comm_destroy(); assert ctx.comm_size() == 1
    with pytest.raises(crux.CruxError) as e:
        ctx.check(ctx.lib.crux_comm_init(ctx.h, 3, 2, uid.ctypes.data_as(L.vp)))
    assert e.value.code == L.EINVAL                                   # rank outside the group
    g = crux.SquashedGaussianPolicy(parity.chain([3, 32, 1], ["relu", "identity"]), np.zeros(1, np.float32), 2.0, ctx=ctx)
    with pytest.raises(crux.CruxError) as e:
        ctx.check(ctx.lib.crux_mlp_set_squash(g.h, -1.0))
    assert e.value.code == L.EINVAL and abs(ctx.lib.crux_mlp_get_squash(g.h) - 2.0) < 1e-7
    q1 = crux.ContinuousNetwork(parity.chain([4, 32, 1], ["relu", "identity"]), ctx=ctx); la = crux.ParamVector([0.0], ctx=ctx)
    cb = crux.ExperienceBuffer(crux.ContinuousSpace(3), crux.ContinuousSpace(1), 64, ctx=ctx)
    cb.push_({"s": rng.normal(0, 1, (3, 64)).astype(np.float32), "a": rng.uniform(-1, 1, (1, 64)).astype(np.float32), "sp": rng.normal(0, 1, (3, 64)).astype(np.float32),
              "r": np.zeros((1, 64), np.float32), "done": np.zeros((1, 64), bool)})
    dy = ctx.alloc(4 * 64)
    with pytest.raises(crux.CruxError) as e:
        ctx.check(ctx.lib.crux_sac_target(g.h, q1.h, q1.h, la.h, cb.h, 0.99, 0, 0, dy))
    assert e.value.code == L.EUNSUP
    with pytest.raises(crux.CruxError) as e:                          # discriminator must map act_dim + obs_dim -> 1
        ctx.check(ctx.lib.crux_gail_d_step(q1.h, cb.h, 0, 32, cb.h, 0, 100, np.zeros(L.INFO_N, np.float32).ctypes.data_as(L.vp)))
    assert e.value.code == L.EINVAL                                   # row range outside the buffer
    with pytest.raises(crux.CruxError):
        cb.push_reservoir_({"s": np.zeros((2, 5), np.float32)})       # wrong obs width
    ctx.free(dy)


def test_episodes_hcat_get_episodes_trim(gpu_ctx):
    """episodes(b) (experience_buffer.jl:194-221; test/gym/sampler_tests.jl:95-100 compares it with the ranges episodes! returns), hcat (:106-116),
    get_episodes (:150-156), trim! (:158-168) on device buffers vs the oracle's episode ranges and plain numpy slicing."""
    rng = np.random.default_rng(23); n = 90
    d = _rand_data(rng, n, 3, 2, True); ee = np.zeros((1, n), bool); ee[0, [9, 10, 40, 77]] = True; d["episode_end"] = ee; d["done"] = ee.copy()
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(3), crux.DiscreteSpace(2), n); ob = O.OBuffer(3, 2, L.ACTION_DISCRETE, n); gb.push_(d); ob.push(d)
    st, en = np.zeros(n, np.int64), np.zeros(n, np.int64); ne = O.lib().orc_buffer_episodes(ob.h, O.vpz(st), O.vpz(en), n)
    eps = crux.episodes(gb)
    assert eps == [(int(a) + 1, int(z) + 1) for a, z in zip(st[:ne], en[:ne])] == [(1, 10), (11, 11), (12, 41), (42, 78), (79, 90)]   # the open tail is closed at length(b)
    assert crux.episodes(gb, episode_checker=lambda b, e: e[1] - e[0] >= 5) == [(1, 10), (12, 41), (42, 78), (79, 90)]
    sub = crux.get_episodes(gb, [eps[1], eps[3]])
    assert len(sub) == 1 + 37 and np.array_equal(sub["s"], np.hstack([d["s"][:, 10:11], d["s"][:, 41:78]])) and np.array_equal(sub["a"], np.hstack([d["a"][:, 10:11], d["a"][:, 41:78]]))
    both = crux.hcat(gb, sub)
    assert len(both) == n + 38 and np.array_equal(both["r"], np.hstack([d["r"], d["r"][:, 10:11], d["r"][:, 41:78]])) and both.keys() == gb.keys()
    t = crux.trim_(gb, 25)
    assert len(t) == 25 and np.array_equal(t["sp"], d["sp"][:, :25]) and np.array_equal(t["episode_end"], ee[:, :25])
    with pytest.raises(crux.CruxError):
        crux.hcat(gb, crux.ExperienceBuffer(crux.ContinuousSpace(3), crux.DiscreteSpace(2), 4, ["return"]))
