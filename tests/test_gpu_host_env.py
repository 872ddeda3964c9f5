"""Caller-stepped environments (north_star: "keeps the ... POMDPs.jl API surface"): step! with an ARBITRARY mdp on the host (src/sampler.jl:71-137) while the
policy forward, the exploration draws, the log-probabilities (crux_policy_explore), the ring write and the GAE / return fills (crux_steps_push) run on the device.

  * crux_policy_explore against the oracle's twin for every head;
  * the restated CartPole driven FROM THE HOST through crux.HostMDP reproduces crux_rollout's buffer bit for bit -- every column, advantages and returns included -- for the
    register-resident policy form (4 -> 64 -> 64 -> 2, C2's actor) and the generic one, with and without the reset at the end of a block, across consecutive blocks of a ring;
  * PPO and DQN `solve` on a HostMDP train exactly as on the device environment."""
import ctypes as C

import numpy as np
import pytest

import host_env as H
import parity
from parity import L, O, crux

pytestmark = pytest.mark.gpu


def _cfg(head, explore=1, i0=0, eps=None, noise=None, logit_div=0.0):
    cfg = parity.rollout_cfg(explore=bool(explore), head=head, i0=i0); cfg.explore = explore
    if eps:
        cfg.eps_start, cfg.eps_stop, cfg.eps_steps = eps
    if noise:
        cfg.noise_sigma, cfg.noise_eps_min, cfg.noise_eps_max, cfg.a_min, cfg.a_max = noise
    cfg.logit_div = logit_div
    return cfg


CASES = [
    # name, dims, acts, kind, n_extra, head, cfg kwargs, squash
    ("categorical_h64", [4, 64, 64, 2], ["relu", "relu", "identity"], "discrete", 0, "categorical", {}, 0.0),
    ("categorical_generic", [8, 32, 4], ["tanh", "identity"], "discrete", 0, "categorical", {}, 0.0),
    ("categorical_softq", [8, 32, 4], ["relu", "identity"], "discrete", 0, "categorical", {"logit_div": 0.5}, 0.0),
    ("always_stochastic_action", [4, 64, 64, 2], ["relu", "relu", "identity"], "discrete", 0, "categorical", {"explore": 2}, 0.0),
    ("greedy", [4, 64, 64, 2], ["relu", "relu", "identity"], "discrete", 0, "categorical", {"explore": 0}, 0.0),
    ("eps_greedy", [2, 8, 4], ["relu", "identity"], "discrete", 0, "greedy_q", {"eps": (1.0, 0.1, 500), "i0": 100}, 0.0),
    ("gaussian_h64", [17, 64, 64, 6], ["tanh", "tanh", "identity"], "gaussian", 6, "gaussian", {}, 0.0),
    ("gaussian_wide", [17, 256, 256, 6], ["relu", "relu", "identity"], "gaussian", 6, "gaussian", {}, 0.0),
    ("squashed_gaussian", [3, 64, 64, 1], ["relu", "relu", "identity"], "gaussian", 1, "gaussian", {}, 2.0),
    ("deterministic_noise", [3, 32, 1], ["relu", "tanh"], "continuous", 0, "deterministic", {"noise": (0.3, -0.5, 0.5, -1.0, 1.0)}, 0.0),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_policy_explore_equals_the_oracle_twin(gpu_ctx, case):
    name, dims, acts, kind, nx, head, kw, squash = case
    E, seed = 37, 0xC0FFEE
    g, o = parity.make_pair(dims, acts, 21, 2, kind, n_extra=nx, extra_init=-0.4)
    if squash:
        g = crux.SquashedGaussianPolicy(parity.chain(dims, acts), np.full(nx, -0.4, np.float32), ascale=squash, seed=21, stream=2); O.chk(O.lib().orc_mlp_set_squash(o.h, squash))
        assert np.array_equal(g.get_params(), o.params)
    rng = np.random.default_rng(3)
    obs = np.asfortranarray(rng.standard_normal((dims[0], E)).astype(np.float32))
    steps = rng.integers(0, 5000, E).astype(np.int64)
    cfg = _cfg(head, **kw)
    ga, glp = crux.policy_explore(g, cfg, obs, seed, steps)
    disc = head in ("categorical", "greedy_q")
    oa = np.zeros((dims[-1], E), np.uint8 if disc else np.float32, order="F"); olp = np.empty(E, np.float32)
    O.chk(O.lib().orc_policy_explore(o.h, C.byref(cfg), E, O.vpz(obs), seed, O.vpz(steps), O.vpz(oa), O.vpz(olp)))
    if disc:
        assert np.array_equal(ga, oa != 0) and (ga.sum(axis=0) == 1).all()
    else:
        assert float(np.abs(ga - oa).max()) < 2e-6
    assert np.array_equal(np.isnan(glp), np.isnan(olp))
    if not np.isnan(olp).all():
        assert float(np.nanmax(np.abs(glp - olp))) < 1e-5
    if kw.get("explore", 1) != 1 or head == "deterministic":
        assert np.isnan(glp).all()      # (action(pi, s), NaN) (sampler.jl:73)
    # the draws depend on (seed, steps_taken[e], e): another seed gives other samples, the same call the same bits
    ga2, glp2 = crux.policy_explore(g, cfg, obs, seed, steps)
    assert np.array_equal(ga, ga2) and np.array_equal(glp.view(np.uint32), glp2.view(np.uint32))
    if kw.get("explore", 1) != 0 and name != "greedy":
        ga3, _ = crux.policy_explore(g, cfg, obs, seed + 1, steps)
        assert not np.array_equal(ga, ga3)


def _host_cartpole(ctx, E, seed):
    return crux.HostMDP(H.cartpole_initialstate(seed), H.cartpole_gen(H.device_step(ctx)), H.cartpole_isterminal, H.cartpole_observation, 4, 2, True, n_envs=E, seed=seed, discount=0.99)


def _bits_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else np.array_equal(a, b)


@pytest.mark.parametrize("zerocopy", ["1", "0"], ids=["zero_copy_staging", "staged_copies"])
@pytest.mark.parametrize("dims,acts", [([4, 64, 64, 2], ["relu", "relu", "identity"]), ([4, 32, 32, 2], ["tanh", "tanh", "identity"])], ids=["h64_policy", "generic_policy"])
def test_cartpole_stepped_from_the_host_reproduces_crux_rollout_bit_for_bit(gpu_ctx, monkeypatch, dims, acts, zerocopy):
    """(round 6: crux_policy_explore reads the observations from and writes the actions into a pinned, device-mapped block in ONE launch, and a small host block is pushed by
    ONE kernel -- CRUX_HOST_ZEROCOPY=0 is the round-5 form with staged uploads and read-backs: both must reproduce the device rollout bit for bit)"""
    monkeypatch.setenv("CRUX_HOST_ZEROCOPY", zerocopy)
    E, T, max_steps, seed = 4, 48, 21, 777
    extras = ["return", "logprob", "advantage", "t", "i", "weight", "cost"]
    S, A = crux.ContinuousSpace(4, mu=np.float32(0.01), sigma=np.float32(1.5)), crux.DiscreteSpace(2)
    cap = 3 * E * T - 40                                                    # the third block wraps around the ring
    mk_pi = lambda: crux.ActorCritic(crux.DiscreteNetwork(parity.chain(dims, acts), [1, 2], seed=31, stream=0), crux.ContinuousNetwork(parity.chain(dims[:-1] + [1], acts), seed=31, stream=1))
    pi_d, pi_h = mk_pi(), mk_pi()
    bd = crux.ExperienceBuffer(S, A, cap, extras); bh = crux.ExperienceBuffer(S, A, cap, extras)
    sd = crux.Sampler(crux.CartPoleMDP(n_envs=E, seed=seed, discount=0.99), pi_d, S=S, max_steps=max_steps, required_columns=extras, lam=0.95)
    sh = crux.Sampler(_host_cartpole(gpu_ctx, E, seed), pi_h, S=S, max_steps=max_steps, required_columns=extras, lam=0.95)
    i = 0
    for blk, reset in enumerate((True, False, True)):
        info_d = crux.steps_(sd, bd, Nsteps=E * T, explore=True, i=i, reset=reset)
        info_h = crux.steps_(sh, bh, Nsteps=E * T, explore=True, i=i, reset=reset)
        i += E * T
        assert len(bd) == len(bh) and bd.next_ind == bh.next_ind
        for k in bd.keys():
            assert _bits_equal(bd[k], bh[k]), "block %d (reset=%s): column :%s differs" % (blk, reset, k)
        assert info_d["n_episode_end"] == info_h["n_episode_end"] and abs(info_d["sum_r"] - info_h["sum_r"]) < 1e-9
    assert bd["episode_end"].sum() > 3 * E and np.isfinite(bd["advantage"]).all() and float(np.abs(bd["advantage"]).max()) > 0
    # the samplers are in the same place: device state == the host Sampler's fields
    st, el, nr = sd.state()
    assert np.array_equal(el, sh.episode_length) and np.array_equal(nr, sh.n_resets) and np.array_equal(st, np.stack(sh.s, axis=1))


def test_ppo_solve_on_a_host_mdp_trains_like_the_device_environment(gpu_ctx):
    E, seed = 4, 4242
    def solver():
        pi = crux.ActorCritic(crux.DiscreteNetwork(parity.chain(parity.ACTOR_DIMS, parity.ACTS), [1, 2], seed=8, stream=0), crux.ContinuousNetwork(parity.chain(parity.CRITIC_DIMS, parity.ACTS), seed=8, stream=1))
        return crux.PPO(pi, crux.ContinuousSpace(4), N=3 * E * 64, dN=E * 64, max_steps=50, lambda_gae=0.95, a_opt={"batch_size": 128, "epochs": 2, "shuffle_seed": 5}, c_opt={"batch_size": 128, "epochs": 2, "shuffle_seed": 6})
    sv_d, sv_h = solver(), solver()
    pd = crux.solve(sv_d, crux.CartPoleMDP(n_envs=E, seed=seed, discount=0.99))
    ph = crux.solve(sv_h, _host_cartpole(gpu_ctx, E, seed))
    assert np.array_equal(pd.A.get_params(), ph.A.get_params()) and np.array_equal(pd.C.get_params(), ph.C.get_params())
    assert len(sv_h.history) == 3 and sv_h.history[-1]["actor_batches_trained"] == sv_d.history[-1]["actor_batches_trained"] > 0
    assert [h["avg_r"] for h in sv_d.history] == [h["avg_r"] for h in sv_h.history]


def test_dqn_solve_on_a_host_mdp_trains_like_the_device_environment(gpu_ctx):
    seed = 99
    def solver():
        q = crux.DiscreteNetwork(parity.chain([4, 32, 2], ["relu", "identity"]), [1, 2], seed=3, stream=0)
        return crux.DQN(q, crux.ContinuousSpace(4), N=400, dN=4, buffer_size=256, buffer_init=64, max_steps=30, c_opt={"batch_size": 32}, pi_explore=crux.EpsGreedyPolicy(crux.LinearDecaySchedule(1.0, 0.1, 200), [1, 2]))
    sv_d, sv_h = solver(), solver()
    pd = crux.solve(sv_d, crux.CartPoleMDP(n_envs=1, seed=seed, discount=0.99))
    ph = crux.solve(sv_h, _host_cartpole(gpu_ctx, 1, seed))
    for k in ("s", "a", "sp", "r", "done"):
        assert _bits_equal(sv_d.buffer[k], sv_h.buffer[k]), k
    assert np.array_equal(pd.get_params(), ph.get_params())


def test_steps_push_checks_its_arguments_before_the_ring_moves(gpu_ctx):
    """crux_steps_push on a buffer with an :advantage column and no critic: CRUX_EINVAL and the ring exactly as it was -- not pushed rows with stale advantages."""
    N = 64; extras = ["return", "logprob", "advantage"]
    b = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), 256, extras, ctx=gpu_ctx)
    rng = np.random.default_rng(3)
    cols = {"s": rng.standard_normal((4, N)).astype(np.float32), "sp": rng.standard_normal((4, N)).astype(np.float32), "a": np.eye(2, dtype=bool)[rng.integers(0, 2, N)].T.copy(),
            "r": np.ones((1, N), np.float32), "done": np.zeros((1, N), bool), "episode_end": np.zeros((1, N), bool), "logprob": np.zeros((1, N), np.float32)}
    cols["episode_end"][0, -1] = True
    ptrs = (C.c_void_p * L.NCOLS)(); keep = []
    for k, v in cols.items():
        arr = np.asfortranarray(v); keep.append(arr); ptrs[L.COL[k]] = arr.ctypes.data
    fr = C.c_int64(-1)
    rc = gpu_ctx.lib.crux_steps_push(b.h, N, ptrs, N, 1, None, 0.95, 0.99, None, None, 0, C.byref(fr))
    assert rc == L.EINVAL and "no critic" in (gpu_ctx.lib.crux_last_error(gpu_ctx.h) or b"").decode()
    assert len(b) == 0 and b.next_ind == 1 and fr.value == -1          # nothing moved (next_ind is 1-based like the reference field)
