"""GPU parity for the round-2 host/ABI fixes: block-local GAE fill in a larger ring, push!'s max_priority snapshot, rand!'s beta(i) and
per-source draws, action() of always_stochastic policies."""
import ctypes as C

import numpy as np
import pytest

import parity
from parity import L, O, crux

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("reset", [True, False])
def test_steps_fills_gae_on_the_pushed_block_of_a_larger_ring(gpu_ctx, reset):
    """steps! computes GAE / returns on the fresh rollout block before push! (src/sampler.jl:53-57,140-152), whatever the destination holds:
    three rollouts of 4 envs x 24 steps into a ring of 250 rows (the third wraps) must equal the oracle's per-block fill, and the rows of
    earlier blocks must keep their values."""
    E, T, seed, cap = 4, 24, 6, 250
    N = E * T
    extras = ["return", "logprob", "advantage"]
    ga, oa = parity.make_pair(parity.ACTOR_DIMS, parity.ACTS, seed, 0, "discrete")
    gc, oc = parity.make_pair(parity.CRITIC_DIMS, parity.ACTS, seed, 1)
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), cap, extras)
    gs = crux.Sampler(crux.CartPoleMDP(n_envs=E, seed=seed), crux.ActorCritic(ga, gc), max_steps=15, required_columns=extras, lam=0.95)
    oe = O.OEnv("cartpole", E, 15, 0.99, seed)
    ring = {k: np.zeros((1, cap), np.float32) for k in ("advantage", "return")}
    ee_ring = np.zeros(cap, bool)
    for it in range(3):
        first = gb.next_ind - 1
        crux.steps_(gs, gb, Nsteps=N, explore=True, i=it * N, reset=reset)
        ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, N, extras)          # the reference's fresh `data` block
        cfg = parity.rollout_cfg(True, reset, "categorical", i0=it * N)
        oe.rollout(oa, cfg, ob, T)
        ee = ob["episode_end"][0].copy()
        adv, ret = np.zeros(N, np.float32), np.zeros(N, np.float32)
        # per-environment segments: terminate_episode! fills the episodes that ended inside the block; open tails stay 0 (mdp_data zeros)
        Vs, Vsp = oc.forward(ob["s"])[0], oc.forward(ob["sp"])[0]
        r, done_u8 = np.ascontiguousarray(ob["r"][0]), np.ascontiguousarray(ob["done"][0]).view(np.uint8)
        Vs, Vsp = np.ascontiguousarray(Vs), np.ascontiguousarray(Vsp)
        for e in range(E):
            lo = e * T; start = lo
            for j in range(lo, lo + T):
                if ee[j]:
                    O.lib().orc_gae_range(O.vpz(r), O.vpz(done_u8), O.vpz(Vs), O.vpz(Vsp), start, j, 0.95, 0.99, O.vpz(adv))
                    O.lib().orc_returns_range(O.vpz(r), start, j, 0.99, O.vpz(ret))
                    start = j + 1
        idx = (first + np.arange(N)) % cap
        ring["advantage"][0, idx] = adv; ring["return"][0, idx] = ret; ee_ring[idx] = ee
        n = len(gb)
        assert np.array_equal(gb["episode_end"][0][idx[idx < n]], ee[idx < n])
        assert np.abs(gb["advantage"][0] - ring["advantage"][0, :n]).max() < 5e-5, it
        assert np.abs(gb["return"][0] - ring["return"][0, :n]).max() < 5e-5, it
    assert len(gb) == cap and gb.next_ind - 1 == (3 * N) % cap


def test_push_gives_every_new_row_the_same_snapshotted_max_priority(gpu_ctx):
    """push!: update_priorities!(b, I, max_priority*ones(N)) (src/experience_buffer.jl:254): all N rows get (m + eps)^alpha for the m read BEFORE the
    call, and max_priority rises once per push -- bit-exact against the oracle over repeated large pushes (many waves racing on the atomic max)."""
    rng = np.random.default_rng(3); cap, n = 200_000, 60_000
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(2), crux.DiscreteSpace(2), cap, prioritized=True)
    ob = O.OBuffer(2, 2, L.ACTION_DISCRETE, cap, ["weight"], prioritized=True, alpha=np.float32(0.6))
    for rep in range(5):
        d = {"s": rng.normal(0, 1, (2, n)).astype(np.float32), "sp": rng.normal(0, 1, (2, n)).astype(np.float32), "a": np.eye(2, dtype=bool)[:, rng.integers(0, 2, n)],
             "r": np.ones((1, n), np.float32), "done": np.zeros((1, n), bool)}
        I = gb.push_(d); ob.push(d)
        pg = gb.priority_params(); pr = np.empty(cap, np.float32); mx, mn = C.c_float(), C.c_float()
        O.chk(O.lib().orc_per_get(ob.h, O.vpz(pr), C.byref(mx), C.byref(mn), None))
        assert np.unique(pg["priorities"][I - 1]).size == 1                  # one value for the whole push
        assert np.array_equal(pg["priorities"], pr) and pg["max_priority"] == mx.value and pg["min_priority"] == mn.value, rep


def test_rand_evaluates_beta_at_i_and_draws_sources_independently(gpu_ctx):
    """rand!(target, sources...; i) (src/experience_buffer.jl:303-315): beta(i) with the caller's i (not the draw counter), and each source gets its own draw."""
    rng = np.random.default_rng(8); n = 400
    S, A = crux.ContinuousSpace(1), crux.DiscreteSpace(2)
    calls = []
    def beta(i):
        calls.append(i); return 0.4 + 0.001 * i
    def filled(prioritized, pp=None):
        b = crux.ExperienceBuffer(S, A, n, prioritized=prioritized, priority_params=pp)
        b.push_({"s": np.arange(n, dtype=np.float32)[None, :], "sp": np.zeros((1, n), np.float32), "a": np.eye(2, dtype=bool)[:, rng.integers(0, 2, n)],
                 "r": np.zeros((1, n), np.float32), "done": np.zeros((1, n), bool)})
        return b
    src = filled(True, {"alpha": 0.6, "beta": beta})
    src.update_priorities_(np.arange(1, n + 1), rng.random(n).astype(np.float32) + 0.1)
    tgt = crux.buffer_like(src, capacity=64)
    crux.rand_(tgt, src, i=123, counter=999_999)
    assert calls and set(calls) == {123}
    osrc = O.OBuffer(1, 2, L.ACTION_DISCRETE, n, ["weight"], prioritized=True, alpha=np.float32(0.6)); otgt = O.OBuffer(1, 2, L.ACTION_DISCRETE, 64, ["weight"], prioritized=True, alpha=np.float32(0.6))
    osrc.push({k: src[k] for k in ("s", "sp", "a", "r", "done")})
    # replay the identical update on the oracle (same rng stream)
    rng2 = np.random.default_rng(8); rng2.integers(0, 2, n); v = rng2.random(n).astype(np.float32) + 0.1
    O.chk(O.lib().orc_per_update(osrc.h, O.vpz(np.arange(n, dtype=np.int64)), O.vpz(v), 0, n))
    O.chk(O.lib().orc_per_sample(otgt.h, osrc.h, 64, None, np.float32(0.4 + 0.123), 999_999, crux.api.SAMPLE_SEED))
    oi = np.empty(64, np.int64); O.chk(O.lib().orc_buffer_indices(otgt.h, O.vpz(oi), 64))
    assert np.array_equal(tgt.indices[:64], oi)
    assert np.allclose(src["weight"], osrc["weight"], rtol=2e-6, atol=0)
    # two uniform sources of equal length: before the fix both halves held the same row numbers
    u1, u2 = filled(False), filled(False)
    t2 = crux.ExperienceBuffer(S, A, 128)
    crux.rand_(t2, u1, u2, i=5)
    rows = t2["s"][0]
    assert not np.array_equal(rows[:64], rows[64:])
    o1 = O.OBuffer(1, 2, L.ACTION_DISCRETE, n); o2 = O.OBuffer(1, 2, L.ACTION_DISCRETE, n); ot = O.OBuffer(1, 2, L.ACTION_DISCRETE, 128)
    o1.push({k: u1[k] for k in ("s", "sp", "a", "r", "done")}); o2.push({k: u2[k] for k in ("s", "sp", "a", "r", "done")})
    O.chk(O.lib().orc_buffer_set_sample_stream(o1.h, 0)); O.chk(O.lib().orc_buffer_set_sample_stream(o2.h, 1))
    O.chk(O.lib().orc_uniform_sample(ot.h, o1.h, 64, None, 5, crux.api.SAMPLE_SEED)); O.chk(O.lib().orc_uniform_sample(ot.h, o2.h, 64, None, 5, crux.api.SAMPLE_SEED))
    assert np.array_equal(t2["s"], ot["s"])


def test_action_of_an_always_stochastic_policy_samples_with_nan_logprob(gpu_ctx):
    """action(pi::DiscreteNetwork, s) = always_stochastic ? exploration(pi, s)[1] : argmax (src/policies.jl:124); step! stores logprob NaN for
    explore=false (src/sampler.jl:73). SoftQ's evaluation rollouts go through this path."""
    E, T, seed = 6, 20, 12
    g, o = parity.make_pair([4, 32, 2], ["relu", "identity"], seed, 0, "discrete")
    g.always_stochastic, g.logit_div = True, 0.7
    extras = ["logprob"]
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), E * T, extras)
    ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, E * T, extras)
    gs = crux.Sampler(crux.CartPoleMDP(n_envs=E, seed=seed), g, max_steps=50)
    crux.steps_(gs, gb, Nsteps=E * T, explore=False, i=0, reset=True)
    cfg = parity.rollout_cfg(False, True, "categorical"); cfg.explore = 2; cfg.logit_div = 0.7
    O.OEnv("cartpole", E, 50, 0.99, seed).rollout(o, cfg, ob, T)
    assert np.array_equal(gb["a"], ob["a"]) and np.isnan(gb["logprob"]).all() and np.isnan(ob["logprob"]).all()
    greedy = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), E * T, extras)
    g.always_stochastic = False
    crux.steps_(crux.Sampler(crux.CartPoleMDP(n_envs=E, seed=seed), g, max_steps=50), greedy, Nsteps=E * T, explore=False, i=0, reset=True)
    assert not np.array_equal(greedy["a"], gb["a"])                       # sampling differs from argmax on this seed


@pytest.mark.parametrize("prioritized", [True, False])
def test_fused_dqn_epoch_equals_the_separate_calls(gpu_ctx, prioritized):
    """crux_dqn_epoch (one fused launch per value_training epoch for wide networks, csrc/exec.hip) against the same epoch made of separate calls:
    same samples, same priorities, same parameters, bit for bit (the op bodies are shared)."""
    def run(fused):
        S, A = crux.ContinuousSpace(8), crux.DiscreteSpace(4)
        q = crux.DiscreteNetwork(parity.chain([8, 256, 256, 4], ["relu", "relu", "identity"]), [1, 2, 3, 4], seed=3)
        mdp = crux.SynthMDP(8, 4, discrete=True, n_envs=1, seed=5, discount=0.97)
        sv = crux.DQN(q, S, N=600, dN=4, buffer_size=2000, prioritized=prioritized, buffer_init=400, max_steps=40, c_opt={"batch_size": 128})
        sv.fused_epochs = fused
        crux.solve(sv, mdp)
        pp = sv.buffer.priority_params() if prioritized else None
        return q.get_params(), sv.agent.pi_minus.get_params(), sv.buffer["s"], pp, sv.history
    pa, ta, sa, ppa, ha = run(True); pb, tb, sb, ppb, hb = run(False)
    assert np.array_equal(sa, sb)
    assert np.array_equal(pa, pb) and np.array_equal(ta, tb), (np.abs(pa - pb).max(), np.abs(ta - tb).max())      # the op bodies are shared: bit for bit
    if prioritized:
        assert np.allclose(ppa["priorities"], ppb["priorities"], rtol=1e-5, atol=1e-7) and abs(ppa["max_priority"] - ppb["max_priority"]) < 1e-5
    assert abs(ha[-1]["critic_loss"] - hb[-1]["critic_loss"]) < 1e-5 * max(1.0, abs(hb[-1]["critic_loss"]))


@pytest.mark.parametrize("B", [256, 512])
def test_fused_sac_epoch_equals_the_separate_calls(gpu_ctx, B):
    """crux_sac_epoch (rand! -> sac_target -> temperature -> twin critics -> actor -> polyak as one fused launch) against the separate calls. B = 256: the target and
    the critic heads are one-block ops and share a block (sequential group); B = 512: the target spans two blocks and keeps its own phase."""
    def run(fused):
        S = crux.ContinuousSpace(3)
        acts = ["relu", "relu", "identity"]
        pi = crux.ActorCritic(crux.GaussianPolicy(parity.chain([3, 256, 256, 1], acts), np.zeros(1, np.float32), seed=2),
                              crux.DoubleNetwork(crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=3), crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=4)))
        sv = crux.SAC(pi, S, N=420 + 2 * B, dN=6, buffer_size=2000, buffer_init=300 + 2 * B, max_steps=50, c_opt={"batch_size": B}, a_opt={"batch_size": B}, SAC_alpha_opt={"batch_size": B})
        sv.fused_epochs = fused
        crux.solve(sv, crux.PendulumMDP(n_envs=1, seed=8))
        return [n.get_params() for n in (pi.A, pi.C.N1, pi.C.N2, sv.agent.pi_minus.C.N1, sv.P["SAC_log_alpha"])], sv.history
    a, ha = run(True); b, hb = run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y), np.abs(x - y).max()
    for k in ("critic_loss", "actor_loss", "temp_loss", "SAC alpha"):
        assert abs(ha[-1][k] - hb[-1][k]) < 1e-5 * max(1.0, abs(hb[-1][k])), k


def test_fused_softq_epochs_equal_the_separate_calls(gpu_ctx):
    """crux_softq_epochs: the DQN epoch chain with softq_target(alpha) (rl/softq.jl:4-13) in place of dqn_target, against the call-by-call value_training."""
    def run(fused):
        S = crux.ContinuousSpace(8)
        q = crux.DiscreteNetwork(parity.chain([8, 256, 256, 4], ["relu", "relu", "identity"]), [1, 2, 3, 4], seed=3)
        sv = crux.SoftQ(q, S, N=560, dN=4, alpha=0.5, buffer_size=2000, buffer_init=400, max_steps=40, c_opt={"batch_size": 128})
        sv.fused_epochs = fused
        crux.solve(sv, crux.SynthMDP(8, 4, discrete=True, n_envs=1, seed=5, discount=0.97))
        return q.get_params(), sv.agent.pi_minus.get_params(), sv.buffer["s"], sv.history
    pa, ta, sa, ha = run(True); pb, tb, sb, hb = run(False)
    assert np.array_equal(sa, sb) and np.array_equal(pa, pb) and np.array_equal(ta, tb), (np.abs(pa - pb).max(), np.abs(ta - tb).max())
    assert abs(ha[-1]["critic_loss"] - hb[-1]["critic_loss"]) < 1e-5 * max(1.0, abs(hb[-1]["critic_loss"]))


@pytest.mark.parametrize("algo", ["ddpg", "td3"])
def test_fused_dpg_epochs_equal_the_separate_calls(gpu_ctx, algo):
    """crux_dpg_epochs (rand! -> ddpg_target | td3_target -> critic(s) -> [actor -> polyak], chained, phases shared between the chains) against the call-by-call
    value_training; TD3 with its delayed policy update (a_opt.update_every = 2)."""
    def run(fused):
        S = crux.ContinuousSpace(3); twin = algo == "td3"
        acts = ["relu", "relu", "identity"]
        q = lambda sd: crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=sd)      # noqa: E731
        # TD3 with a bounded (tanh) action head, DDPG with a linear one: the two shapes of the actor's backward chain in the phase plan
        pi = crux.ActorCritic(crux.ContinuousNetwork(parity.chain([3, 256, 256, 1], ["relu", "relu", "tanh" if twin else "identity"]), seed=2), crux.DoubleNetwork(q(3), q(4)) if twin else q(3))
        ctor = crux.TD3 if twin else crux.DDPG
        a_opt = {"batch_size": 128, "update_every": 2} if twin else {"batch_size": 128}
        sv = ctor(pi, S, N=200, dN=10, buffer_size=1000, buffer_init=140, max_steps=50, c_opt={"batch_size": 128, "epochs": 10}, a_opt=a_opt, noise_seed=5,
                  pi_explore=crux.GaussianNoiseExplorationPolicy(0.3, a_min=-1.0, a_max=1.0))
        sv.fused_epochs = fused
        crux.solve(sv, crux.PendulumMDP(n_envs=1, seed=8))
        nets = [pi.A, sv.agent.pi_minus.A] + ([pi.C.N1, pi.C.N2, sv.agent.pi_minus.C.N1, sv.agent.pi_minus.C.N2] if twin else [pi.C, sv.agent.pi_minus.C])
        return [n.get_params() for n in nets], sv.history
    a, ha = run(True); b, hb = run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y), np.abs(x - y).max()
    for k in ("critic_loss", "actor_loss"):
        assert abs(ha[-1][k] - hb[-1][k]) < 1e-5 * max(1.0, abs(hb[-1][k])), k


def test_small_dqn_solve_in_one_launch_equals_the_call_by_call_loop(gpu_ctx, monkeypatch):
    """The README example's shape (DQN on SimpleGridWorld, 2-8-4, dN = 4, B = 128, buffer 1000): the shape-generic form of crux_dqn_small_solve runs whole solve
    iterations in one workgroup with the bodies of the separate calls -- replay buffer, staging batch, both networks, Adam state and the per-iteration infos must
    be the same bits as the call-by-call loop. (CRUX_SMALL_SOLVE_GENERIC: this shape would otherwise take the wave-resident kernel, tested below.)"""
    monkeypatch.setenv("CRUX_SMALL_SOLVE_GENERIC", "1")
    def run(fast):
        q = crux.DiscreteNetwork(parity.chain([2, 8, 4], ["relu", "identity"]), [1, 2, 3, 4], seed=1)
        sv = crux.DQN(q, crux.ContinuousSpace(2), N=1600, dN=4, max_steps=100, c_opt={"batch_size": 128})
        sv.fused_epochs = fast
        crux.solve(sv, crux.SimpleGridWorld(n_envs=1, seed=3))
        m, v, bp = q.adam_state()
        return q.get_params(), sv.agent.pi_minus.get_params(), m, v, bp, {k: sv.buffer[k] for k in ("s", "a", "sp", "r", "done")}, {k: sv.batch[k] for k in ("s", "a", "r")}, sv.history, sv.i, sv.sampler.state()
    a, b = run(True), run(False)
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    for d1, d2 in ((a[5], b[5]), (a[6], b[6])):
        for k in d1:
            assert np.array_equal(d1[k], d2[k]), k
    assert len(a[7]) == len(b[7]) == (1600 - 200) // 4 and a[8] == b[8]
    for h1, h2 in zip(a[7], b[7]):
        assert h1 == h2
    for x, y in zip(a[9], b[9]):
        assert np.array_equal(x, y)


def test_wave_resident_tiny_dqn_solve_follows_the_call_by_call_loop(gpu_ctx):
    """k_dqn_tiny_solve (one wave owns the README problem: parameters in lane registers, replay ring mirrored in LDS) against the call-by-call loop: a different
    but fixed summation order of the minibatch gradient, so the comparison is to tolerance -- the trajectories (greedy actions of nearly equal Q-networks) and the
    replay contents must still be identical, the networks within 1e-5 over 350 iterations = 1 400 gradient steps."""
    def run(fast):
        q = crux.DiscreteNetwork(parity.chain([2, 8, 4], ["relu", "identity"]), [1, 2, 3, 4], seed=1)
        sv = crux.DQN(q, crux.ContinuousSpace(2), N=1600, dN=4, max_steps=100, c_opt={"batch_size": 128})
        sv.fused_epochs = fast
        crux.solve(sv, crux.SimpleGridWorld(n_envs=1, seed=3))
        m, v, bp = q.adam_state()
        return q.get_params(), sv.agent.pi_minus.get_params(), m, v, bp, {k: sv.buffer[k] for k in ("s", "a", "sp", "r", "done")}, {k: sv.batch[k] for k in ("s", "a", "r")}, sv.history, sv.i, sv.sampler.state()
    a, b = run(True), run(False)
    for k in a[5]:
        assert np.array_equal(a[5][k], b[5][k]), k                      # same transitions in the same ring slots
    for k in a[6]:
        assert np.array_equal(a[6][k], b[6][k]), k                      # the staging batch holds the last minibatch drawn
    d = [float(np.abs(x - y).max()) for x, y in zip(a[:4], b[:4])]
    print("tiny solve vs call-by-call after 1400 gradient steps: max |dtheta| %.3g, target %.3g, m %.3g, v %.3g" % tuple(d))
    assert d[0] < 1e-5 and d[1] < 1e-5 and np.array_equal(a[4], b[4])
    assert len(a[7]) == len(b[7]) and a[8] == b[8]
    for h1, h2 in zip(a[7], b[7]):
        for k in h1:
            assert abs(h1[k] - h2[k]) <= 1e-4 * max(1.0, abs(h2[k])), k
    for x, y in zip(a[9], b[9]):
        assert np.array_equal(x, y)


# ---------------------------------------------------------------------------------------------------- the fast learner's shape list
@pytest.mark.parametrize("bs,T", [(128, 64), (48, 32)])
def test_8_64_64_4_runs_on_the_mfma_learner_and_matches_the_oracle(gpu_ctx, capfd, bs, T):
    """8->64->64->4 used to fall to the generic learner without a word; it is an instantiation of the two-CU kernel now (full minibatches) and of its
    any-mode form (small minibatches, the single steps of the window tests)."""
    res = parity.ppo_iteration_parity(n_envs=8, T=T, batch_size=bs, epochs=2, seed=13, family="synth_8_4", pair=(bs == 128))
    assert res["ok"], res
    assert "outside the MFMA learner family" not in capfd.readouterr().err


def test_narrow_hidden_layers_take_the_dense_engine_and_the_generic_learner_agrees(gpu_ctx, capfd, monkeypatch):
    """a 32-wide network: the dense-engine learner by default, the generic single-workgroup learner under CRUX_FORCE_GENERIC -- both match the oracle"""
    res = parity.ppo_iteration_parity(n_envs=8, T=32, batch_size=64, epochs=1, seed=14, family="synth_8_4_h32")
    assert res["ok"], res
    monkeypatch.setenv("CRUX_FORCE_GENERIC", "1")
    res = parity.ppo_iteration_parity(n_envs=8, T=32, batch_size=64, epochs=1, seed=14, family="synth_8_4_h32")
    assert res["ok"], res
    assert "outside the MFMA learner family" not in capfd.readouterr().err          # forced runs are not announced


@pytest.mark.parametrize("family,bs,kl", [("synth_8_4_h128", 64, -1.0), ("synth_c5_h256", 128, -1.0), ("synth_8_4_h128", 64, 0.0005)])
def test_wide_policies_train_on_the_dense_engine_and_match_the_oracle(gpu_ctx, capfd, family, bs, kl):
    """128- and 256-wide PPO learners: the generic learner took 0.5 ms per minibatch at 64 wide and refused 256 (LDS); the dense-engine learner
    (gather -> tile-GEMM forward -> loss head -> tile-GEMM pullback -> gated Adam per minibatch) takes them, incl. KL early stopping inside an epoch."""
    res = parity.ppo_iteration_parity(n_envs=8, T=64, batch_size=bs, epochs=2, seed=23, family=family, target_kl=kl, pair=(kl < 0))
    assert res["ok"], res
    assert "outside the MFMA learner family" not in capfd.readouterr().err


def test_phase_records_in_global_memory_give_the_same_epochs(gpu_ctx, monkeypatch):
    """The executor launches a phase with its op records inside the kernel arguments (k_phase_k) when they fit, through a device copy of the records (k_phase)
    otherwise; CRUX_EXEC_NO_KERNARG forces the second form for every phase -- sequential one-block groups included -- and the SAC epochs must not change."""
    def run():
        S = crux.ContinuousSpace(3); acts = ["relu", "relu", "identity"]
        pi = crux.ActorCritic(crux.GaussianPolicy(parity.chain([3, 256, 256, 1], acts), np.zeros(1, np.float32), seed=2),
                              crux.DoubleNetwork(crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=3), crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=4)))
        sv = crux.SAC(pi, S, N=420, dN=6, buffer_size=1000, buffer_init=300, max_steps=50, c_opt={"batch_size": 256}, a_opt={"batch_size": 256}, SAC_alpha_opt={"batch_size": 256})
        crux.solve(sv, crux.PendulumMDP(n_envs=1, seed=8))
        return [n.get_params() for n in (pi.A, pi.C.N1, pi.C.N2, sv.agent.pi_minus.C.N1, sv.P["SAC_log_alpha"])]
    a = run()
    monkeypatch.setenv("CRUX_EXEC_NO_KERNARG", "1")
    b = run()
    for x, y in zip(a, b):
        assert np.array_equal(x, y), np.abs(x - y).max()


@pytest.mark.parametrize("dims", [[8, 256, 256, 4], [8, 256, 4], [128, 4]], ids=["L3", "L2", "L1"])
def test_chained_dqn_epochs_equal_single_epoch_calls(gpu_ctx, dims):
    """crux_dqn_epochs (the c_opt.epochs epochs of one value_training recorded into one list, no host round trip between them) == the same epochs as
    separate crux_dqn_epoch calls: sampled rows, priorities, networks and infos, bit for bit -- with prioritized replay on a FULL ring (incremental tree
    refresh inside the chain) and on a ring that is still filling (the chain is cut where the tree needs a plain rebuild). Networks of two and of one Dense layer
    have a backward chain shorter than the replay-tree refresh: the overlap of neighbouring epochs shrinks there (exec.hip `ov`; ADVICE r2: with the full overlap the
    search of epoch e + 1 raced the root-path refresh of epoch e)."""
    no = dims[0]
    def run(chained, fill):
        rng = np.random.default_rng(3); N, B = 20_000, 128
        S, A = crux.ContinuousSpace(no), crux.DiscreteSpace(4)
        buf = crux.ExperienceBuffer(S, A, N, prioritized=True); D = crux.buffer_like(buf, capacity=B)
        n0 = N if fill else N // 2
        a = np.zeros((4, n0), bool); a[rng.integers(0, 4, n0), np.arange(n0)] = True
        buf.push_({"s": rng.normal(0, 1, (no, n0)).astype(np.float32), "a": a, "sp": rng.normal(0, 1, (no, n0)).astype(np.float32), "r": rng.normal(0, 1, (1, n0)).astype(np.float32),
                   "done": rng.random((1, n0)) < 0.02, "episode_end": np.zeros((1, n0), bool)})
        buf.update_priorities_(np.arange(1, n0 + 1), (np.abs(rng.normal(0, 1, n0)) + 1e-3).astype(np.float32))
        q = crux.DiscreteNetwork(parity.chain(dims, ["relu"] * (len(dims) - 2) + ["identity"]), [1, 2, 3, 4], seed=5)
        qm = crux.clone_policy(q); q.attach_optimizer(crux.Adam(np.float32(1e-3)))
        ctx = q.ctx; infos = np.zeros((6, L.INFO_N), np.float32); rows = []
        if chained:
            ctx.check(ctx.lib.crux_dqn_epochs(q.h, qm.h, buf.h, D.h, 0.99, 1, 0.6, 40, 6, O.vpz(infos)))
        else:
            for e in range(6):
                ctx.check(ctx.lib.crux_dqn_epoch(q.h, qm.h, buf.h, D.h, 0.99, 1, 0.6, 40 + e, O.vpz(infos[e])))
        return q.get_params(), buf.priority_params()["priorities"], D.indices.copy(), D["s"], infos
    for fill in (True, False):
        a, b = run(True, fill), run(False, fill)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), fill


def test_chained_sac_epochs_equal_single_epoch_calls(gpu_ctx):
    """crux_sac_epochs (chains of up to 8 epochs per recorded list, update_every handled per epoch) == crux_sac_epoch called epoch by epoch, bit for bit;
    11 epochs = one full chain + a partial one, critic every epoch, actor + target update every second epoch."""
    def run(chained):
        rng = np.random.default_rng(4); B, n = 256, 5000
        S, A = crux.ContinuousSpace(3), crux.ContinuousSpace(1)
        buf = crux.ExperienceBuffer(S, A, n); D = crux.buffer_like(buf, capacity=B)
        buf.push_({"s": rng.normal(0, 1, (3, n)).astype(np.float32), "a": rng.uniform(-2, 2, (1, n)).astype(np.float32), "sp": rng.normal(0, 1, (3, n)).astype(np.float32),
                   "r": rng.normal(-1, 1, (1, n)).astype(np.float32), "done": rng.random((1, n)) < 0.02, "episode_end": np.zeros((1, n), bool)})
        acts = ["relu", "relu", "identity"]
        nets = [crux.GaussianPolicy(parity.chain([3, 256, 256, 1], acts), np.zeros(1, np.float32), seed=2)] + \
               [crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=s) for s in (3, 4, 3, 4)]
        la = crux.ParamVector(np.zeros(1, np.float32))
        for net in nets[:3] + [la]:
            net.attach_optimizer(crux.Adam(np.float32(3e-4)))
        a, q1, q2, t1, t2 = nets; ctx = a.ctx; E = 11
        it, ic, ia = (np.zeros((E, L.INFO_N), np.float32) for _ in range(3))
        if chained:
            ctx.check(ctx.lib.crux_sac_epochs(a.h, q1.h, q2.h, None, t1.h, t2.h, la.h, buf.h, D.h, 0.99, -1.0, 0.005, 0, 0, E, 1, 2, 70, 9, 210, O.vpz(it), O.vpz(ic), O.vpz(ia)))
        else:
            for e in range(E):
                ctx.check(ctx.lib.crux_sac_epoch(a.h, q1.h, q2.h, None, t1.h, t2.h, la.h, buf.h, D.h, 0.99, -1.0, 0.005, 0, 1, 1 if e % 2 == 0 else 0, 70 + e, 9, 210 + 3 * e,
                                                 O.vpz(it[e]), O.vpz(ic[e]), O.vpz(ia[e])))
        return [n_.get_params() for n_ in nets + [la]] + [it, ic, ia[::2]]
    for x, y in zip(run(True), run(False)):
        assert np.array_equal(x, y)
