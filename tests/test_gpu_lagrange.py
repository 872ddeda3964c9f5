"""LagrangePPO (src/model_free/rl/ppo.jl:70-215): the cost channel of the sampler (sampler.jl:65-66,114), the cost critic, and lagrange_ppo_loss with
its PID penalty controller running once per minibatch inside the persistent learner -- GPU against the oracle on the same seeded inputs."""
import ctypes as C

import numpy as np
import pytest

import parity
from parity import L, O, crux

pytestmark = pytest.mark.gpu
EXTRAS = ["return", "advantage", "logprob", "cost_advantage", "cost", "cost_return"]


def _lag(target_cost=0.05, penalty_max=np.inf, Ki=1e-3, Kp=1.0, Kd=0.5, ema=0.95, Ki_max=10.0):
    g = L.Lagrange(); g.target_cost, g.penalty_max, g.Ki_max, g.Ki, g.Kp, g.Kd, g.ema_alpha = target_cost, penalty_max, Ki_max, Ki, Kp, Kd, ema
    return g


def _copy_lag(a):
    b = L.Lagrange(); C.memmove(C.byref(b), C.byref(a), C.sizeof(a)); return b


def _pair(kind, E=6, T=64, max_steps=12, seed=21):
    """the same rollout with cost columns on the GPU and in the oracle; returns buffers, networks and the sampler"""
    if kind == "cartpole":
        od, ad, disc, acts = 4, 2, True, parity.ACTS
        mdp = crux.CartPoleMDP(n_envs=E, seed=seed); oe = O.OEnv("cartpole", E, max_steps, 0.99, seed)
        ga, oa = parity.make_pair([4, 64, 64, 2], acts, 50, 0, "discrete")
    elif kind == "synth84":       # the LunarLander-shaped discrete actor 8-64-64-4 (relu): the third lagrange instantiation of the feature-split learner
        od, ad, disc, acts = 8, 4, True, parity.ACTS
        mdp = crux.SynthMDP(od, ad, n_envs=E, seed=seed, discrete=True); oe = O.OEnv("synth_discrete", E, max_steps, 0.99, seed, so=od, sa=ad)
        ga, oa = parity.make_pair([8, 64, 64, 4], acts, 50, 0, "discrete")
    else:       # "synth": 5-32-32-2 (dense-engine learner); "synth17": the C5 shape 17-64-64-6; "synth31": the Pendulum shape 3-64-64-1 (relu) -- lagrange instantiations
        od, ad, disc, acts = (17, 6, False, ["tanh", "tanh", "identity"]) if kind == "synth17" else (3, 1, False, parity.ACTS) if kind == "synth31" else (5, 2, False, ["tanh", "tanh", "identity"])
        mdp = crux.SynthMDP(od, ad, n_envs=E, seed=seed); oe = O.OEnv("synth", E, max_steps, 0.99, seed, so=od, sa=ad)
        ga, oa = parity.make_pair([od, 32, 32, ad] if kind == "synth" else [od, 64, 64, ad], acts, 50, 0, "gaussian", n_extra=ad, extra_init=-0.5)
    hid = 32 if kind == "synth" else 64
    gc, oc = parity.make_pair([od, hid, hid, 1], acts, 50, 1)
    gv, ov = parity.make_pair([od, hid, hid, 1], acts, 50, 2)               # Vc
    S, A = crux.ContinuousSpace(od), (crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad))
    gb = crux.ExperienceBuffer(S, A, E * T, EXTRAS)
    ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, E * T, EXTRAS)
    gs = crux.Sampler(mdp, crux.ActorCritic(ga, gc), max_steps=max_steps, required_columns=EXTRAS, lam=0.95, Vc=gv)
    crux.steps_(gs, gb, Nsteps=E * T, explore=True, i=0, reset=True)
    head = "categorical" if disc else "gaussian"
    oe.rollout(oa, parity.rollout_cfg(head=head), ob, T)
    ol = O.lib()
    O.chk(ol.orc_fill_gae(ob.h, oc.h, 0.95, 0.99)); O.chk(ol.orc_fill_returns(ob.h, 0.99))
    O.chk(ol.orc_fill_gae_keys(ob.h, ov.h, 0.95, 0.99, L.COL["cost"], L.COL["cost_advantage"]))            # sampler.jl:65
    O.chk(ol.orc_fill_returns_keys(ob.h, 0.99, L.COL["cost"], L.COL["cost_return"]))                      # :66
    return (gb, ob), (ga, oa), (gc, oc), (gv, ov), head


@pytest.mark.parametrize("kind", ["cartpole", "synth"])
def test_sampler_writes_cost_and_fills_cost_advantage_and_cost_return(gpu_ctx, kind):
    (gb, ob), *_ = _pair(kind)
    assert np.array_equal(gb["done"], ob["done"]) and np.array_equal(gb["episode_end"], ob["episode_end"])
    c = gb["cost"]
    assert np.abs(c - ob["cost"]).max() < 1e-5 and float(c.max()) > 0.0            # Float32 of Float64 dynamics; the channel is not trivially zero
    if kind == "cartpole":
        assert np.array_equal(c, ob["cost"]) and set(np.unique(c)) <= {0.0, 1.0}
    for k, tol in (("cost_return", 2e-5), ("cost_advantage", 5e-5), ("advantage", 5e-5), ("return", 2e-5)):
        d = np.abs(gb[k] - ob[k]).max(); s = max(1.0, float(np.abs(ob[k]).max()))
        assert d < tol * s, (k, d)


@pytest.mark.parametrize("bs", [64, 128])
@pytest.mark.parametrize("kind", ["cartpole", "synth", "synth17", "synth84", "synth31"])
def test_lagrange_batch_train_matches_oracle(gpu_ctx, kind, bs):
    """batch_train!(actor, a_opt, P, D) with lagrange_ppo_loss: parameters, the controller's state after every executed minibatch's update, and the infos.
    The 64-wide actors run on the register-resident learners' lagrange instantiations: bs = 128 on the feature-split kernel (train_fs2_kernel.h, LAG: all four of its shapes), bs = 64 on the two-CU kernel (train_mfma_kernel.h, LAG)."""
    (gb, ob), (ga, oa), _, _, head = _pair(kind)
    O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"])); crux.whiten_(gb, "advantage")
    N, epochs = len(gb), 3
    rng = np.random.default_rng(4); perms = np.stack([rng.permutation(N) for _ in range(epochs)])
    glag = _lag(); olag = _copy_lag(glag)
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1, "lagrange": glag}
    p = crux.TrainingParams(loss=crux.lagrange_ppo_loss, batch_size=bs, epochs=epochs, name="actor_")
    info = crux.batch_train_(ga, p, P, gb, perms=perms + 1)
    oa.adam_init(float(np.float32(3e-4)))
    cfg = parity.train_cfg("lagrange_ppo", head, bs, epochs, -1.0, 0); oi = np.zeros(L.INFO_N, np.float32); oe_ = np.zeros((epochs, L.INFO_N), np.float32)
    O.chk(O.lib().orc_batch_train_lagrange(oa.h, ob.h, C.byref(cfg), C.byref(olag), O.vpz(np.ascontiguousarray(perms, np.int64)), O.vpz(oi), O.vpz(oe_)))
    steps = epochs * (N // bs)
    d = float(np.abs(ga.get_params() - oa.params).max())
    print(kind, "lagrange: max |dtheta| after %d steps = %.3g; penalty %.6g / %.6g; I %.6g" % (steps, d, glag.penalty, olag.penalty, glag.I))
    assert info["actor_batches_trained"] == steps
    assert d < parity.param_tol(steps)                     # measured 9e-8 (cartpole) / 3e-8 (synth)
    for f in ("I", "Jc_prev", "smooth_delta", "smooth_Jc", "penalty", "cur_cost", "deriv_term"):
        a, b = getattr(glag, f), getattr(olag, f)
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (f, a, b)
    assert glag.penalty > 0.0 and glag.I > 0.0                                   # the constraint is active in this test (costs above target)
    for k in ("loss", "kl", "entropy", "penalty", "cur_cost", "cost_loss", "p_loss", "clip_fraction"):
        a, b = float(info[k if k in info else "actor_" + k]), float(oi[L.INFO[k]])
        assert abs(a - b) < 2e-5 * max(1.0, abs(b)), (k, a, b)


def test_minibatch_without_an_episode_end_ends_in_nan_like_the_reference(gpu_ctx):
    """Jc = sum(cost) / sum(episode_end) over the MINIBATCH (ppo.jl:86): without an episode end it is Inf, the smoothed terms stay Inf, the next
    derivative term is Inf - Inf = NaN, and train! stops with "NaN detected!" (training.jl:20) -- restated, not repaired. The controller's state keeps
    the updates of the minibatches that ran, like the arrays in the reference's P."""
    (gb, ob), (ga, oa), _, _, head = _pair("synth", E=2, T=64, max_steps=1000, seed=5)      # long episodes: most minibatches of 16 rows hold no episode end
    crux.whiten_(gb, "advantage")
    before = ga.get_params()
    glag = _lag(penalty_max=3.0)
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1, "lagrange": glag}
    with pytest.raises(crux.CruxError) as e:
        crux.batch_train_(ga, crux.TrainingParams(loss=crux.lagrange_ppo_loss, batch_size=16, epochs=1, name="actor_"), P, gb, perms=np.arange(1, 129)[None, :])
    assert e.value.code == L.ENAN
    assert np.isinf(glag.smooth_Jc) and glag.I == 10.0                                          # clamp(I + Ki * Inf, 0, Ki_max)
    assert not np.array_equal(before, ga.get_params())                                          # the first Inf minibatch still trained (penalty = penalty_max = 3)


def test_nan_penalty_inside_the_feature_split_learner_keeps_the_step_before(gpu_ctx):
    """The same Inf -> NaN chain inside k_train_fs2<..., LAG> (17-64-64-6 actor, full minibatches of 128): rows 1..128 hold no episode end (Jc = Inf, penalty = penalty_max,
    the step trains), rows 129..256 end an episode (smooth_Jc stays Inf, the derivative term is Inf - Inf = NaN, so is the penalty and with it the gradient): the second step is
    the error of training.jl:20 and must leave exactly the parameters and Adam moments of the first -- the state a run stopped after one minibatch (max_batches = 1) has. The
    controller keeps what both evaluations did to it."""
    def run(**kw):
        (gb, ob), (ga, oa), _, _, head = _pair("synth17", E=2, T=256, max_steps=1000, seed=5)
        crux.whiten_(gb, "advantage")
        ee = gb["episode_end"][0]; assert not ee[:128].any() and ee[128:256].any()
        glag = _lag(penalty_max=3.0)
        P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1, "lagrange": glag}
        err = None
        try:
            crux.batch_train_(ga, crux.TrainingParams(loss=crux.lagrange_ppo_loss, batch_size=128, epochs=1, name="actor_", **kw), P, gb, perms=np.arange(1, 513)[None, :])
        except crux.CruxError as e:
            err = e
        m, v, bp = ga.adam_state()
        return err, ga.get_params(), m, v, glag
    e2, p2, m2, v2, g2 = run()
    assert e2 is not None and e2.code == L.ENAN
    e1, p1, m1, v1, g1 = run(max_batches=1)
    assert e1 is None and g1.penalty == 3.0 and np.isinf(g1.smooth_Jc)
    for x, y in ((p2, p1), (m2, m1), (v2, v1)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.isfinite(p2).all() and np.isnan(g2.penalty) and np.isnan(g2.deriv_term) and np.isinf(g2.smooth_Jc)


def test_lagrange_ppo_solve_two_iterations_match_the_oracle_loop(gpu_ctx):
    """solve(LagrangePPO(...), mdp) (on_policy.jl:80-109): steps! with cost columns, whiten, actor (lagrange loss, KL stop), critic, cost critic."""
    E, T, ms, seed, bs = 4, 64, 10, 9, 64
    od, ad = 5, 3
    acts = ["tanh", "tanh", "identity"]
    ga, oa = parity.make_pair([od, 32, 32, ad], acts, 60, 0, "discrete")
    gc, oc = parity.make_pair([od, 32, 32, 1], acts, 60, 1); gv, ov = parity.make_pair([od, 32, 32, 1], acts, 60, 2)
    mdp = crux.SynthMDP(od, ad, discrete=True, n_envs=E, seed=seed)
    S = crux.ContinuousSpace(od)
    opt = {"batch_size": bs, "epochs": 2}
    solver = crux.LagrangePPO(crux.ActorCritic(ga, gc), gv, S, N=2 * E * T, dN=E * T, max_steps=ms, target_cost=0.5, Kd=0.25,
                              a_opt=dict(opt, shuffle_seed=3), c_opt=dict(opt, shuffle_seed=4), cost_opt=dict(opt, shuffle_seed=5), target_kl=None)
    crux.solve(solver, mdp)
    # ---- the oracle loop
    ob = O.OBuffer(od, ad, L.ACTION_DISCRETE, E * T, EXTRAS)
    oe = O.OEnv("synth_discrete", E, ms, 0.99, seed, so=od, sa=ad)
    for o in (oa, oc, ov):
        o.adam_init(float(np.float32(3e-4)))
    olag = _lag(target_cost=0.5, Kd=0.25); ol = O.lib(); i = 0
    for it in range(2):
        O.chk(ol.orc_buffer_clear(ob.h))
        cfg = parity.rollout_cfg(head="categorical"); cfg.i0 = i
        oe.rollout(oa, cfg, ob, T)
        O.chk(ol.orc_fill_gae(ob.h, oc.h, 0.95, 0.99)); O.chk(ol.orc_fill_returns(ob.h, 0.99))
        O.chk(ol.orc_fill_gae_keys(ob.h, ov.h, 0.95, 0.99, L.COL["cost"], L.COL["cost_advantage"])); O.chk(ol.orc_fill_returns_keys(ob.h, 0.99, L.COL["cost"], L.COL["cost_return"]))
        O.chk(ol.orc_whiten(ob.h, L.COL["advantage"]))
        info = np.zeros(L.INFO_N, np.float32)
        ca = parity.train_cfg("lagrange_ppo", "categorical", bs, 2, -1.0, 3, 2 * it)
        O.chk(ol.orc_batch_train_lagrange(oa.h, ob.h, C.byref(ca), C.byref(olag), None, O.vpz(info), None))
        cc = parity.train_cfg("value_mse", "deterministic", bs, 2, -1.0, 4, 2 * it); O.chk(ol.orc_batch_train(oc.h, ob.h, C.byref(cc), None, O.vpz(info), None))
        cv = parity.train_cfg("value_mse", "deterministic", bs, 2, -1.0, 5, 2 * it); cv.target_col = L.COL["cost_return"]
        O.chk(ol.orc_batch_train(ov.h, ob.h, C.byref(cv), None, O.vpz(info), None))
        i += E * T
    glag = solver.P["lagrange"]
    for g, o, nm in ((ga, oa, "actor"), (gc, oc, "critic"), (gv, ov, "Vc")):
        d = float(np.abs(g.get_params() - o.params).max()); print(nm, "max |dtheta| = %.3g" % d)
        assert d < 2e-6, nm                                   # measured 1.5e-7 / 3e-8 / 3e-8
    assert abs(glag.penalty - olag.penalty) <= 1e-5 * max(1.0, abs(olag.penalty)) and abs(glag.I - olag.I) <= 1e-5 * max(1.0, abs(olag.I))
    h = solver.history[-1]
    assert "penalty" in h and "cost_critic_loss" in h and "critic_loss" in h and h["cur_cost"] > 0.0
