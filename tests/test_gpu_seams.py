"""The function-valued seams of TrainingParams (src/training.jl:1-25): a user-written loss, the regularizer, an arbitrary early_stopping closure.
The reference lets Zygote differentiate any `loss(pi, P, D)`; the library's fast paths cover a closed list of losses, and everything else goes
through crux_mlp_forward_cached / crux_mlp_backward (the explicit pullback) + crux_adam_apply -- api.CustomLoss is the worked example of that
composition, checked here against the oracle."""
import ctypes as C

import numpy as np
import pytest

import parity
from parity import L, O, crux

pytestmark = pytest.mark.gpu


def _filled(ctx, seed=31, E=4, T=96):
    extras = ["return", "logprob", "advantage"]
    _, oa = parity.make_pair(parity.ACTOR_DIMS, parity.ACTS, 50, 0, "discrete")
    _, oc = parity.make_pair(parity.CRITIC_DIMS, parity.ACTS, 50, 1)
    ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, E * T, extras)
    O.OEnv("cartpole", E, 60, 0.99, seed).rollout(oa, parity.rollout_cfg(), ob, T)
    O.chk(O.lib().orc_fill_gae(ob.h, oc.h, 0.95, 0.99)); O.chk(O.lib().orc_fill_returns(ob.h, 0.99)); O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"]))
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), E * T, extras, ctx=ctx)
    gb.push_({k: ob[k] for k in ob.keys()})
    return gb, ob


def user_value_mse(y, D, P):
    """(pi, P, D) -> Flux.mse(value(pi, D[:s]), D[:return]) (ppo.jl:60) written by a user: the loss and its derivative w.r.t. the network output."""
    d = y - D["return"]
    return float(np.mean(d.astype(np.float64) ** 2)), (2.0 * d / d.size).astype(np.float32)


def test_user_written_mse_loss_trains_like_the_builtin_critic_loss(gpu_ctx):
    """batch_train! with loss = CustomLoss(user_value_mse) == the oracle's batch_train! with the critic loss, same shuffles."""
    ctx = gpu_ctx; bs, epochs = 64, 3
    gb, ob = _filled(ctx); N = len(gb)
    g, o = parity.make_pair(parity.CRITIC_DIMS, parity.ACTS, 8, 1)
    o.adam_init(float(np.float32(3e-4)))
    rng = np.random.default_rng(2); perms = np.stack([rng.permutation(N) for _ in range(epochs)])
    p = crux.TrainingParams(loss=crux.CustomLoss(user_value_mse), batch_size=bs, epochs=epochs, name="critic_")
    info = crux.batch_train_(g, p, {}, gb, perms=perms + 1)
    cfg = parity.train_cfg("value_mse", "deterministic", bs, epochs, -1.0, 0); oi = np.zeros(L.INFO_N, np.float32)
    O.chk(O.lib().orc_batch_train(o.h, ob.h, C.byref(cfg), O.vpz(np.ascontiguousarray(perms, np.int64)), O.vpz(oi), None))
    d = float(np.abs(g.get_params() - o.params).max())
    print("custom mse vs oracle after %d steps: max |dtheta| = %.3g" % (epochs * (N // bs), d))
    assert d < parity.param_tol(epochs * (N // bs))
    assert info["critic_batches_trained"] == epochs * (N // bs)
    assert abs(info["critic_loss"] - float(oi[L.INFO["loss"]])) < 2e-5 * max(1.0, abs(float(oi[L.INFO["loss"]])))
    assert abs(info["critic_grad_norm"] - float(oi[L.INFO["grad_norm"]])) < 2e-5 * max(1.0, abs(float(oi[L.INFO["grad_norm"]])))


def test_regularizer_is_added_to_the_loss_and_its_gradient(gpu_ctx):
    """train!(pi, loss + regularizer) (training.jl:13) with the library's PPO loss and an L2 penalty lam * sum(theta^2): the oracle twin is
    orc_loss_grad -> g += 2 lam theta -> orc_adam_apply."""
    ctx = gpu_ctx; bs, lam = 64, np.float32(1e-2)
    gb, ob = _filled(ctx, seed=33); N = len(gb)
    g, o = parity.make_pair(parity.ACTOR_DIMS, parity.ACTS, 12, 0, "discrete")
    o.adam_init(float(np.float32(3e-4)))
    reg = lambda th: (float(lam * np.sum(th.astype(np.float64) ** 2)), (2 * lam * th).astype(np.float32))    # noqa: E731
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    p = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=1, name="actor_", regularizer=reg)
    cfg = parity.train_cfg("ppo", "categorical", bs, 1, -1.0, 0)
    for st in range(0, 4 * bs, bs):
        ids = np.arange(st, st + bs, dtype=np.int64)
        info = crux.train_(g, p, P, gb, ids + 1)
        oi = np.zeros(L.INFO_N, np.float32)
        O.chk(O.lib().orc_loss_grad(o.h, ob.h, C.byref(cfg), O.vpz(ids), bs, O.vpz(oi)))
        th = o.params.copy(); o.grads[:] = o.grads + 2 * lam * th
        gn = float(np.sqrt(np.sum(o.grads.astype(np.float64) ** 2)))
        O.chk(O.lib().orc_adam_apply(o.h, 1.0))
        want = float(oi[L.INFO["loss"]]) + float(lam * np.sum(th.astype(np.float64) ** 2))
        assert abs(info["actor_loss"] - want) < 2e-5 * max(1.0, abs(want))
        assert abs(info["actor_grad_norm"] - gn) < 2e-5 * max(1.0, gn)
        assert abs(info["kl"] - float(oi[L.INFO["kl"]])) < 1e-5
    d = float(np.abs(g.get_params() - o.params).max())
    print("ppo + L2 regularizer vs oracle after 4 steps: max |dtheta| = %.3g" % d)
    assert d < parity.param_tol(4)
    # the penalty did something: the same steps without it end elsewhere
    g2, _ = parity.make_pair(parity.ACTOR_DIMS, parity.ACTS, 12, 0, "discrete")
    p2 = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=1, name="actor_")
    for st in range(0, 4 * bs, bs):
        crux.train_(g2, p2, P, gb, np.arange(st, st + bs, dtype=np.int64) + 1)
    assert np.abs(g2.get_params() - g.get_params()).max() > 1e-5


def test_early_stopping_closure_sees_the_epoch_infos(gpu_ctx):
    """early_stopping = (infos) -> ... (training.jl:8,46,49): an arbitrary predicate over the aggregated infos ends batch_train!."""
    ctx = gpu_ctx
    gb, _ = _filled(ctx, seed=35); N = len(gb)
    g, _ = parity.make_pair(parity.CRITIC_DIMS, parity.ACTS, 8, 1)
    seen = []
    def stop(infos):
        seen.append(len(infos)); return len(infos) >= 2 and infos[-1]["critic_loss"] < infos[0]["critic_loss"]
    p = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=10, name="critic_", early_stopping=stop)
    info = crux.batch_train_(g, p, {}, gb)
    assert info["_epochs_run"] == 2 and info["critic_batches_trained"] == 2 * (N // 128) - (N // 128 - 1)   # the predicate also fires inside epoch 2 (:46)
    assert max(seen) == 2


def test_user_loss_nan_raises_like_the_reference(gpu_ctx):
    ctx = gpu_ctx
    gb, _ = _filled(ctx, seed=37)
    g, _ = parity.make_pair(parity.CRITIC_DIMS, parity.ACTS, 8, 1)
    before = g.get_params()
    bad = crux.CustomLoss(lambda y, D, P: (float("nan"), np.full_like(y, np.nan)))
    with pytest.raises(crux.CruxError) as e:
        crux.train_(g, crux.TrainingParams(loss=bad, batch_size=32, epochs=1), {}, gb, np.arange(1, 33))
    assert e.value.code == L.ENAN and np.array_equal(before, g.get_params())                      # training.jl:20: no update


def test_param_optimizers_train_a_bare_vector_before_the_actor(gpu_ctx):
    """`param_optimizers` (on_policy.jl:59-61): batch_train!(θs, p_opt, P, D, π_loss=agent.π) for entries that are bare parameter vectors, e.g. a
    multiplier trained by dual ascent on a constraint estimated from the minibatch; the oracle twin applies Flux's Adam to the same gradients."""
    ctx = gpu_ctx; bs = 96
    gb, ob = _filled(ctx, seed=39); N = len(gb)
    lam = crux.ParamVector(np.array([0.5, -0.25], np.float32), ctx=ctx)
    olam = O.OMlp([0], [], 2); olam.params[:] = np.array([0.5, -0.25], np.float32); olam.adam_init(float(np.float32(1e-2)))
    seen = []
    def dual(theta, D, P, pi):          # maximise theta[0] * (mean |advantage| - 0.5) - theta[1]^2: gradient of the negated objective
        c = float(np.mean(np.abs(D["advantage"]))) - P["budget"]
        seen.append(pi is not None)
        return -theta[0] * c + theta[1] ** 2, np.array([-c, 2 * theta[1]], np.float32), {"constraint": c}
    g, _ = parity.make_pair(parity.ACTOR_DIMS, parity.ACTS, 12, 0, "discrete"); gc, _ = parity.make_pair(parity.CRITIC_DIMS, parity.ACTS, 12, 1)
    class _S:
        pass
    s = _S(); s.agent = crux.PolicyParams(crux.ActorCritic(g, gc)); s.P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1, "budget": 0.5}
    s.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=1, name="actor_", shuffle_seed=1)
    s.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=1, name="critic_", shuffle_seed=2)
    p_lam = crux.TrainingParams(loss=crux.ParamLoss(dual), optimizer=crux.Adam(np.float32(1e-2)), batch_size=bs, epochs=2, name="lambda_", shuffle_seed=3)
    s.param_optimizers = [(lam, p_lam)]
    before = g.get_params()
    info = crux.policy_gradient_training(s, gb)
    assert all(seen) and len(seen) == 2 * (N // bs) and "lambda_loss" in info and "constraint" in info and "actor_loss" in info
    assert not np.array_equal(before, g.get_params())                                    # the actor trained after the parameter step
    # oracle twin: the same shuffles (library permutation stream), the same minibatch means, Flux's Adam
    perm = np.empty(N, np.int64); adv = ob["advantage"][0].copy()
    for ep in range(2):
        O.lib().orc_perm(3, ep, N, O.vpz(perm)); adv = adv[perm]
        for st in range(0, N, bs):
            c = float(np.mean(np.abs(adv[st:st + bs]))) - 0.5
            th = olam.params.copy(); olam.grads[:] = np.array([-c, 2 * th[1]], np.float32)
            O.chk(O.lib().orc_adam_apply(olam.h, 1.0))
    d = float(np.abs(lam.get_params() - olam.params).max())
    print("param_optimizers: multiplier after %d steps: %s vs oracle %s (max diff %.3g)" % (2 * (N // bs), lam.get_params(), olam.params, d))
    assert d < 1e-6
