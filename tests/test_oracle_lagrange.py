"""Pins the oracle's restatement of lagrange_ppo_loss (src/model_free/rl/ppo.jl:70-131) on the CPU: the penalty controller against an independent
restatement with Julia's promotion rules spelled out in numpy scalars, and the gradient against central finite differences of the loss value the
oracle reports (the oracle's gradient is hand-derived; the reference gets it from Zygote)."""
import ctypes as C

import numpy as np

import oracle as O
from crux_jl_amd import _lib as L

f32, f64 = np.float32, np.float64
EXTRAS = ["return", "advantage", "logprob", "cost_advantage", "cost", "cost_return"]


def _buffer(rng, od, ad, N, disc):
    ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, N, EXTRAS)
    a = np.zeros((ad, N), bool) if disc else rng.uniform(-1, 1, (ad, N)).astype(f32)
    if disc:
        a[rng.integers(0, ad, N), np.arange(N)] = True
    ob.push({"s": rng.normal(0, 1, (od, N)).astype(f32), "a": a, "sp": rng.normal(0, 1, (od, N)).astype(f32), "r": rng.normal(0, 1, (1, N)).astype(f32),
             "done": np.zeros((1, N), bool), "episode_end": rng.random((1, N)) < 0.2, "return": rng.normal(0, 1, (1, N)).astype(f32),
             "advantage": rng.normal(0, 1, (1, N)).astype(f32), "logprob": rng.normal(-1.0, 0.3, (1, N)).astype(f32),
             "cost_advantage": rng.normal(0, 1, (1, N)).astype(f32), "cost": rng.uniform(0, 1, (1, N)).astype(f32), "cost_return": rng.uniform(0, 3, (1, N)).astype(f32)})
    return ob


def _lag(**kw):
    g = L.Lagrange(); g.target_cost, g.penalty_max, g.Ki_max, g.Ki, g.Kp, g.Kd, g.ema_alpha = 0.025, np.inf, 10.0, 1e-3, 1.0, 0.0, 0.95     # LagrangePPO defaults (ppo.jl:167-176)
    for k, v in kw.items():
        setattr(g, k, v)
    return g


def julia_pid(state, hp, cost, ee):
    """ppo.jl:86-108 with the reference's types: the state arrays are Vector{Float32}, Ki 1f-3, Kp / Kd Int or Float32, ema_alpha Float64."""
    I, Jc_prev, sD, sJ = state
    Jc = f32(f32(np.sum(cost.astype(f64))) / f32(int(np.sum(ee))))                   # Float32 / Int
    D = f32(Jc - hp["target_cost"])
    x = f32(I + f32(hp["Ki"] * D)); I = hp["Ki_max"] if x > hp["Ki_max"] else (f32(0) if x < 0 else x)
    sD = f32(f64(hp["ema"]) * f64(sD) + (1.0 - f64(hp["ema"])) * f64(D))             # Float64 arithmetic, stored into a Float32 array
    sJ = f32(f64(hp["ema"]) * f64(sJ) + (1.0 - f64(hp["ema"])) * f64(Jc))
    d = f32(sJ - Jc_prev); d = d if np.isnan(d) else max(f32(0), d)
    Jc_prev = sJ
    x = f32(f32(f32(hp["Kp"] * sD) + I) + f32(hp["Kd"] * d)); pen = hp["penalty_max"] if x > hp["penalty_max"] else (f32(0) if x < 0 else x)
    return (I, Jc_prev, sD, sJ), pen, Jc, d


def test_penalty_controller_follows_the_reference_arithmetic():
    rng = np.random.default_rng(0); od, ad, N, bs = 3, 2, 256, 32
    ob = _buffer(rng, od, ad, N, True)
    net = O.OMlp([od, 8, ad], ["tanh", "identity"]).init_glorot(3, 0).adam_init(1e-3)
    lag = _lag(Kd=0.5, target_cost=0.3)
    hp = {"target_cost": f32(0.3), "Ki": f32(1e-3), "Ki_max": f32(10.0), "Kp": f32(1.0), "Kd": f32(0.5), "ema": 0.95, "penalty_max": f32(np.inf)}
    cfg = L.TrainCfg(); cfg.loss, cfg.head, cfg.batch_size, cfg.epochs = L.LOSS["lagrange_ppo"], L.HEAD["categorical"], bs, 1
    cfg.eps_clip, cfg.lambda_p, cfg.lambda_e, cfg.target_kl = 0.2, 1.0, 0.1, -1.0
    perm = np.arange(N, dtype=np.int64)[None, :]; ep = np.zeros((1, L.INFO_N), f32); info = np.zeros(L.INFO_N, f32)
    cost, ee = ob["cost"][0], ob["episode_end"][0]
    O.chk(O.lib().orc_batch_train_lagrange(net.h, ob.h, C.byref(cfg), C.byref(lag), O.vpz(perm), O.vpz(info), O.vpz(ep)))
    st = (f32(0), f32(0), f32(0), f32(0))
    for k in range(N // bs):
        st, pen, Jc, d = julia_pid(st, hp, cost[k * bs:(k + 1) * bs], ee[k * bs:(k + 1) * bs])
    assert (f32(lag.I), f32(lag.Jc_prev), f32(lag.smooth_delta), f32(lag.smooth_Jc)) == st
    assert f32(lag.penalty) == pen and f32(lag.cur_cost) == Jc and f32(lag.deriv_term) == d
    assert ep[0, L.INFO["penalty"]] == pen and ep[0, L.INFO["cur_cost"]] == Jc and pen > 0


def _loss_at(net, ob, cfg, lag0, ids, theta):
    net.params[:] = theta
    lag = L.Lagrange(); C.memmove(C.byref(lag), C.byref(lag0), C.sizeof(lag0))
    info = np.zeros(L.INFO_N, f32); perm = ids[None, :].copy(); ep = np.zeros((1, L.INFO_N), f32)
    # one gradient evaluation without an update: epochs = 1 over a buffer of exactly one minibatch, learning rate 0
    O.chk(O.lib().orc_batch_train_lagrange(net.h, ob.h, C.byref(cfg), C.byref(lag), O.vpz(perm), O.vpz(info), O.vpz(ep)))
    return float(ep[0, L.INFO["loss"]]), net.grads.copy()


def test_lagrange_gradient_matches_finite_differences_of_the_loss():
    """d/dtheta [(lambda_p p_loss + lambda_e e_loss + penalty mean(max(r Ac, clamp(r) Ac))) / (1 + penalty)] for both heads"""
    for disc in (True, False):
        rng = np.random.default_rng(5 + disc); od, ad, N = 3, 2, 48
        ob = _buffer(rng, od, ad, N, disc)
        net = O.OMlp([od, 6, ad], ["tanh", "identity"], 0 if disc else ad).init_glorot(7, 0, -0.3).adam_init(0.0)     # eta = 0: Adam leaves theta alone
        cfg = L.TrainCfg(); cfg.loss, cfg.head, cfg.batch_size, cfg.epochs = L.LOSS["lagrange_ppo"], L.HEAD["categorical" if disc else "gaussian"], N, 1
        cfg.eps_clip, cfg.lambda_p, cfg.lambda_e, cfg.target_kl = 0.2, 1.0, 0.1, -1.0
        lag0 = _lag(target_cost=0.1, Kp=2.0)               # a penalty of order 1 so that both halves of the loss matter
        ids = np.arange(N, dtype=np.int64)
        th0 = net.params.copy()
        # old logprobs near the new ones, so that ratios sit on both sides of the clip range
        l0, g = _loss_at(net, ob, cfg, lag0, ids, th0)
        assert np.isfinite(l0) and np.abs(g).max() > 0
        rng2 = np.random.default_rng(1); worst = 0.0
        for _ in range(12):
            i = int(rng2.integers(0, th0.size)); h = 1e-3
            tp, tm = th0.copy(), th0.copy(); tp[i] += h; tm[i] -= h
            lp, _ = _loss_at(net, ob, cfg, lag0, ids, tp); lm, _ = _loss_at(net, ob, cfg, lag0, ids, tm)
            fd = (lp - lm) / (2 * h); worst = max(worst, abs(fd - g[i]) / max(1e-3, abs(fd), abs(g[i])))
        assert worst < 0.03, worst                          # Float32 loss values: ~1e-7 / 1e-3 of noise per difference; kinks of min / max / clamp are measure-zero


def test_lagrange_loss_and_gradient_vs_float64_autograd():
    """lagrange_ppo_loss (ppo.jl:70-131) written out independently in torch float64 and differentiated by autograd, for both heads: the surrogate with the
    cost-advantage term, max instead of min for the cost (a cost is to be decreased), the penalty held constant (ignore_derivatives, :80), the whole sum
    divided by 1 + penalty (:131). The penalty itself comes from julia_pid above."""
    import pytest
    torch = pytest.importorskip("torch")
    for disc in (True, False):
        rng = np.random.default_rng(15 + disc); od, ad, N = 3, 2, 48
        ob = _buffer(rng, od, ad, N, disc)
        dims, acts = [od, 6, ad], ["tanh", "identity"]
        net = O.OMlp(dims, acts, 0 if disc else ad).init_glorot(7, 0, -0.3).adam_init(0.0)
        cfg = L.TrainCfg(); cfg.loss, cfg.head, cfg.batch_size, cfg.epochs = L.LOSS["lagrange_ppo"], L.HEAD["categorical" if disc else "gaussian"], N, 1
        cfg.eps_clip, cfg.lambda_p, cfg.lambda_e, cfg.target_kl = 0.2, 1.0, 0.1, -1.0
        lag0 = _lag(target_cost=0.1, Kp=2.0)
        loss, g = _loss_at(net, ob, cfg, lag0, np.arange(N, dtype=np.int64), net.params.copy())
        hp = {"target_cost": f32(0.1), "Ki": f32(1e-3), "Ki_max": f32(10.0), "Kp": f32(2.0), "Kd": f32(0.0), "ema": 0.95, "penalty_max": f32(np.inf)}
        _, pen, _, _ = julia_pid((f32(0), f32(0), f32(0), f32(0)), hp, ob["cost"][0], ob["episode_end"][0])
        pen = float(pen)
        p = torch.tensor(net.params.copy(), dtype=torch.float64, requires_grad=True)
        off = 0; h = torch.tensor(ob["s"], dtype=torch.float64)
        for l in range(len(acts)):
            i, o = dims[l], dims[l + 1]; W = p[off:off + i * o].reshape(i, o).T; off += i * o; b = p[off:off + o]; off += o
            h = W @ h + b[:, None]; h = torch.tanh(h) if acts[l] == "tanh" else h
        z = h; a = torch.tensor(ob["a"].astype(np.float64)); A = torch.tensor(ob["advantage"][0].astype(np.float64)); Ac = torch.tensor(ob["cost_advantage"][0].astype(np.float64))
        old = torch.tensor(ob["logprob"][0].astype(np.float64))
        if disc:
            pr = torch.softmax(z, 0); newlp = torch.log((pr * a).sum(0)); ent = (-(pr * torch.log(pr + float(np.finfo(np.float32).eps))).sum(0)).mean()
        else:
            ls = p[off:off + ad]; newlp = (-((a - z) ** 2) / (2 * torch.exp(ls)[:, None] ** 2) - 0.9189385332046727 - ls[:, None]).sum(0); ent = 1.4189385332046727 + ls.sum()
        r = torch.exp(newlp - old); rc = torch.clamp(r, 0.8, 1.2)
        p_loss = -torch.minimum(r * A, rc * A).mean(); c_loss = torch.maximum(r * Ac, rc * Ac).mean()
        total = (1.0 * p_loss + 0.1 * (-ent) + pen * c_loss) / (1.0 + pen); total.backward()
        assert abs(loss - total.item()) < 2e-5 * max(1.0, abs(total.item())), (disc, loss, total.item())
        assert np.abs(g - p.grad.numpy()).max() < 5e-6 * max(1.0, np.abs(p.grad.numpy()).max()), (disc, np.abs(g - p.grad.numpy()).max())
