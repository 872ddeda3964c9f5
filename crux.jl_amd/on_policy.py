"""on_policy.py -- OnPolicySolver: policy_gradient_training and its multi-GPU / multi-seed forms, solve, PPO, LagrangePPO, A2C, REINFORCE (src/model_free/on_policy.jl, rl/ppo.jl, rl/a2c.jl, rl/reinforce.jl).

Split out of api.py in round 4; `crux_jl_amd.api` re-exports everything, so `crux.X` and `crux.api.X` resolve as before."""
import ctypes as C
import math
import numpy as np
from . import _lib as L
from .core import (  # noqa: F401
    ActorCritic, Adam, CartPoleMDP, Chain, Context, ContinuousNetwork, ContinuousSpace, CustomLoss, Dense, DiscreteNetwork, DiscreteSpace, DoubleNetwork, EpsGreedyPolicy,
    ExperienceBuffer, GaussianNoiseExplorationPolicy, GaussianPolicy, GymMDP, LinearDecaySchedule, MultitaskDecaySchedule, NetworkPolicy, ParamLoss, ParamVector, PendulumMDP,
    PolicyParams, SAMPLE_SEED, Sampler, SimpleGridWorld, SquashedGaussianPolicy, SynthMDP, TrainingParams, _F32_KEYS, _Loss, _batch_train_seam, _ensure_opt, _fill_block,
    _fill_importance_weights, _info_dict, _leaves, _np_dtype, _rollout_cfg, _train_cfg, _train_seam, _uses_seam, _vp, a2c_loss, actor, batch_train_, buffer_like, capacity,
    clone_policy, copy_buffer, copyto_, cost_value_mse_loss, critic, default_context, dim, discount, discounted_return, episodes, episodes_, extra_columns, failure, fill_gae_,
    fill_returns_, get_episodes, hcat, lagrange_ppo_loss, mdp_data, normalize_, peer_attach_local, polyak_average_, ppo_loss, prioritized_sample_, rand_, reinforce_loss,
    reload_switches, set_default_context, set_sample_stream_, shuffle_device_, split, split_batches, steps_, steps_multi_, train_, trim_, undiscounted_return, uniform_sample_,
    value, value_mse_loss, whiten_, whiten_multi_)


# --------------------------------------------------------------------------------------------------------------
# on-policy solver + PPO (src/model_free/on_policy.jl, src/model_free/rl/ppo.jl:40-65)
# --------------------------------------------------------------------------------------------------------------
class OnPolicySolver:
    """OnPolicySolver(; agent, S, N, dN, max_steps, a_opt, c_opt, P, lambda_gae, required_columns, post_batch_callback)
    (src/model_free/on_policy.jl:31-54)."""

    def __init__(self, agent, S, N=1000, dN=200, max_steps=100, a_opt=None, c_opt=None, P=None, lambda_gae=0.95,
                 required_columns=(), post_batch_callback=None, post_sample_callback=None, i=0, log=None, Vc=None, cost_opt=None, param_optimizers=None, interaction_storage=None):
        self.interaction_storage = interaction_storage      # a list: every steps! block is appended to it (on_policy.jl:44,96)
        self.Vc, self.cost_opt = Vc, cost_opt      # cost constraints: a separate value network and its TrainingParams (on_policy.jl:50-53)
        self.param_optimizers = list(param_optimizers or [])     # [(ParamVector, TrainingParams(loss=ParamLoss(...)))]: trained before the actor (on_policy.jl:59-61)
        self.agent, self.S, self.N, self.dN, self.max_steps = agent, S, int(N), int(dN), int(max_steps)
        self.a_opt, self.c_opt, self.P = a_opt, c_opt, P or {}
        self.lambda_gae, self.required_columns = np.float32(lambda_gae), list(required_columns)
        self.post_batch_callback, self.post_sample_callback, self.i = post_batch_callback, post_sample_callback, int(i)
        self.log = log                         # LoggerParams (crux_jl_amd.logging) or None; sampler for the evaluation fns is set at solve time (on_policy.jl:84)
        self.buffer, self.sampler, self.history = None, None, []


def policy_gradient_training(solver, D, perms_a=None, perms_c=None):
    """policy_gradient_training(S, D) (src/model_free/on_policy.jl:56-78): actor batch_train!, then critic batch_train!.
    One C call: the two persistent learner kernels overlap on two CUs whenever that is exact (see cruxhip.h)."""
    info = {}
    for theta, p_opt in getattr(solver, "param_optimizers", []):                                       # on_policy.jl:59-61: batch_train!(θs, p_opt, P, D, π_loss=agent.π)
        p_opt.pi_loss = solver.agent.pi
        pi_ = batch_train_(theta, p_opt, solver.P, D, info={})
        info.update({k: v for k, v in pi_.items() if not k.startswith("_")})
    A, Cn, pa, pc = actor(solver.agent.pi), critic(solver.agent.pi), solver.a_opt, solver.c_opt
    if pc is None:
        return batch_train_(A, pa, solver.P, D, info=info, perms=perms_a)
    if pa.loss.name == "lagrange_ppo" or getattr(solver, "cost_opt", None) is not None or _uses_seam(pa) or _uses_seam(pc):
        # the sequential form of on_policy.jl:63-76: actor, critic, then the cost critic (the penalty controller rides in the actor's learner kernel)
        batch_train_(A, pa, solver.P, D, info=info, perms=perms_a)
        po = getattr(solver, "cost_opt", None)
        if (po is not None and perms_c is None and not _uses_seam(pc) and not _uses_seam(po) and pc.target_kl is None and po.target_kl is None
                and pc.max_batches == math.inf and po.max_batches == math.inf):
            # the critic and the cost critic (on_policy.jl:66-76) as ONE pair call: two learners whose shuffle chains follow each other, run side by side where that is exact
            # (crux_policy_gradient_training is not tied to an actor: the second learner's order chain starts from the first one's last order)
            Vc = solver.Vc
            _ensure_opt(Cn, pc); _ensure_opt(Vc, po)
            cc, cv = _train_cfg(Cn, pc, solver.P), _train_cfg(Vc, po, solver.P)
            rc_, rv = np.zeros(L.INFO_N, np.float32), np.zeros(L.INFO_N, np.float32)
            ec, ev = np.zeros((pc.epochs, L.INFO_N), np.float32), np.zeros((po.epochs, L.INFO_N), np.float32)
            Cn.ctx.check(Cn.ctx.lib.crux_policy_gradient_training(Cn.h, Vc.h, D.h, C.byref(cc), C.byref(cv), None, None, _vp(rc_), _vp(rv), _vp(ec), _vp(ev)))
            for p, raw in ((pc, rc_), (po, rv)):
                p.shuffle_counter += int(raw[L.INFO["epochs_run"]])
                d = {k: v for k, v in _info_dict(p, raw).items() if k.startswith(p.name)}
                info.update(d); info[p.name + "batches_trained"] = int(raw[L.INFO["batches_trained"]])
            return info
        ci = batch_train_(Cn, pc, solver.P, D, info={}, perms=perms_c)
        info.update({k: v for k, v in ci.items() if k.startswith(pc.name)})
        if po is not None:
            vi = batch_train_(solver.Vc, po, solver.P, D, info={})
            info.update({k: v for k, v in vi.items() if k.startswith(po.name)})
        return info
    _ensure_opt(A, pa); _ensure_opt(Cn, pc)
    ca, cc = _train_cfg(A, pa, solver.P), _train_cfg(Cn, pc, solver.P)
    ra, rc_ = np.zeros(L.INFO_N, np.float32), np.zeros(L.INFO_N, np.float32)
    ea, ec = np.zeros((pa.epochs, L.INFO_N), np.float32), np.zeros((pc.epochs, L.INFO_N), np.float32)
    p1 = None if perms_a is None else np.ascontiguousarray(np.asarray(perms_a, np.int64) - 1)
    p2 = None if perms_c is None else np.ascontiguousarray(np.asarray(perms_c, np.int64) - 1)
    A.ctx.check(A.ctx.lib.crux_policy_gradient_training(A.h, Cn.h, D.h, C.byref(ca), C.byref(cc), _vp(p1), _vp(p2), _vp(ra), _vp(rc_), _vp(ea), _vp(ec)))
    for p, raw in ((pa, ra), (pc, rc_)):
        p.shuffle_counter += int(raw[L.INFO["epochs_run"]])
        d = _info_dict(p, raw)
        if p is pc:
            d = {k: v for k, v in d.items() if k.startswith(p.name)}
        info.update(d); info[p.name + "batches_trained"] = int(raw[L.INFO["batches_trained"]])
    return info


def policy_gradient_training_synced(solver, D, sync_every=1):
    """policy_gradient_training for environment-shard replicas: every `sync_every` epochs the replica group attached to the context
    (Context.comm_init) averages actor/critic parameters and Adam moments with one RCCL all-reduce enqueued behind the learner kernels.
    Without a group it equals policy_gradient_training bit for bit."""
    A, Cn, pa, pc = actor(solver.agent.pi), critic(solver.agent.pi), solver.a_opt, solver.c_opt
    _ensure_opt(A, pa); _ensure_opt(Cn, pc)
    ca, cc = _train_cfg(A, pa, solver.P), _train_cfg(Cn, pc, solver.P)
    ra, rc_ = np.zeros(L.INFO_N, np.float32), np.zeros(L.INFO_N, np.float32)
    A.ctx.check(A.ctx.lib.crux_policy_gradient_training_synced(A.h, Cn.h, D.h, C.byref(ca), C.byref(cc), int(sync_every), _vp(ra), _vp(rc_)))
    info = {}
    for p, raw in ((pa, ra), (pc, rc_)):
        p.shuffle_counter += int(raw[L.INFO["epochs_run"]])
        d = _info_dict(p, raw)
        if p is pc:
            d = {k: v for k, v in d.items() if k.startswith(p.name)}
        info.update(d); info[p.name + "batches_trained"] = int(raw[L.INFO["batches_trained"]])
    return info


def allreduce_mean_(net):
    """average a network's parameters and Adam moments over the replica group (stream-ordered; no-op without a group)."""
    net.ctx.check(net.ctx.lib.crux_allreduce_mean(net.h))


def policy_gradient_training_multi(pis, a_opt, c_opt, P, buffers):
    """policy_gradient_training (src/model_free/on_policy.jl:56-78) for several independent ActorCritic / buffer pairs of equal shape (multi-seed or
    population training) as two batched launches; replica i shuffles with shuffle_seed + i. Returns one info dict per replica."""
    n = len(pis); ctx = buffers[0].ctx
    for pi in pis:
        _ensure_opt(actor(pi), a_opt); _ensure_opt(critic(pi), c_opt)
    ca, cc = _train_cfg(actor(pis[0]), a_opt, P), _train_cfg(critic(pis[0]), c_opt, P)
    ha = (C.c_void_p * n)(*[actor(pi).h for pi in pis]); hc = (C.c_void_p * n)(*[critic(pi).h for pi in pis]); hb = (C.c_void_p * n)(*[b.h for b in buffers])
    ra, rc_ = np.zeros((n, L.INFO_N), np.float32), np.zeros((n, L.INFO_N), np.float32)
    ctx.check(ctx.lib.crux_policy_gradient_training_multi(n, ha, hc, hb, C.byref(ca), C.byref(cc), _vp(ra), _vp(rc_)))
    a_opt.shuffle_counter += int(ra[0, L.INFO["epochs_run"]]); c_opt.shuffle_counter += int(rc_[0, L.INFO["epochs_run"]])
    out = []
    for i in range(n):
        d = _info_dict(a_opt, ra[i]); d.update({k: v for k, v in _info_dict(c_opt, rc_[i]).items() if k.startswith(c_opt.name)})
        d[a_opt.name + "batches_trained"] = int(ra[i, L.INFO["batches_trained"]]); d[c_opt.name + "batches_trained"] = int(rc_[i, L.INFO["batches_trained"]])
        out.append(d)
    return out


def solve(solver, mdp):
    """POMDPs.solve(S::OnPolicySolver, mdp) (src/model_free/on_policy.jl:80-109), with the pre-train and per-iteration log points (:84,:88,:105; crux.jl_amd/logging.py)."""
    if solver.buffer is None:
        solver.buffer = ExperienceBuffer(solver.S, solver.agent.space, solver.dN, solver.required_columns)
        solver.sampler = Sampler(mdp, solver.agent, S=solver.S, required_columns=solver.required_columns, lam=solver.lambda_gae,
                                 max_steps=solver.max_steps, Vc=getattr(solver, "Vc", None))
    D, s = solver.buffer, solver.sampler
    if solver.log is not None:                                                                                    # :84, :88 log the pre-train performance: log(S.log, S.i, S=S)
        from . import logging as _lg
        if solver.log.sampler is None:
            solver.log.sampler = s
        _lg.log(solver.log, solver.i, S=solver)
    stop = solver.i + solver.N - solver.dN
    i = solver.i
    while i <= stop:
        solver.i = i
        info = steps_(s, D, Nsteps=solver.dN, explore=True, i=i, reset=True, cb=solver.post_sample_callback, store=getattr(solver, "interaction_storage", None))     # :96
        if solver.post_batch_callback:
            solver.post_batch_callback(D, info)                                                                   # :99
        tinfo = policy_gradient_training(solver, D)                                                               # :102
        tinfo.update({k: v for k, v in info.items() if k not in ("sum_r", "n_episode_end")})   # avg_r (record_avgr) and whatever the callback logged
        solver.history.append(tinfo)
        if solver.log is not None:                                                                                # :105 log(S.log, S.i + 1:S.i + S.dN, training_info, S=S)
            from . import logging as _lg
            if solver.log.sampler is None:
                solver.log.sampler = s
            _lg.log(solver.log, (i + 1, i + solver.dN), tinfo, S=solver)
        i += solver.dN
    solver.i += solver.dN
    return solver.agent.pi


def PPO(pi, S, eps=0.2, lambda_p=1.0, lambda_e=0.1, target_kl=0.012, a_opt=None, c_opt=None, required_columns=(), **kw):
    """PPO(; pi::ActorCritic, eps, lambda_p, lambda_e, target_kl, a_opt, c_opt, ...) (src/model_free/rl/ppo.jl:40-65)."""
    a_opt, c_opt = dict(a_opt or {}), dict(c_opt or {})
    cols = list(dict.fromkeys(list(required_columns) + ["return", "logprob", "advantage"]))
    return OnPolicySolver(agent=PolicyParams(pi), S=S, P={"eps": eps, "lambda_p": lambda_p, "lambda_e": lambda_e},
                          a_opt=TrainingParams(loss=ppo_loss, target_kl=target_kl, name="actor_", **a_opt),
                          c_opt=TrainingParams(loss=value_mse_loss, name="critic_", **c_opt),
                          post_batch_callback=lambda D, info: whiten_(D, "advantage"),
                          required_columns=cols, **kw)


def LagrangePPO(pi, Vc, S, eps=0.2, lambda_p=1.0, lambda_e=0.1, lambda_gae=0.95, target_kl=0.012, target_cost=0.025, penalty_scale=1.0, penalty_max=math.inf,
                Ki_max=10.0, Ki=1e-3, Kp=1.0, Kd=0.0, ema_alpha=0.95, a_opt=None, c_opt=None, cost_opt=None, required_columns=(), **kw):
    """LagrangePPO(; pi::ActorCritic, Vc::ContinuousNetwork, ...) (src/model_free/rl/ppo.jl:138-215): PPO whose actor loss carries a PID-controlled cost
    penalty (lagrange_ppo_loss, :70-131), a cost critic Vc regressed on :cost_return, and the cost columns filled by the sampler (sampler.jl:65-66,114).
    The controller's state (I, Jc_prev, smooth_delta, smooth_Jc: the one-element arrays of P, :192-201) lives in P["lagrange"]."""
    a_opt, c_opt, cost_opt = dict(a_opt or {}), dict(c_opt or {}), dict(cost_opt or {})
    lag = L.Lagrange(); lag.target_cost, lag.penalty_max, lag.Ki_max, lag.Ki, lag.Kp, lag.Kd, lag.ema_alpha = target_cost, penalty_max, Ki_max, Ki, Kp, Kd, ema_alpha
    cols = list(dict.fromkeys(list(required_columns) + ["return", "advantage", "logprob", "cost_advantage", "cost", "cost_return"]))
    return OnPolicySolver(agent=PolicyParams(pi), S=S, P={"eps": eps, "lambda_p": lambda_p, "lambda_e": lambda_e, "lagrange": lag, "penalty_scale": penalty_scale},
                          Vc=Vc, lambda_gae=lambda_gae,
                          a_opt=TrainingParams(loss=lagrange_ppo_loss, target_kl=target_kl, name="actor_", **a_opt),
                          c_opt=TrainingParams(loss=value_mse_loss, name="critic_", **c_opt),
                          cost_opt=TrainingParams(loss=cost_value_mse_loss, name="cost_critic_", **cost_opt),
                          post_batch_callback=lambda D, info: whiten_(D, "advantage"),
                          required_columns=cols, **kw)


def A2C(pi, S, lambda_p=1.0, lambda_e=0.1, a_opt=None, c_opt=None, required_columns=(), **kw):
    """A2C(; pi::ActorCritic, a_opt, c_opt, lambda_p=1f0, lambda_e=0.1f0, ...) (src/model_free/rl/a2c.jl:32-52): a2c_loss with the 0.015 KL early stop,
    critic mse, advantages whitened after sampling (post_sample_callback, :48)."""
    a_opt, c_opt = dict(a_opt or {}), dict(c_opt or {})
    a_opt.setdefault("target_kl", 0.015)
    cols = list(dict.fromkeys(list(required_columns) + ["return", "logprob", "advantage"]))
    return OnPolicySolver(agent=PolicyParams(pi), S=S, P={"lambda_p": lambda_p, "lambda_e": lambda_e},
                          a_opt=TrainingParams(loss=a2c_loss, name="actor_", **a_opt), c_opt=TrainingParams(loss=value_mse_loss, name="critic_", **c_opt),
                          post_sample_callback=lambda D, info: whiten_(D, "advantage"), required_columns=cols, **kw)


def REINFORCE(pi, S, a_opt=None, required_columns=(), **kw):
    """REINFORCE(; pi, a_opt, ...) (src/model_free/rl/reinforce.jl:30-42): reinforce_loss with the 0.015 KL early stop; no critic, no GAE."""
    a_opt = dict(a_opt or {}); a_opt.setdefault("target_kl", 0.015)
    cols = list(dict.fromkeys(list(required_columns) + ["return", "logprob"]))
    return OnPolicySolver(agent=PolicyParams(pi), S=S, a_opt=TrainingParams(loss=reinforce_loss, name="actor_", **a_opt), c_opt=None, required_columns=cols, **kw)


_solve_on_policy = solve
