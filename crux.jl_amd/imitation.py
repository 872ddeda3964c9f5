"""imitation.py -- OnPolicyGAIL and the BatchSolver family (BC) (src/model_free/il/on_policy_gail.jl, il/bc.jl, src/model_free/batch.jl).

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
from .on_policy import (  # noqa: F401
    A2C, LagrangePPO, OnPolicySolver, PPO, REINFORCE, allreduce_mean_, policy_gradient_training, policy_gradient_training_multi, policy_gradient_training_synced, solve)


# --------------------------------------------------------------------------------------------------------------
# OnPolicyGAIL (src/model_free/il/on_policy_gail.jl) -- the discriminator is trained by batch_train! over TWO buffers (training.jl:28-44)
# --------------------------------------------------------------------------------------------------------------
gail_d_loss = _Loss("gail_d")     # gail_d_loss(GAN_BCELoss()) (on_policy_gail.jl:1-5, extras/gans.jl:7-9)


def batch_train_gail_d_(Dnet, p, D_expert, D_policy, info=None):
    """batch_train!(D, d_opt, (;), D_demo, deepcopy(D)) (on_policy_gail.jl:47, training.jl:28-55): every epoch shuffles both buffers, zips their
    minibatch partitions (the shorter buffer ends the epoch) and takes one discriminator step per pair. Shuffle k of this TrainingParams uses
    permutation counter 2k for the expert buffer and 2k+1 for the policy buffer. Epoch info = its last minibatch, result = mean over epochs."""
    _ensure_opt(Dnet, p)
    B, infos, total = p.batch_size, [], 0
    nb = min(-(-len(D_expert) // B), -(-len(D_policy) // B))
    stop = False
    for _ in range(p.epochs):
        shuffle_device_(D_expert, p.shuffle_seed, 2 * p.shuffle_counter); shuffle_device_(D_policy, p.shuffle_seed, 2 * p.shuffle_counter + 1)
        p.shuffle_counter += 1
        raw = np.zeros(L.INFO_N, np.float32)
        for k in range(nb):
            ne, npi = min(B, len(D_expert) - k * B), min(B, len(D_policy) - k * B)
            Dnet.ctx.check(Dnet.ctx.lib.crux_gail_d_step(Dnet.h, D_expert.h, k * B, ne, D_policy.h, k * B, npi, _vp(raw)))
            total += 1
            if total >= p.max_batches:
                stop = True; break
        infos.append({p.name + "loss": float(raw[L.INFO["loss"]]), p.name + "grad_norm": float(raw[L.INFO["grad_norm"]])})
        if stop:
            break
    out = {k: float(np.mean([d[k] for d in infos])) for k in infos[0]}
    out[p.name + "batches_trained"] = total
    if info is not None:
        info.update(out)
    return out


def gail_reward_(Dnet, buf, alpha_r=0.5, Rscale=1.0):
    """r = ar*logsigmoid(D(a,s)) - (1-ar)*logcompsigmoid(D(a,s)); buf[:r] .= r .* Rscale; returns mean(r) (on_policy_gail.jl:50-55)."""
    m = np.zeros(1, np.float32)
    Dnet.ctx.check(Dnet.ctx.lib.crux_gail_reward(Dnet.h, buf.h, float(alpha_r), float(Rscale), _vp(m)))
    return float(m[0])


def OnPolicyGAIL(pi, S, gamma, D, demo, lambda_gae=0.95, alpha_r=0.5, normalize_demo=True, solver=None, d_opt=None, Rscale=1.0, **kw):
    """OnPolicyGAIL(; pi, S, gamma, lambda_gae, D_demo, alpha_r, normalize_demo, D::ContinuousNetwork, solver=PPO, gan_loss=GAN_BCELoss(), d_opt, Rscale)
    (src/model_free/il/on_policy_gail.jl:26-69): PPO whose post_batch_callback trains the discriminator on (demo, copy of the fresh batch),
    overwrites the rewards with the discriminator's, and refills GAE / returns / whitened advantages."""
    d = dict(d_opt or {}); d.setdefault("name", "discriminator_")
    dp = TrainingParams(loss=gail_d_loss, **d)
    A = pi.space if hasattr(pi, "space") else ContinuousSpace(actor(pi).network.dims[-1])
    demo = copy_buffer(demo)
    if normalize_demo:
        normalize_(demo, S, A)
    sv = (solver or PPO)(pi=pi, S=S, lambda_gae=lambda_gae, **kw)

    def GAIL_callback(buf, info):
        batch_train_gail_d_(D, dp, demo, copy_buffer(buf), info=info)                    # :47
        info["disc_reward"] = gail_reward_(D, buf, alpha_r, Rscale)                    # :50-56
        fill_gae_(buf, sv.agent.pi, lambda_gae, gamma); fill_returns_(buf, gamma)        # :58-63
        whiten_(buf, "advantage")                                                       # :64
    sv.post_batch_callback = GAIL_callback
    sv.discriminator, sv.d_opt, sv.demo = D, dp, demo
    return sv


mse_action_loss, logpdf_bc_loss = _Loss("mse_action"), _Loss("logpdf_bc")   # src/model_free/il/bc.jl:1,10-18


def loss_value(pi, p, P, D):
    """loss(pi, P, D) evaluated on the whole buffer, no update (the validation error of stop_on_validation_increase, src/utils.jl:59-72)."""
    _ensure_opt(pi, p)
    cfg = _train_cfg(pi, p, P); n = len(D)
    ids = np.arange(n, dtype=np.int64); raw = np.zeros(L.INFO_N, np.float32)
    pi.ctx.check(pi.ctx.lib.crux_loss_grad(pi.h, D.h, C.byref(cfg), _vp(ids), n, _vp(raw)))
    return float(raw[L.INFO["loss"]])


class BatchSolver:
    """BatchSolver(; agent, S, D_train, a_opt, P, ...) (src/model_free/batch.jl:20-36) for the actor-only case used by BC."""

    def __init__(self, agent, S, D_train, a_opt, P=None, early_stopping=None, max_steps=100):
        self.agent, self.S, self.D_train, self.a_opt, self.P = agent, S, D_train, a_opt, dict(P or {})
        self.early_stopping, self.max_steps, self.epoch, self.history = early_stopping, int(max_steps), 0, []


def _solve_batch(solver, mdp=None):
    """POMDPs.solve(S::BatchSolver, mdp) (src/model_free/batch.jl:38-85): per epoch shuffle!, partition, train! per minibatch (one persistent
    launch per epoch here), then the early-stopping test on the list of epoch infos. Note the inclusive range: a_opt.epochs + 1 epochs (:47)."""
    A, p = actor(solver.agent.pi), solver.a_opt
    e_total, first = p.epochs, solver.epoch
    try:
        p.epochs = 1
        for solver.epoch in range(first, first + e_total + 1):
            info = batch_train_(A, p, solver.P, solver.D_train)
            solver.history.append({k: v for k, v in info.items() if not k.startswith("_")})
            if solver.early_stopping and solver.early_stopping(solver.history):
                break
    finally:
        p.epochs = e_total
    return solver.agent.pi


def stop_on_validation_increase(pi, P, D_val, p, window=5):
    """stop_on_validation_increase(pi, P, D_val, loss; window) (src/utils.jl:59-72)."""
    def f(infos):
        infos[-1]["validation_error"] = loss_value(pi, p, P, D_val)
        N = len(infos)
        if N >= 2 * window:
            cur = np.mean([infos[i]["validation_error"] for i in range(N - window, N)])
            old = np.mean([infos[i]["validation_error"] for i in range(N - 2 * window, N - window)])
            return bool(cur >= old)
        return False
    return f


def BC(pi, S, D_demo, normalize_demo=True, loss=None, validation_fraction=0.3, window=100, lambda_e=1e-3, opt=None, shuffle_perm=None, **kw):
    """BC(; pi, S, D_demo, normalize_demo, loss, validation_fraction=0.3, window=100, lambda_e=1f-3, opt) (src/model_free/il/bc.jl:37-70):
    mse_action_loss for a ContinuousNetwork, logpdf_bc_loss otherwise; the demonstrations are normalised, shuffled once and split into
    training / validation parts; early stopping on the validation error. shuffle_perm: optional 1-based permutation for the initial shuffle!."""
    loss = loss or (mse_action_loss if type(pi) is ContinuousNetwork else logpdf_bc_loss)
    A = pi.space if hasattr(pi, "space") else (DiscreteSpace(len(pi.outputs), pi.outputs) if isinstance(pi, DiscreteNetwork) else ContinuousSpace(pi.network.dims[-1]))
    D = buffer_like(D_demo, capacity=len(D_demo)); D.push_({k: D_demo[k] for k in D_demo.keys()})      # deepcopy(D_demo) (:52)
    if normalize_demo:
        normalize_(D, S, A)
    n = len(D)
    perm = np.asarray(shuffle_perm, np.int64) if shuffle_perm is not None else np.random.default_rng(0xBC).permutation(n).astype(np.int64) + 1
    D.shuffle_(perm)                                                                                    # shuffle!(D_demo) (:55)
    D_train, D_val = split(D, [1 - validation_fraction, validation_fraction])                         # (:56)
    P = {"lambda_e": lambda_e, "lambda_p": 1.0}
    o = dict(opt or {}); o.setdefault("name", "")
    p = TrainingParams(loss=loss, **o)
    return BatchSolver(agent=PolicyParams(pi), S=S, D_train=D_train, a_opt=p, P=P,
                       early_stopping=stop_on_validation_increase(pi, P, D_val, p, window=window), **kw)
