"""off_policy.py -- OffPolicySolver: value_training (DQN / SoftQ / SAC / DDPG / TD3 epochs), the asynchronous solve loop, the solver constructors (src/model_free/off_policy.jl, rl/dqn.jl, rl/sac.jl, rl/softq.jl, rl/ddpg.jl, rl/td3.jl).

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
from .imitation import (  # noqa: F401
    BC, BatchSolver, OnPolicyGAIL, _solve_batch, batch_train_gail_d_, gail_d_loss, gail_reward_, logpdf_bc_loss, loss_value, mse_action_loss, stop_on_validation_increase)


td_loss = _Loss("td")                # td_loss() (src/utils.jl:76-87)


class OffPolicySolver:
    """OffPolicySolver(; agent, S, N, dN=4, max_steps=100, c_opt, buffer_size=1000, buffer, buffer_init, target_fn, target_update, priority_fn,
    post_sample_callback, post_batch_callback, pre_train_callback, extra_buffers, buffer_fractions) (src/model_free/off_policy.jl:37-64).

    The function-valued fields (:53-63) accept what the reference accepts:
      target_fn            a built-in name ("dqn", "softq", "sac", "ddpg", "td3": the fused device paths) or a callable (pi_minus, P, D, gamma; i) -> y of B
                           Float32 targets (:56, called at :80)
      priority_fn          None = td_error (utils.jl:112, :60) or a callable (pi, P, D, y) -> B non-negative values (:83)
      target_update        None = polyak_average!(pi_minus, pi, tau) (:55) or a callable (pi_minus, pi; i=None) (:100, :108)
      post_sample_callback (D; S, info) after every steps! with the freshly sampled rows as a dict of host arrays; columns the callback modifies are written
                           back into the ring (:50, :125, :138)
      post_batch_callback  (D; S, info) after every rand! with the staging buffer (:53, :77)
      pre_train_callback   (S; info) once per iteration before value_training (:54, :140)
      extra_buffers / buffer_fractions   further sources of rand! and the share of the minibatch each source gets (:62-63, :71)
    A solver whose seams are all built-ins runs the fused epoch chains; any callable (or an extra buffer) selects the call-by-call form of the same loop, in
    which every piece is its own C call and the callables run on the host between them -- the analogue of the reference calling user code between Flux calls."""

    def __init__(self, agent, S, N=1000, dN=4, max_steps=100, c_opt=None, buffer_size=1000, buffer=None, buffer_init=None, tau=0.005,
                 prioritized=False, weighted_loss=False, i=0, a_opt=None, param_optimizers=None, P=None, target_fn="dqn", noise_seed=0, log=None, sample_seed=SAMPLE_SEED,
                 target_update=None, priority_fn=None, post_sample_callback=None, post_batch_callback=None, pre_train_callback=None, extra_buffers=(),
                 buffer_fractions=None, required_columns=(), interaction_storage=None):
        self.interaction_storage = interaction_storage      # a list: every steps! block is appended to it (off_policy.jl:18,49,126,138)
        self.agent, self.S, self.N, self.dN, self.max_steps, self.c_opt, self.i = agent, S, int(N), int(dN), int(max_steps), c_opt, int(i)
        self.log = log                         # LoggerParams (crux_jl_amd.logging) or None
        self.a_opt, self.param_optimizers, self.P, self.target_fn, self.noise_seed = a_opt, list(param_optimizers or []), dict(P or {}), target_fn, int(noise_seed)
        self.buffer = buffer if buffer is not None else ExperienceBuffer(S, agent.space, buffer_size, list(required_columns), prioritized=prioritized)
        self.buffer_init = buffer_init if buffer_init is not None else max(c_opt.batch_size, 200)
        self.tau, self.weighted_loss, self.sample_seed = float(tau), bool(weighted_loss), int(sample_seed)
        self.target_update, self.priority_fn = target_update, priority_fn
        self.post_sample_callback, self.post_batch_callback, self.pre_train_callback = post_sample_callback, post_batch_callback, pre_train_callback
        self.extra_buffers = list(extra_buffers)
        self.buffer_fractions = list(buffer_fractions) if buffer_fractions is not None else ([1.0] if not self.extra_buffers else None)
        if self.extra_buffers and (self.buffer_fractions is None or len(self.buffer_fractions) != 1 + len(self.extra_buffers)):
            raise ValueError("buffer_fractions needs one entry per source: the buffer and every extra buffer (off_policy.jl:62-63)")
        self.fused_epochs = True              # value_training's epoch loop through crux_dqn_epochs / crux_sac_epochs (recorded op lists run by the executor for wide networks)
        self.sampler, self.batch, self._history = None, None, []
        self._dy = self._derr = None
        # solve() without the host in the loop (cruxhip.h: crux_dqn_epochs_async): the epochs' info rows stay on the device until somebody looks at `history`
        self.async_training = True
        self._async_now = self._async_unsupported = self._async_fell_back = False
        self._dinfos, self._dinfos_rows, self._dinfos_used, self._pending = None, 0, 0, []      # device ring of info rows; (history index, first row, epochs, name) not yet fetched

    @property
    def history(self):
        """One info dict per iteration (the `training_info` the reference logs at off_policy.jl:146). Iterations that ran through the asynchronous chain are fetched
        from the device here, on first access: one synchronisation for all of them. A NaN loss raises the reference's "NaN detected!" (training.jl:20) at that point."""
        self._resolve_history()
        return self._history

    @history.setter
    def history(self, v):
        self._resolve_history(); self._history = v

    def _resolve_history(self):
        if not self._pending:
            return
        ctx = self.buffer.ctx
        rows = np.zeros((self._dinfos_used, L.INFO_N), np.float32)
        ctx.sync(); ctx.d2h(self._dinfos, rows)
        pend, self._pending, self._dinfos_used = self._pending, [], 0
        bad = None
        for hi, r0, n, decode, extra in pend:
            raws = rows[r0:r0 + n]
            infos, nan = decode(raws)
            keys = {k for x in infos for k in x}
            d = {k: float(np.mean([x[k] for x in infos if k in x])) for k in keys}                       # aggregate_info: mean over the dicts that have the key (logging.jl:60-66)
            d.update({k: v for k, v in extra.items() if k not in d})
            self._history[hi] = d
            if bad is None and nan:
                bad = hi
        if bad is not None:
            raise L.CruxError(L.ENAN, "NaN detected! (grad norm is NaN, src/training.jl:20) in iteration %d of this solve (asynchronous chain: reported when the infos were fetched)" % bad)

    def custom_seams(self):
        """True when a function-valued field is not the built-in: value_training then runs call by call with the callables on the host."""
        return (callable(self.target_fn) or self.priority_fn is not None or self.target_update is not None or self.post_batch_callback is not None
                or bool(self.extra_buffers))

    def _sources(self):
        return [self.buffer] + self.extra_buffers

    def _rand(self, D, counter):
        """rand!(D, S.buffer, S.extra_buffers...; fracs=S.buffer_fractions, i=S.i) (:71)"""
        rand_(D, *self._sources(), i=self.i, fracs=self.buffer_fractions if self.extra_buffers else None, counter=counter, seed=self.sample_seed)

    def _update_target(self, final=False):
        """S.target_update(pi_minus, pi) (:100) / S.target_update(pi_minus, pi, i = S.i + 1 : S.i + dN) (:108)"""
        if self.target_update is None:
            polyak_average_(self.agent.pi_minus, self.agent.pi, self.tau)
        elif final:
            self.target_update(self.agent.pi_minus, self.agent.pi, i=range(self.i + 1, self.i + self.dN + 1))
        else:
            self.target_update(self.agent.pi_minus, self.agent.pi)


def _value_training_sac(solver, D, gamma):
    """value_training (src/model_free/off_policy.jl:66-111) with SAC's pieces (src/model_free/rl/sac.jl): per epoch rand! -> sac_target ->
    train!(log_alpha, sac_temp_loss) -> train!(critic, double_Q_loss) -> train!(actor, sac_actor_loss) -> target_update."""
    pi, pim, buf, ctx = solver.agent.pi, solver.agent.pi_minus, solver.buffer, solver.buffer.ctx
    A, Q, Qm, la = pi.A, pi.C, pim.C, solver.P["SAC_log_alpha"]
    c_opt, a_opt = solver.c_opt, solver.a_opt
    (_, t_opt), = solver.param_optimizers                                                               # Flux.params(SAC_log_alpha) => temp_ (sac.jl:101)
    _ensure_opt(Q.N1, c_opt); _ensure_opt(Q.N2, c_opt); _ensure_opt(A, a_opt); _ensure_opt(la, t_opt)
    if buf.isprioritized():
        raise NotImplementedError("SAC with a prioritized buffer: td_error over a DoubleNetwork is not defined in the reference either")
    B = D.capacity
    if solver._dy is None:
        solver._dy = ctx.alloc(4 * B)
    infos, lib, raw = [], ctx.lib, np.zeros(L.INFO_N, np.float32)
    fused = solver.fused_epochs and not solver.custom_seams()
    if fused:
        # the whole epoch loop (:69-104) in one C call: chains of up to 8 epochs per recorded list, no host round trip between them (cruxhip.h: crux_sac_epochs);
        # same pieces, order and draws as the epoch-by-epoch branch below
        _set_stream_for(buf, solver.sample_seed)
        n = c_opt.epochs; ctr0 = solver.i * n
        if getattr(solver, "_async_now", False):
            d_rows, row0 = _info_ring(solver, ctx, 3 * n)
            rc = lib.crux_sac_epochs_async(A.h, Q.N1.h, Q.N2.h, pim.A.h, Qm.N1.h, Qm.N2.h, la.h, buf.h, D.h, float(gamma), float(solver.P["SAC_H_target"]), float(solver.tau),
                                           1 if solver.weighted_loss else 0, 0, n, int(c_opt.update_every), int(a_opt.update_every), ctr0, solver.noise_seed, 3 * ctr0, d_rows)
            if rc == L.OK:
                ce, ae, tn, cn, an = int(c_opt.update_every), int(a_opt.update_every), t_opt.name, c_opt.name, a_opt.name
                def decode(raws):
                    out, nan = [], False
                    for epoch in range(len(raws) // 3):
                        rt_, rq_, ra_ = raws[3 * epoch], raws[3 * epoch + 1], raws[3 * epoch + 2]
                        info = {tn + "loss": float(rt_[0]), tn + "grad_norm": float(rt_[1]), "SAC alpha": float(rt_[L.INFO["alpha"]])}; nan = nan or bool(np.isnan(rt_[1]))
                        if epoch % ce == 0:
                            info.update({cn + "loss": float(rq_[0]), cn + "grad_norm": float(rq_[1]), "Q1avg": float(rq_[L.INFO["q1avg"]]), "Q2avg": float(rq_[L.INFO["q2avg"]])}); nan = nan or bool(np.isnan(rq_[1]))
                        if epoch % ae == 0:
                            info.update({an + "loss": float(ra_[0]), an + "grad_norm": float(ra_[1]), "entropy": float(ra_[L.INFO["entropy"]])}); nan = nan or bool(np.isnan(ra_[1]))
                        out.append(info)
                    return out, nan
                solver._dinfos_used += 3 * n
                return _PendingInfo(row0, 3 * n, decode)
            if rc != L.EUNSUP:
                ctx.check(rc)
            solver._async_now = False; solver._async_fell_back = True
        rt, rq, ra = (np.zeros((n, L.INFO_N), np.float32) for _ in range(3))
        ctx.check(lib.crux_sac_epochs(A.h, Q.N1.h, Q.N2.h, pim.A.h, Qm.N1.h, Qm.N2.h, la.h, buf.h, D.h, float(gamma), float(solver.P["SAC_H_target"]), float(solver.tau),
                                      1 if solver.weighted_loss else 0, 0, n, int(c_opt.update_every), int(a_opt.update_every), ctr0, solver.noise_seed, 3 * ctr0,
                                      _vp(rt), _vp(rq), _vp(ra)))
        for epoch in range(n):
            info = {t_opt.name + "loss": float(rt[epoch, 0]), t_opt.name + "grad_norm": float(rt[epoch, 1]), "SAC alpha": float(rt[epoch, L.INFO["alpha"]])}
            if epoch % c_opt.update_every == 0:
                info.update({c_opt.name + "loss": float(rq[epoch, 0]), c_opt.name + "grad_norm": float(rq[epoch, 1]), "Q1avg": float(rq[epoch, L.INFO["q1avg"]]), "Q2avg": float(rq[epoch, L.INFO["q2avg"]])})
            if epoch % a_opt.update_every == 0:
                info.update({a_opt.name + "loss": float(ra[epoch, 0]), a_opt.name + "grad_norm": float(ra[epoch, 1]), "entropy": float(ra[epoch, L.INFO["entropy"]])})
            infos.append(info)
    for epoch in range(0 if fused else c_opt.epochs):
        ctr = solver.i * c_opt.epochs + epoch                                                          # one Philox counter block per epoch
        upd_c, upd_a = epoch % c_opt.update_every == 0, epoch % a_opt.update_every == 0                # :91, :96
        solver._rand(D, ctr)                                                                           # :71 rand!(D, buffer, extra_buffers...; fracs, i=S.i)
        info = {}
        if solver.post_batch_callback is not None:
            solver.post_batch_callback(D, S=solver, info=info)                                         # :77
        if callable(solver.target_fn):
            _upload_target(solver, D, solver.target_fn(pim, solver.P, D, gamma, i=solver.i))           # :80 with the caller's target
        else:
            ctx.check(lib.crux_sac_target(A.h, Qm.N1.h, Qm.N2.h, la.h, D.h, float(gamma), solver.noise_seed, 3 * ctr, solver._dy))       # :80
        ctx.check(lib.crux_sac_temp_step(A.h, la.h, D.h, float(solver.P["SAC_H_target"]), solver.noise_seed, 3 * ctr + 1, _vp(raw)))     # :86-88
        info.update({t_opt.name + "loss": float(raw[0]), t_opt.name + "grad_norm": float(raw[1]), "SAC alpha": float(raw[L.INFO["alpha"]])})
        if upd_c:                                                                                      # :91
            ctx.check(lib.crux_double_q_step(Q.N1.h, Q.N2.h, D.h, solver._dy, 1 if solver.weighted_loss else 0, _vp(raw)))              # :92
            info.update({c_opt.name + "loss": float(raw[0]), c_opt.name + "grad_norm": float(raw[1]), "Q1avg": float(raw[L.INFO["q1avg"]]), "Q2avg": float(raw[L.INFO["q2avg"]])})
        if upd_a:                                                                                      # :96
            ctx.check(lib.crux_sac_actor_step(A.h, Q.N1.h, Q.N2.h, la.h, D.h, solver.noise_seed, 3 * ctr + 2, _vp(raw)))                 # :97
            info.update({a_opt.name + "loss": float(raw[0]), a_opt.name + "grad_norm": float(raw[1]), "entropy": float(raw[L.INFO["entropy"]])})
            solver._update_target()                                                                    # :100 (target update only when the actor trains)
        infos.append(info)
    keys = {k for d in infos for k in d}
    return {k: float(np.mean([d[k] for d in infos if k in d])) for k in keys}      # aggregate_info: mean over the dicts that have the key (logging.jl:60-66)


def _value_training_dpg(solver, D, gamma):
    """value_training (src/model_free/off_policy.jl:66-111) for DDPG (ddpg.jl) and TD3 (td3.jl): per epoch rand! -> ddpg_target / td3_target ->
    train!(critic, td_loss | double_Q_loss) -> train!(actor, -mean(Q(s, mu(s)))) -> target_update."""
    pi, pim, buf, ctx = solver.agent.pi, solver.agent.pi_minus, solver.buffer, solver.buffer.ctx
    A, Q, Am, Qm = pi.A, pi.C, pim.A, pim.C
    twin = isinstance(Q, DoubleNetwork)
    c_opt, a_opt = solver.c_opt, solver.a_opt
    for q in ((Q.N1, Q.N2) if twin else (Q,)):
        _ensure_opt(q, c_opt)
    _ensure_opt(A, a_opt)
    if buf.isprioritized():
        raise NotImplementedError("DDPG/TD3 with a prioritized buffer is not wired up")
    B = D.capacity
    if solver._dy is None:
        solver._dy = ctx.alloc(4 * B)
    sm = solver.P.get("pi_smooth") if solver.target_fn == "td3" else None
    infos, lib, raw = [], ctx.lib, np.zeros(L.INFO_N, np.float32)
    fused = solver.fused_epochs and (not twin or solver.target_fn == "td3") and not solver.custom_seams()
    if fused:
        # the whole epoch loop (:69-104) in one C call: chains of up to 8 epochs per recorded list (cruxhip.h: crux_dpg_epochs); same pieces, order and draws as below
        _set_stream_for(buf, solver.sample_seed)
        n = c_opt.epochs; ctr0 = solver.i * n
        if getattr(solver, "_async_now", False):
            d_rows, row0 = _info_ring(solver, ctx, 2 * n)
            rc = lib.crux_dpg_epochs_async(A.h, (Q.N1 if twin else Q).h, Q.N2.h if twin else None, Am.h, (Qm.N1 if twin else Qm).h, Qm.N2.h if twin else None, buf.h, D.h,
                                           float(gamma), float(solver.tau), sm.sigma if sm else -1.0, sm.eps_min if sm else 0.0, sm.eps_max if sm else 0.0, sm.a_min if sm else 0.0,
                                           sm.a_max if sm else 0.0, 1 if solver.weighted_loss else 0, 0, n, int(c_opt.update_every), int(a_opt.update_every), ctr0,
                                           solver.noise_seed, ctr0, d_rows)
            if rc == L.OK:
                ce, ae, cn, an, tw = int(c_opt.update_every), int(a_opt.update_every), c_opt.name, a_opt.name, twin
                def decode(raws):
                    out, nan = [], False
                    for epoch in range(len(raws) // 2):
                        rq_, ra_ = raws[2 * epoch], raws[2 * epoch + 1]; info = {}
                        if epoch % ce == 0:
                            info.update({"Q1avg": float(rq_[L.INFO["q1avg"]]), "Q2avg": float(rq_[L.INFO["q2avg"]])} if tw else {"Qavg": float(rq_[L.INFO["q1avg"]])})
                            info.update({cn + "loss": float(rq_[0]), cn + "grad_norm": float(rq_[1])}); nan = nan or bool(np.isnan(rq_[1]))
                        if epoch % ae == 0:
                            info.update({an + "loss": float(ra_[0]), an + "grad_norm": float(ra_[1])}); nan = nan or bool(np.isnan(ra_[1]))
                        out.append(info)
                    return out, nan
                solver._dinfos_used += 2 * n
                return _PendingInfo(row0, 2 * n, decode)
            if rc != L.EUNSUP:
                ctx.check(rc)
            solver._async_now = False; solver._async_fell_back = True
        rq, ra = (np.zeros((n, L.INFO_N), np.float32) for _ in range(2))
        ctx.check(lib.crux_dpg_epochs(A.h, (Q.N1 if twin else Q).h, Q.N2.h if twin else None, Am.h, (Qm.N1 if twin else Qm).h, Qm.N2.h if twin else None, buf.h, D.h,
                                      float(gamma), float(solver.tau), sm.sigma if sm else -1.0, sm.eps_min if sm else 0.0, sm.eps_max if sm else 0.0, sm.a_min if sm else 0.0,
                                      sm.a_max if sm else 0.0, 1 if solver.weighted_loss else 0, 0, n, int(c_opt.update_every), int(a_opt.update_every), ctr0,
                                      solver.noise_seed, ctr0, _vp(rq), _vp(ra)))
        for epoch in range(n):
            info = {}
            if epoch % c_opt.update_every == 0:
                info.update({"Q1avg": float(rq[epoch, L.INFO["q1avg"]]), "Q2avg": float(rq[epoch, L.INFO["q2avg"]])} if twin else {"Qavg": float(rq[epoch, L.INFO["q1avg"]])})
                info.update({c_opt.name + "loss": float(rq[epoch, 0]), c_opt.name + "grad_norm": float(rq[epoch, 1])})
            if epoch % a_opt.update_every == 0:
                info.update({a_opt.name + "loss": float(ra[epoch, 0]), a_opt.name + "grad_norm": float(ra[epoch, 1])})
            infos.append(info)
    for epoch in range(0 if fused else c_opt.epochs):
        ctr = solver.i * c_opt.epochs + epoch
        solver._rand(D, ctr)                                                                           # :71 rand!(D, buffer, extra_buffers...; fracs, i=S.i)
        info = {}
        if solver.post_batch_callback is not None:
            solver.post_batch_callback(D, S=solver, info=info)                                         # :77
        ctx.check(lib.crux_dpg_target(Am.h, (Qm.N1 if twin else Qm).h, Qm.N2.h if (twin and solver.target_fn == "td3") else None, D.h, float(gamma),
                                      sm.sigma if sm else -1.0, sm.eps_min if sm else 0.0, sm.eps_max if sm else 0.0, sm.a_min if sm else 0.0, sm.a_max if sm else 0.0,
                                      solver.noise_seed, ctr, solver._dy))                             # :80
        if epoch % c_opt.update_every == 0:                                                            # :91
            if twin:
                ctx.check(lib.crux_double_q_step(Q.N1.h, Q.N2.h, D.h, solver._dy, 1 if solver.weighted_loss else 0, _vp(raw)))
                info.update({"Q1avg": float(raw[L.INFO["q1avg"]]), "Q2avg": float(raw[L.INFO["q2avg"]])})
            else:
                ctx.check(lib.crux_q_step(Q.h, D.h, solver._dy, 1 if solver.weighted_loss else 0, _vp(raw)))
                info["Qavg"] = float(raw[L.INFO["q1avg"]])
            info.update({c_opt.name + "loss": float(raw[0]), c_opt.name + "grad_norm": float(raw[1])})   # :92
        if epoch % a_opt.update_every == 0:                                                            # :96 (TD3's delayed policy update = a_opt.update_every)
            ctx.check(lib.crux_dpg_actor_step(A.h, (Q.N1 if twin else Q).h, D.h, _vp(raw)))             # :97
            info.update({a_opt.name + "loss": float(raw[0]), a_opt.name + "grad_norm": float(raw[1])})
            solver._update_target()                                                                    # :100
        infos.append(info)
    keys = {k for d in infos for k in d}
    return {k: float(np.mean([d[k] for d in infos if k in d])) for k in keys}                          # aggregate_info: mean over the dicts that have the key (logging.jl:60-66)


def _set_stream_for(buf, seed):
    if seed is not None and int(seed) != getattr(buf, "sample_seed", SAMPLE_SEED):
        set_sample_stream_(buf, int(seed), getattr(buf, "sample_stream", 0))


def _upload_target(solver, D, y):
    """the targets a user target_fn returned (1 x B or B Float32, like the reference's y) into the device block the loss heads read"""
    y = np.ascontiguousarray(np.asarray(y, np.float32).reshape(-1))
    if y.size != D.capacity:
        raise ValueError("target_fn returned %d targets for a batch of %d" % (y.size, D.capacity))
    solver.buffer.ctx.h2d(solver._dy, y)
    return y


def value_training(solver, D, gamma):
    """value_training(S, D, gamma) (src/model_free/off_policy.jl:66-111) for the critic-only (DQN) case: per epoch
    rand! -> post_batch_callback -> target_fn -> [update_priorities!(priority_fn)] -> train!(td_loss); then target_update once (:108)."""
    if solver.target_fn == "sac" or (callable(solver.target_fn) and solver.a_opt is not None and isinstance(solver.agent.pi.A, GaussianPolicy)):
        return _value_training_sac(solver, D, gamma)
    if solver.target_fn in ("ddpg", "td3"):
        return _value_training_dpg(solver, D, gamma)
    pi, pim, buf, p, ctx = solver.agent.pi, solver.agent.pi_minus, solver.buffer, solver.c_opt, solver.buffer.ctx
    _ensure_opt(pi, p)
    B = D.capacity
    if solver._dy is None:
        solver._dy, solver._derr = ctx.alloc(4 * B), ctx.alloc(4 * B)
    infos = []
    fused = solver.target_fn in ("dqn", "softq") and solver.fused_epochs and not solver.custom_seams()
    if fused:
        # the whole epoch loop (:69-93) in one C call: for wide networks all c_opt.epochs epochs are recorded into one list and run without a host round trip
        # between them (cruxhip.h: crux_dqn_epochs); same steps, same order, same draws as the separate calls below
        _set_stream_for(buf, solver.sample_seed)
        beta = float(np.float32(buf.beta(solver.i))) if buf.isprioritized() else 0.0                       # rand!(D, buffer, i=S.i): beta(S.i)
        raws = np.zeros((p.epochs, L.INFO_N), np.float32)
        if getattr(solver, "_async_now", False):
            # no host in the loop: the chain is enqueued and the info rows stay on the device (OffPolicySolver.history fetches them)
            d_rows, _row0 = _info_ring(solver, ctx, p.epochs)
            # the epoch loop AND the target update of :108 (polyak_average!(pi_minus, pi, tau): the built-in target_update of this fused path) in ONE chain: the polyak
            # update rides in the last epoch's final phase instead of a launch of its own (cruxhip.h: crux_dqn_value_training_async)
            rc = ctx.lib.crux_dqn_value_training_async(pi.h, pim.h, buf.h, D.h, float(gamma), float(solver.P["alpha"]) if solver.target_fn == "softq" else 0.0,
                                                       1 if solver.weighted_loss else 0, beta, solver.i * p.epochs, p.epochs, float(np.float32(solver.tau)), d_rows)
            if rc == L.OK:
                name = p.name; row0 = _row0
                def decode(raws):
                    return [{name + "loss": float(r[0]), name + "grad_norm": float(r[1]), "Qavg": float(r[2])} for r in raws], bool(np.isnan(raws[:, 1]).any())
                pend = _PendingInfo(row0, p.epochs, decode); solver._dinfos_used += p.epochs
                return pend                                                                                # (:108 ran inside the chain)
            if rc != L.EUNSUP:
                ctx.check(rc)
            solver._async_now = False; solver._async_fell_back = True      # narrow networks: the synchronous entry point from here on
        if solver.target_fn == "softq":      # softq_target(alpha) in place of dqn_target (rl/softq.jl:4-13)
            ctx.check(ctx.lib.crux_softq_epochs(pi.h, pim.h, buf.h, D.h, float(gamma), float(solver.P["alpha"]), 1 if solver.weighted_loss else 0, beta, solver.i * p.epochs, p.epochs, _vp(raws)))
        else:
            ctx.check(ctx.lib.crux_dqn_epochs(pi.h, pim.h, buf.h, D.h, float(gamma), 1 if solver.weighted_loss else 0, beta, solver.i * p.epochs, p.epochs, _vp(raws)))
        infos = [{p.name + "loss": float(r[0]), p.name + "grad_norm": float(r[1]), "Qavg": float(r[2])} for r in raws]
    for epoch in range(0 if fused else p.epochs):
        raw = np.zeros(L.INFO_N, np.float32); info = {}
        solver._rand(D, solver.i * p.epochs + epoch)                                                   # :71 rand!(D, buffer, extra_buffers...; fracs, i=S.i): beta(S.i); the Philox counter is unique per draw
        if solver.post_batch_callback is not None:
            solver.post_batch_callback(D, S=solver, info=info)                                         # :77
        y_host = None
        if callable(solver.target_fn):
            y_host = _upload_target(solver, D, solver.target_fn(pim, solver.P, D, gamma, i=solver.i))  # :80 with the caller's target
        elif solver.target_fn == "softq":
            ctx.check(ctx.lib.crux_softq_target(pim.h, D.h, float(gamma), float(solver.P["alpha"]), solver._dy))   # :80  softq.jl:4-13
        else:
            ctx.check(ctx.lib.crux_dqn_target(pim.h, D.h, float(gamma), solver._dy))                    # :80  dqn.jl:4-6
        if buf.isprioritized() and solver.priority_fn is not None:                                     # :83 with the caller's priority function
            if y_host is None:
                y_host = np.empty(B, np.float32); ctx.d2h(solver._dy, y_host)
            v = np.ascontiguousarray(np.asarray(solver.priority_fn(pi, solver.P, D, y_host), np.float32).reshape(-1))
            buf.update_priorities_(D.indices[:B] + 1, v)
            ctx.check(ctx.lib.crux_td_step(pi.h, D.h, solver._dy, 1 if solver.weighted_loss else 0, _vp(raw)))   # :91-93
        elif buf.isprioritized():                                                                      # :83 update_priorities!(buffer, D.indices, td_error) and :91-93 train!
            ctx.check(ctx.lib.crux_td_step_with_error(pi.h, D.h, solver._dy, 1 if solver.weighted_loss else 0, solver._derr, _vp(raw)))   # one forward pass for both
            ctx.check(ctx.lib.crux_per_update_device(buf.h, ctx.lib.crux_buffer_indices_ptr(D.h), solver._derr, B))
        else:
            ctx.check(ctx.lib.crux_td_step(pi.h, D.h, solver._dy, 1 if solver.weighted_loss else 0, _vp(raw)))   # :91-93
        info.update({p.name + "loss": float(raw[0]), p.name + "grad_norm": float(raw[1]), "Qavg": float(raw[2])})
        infos.append(info)
    solver._update_target(final=True)                                                                  # :108
    keys = {k for d in infos for k in d}
    return {k: float(np.mean([d[k] for d in infos if k in d])) for k in keys}                          # aggregate_info: mean over the dicts that have the key (logging.jl:60-66)


def value_training_async(solver, D, gamma):
    """value_training without the host in the loop, for callers outside solve() (benchmarks): the chain is enqueued, its info rows stay in the solver's device ring and
    are registered as pending, so `solver.history` (or `_resolve_history`) fetches them later. Falls back to the synchronous call (returns the info dict) where the
    asynchronous entry point does not apply."""
    solver._async_now = solver.async_training and not solver._async_unsupported
    try:
        tinfo = value_training(solver, D, gamma)
    finally:
        solver._async_unsupported = solver._async_unsupported or solver._async_fell_back
        solver._async_now = False
    if isinstance(tinfo, _PendingInfo):
        solver._history.append(None); solver._pending.append((len(solver._history) - 1, tinfo.row0, tinfo.n, tinfo.decode, {}))
    else:
        solver._history.append(tinfo)
    return tinfo


def _info_ring(solver, ctx, nrows):
    """nrows rows of the solver's device info ring (fetching what is pending when it is full); returns the device address of the first one and its row index"""
    if solver._dinfos is None or solver._dinfos_used + nrows > solver._dinfos_rows:
        solver._resolve_history()
        solver._dinfos_used = 0                    # whatever was pending has been fetched; rows handed to callers that never registered them are dropped
        if solver._dinfos is None or nrows > solver._dinfos_rows:
            ctx.sync()                             # chains already enqueued may still write the old ring
            solver._dinfos_rows = max(4096, 8 * nrows); solver._dinfos = ctx.alloc(4 * L.INFO_N * solver._dinfos_rows)
    if solver._dinfos_used + nrows > solver._dinfos_rows:
        raise RuntimeError("info ring: %d rows requested, %d of %d in use" % (nrows, solver._dinfos_used, solver._dinfos_rows))
    base = solver._dinfos.value if hasattr(solver._dinfos, "value") else int(solver._dinfos)
    return C.c_void_p(base + 4 * L.INFO_N * solver._dinfos_used), solver._dinfos_used


class _PendingInfo:
    """value_training's info of an iteration whose chain is still on its way (crux_dqn_epochs_async): rows [row0, row0 + n) of the solver's device info ring."""
    def __init__(self, row0, n, decode):
        self.row0, self.n, self.decode = row0, n, decode      # decode(rows) -> (list of per-epoch info dicts, any NaN norm)


def _solve_small_dqn(solver, D, s, gamma, i, stop):
    """The iterations i, i + dN, ..., stop of solve(::OffPolicySolver) for a small DQN as a few launches of the one-workgroup solve kernel (cruxhip.h:
    crux_dqn_small_solve); returns the first iteration index it did NOT run (== i when the configuration needs the call-by-call loop)."""
    pe, pi, buf = solver.agent.pi_explore, solver.agent.pi, solver.buffer
    if not (s.h is not None and solver.fused_epochs and solver.target_fn == "dqn" and not solver.custom_seams() and solver.post_sample_callback is None and solver.pre_train_callback is None
            and solver.log is None and solver.interaction_storage is None and isinstance(pe, EpsGreedyPolicy) and isinstance(pi, DiscreteNetwork)
            and not buf.isprioritized() and not solver.weighted_loss and max(pi.network.dims) < 128 and D.capacity <= 256 and s.n_envs <= 4 and solver.dN % s.n_envs == 0 and i <= stop):
        return i
    p, ctx = solver.c_opt, buf.ctx
    _ensure_opt(pi, p); _set_stream_for(buf, solver.sample_seed)
    cfg, pi_on = _rollout_cfg(s, True, False, i)
    n_total = (stop - i) // solver.dN + 1
    while n_total > 0:
        n = min(n_total, 8192)
        infos = np.zeros((n, p.epochs, L.INFO_N), np.float32); sr, ne = C.c_double(), C.c_int64()
        rc = ctx.lib.crux_dqn_small_solve(pi.h, solver.agent.pi_minus.h, s.h, C.byref(cfg), buf.h, D.h, n, solver.dN, p.epochs, float(gamma), float(solver.tau), 0, int(i), _vp(infos), C.byref(sr), C.byref(ne))
        if rc == L.EUNSUP:
            return i
        ctx.check(rc)
        for k in range(n):
            solver.history.append({p.name + "loss": float(np.mean([float(x) for x in infos[k, :, 0]])), p.name + "grad_norm": float(np.mean([float(x) for x in infos[k, :, 1]])),
                                   "Qavg": float(np.mean([float(x) for x in infos[k, :, 2]]))})
        i += n * solver.dN; n_total -= n
        solver.i = i - solver.dN
    return i


def _post_sample(solver, n, info):
    """steps!(...; cb = D -> S.post_sample_callback(D, S=S, info=info)) (off_policy.jl:125,138; sampler.jl:151): the callback sees the n rows this steps!
    produced (host copies of the ring's newest rows, oldest first) and whatever it changes in them is written back into the ring."""
    if solver.post_sample_callback is None:
        return
    buf = solver.buffer
    ids = np.asarray(buf.get_last_N_indices(n), np.int64)       # 1-based ring rows of this steps!, oldest first
    rows = buf.minibatch(ids)
    before = {k: v.copy() for k, v in rows.items()}
    solver.post_sample_callback(rows, S=solver, info=info)
    for k, v in rows.items():
        if not np.array_equal(v, before[k], equal_nan=(v.dtype.kind == "f")):
            col = buf[k]; col[..., ids - 1] = v; buf[k] = col


def _solve_off_policy(solver, mdp):
    """POMDPs.solve(S::OffPolicySolver, mdp) (src/model_free/off_policy.jl:113-150), with the log points of :130 and :146 (crux.jl_amd/logging.py)."""
    gamma = np.float32(discount(mdp))
    if solver.batch is None:
        solver.batch = buffer_like(solver.buffer, capacity=solver.c_opt.batch_size)                     # :115
        solver.sampler = Sampler(mdp, solver.agent, S=solver.S, max_steps=solver.max_steps, required_columns=extra_columns(solver.buffer))
    D, s = solver.batch, solver.sampler
    istart = solver.i
    nfill = max(0, solver.buffer_init - len(solver.buffer))                                            # :122
    fill_info = {}
    if nfill > 0:
        solver.i += nfill                                                                              # :125 (Q12: advanced BEFORE sampling)
        if solver.interaction_storage is not None:      # :126 store=S.interaction_storage: the block goes to the storage after the callback, like the reference's `data`
            first = solver.buffer.next_ind - 1
            steps_(s, solver.buffer, Nsteps=nfill, explore=True, i=solver.i, want_info=False)
            _post_sample(solver, nfill, fill_info)
            solver.interaction_storage.append(solver.buffer.minibatch((first + np.arange(nfill)) % solver.buffer.capacity + 1))
        else:
            steps_(s, solver.buffer, Nsteps=nfill, explore=True, i=solver.i, want_info=False)
            _post_sample(solver, nfill, fill_info)
    if solver.log is not None:                                                                         # :130 log the pre-train performance: log(S.log, S.i, info, S=S)
        from . import logging as _lg
        if solver.log.sampler is None:
            solver.log.sampler = s
        _lg.log(solver.log, solver.i, fill_info, S=solver)
    i = solver.i
    stop = istart + solver.N - solver.dN
    i = _solve_small_dqn(solver, D, s, gamma, i, stop)                                                 # whole iterations in one launch where the configuration allows it
    while i <= stop:                                                                                   # :133
        solver.i = i
        first_ = solver.buffer.next_ind - 1
        steps_(s, solver.buffer, Nsteps=solver.dN, explore=True, i=i, want_info=False)                # :138 (its info is not used by this loop)
        it_info = {}
        # asynchronous chains when nothing on the host looks at an iteration's result before the next one starts: no logger, no callbacks, built-in seams
        solver._async_now = (solver.async_training and solver.log is None and solver.post_sample_callback is None and solver.pre_train_callback is None
                             and not solver.custom_seams() and solver.fused_epochs and not getattr(solver, "_async_unsupported", False) and solver.interaction_storage is None)
        _post_sample(solver, solver.dN, it_info)                                                      # :138 cb = D -> S.post_sample_callback(D, S=S, info=info)
        if solver.interaction_storage is not None:                                                    # :138 store=S.interaction_storage (after the callback, sampler.jl:150-151)
            solver.interaction_storage.append(solver.buffer.minibatch((first_ + np.arange(solver.dN)) % solver.buffer.capacity + 1))
        if solver.pre_train_callback is not None:
            solver.pre_train_callback(solver, info=it_info)                                           # :140
        try:
            tinfo = value_training(solver, D, gamma)                                                  # :143
        finally:
            solver._async_unsupported = solver._async_unsupported or solver._async_fell_back      # CRUX_EUNSUP once: the synchronous entry point from here on
            solver._async_now = False              # the request covers this call only: a later direct value_training(solver, ...) gets the info dict
        if isinstance(tinfo, _PendingInfo):          # the chain was only enqueued: the host goes on to the next iteration, `history` fetches the rows when asked
            solver._history.append(None); solver._pending.append((len(solver._history) - 1, tinfo.row0, tinfo.n, tinfo.decode, dict(it_info)))
        else:
            solver._history.append(tinfo)
            solver._history[-1].update({k: v for k, v in it_info.items() if k not in solver._history[-1]})  # :146 log(..., training_info, info)
        if solver.log is not None:                                                                     # :146 log(S.log, S.i, infos..., S=S)
            from . import logging as _lg
            if solver.log.sampler is None:
                solver.log.sampler = s
            _lg.log(solver.log, (i + 1, i + solver.dN), solver.history[-1], S=solver)
        i += solver.dN
    solver.i += solver.dN
    solver._resolve_history()                  # one synchronisation at the end: the infos of the asynchronous iterations, and their NaN check
    return solver.agent.pi


def DQN(pi, S, N, dN=4, pi_explore=None, c_opt=None, **kw):
    """DQN(; pi::DiscreteNetwork, N, dN=4, pi_explore=eps-greedy(LinearDecaySchedule(1., 0.1, N/2)), c_opt, ...) (src/model_free/rl/dqn.jl:27-46)."""
    import copy
    pe = pi_explore or EpsGreedyPolicy(LinearDecaySchedule(1.0, 0.1, N // 2), pi.outputs)
    pim = DiscreteNetwork(pi.network, pi.outputs, ctx=pi.ctx); copyto_(pim, pi)                          # pi_minus = deepcopy(pi)
    c = dict(c_opt or {}); c.setdefault("name", "critic_")
    return OffPolicySolver(agent=PolicyParams(pi, pi_explore=pe, pi_minus=pim), S=S, N=N, dN=dN,
                           c_opt=TrainingParams(loss=td_loss, epochs=dN, **c), **kw)


sac_actor_loss, sac_temp_loss, double_Q_loss = _Loss("sac_actor"), _Loss("sac_temp"), _Loss("double_q")   # sac.jl:34-52, utils.jl:89-96


def SAC(pi, S, N, dN=50, SAC_alpha=1.0, SAC_H_target=None, pi_explore=None, SAC_alpha_opt=None, a_opt=None, c_opt=None, **kw):
    """SAC(; pi::ActorCritic{GaussianPolicy, DoubleNetwork}, dN=50, SAC_alpha=1f0, SAC_H_target=-dim(A), pi_explore=GaussianNoiseExplorationPolicy(0.1f0),
    SAC_alpha_opt, a_opt, c_opt(epochs=dN), ...) (src/model_free/rl/sac.jl:75-106)."""
    if not (isinstance(pi, ActorCritic) and isinstance(pi.A, GaussianPolicy) and isinstance(pi.C, DoubleNetwork)):
        raise TypeError("SAC: pi must be ActorCritic(GaussianPolicy, DoubleNetwork(ContinuousNetwork, ContinuousNetwork))")
    ad = pi.A.network.dims[-1]
    P = {"SAC_log_alpha": ParamVector([np.log(np.float32(SAC_alpha))], ctx=pi.A.ctx), "SAC_H_target": np.float32(-ad if SAC_H_target is None else SAC_H_target)}
    c = dict(c_opt or {}); c.setdefault("name", "critic_"); c.setdefault("epochs", dN)
    a = dict(a_opt or {}); a.setdefault("name", "actor_")
    t = dict(SAC_alpha_opt or {}); t.setdefault("name", "temp_")
    return OffPolicySolver(agent=PolicyParams(pi, pi_explore=pi_explore or GaussianNoiseExplorationPolicy(0.1), pi_minus=clone_policy(pi)), S=S, N=N, dN=dN, P=P,
                           param_optimizers=[(P["SAC_log_alpha"], TrainingParams(loss=sac_temp_loss, **t))],
                           a_opt=TrainingParams(loss=sac_actor_loss, **a), c_opt=TrainingParams(loss=double_Q_loss, **c), target_fn="sac", **kw)


def SoftQ(pi, S, N, dN=4, c_opt=None, alpha=1.0, **kw):
    """SoftQ(; pi::DiscreteNetwork, N, dN=4, c_opt=(epochs=4,), alpha=1f0) (src/model_free/rl/softq.jl:31-58): the policy samples from
    softmax(Q ./ alpha) (always_stochastic, :52-53), target = softq_target(alpha)."""
    pi.always_stochastic, pi.logit_div = True, float(np.float32(alpha))
    pim = DiscreteNetwork(pi.network, pi.outputs, ctx=pi.ctx); copyto_(pim, pi)
    c = dict(c_opt or {}); c.setdefault("name", "critic_"); c.setdefault("epochs", 4)
    return OffPolicySolver(agent=PolicyParams(pi, pi_minus=pim), S=S, N=N, dN=dN, c_opt=TrainingParams(loss=td_loss, **c), target_fn="softq", P={"alpha": np.float32(alpha)}, **kw)


ddpg_actor_loss, td3_actor_loss = _Loss("ddpg_actor"), _Loss("td3_actor")   # ddpg.jl:26, td3.jl:12


def _dpg_solver(pi, S, N, dN, pi_explore, a_opt, c_opt, target_fn, a_loss, c_loss, pi_smooth, kw):
    c = dict(c_opt or {}); c.setdefault("name", "critic_"); c.setdefault("epochs", dN)
    a = dict(a_opt or {}); a.setdefault("name", "actor_")
    return OffPolicySolver(agent=PolicyParams(pi, pi_explore=pi_explore or GaussianNoiseExplorationPolicy(0.1), pi_minus=clone_policy(pi)), S=S, N=N, dN=dN,
                           P={"pi_smooth": pi_smooth or GaussianNoiseExplorationPolicy(0.1, eps_min=-0.5, eps_max=0.5)},
                           a_opt=TrainingParams(loss=a_loss, **a), c_opt=TrainingParams(loss=c_loss, **c), target_fn=target_fn, **kw)


def DDPG(pi, S, N, dN=50, pi_explore=None, a_opt=None, c_opt=None, pi_smooth=None, **kw):
    """DDPG(; pi::ActorCritic{ContinuousNetwork, ContinuousNetwork}, dN=50, pi_explore=GaussianNoiseExplorationPolicy(0.1f0), a_opt, c_opt(epochs=dN), ...)
    (src/model_free/rl/ddpg.jl:46-70): ddpg_target, td_loss critic, ddpg_actor_loss."""
    if not (isinstance(pi, ActorCritic) and isinstance(pi.A, ContinuousNetwork) and isinstance(pi.C, ContinuousNetwork)):
        raise TypeError("DDPG: pi must be ActorCritic(ContinuousNetwork, ContinuousNetwork)")
    return _dpg_solver(pi, S, N, dN, pi_explore, a_opt, c_opt, "ddpg", ddpg_actor_loss, td_loss, pi_smooth, kw)


def TD3(pi, S, N, dN=50, pi_explore=None, a_opt=None, c_opt=None, pi_smooth=None, **kw):
    """TD3(; pi::ActorCritic{ContinuousNetwork, DoubleNetwork}, dN=50, pi_smooth=GaussianNoiseExplorationPolicy(0.1f0, eps_min=-0.5f0, eps_max=0.5f0), ...)
    (src/model_free/rl/td3.jl:30-58): td3_target, double_Q_loss critic, td3_actor_loss through critic.N1."""
    if not (isinstance(pi, ActorCritic) and isinstance(pi.A, ContinuousNetwork) and isinstance(pi.C, DoubleNetwork)):
        raise TypeError("TD3: pi must be ActorCritic(ContinuousNetwork, DoubleNetwork(ContinuousNetwork, ContinuousNetwork))")
    return _dpg_solver(pi, S, N, dN, pi_explore, a_opt, c_opt, "td3", td3_actor_loss, double_Q_loss, pi_smooth, kw)
