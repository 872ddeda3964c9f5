"""Host-side mirror of the Crux.jl interface for the actor-learner hot path, over the C ABI of libcruxhip.so.

Julia is not installed where this is built, so the host layer that `north_star` asks to keep in Julia is mirrored here
in Python with the reference's names and argument meaning (Julia's `f!` is spelled `f_`). The Julia shim that binds the
same C symbols with `ccall` is shown in INTEGRATION.md. Every class/function cites the reference definition it mirrors
(paths under sisl/Crux.jl v0.1.4).

Array convention: like the Julia arrays, columns are (features, batch) with the batch LAST; numpy arrays returned here
are Fortran-ordered so their memory is identical to the Julia array's (and to what crosses the C ABI).

Round 4: the definitions live in four modules by solver family -- core.py (contexts, networks, buffers, sampling, steps!, losses, train! / batch_train!), on_policy.py
(OnPolicySolver, PPO, LagrangePPO, A2C, REINFORCE), imitation.py (OnPolicyGAIL, BC), off_policy.py (OffPolicySolver, DQN, SoftQ, SAC, DDPG, TD3); this module re-exports all of
them (private helpers included) and holds the `solve` dispatcher, so `crux.X` and `crux.api.X` resolve as before.
"""
import ctypes as C   # noqa: F401
import numpy as np   # noqa: F401

from . import _lib as L   # noqa: F401
from .core import *   # noqa: F401,F403
from .core import (  # noqa: F401
    ActorCritic, Adam, CartPoleMDP, Chain, Context, ContinuousNetwork, ContinuousSpace, CustomLoss, Dense, DiscreteNetwork, DiscreteSpace, DoubleNetwork, EpsGreedyPolicy,
    ExperienceBuffer, GaussianNoiseExplorationPolicy, GaussianPolicy, GymMDP, HostMDP, LinearDecaySchedule, MultitaskDecaySchedule, NetworkPolicy, ParamLoss, ParamVector, PendulumMDP,
    PolicyParams, SAMPLE_SEED, Sampler, SimpleGridWorld, SquashedGaussianPolicy, SynthMDP, TrainingParams, _F32_KEYS, _Loss, _batch_train_seam, _ensure_opt, _fill_block,
    _fill_importance_weights, _info_dict, _leaves, _np_dtype, _rollout_cfg, _train_cfg, _train_seam, _uses_seam, _vp, a2c_loss, actor, batch_train_, buffer_like, capacity,
    clone_policy, copy_buffer, copyto_, cost_value_mse_loss, critic, default_context, dim, discount, discounted_return, episodes, episodes_, extra_columns, failure, fill_gae_,
    fill_returns_, get_episodes, hcat, lagrange_ppo_loss, mdp_data, normalize_, peer_attach_local, policy_explore, polyak_average_, ppo_loss, prioritized_sample_, rand_, reinforce_loss,
    reload_switches, set_default_context, set_sample_stream_, shuffle_device_, split, split_batches, steps_, steps_multi_, train_, trim_, undiscounted_return, uniform_sample_,
    value, value_mse_loss, whiten_, whiten_multi_)
from .on_policy import *   # noqa: F401,F403
from .on_policy import (  # noqa: F401
    A2C, LagrangePPO, OnPolicySolver, PPO, REINFORCE, allreduce_mean_, policy_gradient_training, policy_gradient_training_multi, policy_gradient_training_synced, solve)
from .imitation import *   # noqa: F401,F403
from .imitation import (  # noqa: F401
    BC, BatchSolver, OnPolicyGAIL, _solve_batch, batch_train_gail_d_, gail_d_loss, gail_reward_, logpdf_bc_loss, loss_value, mse_action_loss, stop_on_validation_increase)
from .off_policy import *   # noqa: F401,F403
from .off_policy import (  # noqa: F401
    DDPG, DQN, OffPolicySolver, SAC, SoftQ, TD3, _PendingInfo, _dpg_solver, _info_ring, _post_sample, _set_stream_for, _solve_off_policy, _solve_small_dqn, _upload_target,
    _value_training_dpg, _value_training_sac, ddpg_actor_loss, double_Q_loss, sac_actor_loss, sac_temp_loss, td3_actor_loss, td_loss, value_training, value_training_async)
from .on_policy import _solve_on_policy
from . import core, on_policy, imitation, off_policy   # noqa: F401


def solve(solver, mdp=None):  # noqa: F811
    """POMDPs.solve(solver, mdp) for OnPolicySolver (on_policy.jl:80-109) and OffPolicySolver (off_policy.jl:113-150)."""
    if isinstance(solver, BatchSolver):
        return _solve_batch(solver, mdp)
    return _solve_off_policy(solver, mdp) if isinstance(solver, OffPolicySolver) else _solve_on_policy(solver, mdp)
