"""ctypes binding of libcruxhip.so (include/cruxhip.h). No torch types cross this boundary.

The library is the product: if it is missing, or no MI355X is visible when a context is created, the
package fails loudly -- there is no CPU fallback here (the CPU restatement lives in oracle/ and is test-only).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRUXHIP_LIB") or os.path.join(_HERE, "libcruxhip.so")   # CRUXHIP_LIB: an alternative build of the same ABI (compiler-flag experiments)

i32, i64, u32, u64, f32, f64, vp, cp = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_void_p, C.c_char_p
P = C.POINTER


class RolloutCfg(C.Structure):
    """crux_rollout_cfg (include/cruxhip.h)."""
    _fields_ = [("explore", i32), ("reset_at_end", i32), ("head", i32), ("eps_start", f64), ("eps_stop", f64),
                ("eps_steps", i64), ("noise_sigma", f32), ("noise_eps_min", f32), ("noise_eps_max", f32),
                ("a_min", f32), ("a_max", f32), ("logit_div", f32), ("i0", u64)]


class Lagrange(C.Structure):
    """crux_lagrange (include/cruxhip.h): LagrangePPO's penalty controller -- hyper-parameters, PID state, last values."""
    _fields_ = [("target_cost", f32), ("penalty_max", f32), ("Ki_max", f32), ("Ki", f32), ("Kp", f32), ("Kd", f32), ("ema_alpha", C.c_double),
                ("I", f32), ("Jc_prev", f32), ("smooth_delta", f32), ("smooth_Jc", f32),
                ("penalty", f32), ("cur_cost", f32), ("deriv_term", f32), ("reserved", f32)]


class TrainCfg(C.Structure):
    """crux_train_cfg (include/cruxhip.h)."""
    _fields_ = [("loss", i32), ("head", i32), ("batch_size", i32), ("epochs", i32), ("max_batches", i64),
                ("eps_clip", f32), ("lambda_p", f32), ("lambda_e", f32), ("target_kl", f32), ("shuffle_seed", u64),
                ("shuffle_counter", u64), ("reserved0", i32), ("target_col", i32)]


# name -> (restype, argtypes); every symbol declared in include/cruxhip.h appears here (checked by tests).
SIGNATURES = {
    "crux_ctx_create": (i32, [i32, vp, P(vp)]),
    "crux_ctx_destroy": (i32, [vp]),
    "crux_policy_explore": (i32, [vp, P(RolloutCfg), i32, vp, u64, vp, vp, vp]),
    "crux_steps_push": (i32, [vp, i64, P(vp), i64, i32, vp, f32, f32, vp, vp, i32, P(i64)]),
    "crux_last_error": (cp, [vp]),
    "crux_sync": (i32, [vp]),
    "crux_version": (cp, []),
    "crux_device_alloc": (i32, [vp, i64, P(vp)]),
    "crux_device_free": (i32, [vp, vp]),
    "crux_memcpy_h2d": (i32, [vp, vp, vp, i64]),
    "crux_memcpy_d2h": (i32, [vp, vp, vp, i64]),
    "crux_prof_enable": (i32, [vp, i32]),
    "crux_prof_reset": (i32, [vp]),
    "crux_prof_get": (i32, [vp, i32, P(f64), P(i64)]),
    "crux_mlp_create": (i32, [vp, i32, P(i32), P(i32), i32, P(vp)]),
    "crux_mlp_destroy": (i32, [vp]),
    "crux_mlp_n_params": (i64, [vp]),
    "crux_mlp_set_params": (i32, [vp, vp, i64]),
    "crux_mlp_get_params": (i32, [vp, vp, i64]),
    "crux_mlp_params_ptr": (vp, [vp]),
    "crux_mlp_grads_ptr": (vp, [vp]),
    "crux_mlp_init_glorot": (i32, [vp, u64, u32, f32]),
    "crux_mlp_forward": (i32, [vp, vp, i64, vp]),
    "crux_mlp_forward_host": (i32, [vp, vp, i64, vp]),
    "crux_mlp_copy": (i32, [vp, vp]),
    "crux_polyak": (i32, [vp, vp, f32]),
    "crux_adam_init": (i32, [vp, f64, f64, f64, f64]),
    "crux_adam_get_state": (i32, [vp, vp, vp, vp]),
    "crux_adam_set_state": (i32, [vp, vp, vp, vp]),
    "crux_adam_state_ptrs": (i32, [vp, P(vp), P(vp)]),
    "crux_buffer_create": (i32, [vp, i32, i32, i32, i64, u32, i32, f32, P(vp)]),
    "crux_buffer_destroy": (i32, [vp]),
    "crux_buffer_len": (i64, [vp]),
    "crux_buffer_capacity": (i64, [vp]),
    "crux_buffer_next_ind": (i64, [vp]),
    "crux_buffer_total_count": (i64, [vp]),
    "crux_buffer_has_column": (i32, [vp, i32]),
    "crux_buffer_clear": (i32, [vp]),
    "crux_buffer_column_info": (i32, [vp, i32, P(i32), P(i32)]),
    "crux_buffer_column_ptr": (i32, [vp, i32, P(vp)]),
    "crux_buffer_push_host": (i32, [vp, i64, P(vp), vp]),
    "crux_buffer_push_buffer": (i32, [vp, vp, vp, i64, vp]),
    "crux_buffer_read_column": (i32, [vp, i32, vp, i64]),
    "crux_buffer_write_column": (i32, [vp, i32, vp, i64]),
    "crux_buffer_permute": (i32, [vp, vp]),
    "crux_buffer_last_n_indices": (i64, [vp, i64, vp]),
    "crux_buffer_gather_host": (i32, [vp, vp, i64, P(vp)]),
    "crux_buffer_indices": (i32, [vp, vp, i64]),
    "crux_buffer_indices_ptr": (vp, [vp]),
    "crux_per_update": (i32, [vp, vp, vp, i32, i64]),
    "crux_per_update_device": (i32, [vp, vp, vp, i64]),
    "crux_per_sample": (i32, [vp, vp, i64, vp, f32, u64]),
    "crux_uniform_sample": (i32, [vp, vp, i64, vp, u64]),
    "crux_per_get": (i32, [vp, vp, P(f32), P(f32), vp]),
    "crux_buffer_set_sample_stream": (i32, [vp, u64, u32]),
    "crux_dqn_small_solve": (i32, [vp, vp, vp, P(RolloutCfg), vp, vp, i32, i32, i32, f32, f32, i32, u64, vp, P(f64), P(i64)]),
    "crux_dqn_epoch": (i32, [vp, vp, vp, vp, f32, i32, f32, u64, vp]),
    "crux_dqn_epochs": (i32, [vp, vp, vp, vp, f32, i32, f32, u64, i32, vp]),
    "crux_dqn_epochs_async": (i32, [vp, vp, vp, vp, f32, i32, f32, u64, i32, vp]), "crux_dqn_value_training_async": (i32, [vp, vp, vp, vp, f32, f32, i32, f32, u64, i32, f32, vp]),
    "crux_softq_epochs": (i32, [vp, vp, vp, vp, f32, f32, i32, f32, u64, i32, vp]),
    "crux_softq_epochs_async": (i32, [vp, vp, vp, vp, f32, f32, i32, f32, u64, i32, vp]),
    "crux_sac_epochs": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, i32, i32, i32, i32, i32, u64, u64, u64, vp, vp, vp]),
    "crux_sac_epochs_async": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, i32, i32, i32, i32, i32, u64, u64, u64, vp]),
    "crux_dpg_epochs": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i32, i32, i32, i32, i32, u64, u64, u64, vp, vp]),
    "crux_dpg_epochs_async": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i32, i32, i32, i32, i32, u64, u64, u64, vp]),
    "crux_sac_epoch": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, i32, i32, i32, u64, u64, u64, vp, vp, vp]),
    "crux_env_create": (i32, [vp, i32, i32, i32, f32, vp, vp, u64, i32, i32, P(vp)]),
    "crux_env_destroy": (i32, [vp]),
    "crux_env_obs_dim": (i32, [vp]),
    "crux_env_act_dim": (i32, [vp]),
    "crux_env_state_dim": (i32, [vp]),
    "crux_env_reset": (i32, [vp]),
    "crux_env_get_state": (i32, [vp, vp, vp, vp]),
    "crux_rollout": (i32, [vp, vp, P(RolloutCfg), vp, i64, P(f64), P(i64)]),
    "crux_env_step_host": (i32, [vp, i32, i64, vp, vp, vp, vp, vp, vp, vp]),
    "crux_fill_gae": (i32, [vp, vp, f32, f32]),
    "crux_fill_returns": (i32, [vp, f32]),
    "crux_fill_gae_rows": (i32, [vp, vp, f32, f32, i64, i64, i64, i32]),
    "crux_fill_returns_rows": (i32, [vp, f32, i64, i64, i64, i32]),
    "crux_whiten": (i32, [vp, i32]),
    "crux_batch_train": (i32, [vp, vp, P(TrainCfg), vp, vp, vp]),
    "crux_batch_train_lagrange": (i32, [vp, vp, P(TrainCfg), P(Lagrange), vp, vp, vp]),
    "crux_fill_gae_keys": (i32, [vp, vp, f32, f32, i32, i32]),
    "crux_fill_returns_keys": (i32, [vp, f32, i32, i32]),
    "crux_fill_gae_rows_keys": (i32, [vp, vp, f32, f32, i64, i64, i64, i32, i32, i32]),
    "crux_fill_returns_rows_keys": (i32, [vp, f32, i64, i64, i64, i32, i32, i32]),
    "crux_policy_gradient_training": (i32, [vp, vp, vp, P(TrainCfg), P(TrainCfg), vp, vp, vp, vp, vp, vp]),
    "crux_train_step": (i32, [vp, vp, P(TrainCfg), vp, i64, vp]),
    "crux_loss_grad": (i32, [vp, vp, P(TrainCfg), vp, i64, vp]),
    "crux_loss_grad_device_ids": (i32, [vp, vp, P(TrainCfg), vp, i64, vp]),
    "crux_adam_apply": (i32, [vp, f32]),
    "crux_rollout_multi": (i32, [i32, vp, vp, P(RolloutCfg), vp, i64, vp, vp]),
    "crux_policy_gradient_training_multi": (i32, [i32, vp, vp, vp, P(TrainCfg), P(TrainCfg), vp, vp]),
    "crux_policy_gradient_training_synced": (i32, [vp, vp, vp, P(TrainCfg), P(TrainCfg), i32, vp, vp]),
    "crux_mlp_set_squash": (i32, [vp, f32]),
    "crux_mlp_get_squash": (f32, [vp]),
    "crux_buffer_push_reservoir": (i32, [vp, i64, vp, i32, u64, u64]),
    "crux_buffer_shuffle": (i32, [vp, u64, u64]),
    "crux_gail_d_step": (i32, [vp, vp, i64, i64, vp, i64, i64, vp]),
    "crux_gail_reward": (i32, [vp, vp, f32, f32, vp]),
    "crux_fill_gae_multi": (i32, [i32, vp, vp, f32, f32, i32]),
    "crux_whiten_multi": (i32, [i32, vp, i32]),
    "crux_ctx_set_learner_cus": (i32, [vp, i32]),
    "crux_comm_unique_id": (i32, [vp, vp]),
    "crux_comm_init": (i32, [vp, i32, i32, vp]),
    "crux_comm_destroy": (i32, [vp]),
    "crux_comm_size": (i32, [vp]),
    "crux_peer_export": (i32, [vp, vp]), "crux_peer_attach": (i32, [vp, i32, i32, vp]), "crux_peer_attach_local": (i32, [P(vp), i32]),
    "crux_peer_detach": (i32, [vp]), "crux_peer_hist_enable": (i32, [vp, i32]), "crux_peer_set_sync_every": (i32, [vp, i32]), "crux_reload_switches": (i32, []), "crux_peer_sync_every": (i32, [vp]), "crux_peer_set_timeout_ms": (i32, [vp, i32]), "crux_peer_set_budget_ms": (i32, [vp, i32]), "crux_peer_abort": (i32, [vp]), "crux_peer_abort_clear": (i32, [vp]), "crux_abort_all": (i32, []), "crux_peer_abort_reason": (i32, [vp, vp]), "crux_peer_probe": (i32, [vp, i32, i32, i32, vp]), "crux_peer_wait_hist": (i32, [vp, vp, i32]), "crux_peer_size": (i32, [vp]), "crux_peer_rank": (i32, [vp]),
    "crux_allreduce_mean": (i32, [vp]),
    "crux_allreduce_grads": (i32, [vp]),
    "crux_first_episode_metrics": (i32, [vp, i32, i64, f32, vp, vp, vp, vp]),
    "crux_dqn_target": (i32, [vp, vp, f32, vp]),
    "crux_td_error": (i32, [vp, vp, vp, vp]),
    "crux_softq_target": (i32, [vp, vp, f32, f32, vp]),
    "crux_td_step": (i32, [vp, vp, vp, i32, vp]),
    "crux_td_step_with_error": (i32, [vp, vp, vp, i32, vp, vp]),
    "crux_importance_weight_rows": (i32, [vp, vp, i32, i64, i64]),
    "crux_fill_importance_weights_rows": (i32, [vp, i64, i64, i64, i32]),
    "crux_mlp_forward_cached": (i32, [vp, vp, i64, vp]),
    "crux_mlp_backward": (i32, [vp, vp, i64, vp, f32, i32, vp]),
    "crux_sac_target": (i32, [vp, vp, vp, vp, vp, f32, u64, u64, vp]),
    "crux_sac_temp_step": (i32, [vp, vp, vp, f32, u64, u64, vp]),
    "crux_double_q_step": (i32, [vp, vp, vp, vp, i32, vp]),
    "crux_sac_actor_step": (i32, [vp, vp, vp, vp, vp, u64, u64, vp]),
    "crux_dpg_target": (i32, [vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, u64, u64, vp]),
    "crux_q_step": (i32, [vp, vp, vp, i32, vp]),
    "crux_dpg_actor_step": (i32, [vp, vp, vp, vp]),
}

_lib = None


def bind(lib, signatures=SIGNATURES, prefix_from="crux_", prefix_to=None):
    for name, (res, args) in signatures.items():
        sym = name if prefix_to is None else prefix_to + name[len(prefix_from):]
        fn = getattr(lib, sym)
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load libcruxhip.so (built by __graft_entry__.build() / make -C crux.jl_amd/csrc). Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libcruxhip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C crux.jl_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib


# enums (mirrors of include/cruxhip.h) ---------------------------------------------------------------
OK, EINVAL, ENAN, EHIP, ERCCL, ENOMEM, EUNSUP = 0, -1, -2, -3, -4, -5, -6
ACT = {"identity": 0, "relu": 1, "tanh": 2}
COL = {"s": 0, "a": 1, "sp": 2, "r": 3, "done": 4, "episode_end": 5, "return": 6, "logprob": 7, "advantage": 8,
       "weight": 9, "t": 10, "i": 11, "value": 12, "cost": 13, "cost_advantage": 14, "cost_return": 15,
       "importance_weight": 16, "fwd_importance_weight": 17, "rev_importance_weight": 18, "cum_importance_weight": 19, "traj_importance_weight": 20}
NCOLS = 21
ACTION_DISCRETE, ACTION_CONTINUOUS = 0, 1
ENV = {"cartpole": 0, "pendulum": 1, "gridworld": 2, "synth": 3, "synth_discrete": 4}
HEAD = {"categorical": 0, "gaussian": 1, "greedy_q": 2, "deterministic": 3}
LOSS = {"ppo": 0, "value_mse": 1, "a2c": 3, "reinforce": 4, "logpdf_bc": 5, "mse_action": 6, "lagrange_ppo": 7}
INFO = {"loss": 0, "grad_norm": 1, "entropy": 2, "kl": 3, "clip_fraction": 4, "avg_advantage": 5, "avg_return": 6,
        "batches_trained": 7, "epochs_run": 8, "q1avg": 9, "q2avg": 10, "alpha": 11, "penalty": 12, "cur_cost": 13, "cost_loss": 14, "p_loss": 15}
INFO_N = 16
PROF = {"rollout": 0, "values": 1, "gae": 2, "whiten": 3, "train_actor": 4, "train_critic": 5, "per_scan": 6,
        "per_search": 7, "gather": 8, "td_step": 9, "tiny_solve": 10}


class CruxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cruxhip error %d: %s" % (code, msg))
        self.code = code
