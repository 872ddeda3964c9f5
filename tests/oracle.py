"""ctypes binding of oracle/libcruxoracle.so -- TEST INFRASTRUCTURE ONLY (the CPU restatement of the reference)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crux_jl_amd as crux  # noqa: E402
from crux_jl_amd import _lib as L  # noqa: E402

ORACLE_PATH = os.path.join(ROOT, "oracle", "libcruxoracle.so")
i32, i64, u32, u64, f32, f64, vp = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_void_p
P = C.POINTER

_SIG = {
    "orc_mlp_create": (vp, [i32, P(i32), P(i32), i32]), "orc_mlp_destroy": (None, [vp]), "orc_mlp_n_params": (i64, [vp]),
    "orc_mlp_params": (P(f32), [vp]), "orc_mlp_grads": (P(f32), [vp]), "orc_mlp_init_glorot": (i32, [vp, u64, u32, f32]),
    "orc_mlp_forward": (i32, [vp, vp, i64, vp]), "orc_mlp_copy": (i32, [vp, vp]), "orc_polyak": (i32, [vp, vp, f32]),
    "orc_adam_init": (i32, [vp, f64, f64, f64, f64]), "orc_adam_get_state": (i32, [vp, vp, vp, vp]), "orc_adam_set_state": (i32, [vp, vp, vp, vp]), "orc_adam_apply": (i32, [vp, f32]),
    "orc_buffer_create": (vp, [i32, i32, i32, i64, u32, i32, f32]), "orc_buffer_destroy": (None, [vp]), "orc_buffer_len": (i64, [vp]),
    "orc_buffer_capacity": (i64, [vp]), "orc_buffer_next_ind": (i64, [vp]), "orc_buffer_total_count": (i64, [vp]),
    "orc_buffer_has_column": (i32, [vp, i32]), "orc_buffer_clear": (i32, [vp]), "orc_buffer_column_info": (i32, [vp, i32, P(i32), P(i32)]),
    "orc_buffer_column": (vp, [vp, i32]), "orc_buffer_push_host": (i32, [vp, i64, P(vp), vp]), "orc_buffer_push_buffer": (i32, [vp, vp, vp, i64, vp]),
    "orc_buffer_permute": (i32, [vp, vp]), "orc_buffer_last_n_indices": (i64, [vp, i64, vp]), "orc_buffer_gather_host": (i32, [vp, vp, i64, P(vp)]),
    "orc_buffer_indices": (i32, [vp, vp, i64]), "orc_buffer_episodes": (i64, [vp, vp, vp, i64]), "orc_split_batches": (None, [i64, vp, i32, vp]),
    "orc_circ_inds": (None, [i64, i64, i64, vp]), "orc_per_update": (i32, [vp, vp, vp, i32, i64]),
    "orc_per_sample": (i32, [vp, vp, i64, vp, f32, u64, u64]), "orc_uniform_sample": (i32, [vp, vp, i64, vp, u64, u64]),
    "orc_per_get": (i32, [vp, vp, P(f32), P(f32), vp]), "orc_buffer_set_sample_stream": (i32, [vp, u32]), "orc_pairwise_cumsum_f32": (None, [vp, i64, vp]),
    "orc_env_create": (vp, [i32, i32, i32, f32, vp, vp, u64, i32, i32]), "orc_env_destroy": (None, [vp]), "orc_env_obs_dim": (i32, [vp]),
    "orc_env_act_dim": (i32, [vp]), "orc_env_state_dim": (i32, [vp]), "orc_env_reset": (i32, [vp]), "orc_env_get_state": (i32, [vp, vp, vp, vp]),
    "orc_rollout": (i32, [vp, vp, P(L.RolloutCfg), vp, i64, P(f64), P(i64)]), "orc_policy_explore": (i32, [vp, P(L.RolloutCfg), i32, vp, u64, vp, vp, vp]), "orc_env_step_host": (i32, [i32, i64, vp, vp, vp, vp, vp, vp, vp]),
    "orc_fill_gae": (i32, [vp, vp, f32, f32]), "orc_fill_returns": (i32, [vp, f32]), "orc_importance_weight": (i32, [vp, vp, i32]), "orc_fill_importance_weights": (i32, [vp]), "orc_whiten": (i32, [vp, i32]),
    "orc_fill_gae_keys": (i32, [vp, vp, f32, f32, i32, i32]), "orc_fill_returns_keys": (i32, [vp, f32, i32, i32]),
    "orc_batch_train_lagrange": (i32, [vp, vp, P(L.TrainCfg), P(L.Lagrange), vp, vp, vp]),
    "orc_gae_range": (None, [vp, vp, vp, vp, i64, i64, f32, f32, vp]), "orc_returns_range": (None, [vp, i64, i64, f32, vp]),
    "orc_batch_train": (i32, [vp, vp, P(L.TrainCfg), vp, vp, vp]), "orc_train_step": (i32, [vp, vp, P(L.TrainCfg), vp, i64, vp]),
    "orc_loss_grad": (i32, [vp, vp, P(L.TrainCfg), vp, i64, vp]), "orc_first_episode_metrics": (i32, [vp, i32, i64, f32, vp, vp, vp, vp]), "orc_dqn_target": (i32, [vp, vp, f32, vp]), "orc_softq_target": (i32, [vp, vp, f32, f32, vp]), "orc_td_error": (i32, [vp, vp, vp, vp]),
    "orc_td_step": (i32, [vp, vp, vp, i32, vp]),
    "orc_sac_target": (i32, [vp, vp, vp, vp, vp, f32, u64, u64, vp]), "orc_sac_temp_step": (i32, [vp, vp, vp, f32, u64, u64, vp]),
    "orc_double_q_step": (i32, [vp, vp, vp, vp, i32, vp]), "orc_sac_actor_step": (i32, [vp, vp, vp, vp, vp, u64, u64, vp]),
    "orc_dpg_target": (i32, [vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, u64, u64, vp]), "orc_q_step": (i32, [vp, vp, vp, i32, vp]), "orc_dpg_actor_step": (i32, [vp, vp, vp, vp]),
    "orc_mlp_set_squash": (i32, [vp, f32]),
    "orc_buffer_push_reservoir": (i32, [vp, i64, P(vp), i32, u64, u64]),
    "orc_gail_d_step": (i32, [vp, vp, i64, i64, vp, i64, i64, vp]), "orc_gail_reward": (i32, [vp, vp, f32, f32, vp]),
    "orc_linear_decay": (f64, [f64, f64, i64, i64]),
    "orc_jl_sum_f32": (f32, [vp, i64]), "orc_jl_mean_f32": (f32, [vp, i64]), "orc_jl_std_f32": (f32, [vp, i64]),
    "orc_perm": (None, [u64, u64, u32, vp]), "orc_philox": (None, [u64, u64, u32, u32, vp]),
}

_lib = None
_libs = {}
OMP_PATH = os.path.join(ROOT, "oracle", "libcruxoracle_omp.so")      # the same restatement built with -fopenmp (bench.py's all-cores CPU baseline only)


def _load(path):
    l = C.CDLL(path)
    for name, (res, args) in _SIG.items():
        fn = getattr(l, name); fn.restype = res; fn.argtypes = args
    return l


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_PATH):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        _lib = _libs["scalar"] = _load(ORACLE_PATH)
    return _lib


def have(which):
    return which == "scalar" or (which == "omp" and os.path.exists(OMP_PATH))


def select(which):
    """switch the library the wrappers below bind to ("scalar" = the parity oracle, "omp" = its OpenMP build); objects made before a switch stay with their library's handles and must not be mixed."""
    global _lib
    lib()
    if which not in _libs:
        _libs[which] = _load(OMP_PATH); _libs[which].orc_omp_threads.restype = C.c_int32
    _lib = _libs[which]


def vpz(a):
    return a.ctypes.data_as(vp) if a is not None else None


def chk(rc):
    if rc != 0:
        raise L.CruxError(rc, "oracle")
    return rc


class OMlp:
    def __init__(self, dims, acts, n_extra=0):
        self.dims, self.acts, self.n_extra = list(dims), [L.ACT[a] if isinstance(a, str) else a for a in acts], n_extra
        self.h = lib().orc_mlp_create(len(acts), (i32 * len(dims))(*dims), (i32 * len(acts))(*self.acts), n_extra)
        self.n = int(lib().orc_mlp_n_params(self.h))

    @property
    def params(self):
        return np.ctypeslib.as_array(lib().orc_mlp_params(self.h), shape=(self.n,))

    @property
    def grads(self):
        return np.ctypeslib.as_array(lib().orc_mlp_grads(self.h), shape=(self.n,))

    def init_glorot(self, seed, stream=0, extra_init=0.0):
        chk(lib().orc_mlp_init_glorot(self.h, seed, stream, extra_init)); return self

    def forward(self, x):
        x = np.asfortranarray(np.asarray(x, np.float32)); B = x.shape[1]
        y = np.empty((self.dims[-1], B), np.float32, order="F")
        chk(lib().orc_mlp_forward(self.h, vpz(x), B, vpz(y))); return y

    def adam_init(self, eta, b1=0.9, b2=0.999, eps=1e-8):
        chk(lib().orc_adam_init(self.h, eta, b1, b2, eps)); return self

    def adam_state(self):
        m, v, bp = np.empty(self.n, np.float32), np.empty(self.n, np.float32), np.empty(2, np.float64)
        chk(lib().orc_adam_get_state(self.h, vpz(m), vpz(v), vpz(bp))); return m, v, bp

    def set_adam_state(self, m, v):
        m, v = np.ascontiguousarray(m, np.float32), np.ascontiguousarray(v, np.float32)
        chk(lib().orc_adam_set_state(self.h, vpz(m), vpz(v), None)); return self

    def __del__(self):
        try:
            lib().orc_mlp_destroy(self.h)
        except Exception:
            pass


class OBuffer:
    def __init__(self, obs_dim, act_dim, act_kind, capacity, extras=(), prioritized=False, alpha=0.6):
        mask = 0
        for k in extras:
            mask |= 1 << L.COL[k]
        self.obs_dim, self.act_dim, self.act_kind = obs_dim, act_dim, act_kind
        self.h = lib().orc_buffer_create(obs_dim, act_dim, act_kind, capacity, mask, 1 if prioritized else 0, alpha)
        self.cap = capacity

    def __len__(self):
        return int(lib().orc_buffer_len(self.h))

    def haskey(self, k):
        return bool(lib().orc_buffer_has_column(self.h, L.COL[k]))

    def keys(self):
        return [k for k in L.COL if self.haskey(k)]

    def col(self, k, n=None):
        """numpy view (features, capacity) of the oracle's column memory."""
        from crux_jl_amd.api import _np_dtype
        rows = self.obs_dim if k in ("s", "sp") else self.act_dim if k == "a" else 1
        dt = np.dtype(_np_dtype(k, self.act_kind))
        ptr = lib().orc_buffer_column(self.h, L.COL[k])
        buf = (C.c_char * (dt.itemsize * rows * self.cap)).from_address(ptr)
        a = np.frombuffer(buf, dtype=dt).reshape((rows, self.cap), order="F")
        return a if n is None else a[:, :n]

    def __getitem__(self, k):
        return self.col(k, len(self)).copy(order="F")

    def push(self, data):
        N = np.asarray(next(iter(data.values()))).shape[-1]
        from crux_jl_amd.api import _np_dtype
        cols = (vp * L.NCOLS)(); keep = []
        for k, v in data.items():
            if k in L.COL and self.haskey(k):
                rows = self.obs_dim if k in ("s", "sp") else self.act_dim if k == "a" else 1
                a = np.asfortranarray(np.asarray(v).reshape(rows, N).astype(_np_dtype(k, self.act_kind))); keep.append(a); cols[L.COL[k]] = a.ctypes.data
        I = np.empty(N, np.int64)
        chk(lib().orc_buffer_push_host(self.h, N, cols, vpz(I))); return I + 1

    def push_reservoir(self, data, weighted, seed, counter):
        N = np.asarray(next(iter(data.values()))).shape[-1]
        from crux_jl_amd.api import _np_dtype
        cols = (vp * L.NCOLS)(); keep = []
        for k, v in data.items():
            if k in L.COL and self.haskey(k):
                rows = self.obs_dim if k in ("s", "sp") else self.act_dim if k == "a" else 1
                a = np.asfortranarray(np.asarray(v).reshape(rows, N).astype(_np_dtype(k, self.act_kind))); keep.append(a); cols[L.COL[k]] = a.ctypes.data
        chk(lib().orc_buffer_push_reservoir(self.h, N, cols, 1 if weighted else 0, seed, counter))

    def push_buffer(self, src, ids=None):
        ids0 = None if ids is None else np.ascontiguousarray(np.asarray(ids, np.int64) - 1)
        n = len(src) if ids is None else ids0.size
        I = np.empty(n, np.int64)
        chk(lib().orc_buffer_push_buffer(self.h, src.h, vpz(ids0), n, vpz(I))); return I + 1

    def permute(self, perm1):
        p = np.ascontiguousarray(np.asarray(perm1, np.int64) - 1); chk(lib().orc_buffer_permute(self.h, vpz(p)))

    def __del__(self):
        try:
            lib().orc_buffer_destroy(self.h)
        except Exception:
            pass


class OEnv:
    def __init__(self, kind, n_envs, max_steps, gamma=0.99, seed=0, mu=None, sigma=None, so=0, sa=0):
        self.h = lib().orc_env_create(L.ENV[kind], n_envs, max_steps, gamma, vpz(mu), vpz(sigma), seed, so, sa)
        assert self.h, "oracle env kind unsupported"
        self.n_envs = n_envs

    def state(self):
        sd = lib().orc_env_state_dim(self.h); E = self.n_envs
        st, el, nr = np.empty((sd, E), np.float64, order="F"), np.empty(E, np.int64), np.empty(E, np.int64)
        chk(lib().orc_env_get_state(self.h, vpz(st), vpz(el), vpz(nr))); return st, el, nr

    def rollout(self, pol, cfg, buf, T):
        sr, ne = f64(), i64()
        chk(lib().orc_rollout(self.h, pol.h, C.byref(cfg), buf.h, T, C.byref(sr), C.byref(ne))); return sr.value, ne.value

    def __del__(self):
        try:
            lib().orc_env_destroy(self.h)
        except Exception:
            pass
