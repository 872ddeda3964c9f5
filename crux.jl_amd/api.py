"""Host-side mirror of the Crux.jl interface for the actor-learner hot path, over the C ABI of libcruxhip.so.

Julia is not installed where this is built, so the host layer that `north_star` asks to keep in Julia is mirrored here
in Python with the reference's names and argument meaning (Julia's `f!` is spelled `f_`). The Julia shim that binds the
same C symbols with `ccall` is shown in INTEGRATION.md. Every class/function cites the reference definition it mirrors
(paths under sisl/Crux.jl v0.1.4).

Array convention: like the Julia arrays, columns are (features, batch) with the batch LAST; numpy arrays returned here
are Fortran-ordered so their memory is identical to the Julia array's (and to what crosses the C ABI).
"""
import ctypes as C
import math
import numpy as np

from . import _lib as L

# --------------------------------------------------------------------------------------------------------------
# context
# --------------------------------------------------------------------------------------------------------------
_default_ctx = None


class Context:
    """One HIP stream + error slot (crux_ctx). `stream` may be a raw hipStream_t (e.g. torch's current stream)."""

    def __init__(self, device=0, stream=None):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.crux_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise L.CruxError(rc, "no usable MI355X/HIP device %d (libcruxhip has no CPU fallback)" % device)
        self.h = h
        self.device = device

    def check(self, rc):
        if rc != 0:
            raise L.CruxError(rc, (self.lib.crux_last_error(self.h) or b"").decode())
        return rc

    # ---- replica group (RCCL over xGMI, one process per GPU; cruxhip.h "multi-GPU") -------------------------------------------
    def set_learner_cus(self, cus):
        """0 = automatic, 1 = one CU per small-MLP learner, 2 = two CUs of one XCD per learner (cruxhip.h)."""
        self.check(self.lib.crux_ctx_set_learner_cus(self.h, int(cus)))

    def comm_unique_id(self):
        """128-byte RCCL id; rank 0 creates it and ships it to the other ranks (torch.distributed.broadcast, a file, MPI ...)."""
        b = np.zeros(128, np.uint8); self.check(self.lib.crux_comm_unique_id(self.h, _vp(b))); return b

    def comm_init(self, rank, nranks, uid):
        uid = np.ascontiguousarray(np.asarray(uid, np.uint8)); assert uid.size == 128
        self.check(self.lib.crux_comm_init(self.h, int(rank), int(nranks), _vp(uid)))

    def comm_destroy(self):
        self.check(self.lib.crux_comm_destroy(self.h))

    def comm_size(self):
        return int(self.lib.crux_comm_size(self.h))

    # ---- replica group with direct peer slots: in-kernel SUM all-reduce of every minibatch gradient over xGMI (cruxhip.h) ----------------
    def peer_export(self):
        """64-byte IPC handle of this context's peer region; every rank ships its handle to every other rank."""
        b = np.zeros(64, np.uint8); self.check(self.lib.crux_peer_export(self.h, _vp(b))); return b

    def peer_attach(self, rank, nranks, handles):
        """handles: (nranks, 64) uint8, row r = rank r's peer_export(). All ranks must have attached before any of them trains."""
        h = np.ascontiguousarray(np.asarray(handles, np.uint8).reshape(int(nranks), 64))
        self.check(self.lib.crux_peer_attach(self.h, int(rank), int(nranks), _vp(h)))

    def peer_detach(self):
        self.check(self.lib.crux_peer_detach(self.h))

    def peer_size(self):
        return int(self.lib.crux_peer_size(self.h))

    def peer_set_sync_every(self, k):
        """k = 1: gradient exchange every minibatch (the exact form); k > 1: local Adam steps, theta / m / v averaged in the learner kernel after every k-th (cruxhip.h: crux_peer_set_sync_every)"""
        self.check(self.lib.crux_peer_set_sync_every(self.h, int(k)))

    def peer_sync_every(self):
        return int(self.lib.crux_peer_sync_every(self.h))

    def peer_hist_enable(self, on=True):
        """record, per learner workgroup, how long every in-kernel exchange waited for the slowest peer's flag (cruxhip.h: crux_peer_hist_enable)"""
        self.check(self.lib.crux_peer_hist_enable(self.h, 1 if on else 0))

    def peer_wait_hist(self, reset=True):
        """uint32 [2 learner streams][2 workgroups][32]: log2 bins of the flag waits in 10 ns ticks"""
        out = np.zeros((2, 2, 32), np.uint32); self.check(self.lib.crux_peer_wait_hist(self.h, _vp(out), 1 if reset else 0)); return out

    def sync(self):
        self.check(self.lib.crux_sync(self.h))

    def prof_enable(self, on=True):
        self.check(self.lib.crux_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self.check(self.lib.crux_prof_reset(self.h))

    def prof_get(self, slot):
        ms, n = C.c_double(), C.c_int64()
        self.check(self.lib.crux_prof_get(self.h, L.PROF[slot] if isinstance(slot, str) else slot, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.crux_device_alloc(self.h, int(nbytes), C.byref(p)))
        return p

    def free(self, p):
        self.lib.crux_device_free(self.h, p)

    def d2h(self, d_ptr, arr):
        self.check(self.lib.crux_memcpy_d2h(self.h, arr.ctypes.data_as(C.c_void_p), d_ptr, arr.nbytes))
        return arr

    def h2d(self, d_ptr, arr):
        if not (arr.flags.c_contiguous or arr.flags.f_contiguous):
            arr = np.ascontiguousarray(arr)          # raw memory is copied: column-major (Julia layout) arrays go up as they are
        self.check(self.lib.crux_memcpy_h2d(self.h, d_ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def close(self):
        if self.h:
            self.lib.crux_ctx_destroy(self.h)
            self.h = None


def reload_switches():
    """re-read the CRUX_* environment switches (the library reads them when a context is created; cruxhip.h: crux_reload_switches)"""
    L.load().crux_reload_switches()


def peer_attach_local(contexts):
    """Wire the contexts of ONE process into a replica group (contexts[r] = rank r): a multi-GPU single-process host, or replicas sharing a device."""
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    rc = contexts[0].lib.crux_peer_attach_local(arr, len(contexts))
    if rc != 0:      # the library records the reason on the context that failed, not necessarily the first one
        msgs = [(c.lib.crux_last_error(c.h) or b"").decode() for c in contexts]
        hit = [m for m in msgs if "peer_attach" in m or "hardware queue" in m] or msgs
        raise L.CruxError(rc, hit[-1])


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# --------------------------------------------------------------------------------------------------------------
# spaces  (src/spaces.jl:1-43)
# --------------------------------------------------------------------------------------------------------------
class DiscreteSpace:
    """DiscreteSpace(N, vals) -- actions stored as Bool one-hot columns (src/spaces.jl:2-8,18,24)."""

    def __init__(self, N, vals=None):
        if not isinstance(N, (int, np.integer)):
            vals = list(N); N = len(vals)
        self.N = int(N)
        self.vals = list(range(1, self.N + 1)) if vals is None else list(vals)


class ContinuousSpace:
    """ContinuousSpace(dims, type; mu, sigma) (src/spaces.jl:10-16); tovec whitens with (v-mu)/sigma (:25)."""

    def __init__(self, dims, type=np.float32, mu=0.0, sigma=1.0):
        self.dims = (int(dims),) if isinstance(dims, (int, np.integer)) else tuple(int(d) for d in dims)
        self.type = type
        self.mu, self.sigma = mu, sigma


def dim(S):
    return (S.N,) if isinstance(S, DiscreteSpace) else S.dims


# --------------------------------------------------------------------------------------------------------------
# networks  (src/policies.jl:68-157, 246-276, 315-350)
# --------------------------------------------------------------------------------------------------------------
class Dense:
    """Flux.Dense(in => out, act)."""

    def __init__(self, inp, out, act="identity"):
        self.inp, self.out, self.act = int(inp), int(out), act if isinstance(act, str) else getattr(act, "__name__", "identity")


class Chain:
    """Flux.Chain(Dense...)."""

    def __init__(self, *layers):
        self.layers = list(layers)
        for a, b in zip(self.layers[:-1], self.layers[1:]):
            if a.out != b.inp:
                raise ValueError("Chain: layer widths do not match (%d -> %d)" % (a.out, b.inp))

    @property
    def dims(self):
        return [self.layers[0].inp] + [l.out for l in self.layers]

    @property
    def acts(self):
        return [L.ACT[l.act] for l in self.layers]


class NetworkPolicy:
    """Device-resident Chain(Dense...) + optional trailing trainables; the crux_mlp handle."""

    def __init__(self, network, n_extra=0, extra_init=0.0, ctx=None, seed=0, stream=0):
        self.ctx = ctx or default_context()
        self.network = network
        self.n_extra = int(n_extra)
        dims = (C.c_int32 * len(network.dims))(*network.dims)
        acts = (C.c_int32 * len(network.acts))(*network.acts)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.crux_mlp_create(self.ctx.h, len(network.layers), dims, acts, self.n_extra, C.byref(h)))
        self.h = h
        self.ctx.check(self.ctx.lib.crux_mlp_init_glorot(self.h, int(seed), int(stream), float(extra_init)))
        self.optimizer = None

    # Flux.params(pi) as one flat Float32 vector in Flux order (W1,b1,W2,b2,...,extras)
    @property
    def n_params(self):
        return int(self.ctx.lib.crux_mlp_n_params(self.h))

    def get_params(self):
        out = np.empty(self.n_params, np.float32)
        self.ctx.check(self.ctx.lib.crux_mlp_get_params(self.h, _vp(out), out.size))
        return out

    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        self.ctx.check(self.ctx.lib.crux_mlp_set_params(self.h, _vp(flat), flat.size))

    def params(self):
        """List of arrays like Flux.params: W (out,in) column-major, b (out,), ..., extras."""
        flat, out, off = self.get_params(), [], 0
        d = self.network.dims
        for l in range(len(d) - 1):
            n = d[l + 1] * d[l]
            out.append(flat[off:off + n].reshape((d[l + 1], d[l]), order="F")); off += n
            out.append(flat[off:off + d[l + 1]].copy()); off += d[l + 1]
        if self.n_extra:
            out.append(flat[off:off + self.n_extra].copy())
        return out

    def forward(self, s):
        """value(pi, s) for ContinuousNetwork / raw logits for DiscreteNetwork (src/policies.jl:94,120)."""
        s = np.asarray(s, np.float32)
        d_in, d_out = self.network.dims[0], self.network.dims[-1]
        if s.ndim == 1:
            s = s.reshape(d_in, 1)
        if s.shape[0] != d_in:
            raise ValueError("value: input has %d rows, network expects %d" % (s.shape[0], d_in))
        B = s.shape[1]
        x = np.asfortranarray(s)
        y = np.empty((d_out, B), np.float32, order="F")
        self.ctx.check(self.ctx.lib.crux_mlp_forward_host(self.h, _vp(x), B, _vp(y)))
        return y

    def attach_optimizer(self, opt):
        self.optimizer = opt
        self.ctx.check(self.ctx.lib.crux_adam_init(self.h, opt.eta, opt.beta[0], opt.beta[1], opt.epsilon))

    def adam_state(self):
        m, v, bp = np.empty(self.n_params, np.float32), np.empty(self.n_params, np.float32), np.empty(2, np.float64)
        self.ctx.check(self.ctx.lib.crux_adam_get_state(self.h, _vp(m), _vp(v), _vp(bp)))
        return m, v, bp

    def set_adam_state(self, m, v, beta_pow):
        """Load Adam's (m, v, [beta1^t, beta2^t]) -- the IdDict entry Flux keeps per parameter array (checkpoint restore, parity tests)."""
        m, v, bp = np.ascontiguousarray(m, np.float32), np.ascontiguousarray(v, np.float32), np.ascontiguousarray(beta_pow, np.float64)
        self.ctx.check(self.ctx.lib.crux_adam_set_state(self.h, _vp(m), _vp(v), _vp(bp)))

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.ctx.h:
                self.ctx.lib.crux_mlp_destroy(self.h)
        except Exception:
            pass


class ContinuousNetwork(NetworkPolicy):
    """ContinuousNetwork(network, output_dim) (src/policies.jl:68-98)."""
    head = "deterministic"

    def __init__(self, network, output_dim=None, **kw):
        super().__init__(network, **kw)
        self.output_dim = output_dim or network.dims[-1]


class DiscreteNetwork(NetworkPolicy):
    """DiscreteNetwork(network, outputs): softmax logit_conversion, categorical sampling (src/policies.jl:104-157)."""
    head = "categorical"

    def __init__(self, network, outputs, always_stochastic=False, **kw):
        super().__init__(network, **kw)
        self.outputs = list(outputs)
        self.always_stochastic = always_stochastic
        if len(self.outputs) != network.dims[-1]:
            raise ValueError("DiscreteNetwork: %d outputs for %d logits" % (len(self.outputs), network.dims[-1]))


class GaussianPolicy(NetworkPolicy):
    """GaussianPolicy(mu::ContinuousNetwork, logSigma::AbstractArray): constant trainable log-std (src/policies.jl:315-350)."""
    head = "gaussian"

    def __init__(self, mu_chain, logSigma, **kw):
        logSigma = np.asarray(logSigma, np.float32).reshape(-1)
        super().__init__(mu_chain, n_extra=logSigma.size, **kw)
        p = self.get_params(); p[-logSigma.size:] = logSigma; self.set_params(p)



class SquashedGaussianPolicy(GaussianPolicy):
    """SquashedGaussianPolicy(mu::ContinuousNetwork, logSigma::Array, ascale=1f0) (src/policies.jl:353-400): a = ascale*tanh(mu + sigma*eps),
    sigma = exp(clamp(logSigma, -5, 2)), logpdf with the tanh correction; the constant-logSigma form the reference's examples use
    (examples/rl/pendulum.jl:20, half_cheetah_mujoco.jl:42). Greedy action = ascale*tanh(mu(s)) (:372)."""

    def __init__(self, mu_chain, logSigma, ascale=1.0, **kw):
        super().__init__(mu_chain, logSigma, **kw)
        self.ascale = float(np.float32(ascale))
        self.ctx.check(self.ctx.lib.crux_mlp_set_squash(self.h, self.ascale))

class ParamVector(NetworkPolicy):
    """A bare trainable vector with its own optimiser state (ConstantLayer, src/utils.jl:31-36; P[:SAC_log_alpha], sac.jl:96):
    the crux_mlp handle with n_layers = 0."""
    head = None

    def __init__(self, values, ctx=None):
        values = np.asarray(values, np.float32).reshape(-1)
        self.ctx = ctx or default_context()
        self.network, self.n_extra, self.optimizer = None, values.size, None
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.crux_mlp_create(self.ctx.h, 0, None, None, values.size, C.byref(h)))
        self.h = h
        self.set_params(values)

    def params(self):
        return [self.get_params()]


class DoubleNetwork:
    """DoubleNetwork(N1, N2) (src/policies.jl:162-187): value(pi, s, a) = (value(N1, s, a), value(N2, s, a))."""

    def __init__(self, N1, N2):
        self.N1, self.N2 = N1, N2
        self.ctx = N1.ctx


class ActorCritic:
    """ActorCritic(A, C) (src/policies.jl:246-276): actor(pi)=A, critic(pi)=C, value(pi,s)=value(C,s)."""

    def __init__(self, A, C_):
        self.A, self.C = A, C_


def actor(pi):
    return pi.A if isinstance(pi, ActorCritic) else pi


def critic(pi):
    return pi.C if isinstance(pi, ActorCritic) else pi


def value(pi, s):
    """POMDPs.value(pi, s) (src/policies.jl:94,120,265)."""
    return critic(pi).forward(s)


def _leaves(pi):
    """layers(pi) flattened to the crux_mlp handles it is made of (src/policies.jl:171,255)."""
    if isinstance(pi, ActorCritic):
        return _leaves(pi.A) + _leaves(pi.C)
    if isinstance(pi, DoubleNetwork):
        return _leaves(pi.N1) + _leaves(pi.N2)
    return [pi]


def polyak_average_(to, frm, tau=1.0):
    """polyak_average!(to, from, tau) (src/policies.jl:48-59) over every layer of the (possibly composite) policy."""
    for t, f in zip(_leaves(to), _leaves(frm)):
        t.ctx.check(t.ctx.lib.crux_polyak(t.h, f.h, float(tau)))


def copyto_(to, frm):
    """Base.copyto!(to, from) on network parameters (src/policies.jl:61-65)."""
    for t, f in zip(_leaves(to), _leaves(frm)):
        t.ctx.check(t.ctx.lib.crux_mlp_copy(t.h, f.h))


def clone_policy(pi):
    """deepcopy(pi) for pi_minus (src/policies.jl:24-36): same architecture, copied parameters."""
    if isinstance(pi, ActorCritic):
        return ActorCritic(clone_policy(pi.A), clone_policy(pi.C))
    if isinstance(pi, DoubleNetwork):
        return DoubleNetwork(clone_policy(pi.N1), clone_policy(pi.N2))
    if isinstance(pi, SquashedGaussianPolicy):
        new = SquashedGaussianPolicy(pi.network, np.zeros(pi.n_extra, np.float32), pi.ascale, ctx=pi.ctx)
    elif isinstance(pi, GaussianPolicy):
        new = GaussianPolicy(pi.network, np.zeros(pi.n_extra, np.float32), ctx=pi.ctx)
    elif isinstance(pi, DiscreteNetwork):
        new = DiscreteNetwork(pi.network, pi.outputs, ctx=pi.ctx)
    else:
        new = ContinuousNetwork(pi.network, ctx=pi.ctx)
    copyto_(new, pi)
    return new


class PolicyParams:
    """PolicyParams(pi; space, pi_explore, pi_minus) (src/policies.jl:12-19)."""

    def __init__(self, pi, space=None, pi_explore=None, pi_minus=None, pa=None):
        self.pi, self.pi_explore, self.pi_minus = pi, pi_explore if pi_explore is not None else pi, pi_minus
        self.pa = pa                                   # nominal action policy (policies.jl:17): the reference distribution of :importance_weight (sampler.jl:108-111)
        a = actor(pi)
        self.space = space or (DiscreteSpace(len(a.outputs), a.outputs) if isinstance(a, DiscreteNetwork) else ContinuousSpace(a.network.dims[-1]))


class Adam:
    """Flux.Optimise.Adam(eta, beta, epsilon): Float64 fields; Adam(3f-4) stores Float64(3f-4)."""

    def __init__(self, eta=0.001, beta=(0.9, 0.999), epsilon=1e-8):
        self.eta = float(np.float32(eta)) if isinstance(eta, np.float32) else float(eta)
        self.beta, self.epsilon = (float(beta[0]), float(beta[1])), float(epsilon)


# --------------------------------------------------------------------------------------------------------------
# experience buffer (src/experience_buffer.jl)
# --------------------------------------------------------------------------------------------------------------
_F32_KEYS = ["return", "logprob", "advantage", "value", "cost", "cost_advantage", "cost_return",
             "importance_weight", "fwd_importance_weight", "rev_importance_weight", "cum_importance_weight", "traj_importance_weight"]      # the last five start at 1 (experience_buffer.jl:17-19)


def _np_dtype(key, act_kind):
    if key in ("s", "sp", "r", "weight") or key in _F32_KEYS:
        return np.float32
    if key == "a":
        return np.bool_ if act_kind == L.ACTION_DISCRETE else np.float32
    if key in ("done", "episode_end"):
        return np.bool_
    return np.int64


def mdp_data(S, A, capacity, extras=()):
    """mdp_data(S, A, capacity, extras) (src/experience_buffer.jl:4-35): host Dict of zero (weight: one) columns."""
    od, ad = int(np.prod(dim(S))), int(np.prod(dim(A)))
    kind = L.ACTION_DISCRETE if isinstance(A, DiscreteSpace) else L.ACTION_CONTINUOUS
    d = {"s": np.zeros((od, capacity), np.float32, order="F"), "a": np.zeros((ad, capacity), _np_dtype("a", kind), order="F"),
         "sp": np.zeros((od, capacity), np.float32, order="F"), "r": np.zeros((1, capacity), np.float32, order="F"),
         "done": np.zeros((1, capacity), np.bool_, order="F"), "episode_end": np.zeros((1, capacity), np.bool_, order="F")}
    for k in extras:
        if k in _F32_KEYS and not k.endswith("importance_weight"):
            d[k] = np.zeros((1, capacity), np.float32, order="F")
        elif k == "weight" or k.endswith("importance_weight"):      # :17-19 fill(one(R), 1, capacity)
            d[k] = np.ones((1, capacity), np.float32, order="F")
        elif k in ("t", "i"):
            d[k] = np.zeros((1, capacity), np.int64, order="F")
        else:
            raise KeyError("Unrecognized key: %s" % k)
    return d


class ExperienceBuffer:
    """ExperienceBuffer(S, A, capacity, extras; prioritized, priority_params) (src/experience_buffer.jl:53-80).

    Columns live in HBM as separate arrays (SoA across keys, one transition's features contiguous)."""

    def __init__(self, S, A, capacity, extras=(), prioritized=False, priority_params=None, ctx=None):
        self.ctx = ctx or default_context()
        self.S, self.A = S, A
        self.obs_dim, self.act_dim = int(np.prod(dim(S))), int(np.prod(dim(A)))
        self.act_kind = L.ACTION_DISCRETE if isinstance(A, DiscreteSpace) else L.ACTION_CONTINUOUS
        mask = 0
        for k in extras:
            mask |= 1 << L.COL[k]
        pp = priority_params or {}
        self.alpha = float(pp.get("alpha", 0.6))
        self.beta = pp.get("beta", lambda i: 0.5)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.crux_buffer_create(self.ctx.h, self.obs_dim, self.act_dim, self.act_kind, int(capacity), mask,
                                                       1 if prioritized else 0, self.alpha, C.byref(h)))
        self.h = h
        self.prioritized = bool(prioritized)

    # Base functions (:176-192)
    def __len__(self):
        return int(self.ctx.lib.crux_buffer_len(self.h))

    @property
    def capacity(self):
        return int(self.ctx.lib.crux_buffer_capacity(self.h))

    @property
    def next_ind(self):
        """1-based like the reference field."""
        return int(self.ctx.lib.crux_buffer_next_ind(self.h)) + 1

    @property
    def total_count(self):
        return int(self.ctx.lib.crux_buffer_total_count(self.h))

    def haskey(self, k):
        return k in L.COL and bool(self.ctx.lib.crux_buffer_has_column(self.h, L.COL[k]))

    def keys(self):
        return [k for k in L.COL if self.haskey(k)]

    def _shape(self, k, n):
        rows = self.obs_dim if k in ("s", "sp") else self.act_dim if k == "a" else 1
        return (rows, n)

    def __getitem__(self, k):
        """b[key] = view of the first length(b) columns (:176); returned as a host copy."""
        n = len(self)
        out = np.empty(self._shape(k, n), _np_dtype(k, self.act_kind), order="F")
        self.ctx.check(self.ctx.lib.crux_buffer_read_column(self.h, L.COL[k], _vp(out), n))
        return out

    def __setitem__(self, k, v):
        """b[key] .= v."""
        n = len(self)
        v = np.asfortranarray(np.broadcast_to(np.asarray(v, _np_dtype(k, self.act_kind)), self._shape(k, n)))
        self.ctx.check(self.ctx.lib.crux_buffer_write_column(self.h, L.COL[k], _vp(v), n))

    def column_ptr(self, k):
        p = C.c_void_p()
        self.ctx.check(self.ctx.lib.crux_buffer_column_ptr(self.h, L.COL[k], C.byref(p)))
        return p.value

    def isprioritized(self):
        return self.prioritized

    def clear_(self):
        self.ctx.check(self.ctx.lib.crux_buffer_clear(self.h)); return self

    def push_(self, data, ids=None):
        """push!(b, data; ids) (:232-259). `data` is a dict of (features, N) arrays or another ExperienceBuffer.
        Returns the (1-based) destination indices I like the reference."""
        if isinstance(data, ExperienceBuffer):
            if ids is None:
                ids0 = None; n = len(data)
            else:
                ids0 = np.ascontiguousarray(np.asarray(ids, np.int64) - 1); n = ids0.size
            I = np.empty(n, np.int64)
            self.ctx.check(self.ctx.lib.crux_buffer_push_buffer(self.h, data.h, _vp(ids0), n, _vp(I)))
            return I + 1
        first = next(iter(data.values()))
        N = np.asarray(first).shape[-1]
        cols = (C.c_void_p * L.NCOLS)()
        keep = []
        for k, v in data.items():
            if k not in L.COL or not self.haskey(k):
                continue
            arr = np.asarray(v)
            arr = arr.reshape(self._shape(k, N)) if arr.ndim == 1 else arr
            if arr.shape[:-1] != self._shape(k, N)[:-1]:
                raise L.CruxError(L.EINVAL, "push!: column :%s has shape %s, buffer expects %s (@assert size(v1)[1:end-1] == size(v2)[1:end-1])" % (k, arr.shape, self._shape(k, N)))
            arr = np.asfortranarray(arr.astype(_np_dtype(k, self.act_kind)))
            if ids is not None:
                arr = np.asfortranarray(arr[:, np.asarray(ids) - 1])
            keep.append(arr); cols[L.COL[k]] = arr.ctypes.data
        if ids is not None:
            N = len(ids)
        I = np.empty(N, np.int64)
        self.ctx.check(self.ctx.lib.crux_buffer_push_host(self.h, N, cols, _vp(I)))
        return I + 1

    def push_reservoir_(self, data, weighted=False, seed=0, counter=0):
        """push_reservoir!(buffer, data; weighted) (:262-288) for a dict of (features, N) host arrays; row i draws Philox(seed, counter + i) (crux_rng.h)."""
        first = next(iter(data.values())); N = np.asarray(first).shape[-1]
        cols = (C.c_void_p * L.NCOLS)(); keep = []
        for k, v in data.items():
            if k not in L.COL or not self.haskey(k):
                continue
            arr = np.asarray(v); arr = arr.reshape(self._shape(k, N)) if arr.ndim == 1 else arr
            if arr.shape[:-1] != self._shape(k, N)[:-1]:
                raise L.CruxError(L.EINVAL, "push_reservoir!: column :%s has shape %s, buffer expects %s" % (k, arr.shape, self._shape(k, N)))
            arr = np.asfortranarray(arr.astype(_np_dtype(k, self.act_kind))); keep.append(arr); cols[L.COL[k]] = arr.ctypes.data
        self.ctx.check(self.ctx.lib.crux_buffer_push_reservoir(self.h, N, cols, 1 if weighted else 0, int(seed), int(counter)))
        return self

    def shuffle_(self, perm):
        """shuffle!(b) with an explicit 1-based permutation (:118-124)."""
        p = np.ascontiguousarray(np.asarray(perm, np.int64) - 1)
        self.ctx.check(self.ctx.lib.crux_buffer_permute(self.h, _vp(p))); return self

    def minibatch(self, indices):
        """minibatch_copy(b, indices) (:171) with 1-based indices -> dict of host arrays."""
        ids = np.ascontiguousarray(np.asarray(indices, np.int64) - 1)
        outs = (C.c_void_p * L.NCOLS)(); res = {}
        for k in self.keys():
            res[k] = np.empty(self._shape(k, ids.size), _np_dtype(k, self.act_kind), order="F"); outs[L.COL[k]] = res[k].ctypes.data
        self.ctx.check(self.ctx.lib.crux_buffer_gather_host(self.h, _vp(ids), ids.size, outs))
        return res

    def get_last_N_indices(self, N):
        """get_last_N_indices(b, N) (:223-229), 1-based."""
        out = np.empty(max(1, min(N, len(self))), np.int64)
        n = self.ctx.lib.crux_buffer_last_n_indices(self.h, int(N), _vp(out))
        return out[:n] + 1

    @property
    def indices(self):
        n = self.capacity; out = np.empty(n, np.int64)
        self.ctx.check(self.ctx.lib.crux_buffer_indices(self.h, _vp(out), n))
        return out

    def update_priorities_(self, I, v):
        """update_priorities!(b, I, v) (:290-301); I 1-based; v Float64 or Float32 array (dtype is significant)."""
        I0 = np.ascontiguousarray(np.asarray(I, np.int64) - 1)
        v = np.ascontiguousarray(v)
        is64 = v.dtype == np.float64
        if not is64:
            v = v.astype(np.float32)
        self.ctx.check(self.ctx.lib.crux_per_update(self.h, _vp(I0), _vp(v), 1 if is64 else 0, I0.size))

    def cumsum(self):
        out = np.empty(len(self), np.float32)
        self.ctx.check(self.ctx.lib.crux_per_get(self.h, None, None, None, _vp(out)))
        return out

    def priority_params(self):
        pr = np.empty(self.capacity, np.float32); mx, mn = C.c_float(), C.c_float()
        self.ctx.check(self.ctx.lib.crux_per_get(self.h, _vp(pr), C.byref(mx), C.byref(mn), None))
        return {"priorities": pr, "max_priority": mx.value, "min_priority": mn.value, "alpha": self.alpha}

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.ctx.h:
                self.ctx.lib.crux_buffer_destroy(self.h)
        except Exception:
            pass


def capacity(b):
    return b.capacity


SAMPLE_SEED = 0x5EED5A3F   # Philox key of the library's replay-sampling draws (fixed; the counter is the caller's `i`)


def set_sample_stream_(source, seed=SAMPLE_SEED, stream=0):
    """Philox key / stream of the draws that sample FROM `source` (cruxhip.h: crux_buffer_set_sample_stream)."""
    source.ctx.check(source.ctx.lib.crux_buffer_set_sample_stream(source.h, int(seed), int(stream)))
    source.sample_seed, source.sample_stream = int(seed), int(stream)


def uniform_sample_(target, source, B=None, ids=None, i=0):
    """uniform_sample!(target, source; B) (src/experience_buffer.jl:317-321). ids: optional explicit 1-based rows (else the Philox draw with counter i)."""
    B = B or target.capacity
    ids0 = None if ids is None else np.ascontiguousarray(np.asarray(ids, np.int64) - 1)
    target.ctx.check(target.ctx.lib.crux_uniform_sample(target.h, source.h, B, _vp(ids0), int(i)))
    return target.indices[:B] + 1


def prioritized_sample_(target, source, B=None, i=1, rands=None, counter=None):
    """prioritized_sample!(target, source; i, B) (src/experience_buffer.jl:324-349). `i` is the reference's keyword: the interaction count at which
    the importance-sampling exponent beta(i) is evaluated (:344,346). `counter` is the Philox counter of this draw (defaults to i; the reference
    advances Julia's global RNG instead). rands: optional B Float64 uniforms."""
    B = B or target.capacity
    r = None if rands is None else np.ascontiguousarray(rands, np.float64)
    beta = np.float32(source.beta(i))
    target.ctx.check(target.ctx.lib.crux_per_sample(target.h, source.h, B, _vp(r), float(beta), int(i if counter is None else counter)))
    return target.indices[:B] + 1


def rand_(target, *sources, i=1, fracs=None, counter=None, seed=None):
    """Random.rand!(target, sources...; i, fracs) (src/experience_buffer.jl:303-315). `i` goes to prioritized_sample! unchanged (beta(i), :312);
    `counter` numbers this call's draws (defaults to i) and `seed` keys them (defaults to each source's own key). With several sources, source k
    draws from Philox stream k, so the per-source samples are independent like the reference's successive rand calls."""
    fr = list(fracs) if fracs is not None else [1.0 / len(sources)] * len(sources)
    lens = [len(s) for s in sources]
    if any(l == 0 for l in lens):
        fr = [0.0 if l == 0 else f for f, l in zip(fr, lens)]; tot = sum(fr); fr = [f / tot for f in fr]
    batches = split_batches(target.capacity, fr)
    ctr = i if counter is None else counter
    for k, (b, B) in enumerate(zip(sources, batches)):
        if B == 0:
            continue
        had = (getattr(b, "sample_seed", SAMPLE_SEED), getattr(b, "sample_stream", 0))
        want = (int(seed) if seed is not None else had[0], k if len(sources) > 1 else had[1])
        if want != had:
            set_sample_stream_(b, *want)
        try:
            prioritized_sample_(target, b, B=B, i=i, counter=ctr) if b.isprioritized() else uniform_sample_(target, b, B=B, i=ctr)
        finally:
            if want != had:      # the re-keying is this call's only: a later single-source draw from b uses b's own (seed, stream) again
                set_sample_stream_(b, *had)


def split_batches(N, fracs):
    """split_batches(N, fracs) (src/experience_buffer.jl:126-131)."""
    if not isinstance(fracs, (list, tuple, np.ndarray)) or not math.isclose(sum(fracs), 1.0, rel_tol=1e-8):
        raise AssertionError("sum(fracs) must be 1")
    b = [int(math.floor(N * f)) for f in fracs]
    b[0] += N - sum(b)
    return b


# --------------------------------------------------------------------------------------------------------------
# sampler (src/sampler.jl)
# --------------------------------------------------------------------------------------------------------------
class GymMDP:
    """Stand-in for POMDPGym's GymPOMDP(:CartPole) etc.: names a dynamics kind that runs inside the rollout kernel.
    n_envs independent-seed copies are stepped together (env-major Vector{Sampler} semantics)."""

    def __init__(self, kind, n_envs=1, seed=0, discount=0.99, obs_dim=None, act_dim=None):
        self.kind, self.n_envs, self.seed, self.discount = kind, int(n_envs), int(seed), float(discount)
        if kind in ("synth", "synth_discrete"):      # the library's synthetic dynamics (include/cruxhip.h): any obs/act width up to 32
            self.obs_dim, self.act_dim, self.discrete = int(obs_dim), int(act_dim), kind == "synth_discrete"
        else:
            self.obs_dim, self.act_dim, self.discrete = {"cartpole": (4, 2, True), "pendulum": (3, 1, False), "gridworld": (2, 4, True)}[kind]

    def state_space(self, mu=0.0, sigma=1.0):
        """state_space(mdp; mu, sigma) (src/spaces.jl:34-43)."""
        return ContinuousSpace(self.obs_dim, np.float32, mu, sigma)

    def action_space(self):
        return DiscreteSpace(self.act_dim) if self.discrete else ContinuousSpace(self.act_dim)


def CartPoleMDP(**kw):
    return GymMDP("cartpole", **kw)


def PendulumMDP(**kw):
    return GymMDP("pendulum", **kw)


def SynthMDP(obs_dim, act_dim, discrete=False, **kw):
    """The library's synthetic environment for the LunarLander- (8 obs / 4 discrete actions) and HalfCheetah-shaped (17 obs / 6 continuous actions) configs."""
    return GymMDP("synth_discrete" if discrete else "synth", obs_dim=obs_dim, act_dim=act_dim, **kw)


def SimpleGridWorld(**kw):
    """POMDPModels.SimpleGridWorld(size=(10,10), tprob=.7) of the README example (discount 0.95)."""
    kw.setdefault("discount", 0.95)
    return GymMDP("gridworld", **kw)


def discount(mdp):
    return mdp.discount


class LinearDecaySchedule:
    """LinearDecaySchedule(start, stop, steps) (src/utils.jl:116-126)."""

    def __init__(self, start, stop, steps):
        self.start, self.stop, self.steps = float(start), float(stop), int(steps)

    def __call__(self, i):
        rate = (self.start - self.stop) / self.steps
        return max(self.stop, self.start - i * rate)


class MultitaskDecaySchedule:
    """MultitaskDecaySchedule(steps, task_ids; start=1.0, stop=0.1) (src/utils.jl:128-138): a LinearDecaySchedule restarted per task, continuing
    where the previous visit of the same task id stopped; before the first task -> start, after the last -> stop."""

    def __init__(self, steps, task_ids, start=1.0, stop=0.1):
        self.steps, self.task_ids, self.start, self.stop = int(steps), list(task_ids), float(start), float(stop)
        self.schedule = LinearDecaySchedule(start, stop, steps)

    def __call__(self, i):
        taskindex = -(-int(i) // self.steps)                 # ceil(Int, i / steps)
        if taskindex < 1:
            return self.start
        if taskindex > len(self.task_ids):
            return self.stop
        taskid = self.task_ids[taskindex - 1]
        used = self.steps * sum(1 for t in self.task_ids[:taskindex - 1] if t == taskid)
        return self.schedule(used + ((int(i) - 1) % self.steps) + 1)       # mod1(i, steps)


class EpsGreedyPolicy:
    """ϵGreedyPolicy(eps, actions) = MixedPolicy(eps, uniform random action) (src/policies.jl:466-494)."""

    def __init__(self, eps, actions):
        self.eps = eps if isinstance(eps, LinearDecaySchedule) else LinearDecaySchedule(eps, eps, 1)
        self.actions = list(actions)


class GaussianNoiseExplorationPolicy:
    """GaussianNoiseExplorationPolicy(sigma; a_min, a_max, eps_min, eps_max) (src/policies.jl:499-514)."""

    def __init__(self, sigma=0.01, a_min=-np.inf, a_max=np.inf, eps_min=-np.inf, eps_max=np.inf):
        self.sigma, self.a_min, self.a_max, self.eps_min, self.eps_max = float(sigma), float(a_min), float(a_max), float(eps_min), float(eps_max)


class Sampler:
    """Sampler(mdp, agent; max_steps, required_columns, lambda, S) (src/sampler.jl:1-29) for mdp.n_envs environments."""

    def __init__(self, mdp, agent, S=None, max_steps=100, required_columns=(), lam=float("nan"), ctx=None, Vc=None, traj_weight_fn=None):
        self.ctx = ctx or default_context()
        self.mdp = mdp
        self.traj_weight_fn = traj_weight_fn     # weight of a trajectory (Sampler.traj_weight_fn, src/sampler.jl:21): (agent, data, ep) -> the :traj_importance_weight of the episode's rows (:62)
        self.Vc = Vc                             # cost value network (Sampler.Vc, src/sampler.jl:20): fill_gae!(..., source=:cost, target=:cost_advantage) (:65)
        self.agent = agent if isinstance(agent, PolicyParams) else PolicyParams(agent)
        self.S = S or mdp.state_space()
        self.max_steps, self.required_columns = int(max_steps), list(required_columns)
        self.gamma, self.lam = np.float32(discount(mdp)), np.float32(lam)
        od = mdp.obs_dim
        mu = np.ascontiguousarray(np.broadcast_to(np.asarray(self.S.mu, np.float32), (od,)))
        sg = np.ascontiguousarray(np.broadcast_to(np.asarray(self.S.sigma, np.float32), (od,)))
        h = C.c_void_p()
        synth = mdp.kind in ("synth", "synth_discrete")
        self.ctx.check(self.ctx.lib.crux_env_create(self.ctx.h, L.ENV[mdp.kind], mdp.n_envs, self.max_steps, float(self.gamma), _vp(mu), _vp(sg),
                                                    mdp.seed, mdp.obs_dim if synth else 0, mdp.act_dim if synth else 0, C.byref(h)))
        self.h = h

    @property
    def n_envs(self):
        return self.mdp.n_envs

    def state(self):
        sd = int(self.ctx.lib.crux_env_state_dim(self.h)); E = self.n_envs
        st, el, nr = np.empty((sd, E), np.float64, order="F"), np.empty(E, np.int64), np.empty(E, np.int64)
        self.ctx.check(self.ctx.lib.crux_env_get_state(self.h, _vp(st), _vp(el), _vp(nr)))
        return st, el, nr

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.ctx.h:
                self.ctx.lib.crux_env_destroy(self.h)
        except Exception:
            pass


def _rollout_cfg(sampler, explore, reset, i):
    cfg = L.RolloutCfg()
    pi_on, pe = actor(sampler.agent.pi), sampler.agent.pi_explore
    cfg.explore, cfg.reset_at_end, cfg.i0 = int(bool(explore)), int(bool(reset)), int(i)
    cfg.eps_steps, cfg.noise_sigma = 0, -1.0
    cfg.noise_eps_min, cfg.noise_eps_max, cfg.a_min, cfg.a_max = -np.inf, np.inf, -np.inf, np.inf
    if isinstance(pe, EpsGreedyPolicy):
        cfg.head = L.HEAD["greedy_q"]; cfg.eps_start, cfg.eps_stop, cfg.eps_steps = pe.eps.start, pe.eps.stop, pe.eps.steps
    elif isinstance(pe, GaussianNoiseExplorationPolicy):
        cfg.head = L.HEAD["deterministic"]; cfg.noise_sigma = pe.sigma
        cfg.noise_eps_min, cfg.noise_eps_max, cfg.a_min, cfg.a_max = pe.eps_min, pe.eps_max, pe.a_min, pe.a_max
    else:
        cfg.head = L.HEAD[pi_on.head]
        cfg.logit_div = float(getattr(pi_on, "logit_div", 0.0))          # SoftQ: softmax(value ./ alpha) (softq.jl:53)
        if not explore and getattr(pi_on, "always_stochastic", False):   # action(pi, s) = exploration(pi, s)[1] (policies.jl:124): sample, logprob NaN
            cfg.explore = 2
    return cfg, pi_on


def steps_(sampler, buffer=None, Nsteps=1, explore=False, i=0, reset=False, cb=None, want_info=True, store=None):
    """steps!(sampler, buffer; Nsteps, explore, i, reset, cb, store) (src/sampler.jl:139-173). `store`: a list that receives a host copy of the block's columns after the
    callback ran on them (`!isnothing(store) && push!(store, data)`, :151 -- the solvers' interaction_storage).

    Nsteps counts transitions over all of the sampler's environments (Nsteps/n_envs per environment, env-major).
    GAE / returns are filled on the block this call produced, like terminate_episode! does before push! (:53-57,148-152), whatever the
    destination buffer's capacity or previous contents: the scans run on the ring rows the block was pushed to. Returns the info dict
    (avg_r as record_avgr)."""
    E = sampler.n_envs
    if Nsteps % E:
        raise ValueError("steps!: Nsteps=%d is not a multiple of n_envs=%d" % (Nsteps, E))
    cfg, pi_on = _rollout_cfg(sampler, explore, reset, i)
    sr, ne = C.c_double(), C.c_int64()
    first = buffer.next_ind - 1                                   # 0-based ring row the block starts at (push!, experience_buffer.jl:236)
    if not want_info and cb is None and store is None:      # callers that do not look at the rewards (the off-policy solve loop): the rollout stays asynchronous, no read-back to wait for
        sampler.ctx.check(sampler.ctx.lib.crux_rollout(sampler.h, pi_on.h, C.byref(cfg), buffer.h, Nsteps // E, None, None))
        _fill_block(sampler, buffer, first, Nsteps, reset)
        return {}
    sampler.ctx.check(sampler.ctx.lib.crux_rollout(sampler.h, pi_on.h, C.byref(cfg), buffer.h, Nsteps // E, C.byref(sr), C.byref(ne)))
    _fill_block(sampler, buffer, first, Nsteps, reset)
    info = {"sum_r": sr.value, "n_episode_end": ne.value, "avg_r": sr.value / ne.value if ne.value else float("nan")}
    if cb:
        cb(buffer, info)
    if store is not None:                                         # :151, after the callback, before push!(buffer, data): the block as the reference's `data` Dict
        store.append(buffer.minibatch((first + np.arange(Nsteps)) % buffer.capacity + 1))
    return info


def _fill_block(sampler, buffer, first, Nsteps, reset):
    """terminate_episode!'s fill_gae! / fill_returns! (src/sampler.jl:56-57) on the rows [first, first + Nsteps) mod capacity of `buffer`."""
    if Nsteps > buffer.capacity:
        if buffer.haskey("advantage") or buffer.haskey("return") or buffer.haskey("cost_advantage") or buffer.haskey("cost_return"):
            raise L.CruxError(L.EINVAL, "steps!: a block of %d transitions does not fit the buffer (capacity %d) whose :advantage / :return columns it must fill" % (Nsteps, buffer.capacity))
        return
    lib = buffer.ctx.lib
    if buffer.haskey("advantage"):
        buffer.ctx.check(lib.crux_fill_gae_rows(buffer.h, critic(sampler.agent.pi).h, float(sampler.lam), float(sampler.gamma), int(first), int(Nsteps), int(Nsteps // sampler.n_envs), 1 if reset else 0))
    if buffer.haskey("return"):
        buffer.ctx.check(lib.crux_fill_returns_rows(buffer.h, float(sampler.gamma), int(first), int(Nsteps), int(Nsteps // sampler.n_envs), 1 if reset else 0))
    # importance weights: the per-step ratio against the nominal action policy (step!, sampler.jl:108-111), then its running products per episode (:58-62, 283-308)
    if buffer.haskey("importance_weight") or any(buffer.haskey(k) for k in ("fwd_importance_weight", "cum_importance_weight", "rev_importance_weight", "traj_importance_weight")):
        _fill_importance_weights(sampler, buffer, first, Nsteps, reset)
    # cost constraints (sampler.jl:65-66)
    if buffer.haskey("cost_advantage"):
        if sampler.Vc is None:
            raise L.CruxError(L.EINVAL, "steps!: the buffer has a :cost_advantage column but the sampler has no Vc")
        buffer.ctx.check(lib.crux_fill_gae_rows_keys(buffer.h, sampler.Vc.h, float(sampler.lam), float(sampler.gamma), int(first), int(Nsteps), int(Nsteps // sampler.n_envs),
                                                     1 if reset else 0, L.COL["cost"], L.COL["cost_advantage"]))
    if buffer.haskey("cost_return"):
        buffer.ctx.check(lib.crux_fill_returns_rows_keys(buffer.h, float(sampler.gamma), int(first), int(Nsteps), int(Nsteps // sampler.n_envs), 1 if reset else 0,
                                                         L.COL["cost"], L.COL["cost_return"]))


def _fill_importance_weights(sampler, buffer, first, Nsteps, reset):
    lib, ctx = buffer.ctx.lib, buffer.ctx
    if buffer.haskey("importance_weight"):
        pa = getattr(sampler.agent, "pa", None)
        if pa is None:
            raise L.CruxError(L.EINVAL, "steps!: the buffer has an :importance_weight column but the agent has no nominal action policy `pa` (sampler.jl:109)")
        if not buffer.haskey("logprob"):
            raise L.CruxError(L.EINVAL, "steps!: :importance_weight needs the :logprob column of the exploration policy (sampler.jl:110)")
        head = L.HEAD["categorical"] if isinstance(pa, DiscreteNetwork) else L.HEAD["gaussian"]
        C_ = buffer.capacity; n1 = min(Nsteps, C_ - first)
        ctx.check(lib.crux_importance_weight_rows(buffer.h, pa.h, head, int(first), int(n1)))
        if n1 < Nsteps:
            ctx.check(lib.crux_importance_weight_rows(buffer.h, pa.h, head, 0, int(Nsteps - n1)))
    if any(buffer.haskey(k) for k in ("fwd_importance_weight", "cum_importance_weight", "rev_importance_weight")):
        ctx.check(lib.crux_fill_importance_weights_rows(buffer.h, int(first), int(Nsteps), int(Nsteps // sampler.n_envs), 1 if reset else 0))
    if buffer.haskey("traj_importance_weight"):
        # data[:traj_importance_weight][1, ep] .= sampler.traj_weight_fn(sampler.agent, data, ep) (sampler.jl:62): a host function of the episode's rows
        fn = getattr(sampler, "traj_weight_fn", None)
        if fn is None:
            raise L.CruxError(L.EINVAL, "steps!: the buffer has a :traj_importance_weight column but the sampler has no traj_weight_fn (sampler.jl:21,62)")
        C_ = buffer.capacity; ids = (first + np.arange(Nsteps)) % C_ + 1
        rows = buffer.minibatch(ids); ee = rows["episode_end"].reshape(-1).astype(bool); seg = Nsteps // sampler.n_envs
        out = rows["traj_importance_weight"].reshape(-1).copy(); start = 0
        for j in range(Nsteps):
            last_of_seg = (j + 1) % seg == 0
            if ee[j] or (last_of_seg and reset):
                ep = np.arange(start, j + 1); out[ep] = np.float32(fn(sampler.agent, rows, ep)); start = j + 1
            elif last_of_seg:
                out[start:j + 1] = 1.0; start = j + 1          # an episode left open: the fresh data block's ones
        col = buffer["traj_importance_weight"]; col[..., ids - 1] = out; buffer["traj_importance_weight"] = col


def episodes_(sampler, Neps=1, explore=False, i=0, seed_offset=0x45564C):
    """episodes!(sampler; Neps, explore, i) (src/sampler.jl:175-200) as a batched evaluation: Neps freshly reset copies of the sampler's
    environment are rolled out in parallel (one wave each) for max_steps steps; the first episode of each copy is one evaluation episode.
    Returns (data::ExperienceBuffer, metrics) with per-episode undiscounted / discounted returns, lengths and completion flags."""
    mdp = sampler.mdp
    em = GymMDP(mdp.kind, n_envs=int(Neps), seed=mdp.seed + int(seed_offset), discount=mdp.discount, obs_dim=mdp.obs_dim, act_dim=mdp.act_dim)
    es = Sampler(em, sampler.agent, S=sampler.S, max_steps=sampler.max_steps, required_columns=(), ctx=sampler.ctx)
    T = sampler.max_steps
    data = ExperienceBuffer(sampler.S, sampler.agent.space, int(Neps) * T, ctx=sampler.ctx)
    steps_(es, data, Nsteps=int(Neps) * T, explore=explore, i=i, reset=True)
    und, dis = np.empty(Neps, np.float32), np.empty(Neps, np.float32)
    ln, ok = np.empty(Neps, np.int64), np.empty(Neps, np.uint8)
    sampler.ctx.check(sampler.ctx.lib.crux_first_episode_metrics(data.h, int(Neps), T, float(np.float32(discount(mdp))), _vp(und), _vp(dis), _vp(ln), _vp(ok)))
    return data, {"undiscounted": und, "discounted": dis, "length": ln, "complete": ok.astype(bool)}


def undiscounted_return(sampler, Neps=100, **kw):
    """undiscounted_return(s::Sampler; Neps) (src/sampler.jl:219-220): sum of rewards per evaluation episode, averaged."""
    return float(episodes_(sampler, Neps=Neps, **kw)[1]["undiscounted"].astype(np.float64).sum() / Neps)


def discounted_return(sampler, Neps=100, **kw):
    """discounted_return(s::Sampler; Neps) (src/sampler.jl:231-234)."""
    return float(np.mean(episodes_(sampler, Neps=Neps, **kw)[1]["discounted"]))


def failure(sampler, threshold=0.0, Neps=100, **kw):
    """failure(s::Sampler; threshold, Neps) (src/sampler.jl:237-242): fraction of evaluation episodes whose undiscounted return is below threshold."""
    return float(np.mean(episodes_(sampler, Neps=Neps, **kw)[1]["undiscounted"] < threshold))


def steps_multi_(samplers, buffers, Nsteps=1, explore=False, i=0, reset=False):
    """steps! for several independent samplers of equal shape in one launch (the rollout half of a multi-seed run); GAE / returns are then
    filled per buffer exactly as steps_ does. Returns one info dict per sampler."""
    s0 = samplers[0]; E = s0.n_envs; n = len(samplers)
    if Nsteps % E:
        raise ValueError("steps!: Nsteps=%d is not a multiple of n_envs=%d" % (Nsteps, E))
    cfg, _ = _rollout_cfg(s0, explore, reset, i)
    he = (C.c_void_p * n)(*[s.h for s in samplers]); hp = (C.c_void_p * n)(*[actor(s.agent.pi).h for s in samplers]); hb = (C.c_void_p * n)(*[b.h for b in buffers])
    sr, ne = np.zeros(n, np.float64), np.zeros(n, np.int64)
    firsts = [b.next_ind - 1 for b in buffers]
    s0.ctx.check(s0.ctx.lib.crux_rollout_multi(n, he, hp, C.byref(cfg), hb, Nsteps // E, _vp(sr), _vp(ne)))
    if reset and all(b.haskey("advantage") and len(b) == Nsteps and b.capacity == Nsteps for b in buffers):     # every buffer IS its block: batched scans
        hc = (C.c_void_p * n)(*[critic(s.agent.pi).h for s in samplers])
        with_ret = all(b.haskey("return") for b in buffers)
        s0.ctx.check(s0.ctx.lib.crux_fill_gae_multi(n, hb, hc, float(s0.lam), float(s0.gamma), 1 if with_ret else 0))
        if not with_ret:
            for s, b in zip(samplers, buffers):
                if b.haskey("return"):
                    fill_returns_(b, s.gamma)
    else:
        for s, b, f in zip(samplers, buffers, firsts):
            _fill_block(s, b, f, Nsteps, reset)
    return [{"sum_r": float(sr[k]), "n_episode_end": int(ne[k]), "avg_r": float(sr[k] / ne[k]) if ne[k] else float("nan")} for k in range(n)]


def fill_gae_(buffer, V, lam, gamma):
    """fill_gae!(d::ExperienceBuffer, V, lambda, gamma) (src/sampler.jl:255-273)."""
    buffer.ctx.check(buffer.ctx.lib.crux_fill_gae(buffer.h, critic(V).h, float(lam), float(gamma)))


def fill_returns_(buffer, gamma):
    """fill_returns! over episodes(buffer) (src/sampler.jl:275-281)."""
    buffer.ctx.check(buffer.ctx.lib.crux_fill_returns(buffer.h, float(gamma)))


def whiten_(buffer, key="advantage"):
    """buffer[key] .= whiten(buffer[key]) (src/utils.jl:41-42, ppo.jl:61)."""
    buffer.ctx.check(buffer.ctx.lib.crux_whiten(buffer.h, L.COL[key]))


def whiten_multi_(buffers, key="advantage"):
    """whiten_ for several buffers of equal length in one launch."""
    n = len(buffers); hb = (C.c_void_p * n)(*[b.h for b in buffers])
    buffers[0].ctx.check(buffers[0].ctx.lib.crux_whiten_multi(n, hb, L.COL[key]))


# --------------------------------------------------------------------------------------------------------------
# training (src/training.jl)
# --------------------------------------------------------------------------------------------------------------
class _Loss:
    def __init__(self, name):
        self.name = name


ppo_loss = _Loss("ppo")            # src/model_free/rl/ppo.jl:4-21
value_mse_loss = _Loss("value_mse")  # (pi, P, D) -> Flux.mse(value(pi, D[:s]), D[:return])  ppo.jl:60
a2c_loss = _Loss("a2c")            # src/model_free/rl/a2c.jl:4-15
reinforce_loss = _Loss("reinforce")  # src/model_free/rl/reinforce.jl:4-13
lagrange_ppo_loss = _Loss("lagrange_ppo")   # src/model_free/rl/ppo.jl:70-131; P carries the penalty controller ("lagrange": _lib.Lagrange)
cost_value_mse_loss = _Loss("cost_value_mse")   # (pi, P, D) -> Flux.mse(value(pi, D[:s]), D[:cost_return]) (ppo.jl:210)


class CustomLoss(_Loss):
    """A user-written loss in the `loss` field of TrainingParams (src/training.jl:2) for losses outside the library's closed list. The reference
    differentiates `loss(pi, P, D)` with Zygote (training.jl:16-18); here the user supplies the one piece Zygote would derive, d(loss)/d(network
    output), and the library supplies the pullback through the network (crux_mlp_forward_cached / crux_mlp_backward on the MFMA dense engine):

        fn(y, D, P) -> (loss, dloss_dy)  or  (loss, dloss_dy, info_dict)

    y = value(pi, D[:s]) as a host array [out x B]; D = minibatch(D, indices) as a dict of host arrays; dloss_dy has y's shape. Host-mediated (one
    round trip per minibatch): the escape hatch, not the fast path."""

    def __init__(self, fn, name="custom"):
        super().__init__(name)
        self.fn = fn


class ParamLoss(_Loss):
    """A user-written loss over a bare parameter vector (a `param_optimizers` entry, on_policy.jl:59-61 / off_policy.jl:86-88: `batch_train!(θs, p_opt, P, D, π_loss=agent.π)`):

        fn(theta, D, P, pi) -> (loss, dloss_dtheta)  or  (loss, dloss_dtheta, info_dict)

    theta = the ParamVector's values (host copy), D = minibatch(D, indices) as a dict of host arrays, pi = the agent's policy (for value(pi, ...) on the host
    side of the loss). There is no network between the parameters and the loss, so the gradient the user returns IS the pullback."""

    def __init__(self, fn, name="param"):
        super().__init__(name)
        self.fn = fn


class TrainingParams:
    """TrainingParams(; loss, optimizer=Adam(3f-4), regularizer, batch_size=128, epochs=80, early_stopping, name, max_batches)
    (src/training.jl:1-11). PPO's early_stopping (`infos[end][:kl] > target_kl`, ppo.jl:59) is expressed as target_kl and runs inside the
    persistent learner kernel. The function-valued seams of the reference take the host-driven path of _train_seam / _batch_train_seam:
    regularizer(theta) -> (value, gradient) over the flat parameter vector (training.jl:4,13: loss + regularizer(pi)); early_stopping(infos) ->
    Bool over the per-epoch info dicts (:8,46,49); loss = CustomLoss(fn)."""

    def __init__(self, loss, optimizer=None, batch_size=128, epochs=80, target_kl=None, name="", max_batches=math.inf,
                 shuffle_seed=0, update_every=1, regularizer=None, early_stopping=None):
        self.loss = loss
        self.regularizer, self.early_stopping = regularizer, early_stopping
        self.update_every = int(update_every)        # off-policy solvers: train this network every update_every-th epoch (off_policy.jl:91,96)
        self.optimizer = optimizer or Adam(np.float32(3e-4))
        self.batch_size, self.epochs, self.target_kl, self.name, self.max_batches = int(batch_size), int(epochs), target_kl, name, max_batches
        self.shuffle_seed, self.shuffle_counter = int(shuffle_seed), 0


def _uses_seam(p):
    return isinstance(p.loss, (CustomLoss, ParamLoss)) or getattr(p, "regularizer", None) is not None or getattr(p, "early_stopping", None) is not None


def _train_seam(pi, p, P, D, ids0, info):
    """train!(pi, loss + regularizer, p) (src/training.jl:13-25) with the function-valued pieces evaluated on the host:
    pullback (:16-18) = library loss gradient (crux_loss_grad) or, for a CustomLoss, crux_mlp_forward_cached -> fn -> crux_mlp_backward;
    the regularizer's gradient is added to the flat gradient; norm / NaN check (:19-20); Flux.update! = crux_adam_apply (:21)."""
    ctx, lib = pi.ctx, pi.ctx.lib
    n = pi.n_params; extra = {}
    if isinstance(p.loss, ParamLoss):                                      # a bare vector: the user's gradient is the pullback
        res = p.loss.fn(pi.get_params(), D.minibatch(ids0 + 1), P, getattr(p, "pi_loss", None))
        l, g0 = float(res[0]), np.ascontiguousarray(np.asarray(res[1], np.float32).reshape(-1))
        if len(res) > 2:
            extra = dict(res[2])
        if g0.size != n:
            raise ValueError("ParamLoss: the gradient must have %d entries" % n)
        ctx.h2d(lib.crux_mlp_grads_ptr(pi.h), g0)
        raw = None
    elif isinstance(p.loss, CustomLoss):
        mb = D.minibatch(ids0 + 1); x = np.asfortranarray(mb["s"], dtype=np.float32); B = x.shape[1]; out = pi.network.dims[-1]
        d_x, d_y = ctx.alloc(x.nbytes), ctx.alloc(4 * out * B)
        try:
            ctx.h2d(d_x, x)
            ctx.check(lib.crux_mlp_forward_cached(pi.h, d_x, B, d_y))
            y = np.empty((out, B), np.float32, order="F"); ctx.d2h(d_y, y)
            res = p.loss.fn(y, mb, P)
            l, dy = float(res[0]), np.asfortranarray(res[1], dtype=np.float32)
            if len(res) > 2:
                extra = dict(res[2])
            if dy.shape != y.shape:
                raise ValueError("CustomLoss: dloss_dy must have the shape of the network output %r" % (y.shape,))
            ctx.h2d(d_y, dy)
            ctx.check(lib.crux_mlp_backward(pi.h, d_x, B, d_y, 1.0, 1, None))
        finally:
            ctx.free(d_x); ctx.free(d_y)
        raw = None
    else:
        raw = np.zeros(L.INFO_N, np.float32); cfg = _train_cfg(pi, p, P)
        ctx.check(lib.crux_loss_grad(pi.h, D.h, C.byref(cfg), _vp(ids0), ids0.size, _vp(raw)))
        l = float(raw[L.INFO["loss"]])
    g = np.empty(n, np.float32); ctx.d2h(lib.crux_mlp_grads_ptr(pi.h), g)
    if p.regularizer is not None:
        rv, rg = p.regularizer(pi.get_params())
        l = float(np.float32(l) + np.float32(rv)); g = (g + np.asarray(rg, np.float32).reshape(-1)).astype(np.float32)
        ctx.h2d(lib.crux_mlp_grads_ptr(pi.h), g)
    gnorm = float(np.float32(np.sqrt(np.sum(g.astype(np.float64) ** 2))))
    if math.isnan(gnorm):
        raise L.CruxError(L.ENAN, "NaN detected! Loss: %r" % l)                                   # training.jl:20
    ctx.check(lib.crux_adam_apply(pi.h, 1.0))
    if raw is not None:
        info.update(_info_dict(p, raw))
    info.update(extra)
    info[p.name + "loss"], info[p.name + "grad_norm"] = l, gnorm
    return info


def _batch_train_seam(pi, p, P, D, info, perms):
    """batch_train! (src/training.jl:28-55) driven from the host, for TrainingParams with a CustomLoss, a regularizer or an early_stopping closure:
    epochs x (shuffle!, partition(1:length(D), batch_size), train!), max_batches (:45,50), early_stopping over the aggregated infos (:46,49; the
    aliased info dict makes aggregate_info(minibatch_infos) the latest minibatch's)."""
    infos, total, N = [], 0, len(D)
    stop_fn = p.early_stopping or ((lambda infos_: infos_[-1].get("kl", 0.0) > p.target_kl) if p.target_kl is not None else (lambda infos_: False))
    maxb = math.inf if p.max_batches in (None, math.inf) else int(p.max_batches)
    cur = {}
    for ep in range(p.epochs):
        if perms is not None:
            D.shuffle_(np.asarray(perms[ep], np.int64))
        else:
            shuffle_device_(D, p.shuffle_seed, p.shuffle_counter); p.shuffle_counter += 1
        for st in range(0, N, p.batch_size):
            ids0 = np.arange(st, min(N, st + p.batch_size), dtype=np.int64)
            cur = _train_seam(pi, p, P, D, ids0, cur)
            total += 1
            if total >= maxb or stop_fn(infos + [dict(cur)]):
                break
        infos.append(dict(cur))
        if stop_fn(infos) or total >= maxb:
            break
    info = info if info is not None else {}
    agg = {k: float(np.mean([d[k] for d in infos])) for k in infos[0]} if infos else {}
    info.update(agg)
    info[p.name + "batches_trained"] = total; info["_epochs_run"] = len(infos)
    return info


def _train_cfg(pi, p, P):
    cfg = L.TrainCfg()
    if p.loss.name == "cost_value_mse":              # Flux.mse(value(pi, D[:s]), D[:cost_return]) (ppo.jl:210): the critic loss against another column
        cfg.loss, cfg.target_col = L.LOSS["value_mse"], L.COL["cost_return"]
    else:
        cfg.loss = L.LOSS[p.loss.name]
    cfg.head = L.HEAD.get(getattr(pi, "head", "deterministic"), 3)
    cfg.batch_size, cfg.epochs = p.batch_size, p.epochs
    cfg.max_batches = 0 if p.max_batches in (None, math.inf) else int(p.max_batches)
    cfg.eps_clip, cfg.lambda_p, cfg.lambda_e = float(P.get("eps", 0.2)), float(P.get("lambda_p", 1.0)), float(P.get("lambda_e", 0.1))
    cfg.target_kl = -1.0 if p.target_kl is None else float(p.target_kl)
    cfg.shuffle_seed, cfg.shuffle_counter = p.shuffle_seed, p.shuffle_counter
    return cfg


def _info_dict(p, raw, extra=True):
    d = {p.name + "loss": float(raw[L.INFO["loss"]]), p.name + "grad_norm": float(raw[L.INFO["grad_norm"]])}
    if p.loss.name == "logpdf_bc":
        d["entropy"], d["logpdf"] = float(raw[L.INFO["entropy"]]), float(raw[L.INFO["kl"]])          # info[:logpdf] = -mean(logpdf) (bc.jl:15)
    if p.loss.name in ("a2c", "reinforce"):
        for k in ("entropy", "kl"):
            d[k] = float(raw[L.INFO[k]])
    if p.loss.name in ("ppo", "lagrange_ppo"):
        for k in ("entropy", "kl", "clip_fraction", "avg_advantage", "avg_return"):
            d[k] = float(raw[L.INFO[k]])
    if p.loss.name == "lagrange_ppo":                                                  # info["penalty"], ["cur_cost"], ["cost_loss"], ["p_loss"] (ppo.jl:111-127)
        for k in ("penalty", "cur_cost", "cost_loss", "p_loss"):
            d[k] = float(raw[L.INFO[k]])
    return d


def _ensure_opt(pi, p):
    if pi.optimizer is not p.optimizer:
        pi.attach_optimizer(p.optimizer)


def train_(pi, p, P, D, indices, info=None):
    """Flux.Optimise.train!(pi, loss, p; info) on minibatch(D, indices) (src/training.jl:13-25); 1-based indices.
    Raises CruxError(ENAN) like `error("NaN detected!")` (:20)."""
    _ensure_opt(pi, p)
    ids = np.ascontiguousarray(np.asarray(indices, np.int64) - 1)
    if _uses_seam(p):
        return _train_seam(pi, p, P, D, ids, info if info is not None else {})
    raw = np.zeros(L.INFO_N, np.float32)
    cfg = _train_cfg(pi, p, P)
    pi.ctx.check(pi.ctx.lib.crux_train_step(pi.h, D.h, C.byref(cfg), _vp(ids), ids.size, _vp(raw)))
    info = info if info is not None else {}
    info.update(_info_dict(p, raw))
    return info


def batch_train_(pi, p, P, D, info=None, perms=None):
    """batch_train!(pi, p, P, D; info) (src/training.jl:28-55): epochs x (shuffle!, partition, train!) with max_batches and
    early stopping, as ONE persistent kernel. perms: optional (epochs, len) 1-based permutations (else Philox)."""
    _ensure_opt(pi, p)
    if _uses_seam(p):
        return _batch_train_seam(pi, p, P, D, info, perms)
    cfg = _train_cfg(pi, p, P)
    pp = None
    if perms is not None:
        pp = np.ascontiguousarray(np.asarray(perms, np.int64) - 1)
        if pp.shape != (p.epochs, len(D)):
            raise ValueError("batch_train!: perms must have shape (epochs, length(D))")
    raw = np.zeros(L.INFO_N, np.float32)
    ep = np.zeros((p.epochs, L.INFO_N), np.float32)
    if p.loss.name == "lagrange_ppo":
        pi.ctx.check(pi.ctx.lib.crux_batch_train_lagrange(pi.h, D.h, C.byref(cfg), C.byref(P["lagrange"]), _vp(pp), _vp(raw), _vp(ep)))
    else:
        pi.ctx.check(pi.ctx.lib.crux_batch_train(pi.h, D.h, C.byref(cfg), _vp(pp), _vp(raw), _vp(ep)))
    p.shuffle_counter += int(raw[L.INFO["epochs_run"]])
    info = info if info is not None else {}
    info.update(_info_dict(p, raw))
    info[p.name + "batches_trained"] = int(raw[L.INFO["batches_trained"]])
    info["_epochs_run"] = int(raw[L.INFO["epochs_run"]])
    info["_epoch_infos"] = ep[: info["_epochs_run"]]
    return info


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
    """POMDPs.solve(S::OnPolicySolver, mdp) (src/model_free/on_policy.jl:80-109), logging left out (SURVEY #14)."""
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


# --------------------------------------------------------------------------------------------------------------
# OnPolicyGAIL (src/model_free/il/on_policy_gail.jl) -- the discriminator is trained by batch_train! over TWO buffers (training.jl:28-44)
# --------------------------------------------------------------------------------------------------------------
gail_d_loss = _Loss("gail_d")     # gail_d_loss(GAN_BCELoss()) (on_policy_gail.jl:1-5, extras/gans.jl:7-9)


def copy_buffer(b):
    """deepcopy(b::ExperienceBuffer): same columns, same rows, same order."""
    out = buffer_like(b, capacity=b.capacity)
    if len(b):
        out.push_(b, ids=np.arange(1, len(b) + 1))
    return out


def shuffle_device_(b, seed, counter):
    """shuffle!(b) with the library's permutation stream (crux_rng.h), composed and applied on the device."""
    b.ctx.check(b.ctx.lib.crux_buffer_shuffle(b.h, int(seed), int(counter))); return b


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


def normalize_(b, S, A):
    """normalize!(b, S, A) (src/experience_buffer.jl:143-148): s, sp (and a for ContinuousSpace) replaced by tovec(., space) = (v - mu) / sigma (spaces.jl:25)."""
    def tovec(v, sp):
        mu = np.broadcast_to(np.asarray(sp.mu, np.float32), (v.shape[0],))[:, None]; sg = np.broadcast_to(np.asarray(sp.sigma, np.float32), (v.shape[0],))[:, None]
        return ((v - mu) / sg).astype(np.float32)
    for k in ("s", "sp"):
        if b.haskey(k):
            b[k] = tovec(b[k], S)
    if isinstance(A, ContinuousSpace):
        b["a"] = tovec(b["a"], A)
    return b


def split(b, fracs):
    """split(b::ExperienceBuffer, fracs) (src/experience_buffer.jl:133-141): consecutive row ranges of sizes split_batches(length(b), fracs)."""
    out, start = [], 0
    extras = extra_columns(b)
    for n in split_batches(len(b), fracs):
        nb = ExperienceBuffer(b.S, b.A, max(int(n), 1), extras, ctx=b.ctx)
        if n > 0:
            nb.push_({k: b[k][:, start:start + n] for k in b.keys()})
        out.append(nb); start += n
    return out


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


# --------------------------------------------------------------------------------------------------------------
# off-policy solver + DQN (src/model_free/off_policy.jl, src/model_free/rl/dqn.jl)
# --------------------------------------------------------------------------------------------------------------
def buffer_like(b, capacity=None):
    """buffer_like(b; capacity) (src/experience_buffer.jl:82-85): same columns, prioritized if b is (:84)."""
    extras = [k for k in b.keys() if k not in ("s", "a", "sp", "r", "done", "episode_end")]
    return ExperienceBuffer(b.S, b.A, capacity or b.capacity, extras, prioritized=b.isprioritized(), priority_params={"alpha": b.alpha, "beta": b.beta}, ctx=b.ctx)


def episodes(b, use_done=False, episode_checker=None):
    """episodes(b::ExperienceBuffer, use_done, episode_checker) (src/experience_buffer.jl:194-221): 1-based inclusive (start, stop) pairs from :episode_end
    (or :t == 1 starts, or :done when asked); a trailing open episode is closed at length(b)."""
    n = len(b)
    if b.haskey("episode_end"):
        ends = list(np.flatnonzero(b["episode_end"][0]) + 1); starts = [1] + [e + 1 for e in ends[:-1]]
    elif b.haskey("t"):
        starts = list(np.flatnonzero(b["t"][0] == 1) + 1); ends = [s_ - 1 for s_ in starts[1:]] + [n]
    elif use_done:
        ends = list(np.flatnonzero(b["done"][0]) + 1); starts = [1] + [e + 1 for e in ends[:-1]]
    else:
        raise ValueError("Need :episode_end flag or :t column to determine episodes")
    if not ends:                                     # the reference would index an empty array here; an un-terminated buffer is one open episode
        starts, ends = ([1], [n]) if n > 0 else ([], [])
    elif n > 0 and ends[-1] != n:
        starts.append(ends[-1] + 1); ends.append(n)
    eps = [(int(a), int(z)) for a, z in zip(starts, ends)]
    return [e for e in eps if episode_checker(b, e)] if episode_checker is not None else eps


def hcat(*buffers, capacity=None):
    """hcat(buffers::ExperienceBuffer...) (:106-116): a new buffer holding the rows of every argument in order (same columns required)."""
    b0 = buffers[0]
    for b in buffers[1:]:
        if sorted(b.keys()) != sorted(b0.keys()):
            raise L.CruxError(L.EINVAL, "hcat: buffers have different columns (@assert keys(data) == keys(b))")
    n = sum(len(b) for b in buffers)
    out = buffer_like(b0, capacity=max(1, capacity or n))
    for b in buffers:
        if len(b):
            out.push_(b, ids=np.arange(1, len(b) + 1))
    return out


def get_episodes(b, eps):
    """get_episodes(b, episodes) (:150-156): the rows of the listed (start, stop) episodes, concatenated."""
    ids = np.concatenate([np.arange(a, z + 1) for a, z in eps]) if len(eps) else np.zeros(0, np.int64)
    out = buffer_like(b, capacity=max(1, ids.size))
    if ids.size:
        out.push_(b, ids=ids)
    return out


def trim_(b, n):
    """trim!(b, 1:n) (:158-168) as the samplers use it (sampler.jl:144,193): keep the first n rows. Returns a buffer of capacity n (the device columns
    are fixed-size allocations, so the trimmed view is a new handle)."""
    n = int(min(n, len(b)))
    out = buffer_like(b, capacity=max(1, n))
    if n:
        out.push_(b, ids=np.arange(1, n + 1))
    return out


def extra_columns(b):
    """extra_columns(b) (src/experience_buffer.jl:178)."""
    return [k for k in b.keys() if k not in ("s", "a", "sp", "r", "done", "episode_end")]


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
            if solver.target_fn == "softq":
                rc = ctx.lib.crux_softq_epochs_async(pi.h, pim.h, buf.h, D.h, float(gamma), float(solver.P["alpha"]), 1 if solver.weighted_loss else 0, beta, solver.i * p.epochs, p.epochs, d_rows)
            else:
                rc = ctx.lib.crux_dqn_epochs_async(pi.h, pim.h, buf.h, D.h, float(gamma), 1 if solver.weighted_loss else 0, beta, solver.i * p.epochs, p.epochs, d_rows)
            if rc == L.OK:
                name = p.name; row0 = _row0
                def decode(raws):
                    return [{name + "loss": float(r[0]), name + "grad_norm": float(r[1]), "Qavg": float(r[2])} for r in raws], bool(np.isnan(raws[:, 1]).any())
                pend = _PendingInfo(row0, p.epochs, decode); solver._dinfos_used += p.epochs
                solver._update_target(final=True)                                                          # :108
                return pend
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
    if not (solver.fused_epochs and solver.target_fn == "dqn" and not solver.custom_seams() and solver.post_sample_callback is None and solver.pre_train_callback is None
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
    """POMDPs.solve(S::OffPolicySolver, mdp) (src/model_free/off_policy.jl:113-150), logging left out."""
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


_solve_on_policy = solve


def solve(solver, mdp=None):  # noqa: F811
    """POMDPs.solve(solver, mdp) for OnPolicySolver (on_policy.jl:80-109) and OffPolicySolver (off_policy.jl:113-150)."""
    if isinstance(solver, BatchSolver):
        return _solve_batch(solver, mdp)
    return _solve_off_policy(solver, mdp) if isinstance(solver, OffPolicySolver) else _solve_on_policy(solver, mdp)
