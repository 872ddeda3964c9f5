"""core.py -- contexts, spaces, networks, buffers, sampling, environments, the Sampler and steps!, losses, TrainingParams, train! / batch_train! (shared by every solver family).

Split out of api.py in round 4; `crux_jl_amd.api` re-exports everything, so `crux.X` and `crux.api.X` resolve as before."""
import ctypes as C
import math
import numpy as np
from . import _lib as L


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

    def peer_set_timeout_ms(self, ms):
        """in-kernel flag-wait timeout of one exchange (default 30 s): a rank whose peer died gets CruxError(EHIP) after this long instead of a hung GPU"""
        self.check(self.lib.crux_peer_set_timeout_ms(self.h, int(ms)))

    def peer_set_budget_ms(self, ms):
        """what the flag waits of ONE learner launch may add up to (default 60 s; 0 = no budget): replicas that answer, but a scheduling quantum late (cruxhip.h)"""
        self.check(self.lib.crux_peer_set_budget_ms(self.h, int(ms)))

    def peer_abort(self):
        """call this context's replica-group launches off from the host (no GPU work; any thread): they return CruxError(EHIP) within ~100 us"""
        self.lib.crux_peer_abort(self.h)

    def peer_abort_clear(self):
        self.lib.crux_peer_abort_clear(self.h)

    def peer_abort_reason(self):
        """the abort words of this rank's region, one per learner stream: 0 none, 1 timeout, 2 budget, 3 passed on, 4 host, 5 a peer left on a NaN step"""
        out = np.zeros(2, np.int32); self.check(self.lib.crux_peer_abort_reason(self.h, _vp(out))); return [int(x) for x in out]

    def peer_probe(self, rounds=64, first_bound_ms=2000, round_bound_ms=20):
        """COLLECTIVE rendezvous probe (cruxhip.h: crux_peer_probe): [first-round wait, longest later wait] in us for learner stream 0, then 1; CruxError(EHIP) if the replicas did not meet"""
        out = np.zeros(4, np.float32); self.check(self.lib.crux_peer_probe(self.h, int(rounds), int(first_bound_ms), int(round_bound_ms), _vp(out))); return [float(x) for x in out]

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


def abort_all():
    """raise the host abort word of every live context of this process (cruxhip.h: crux_abort_all): replica-group launches waiting for a peer return CruxError(EHIP). For watchdogs."""
    return int(L.load().crux_abort_all())


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


class HostMDP:
    """An mdp that exists only on the HOST -- any POMDPs.jl-style generative model (LunarLander, MuJoCo, a user's simulator) -- as n_envs independent copies the CALLER steps.
    The reference's step! calls `sp, r = @gen(:sp,:r)(mdp, s, a)`, `isterminal(mdp, sp)`, `convert_s(AbstractArray, sp, mdp)` (src/sampler.jl:89-97) and reset_sampler!
    draws `rand(initialstate(mdp))` (:31-43) on whatever mdp the user handed to solve; this class carries those four functions:

        initialstate(e, n_resets) -> s            rand(initialstate(mdp)) for copy e; n_resets = episodes copy e has started so far (for seeded environments)
        gen(e, s, a, n_steps)     -> (sp, r) | (sp, r, info)      @gen(:sp,:r)(mdp, s, a[; info]); a = the action index (DiscreteSpace, 0-based) or the Float32 action vector;
                                                   n_steps = steps copy e has taken so far; info["cost"] feeds the :cost column (:114)
        isterminal(sp)            -> bool
        observation(s)            -> (obs_dim,) array           convert_s(AbstractArray, s, mdp)

    The policy forward, exploration draws, log-probabilities (crux_policy_explore) and the buffer write with the GAE / return / importance-weight fills (crux_steps_push) run on
    the device; the Sampler state (s, svec, episode_length, was_reset) stays here, as it stays in Julia in the reference."""

    kind = "host"

    def __init__(self, initialstate, gen, isterminal, observation, obs_dim, act_dim, discrete, n_envs=1, seed=0, discount=0.99):
        self.initialstate, self.gen, self.isterminal, self.observation = initialstate, gen, isterminal, observation
        self.obs_dim, self.act_dim, self.discrete = int(obs_dim), int(act_dim), bool(discrete)
        self.n_envs, self.seed, self.discount = int(n_envs), int(seed), float(discount)

    def state_space(self, mu=0.0, sigma=1.0):
        return ContinuousSpace(self.obs_dim, np.float32, mu, sigma)

    def action_space(self):
        return DiscreteSpace(self.act_dim) if self.discrete else ContinuousSpace(self.act_dim)


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
        self.h = None
        if isinstance(mdp, HostMDP):      # caller-stepped environments: the Sampler fields of src/sampler.jl:1-22 live here, one entry per copy
            E = mdp.n_envs
            self._mu, self._sigma = mu, sg
            self.s = [None] * E; self.svec = np.zeros((od, E), np.float32, order="F")
            self.episode_length = np.zeros(E, np.int64); self.n_resets = np.zeros(E, np.int64); self.steps_taken = np.zeros(E, np.int64); self.was_reset = np.zeros(E, np.bool_)
            for e in range(E):
                self._reset_one(e)
            self.was_reset[:] = False      # (a fresh Sampler has was_reset = false, sampler.jl:14)
            return
        h = C.c_void_p()
        synth = mdp.kind in ("synth", "synth_discrete")
        self.ctx.check(self.ctx.lib.crux_env_create(self.ctx.h, L.ENV[mdp.kind], mdp.n_envs, self.max_steps, float(self.gamma), _vp(mu), _vp(sg),
                                                    mdp.seed, mdp.obs_dim if synth else 0, mdp.act_dim if synth else 0, C.byref(h)))
        self.h = h

    @property
    def n_envs(self):
        return self.mdp.n_envs

    def _tovec(self, o):
        """tovec(v, S::ContinuousSpace) = (v .- mu) ./ sigma in Float32 (src/spaces.jl:25)"""
        return (np.asarray(o, np.float32).reshape(-1) - self._mu) / self._sigma

    def _reset_one(self, e):
        """reset_sampler! (src/sampler.jl:31-43) of copy e of a HostMDP"""
        if self.was_reset[e]:
            return
        self.s[e] = self.mdp.initialstate(e, int(self.n_resets[e])); self.n_resets[e] += 1
        self.svec[:, e] = self._tovec(self.mdp.observation(self.s[e]))
        self.episode_length[e] = 0; self.was_reset[e] = True

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
    if isinstance(sampler.mdp, HostMDP):
        return _steps_host(sampler, buffer, Nsteps, cfg, pi_on, explore, i, reset, cb, store)
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


def policy_explore(pi, cfg, svec, seed, steps_taken):
    """crux_policy_explore (cruxhip.h): `a, logprob = exploration(pi_explore, svec; pi_on, i)` / `(action(pi, svec), NaN)` (src/sampler.jl:73) for the columns of svec at once.
    Returns (actions [act_dim x E] Bool one-hot or Float32, logprob [E])."""
    svec = np.asfortranarray(svec, np.float32); E = svec.shape[1]; nout = pi.network.dims[-1]
    disc = cfg.head in (L.HEAD["categorical"], L.HEAD["greedy_q"])
    a = np.zeros((nout, E), np.bool_ if disc else np.float32, order="F"); lp = np.empty(E, np.float32)
    st = np.ascontiguousarray(steps_taken, np.int64)
    pi.ctx.check(pi.ctx.lib.crux_policy_explore(pi.h, C.byref(cfg), E, _vp(svec), int(seed), _vp(st), _vp(a), _vp(lp)))
    return a, lp


def _steps_host(sampler, buffer, Nsteps, cfg, pi_on, explore, i, reset, cb, store):
    """steps! / step! (src/sampler.jl:71-155) for the copies of a HostMDP, statement by statement; the block is env-major like the device rollout (copy e owns rows
    [e*T, (e+1)*T), == hcat of E single-Sampler rollouts) and the interaction counter env-minor (i + t*E + e, :161-163)."""
    mdp = sampler.mdp; E = sampler.n_envs; T = Nsteps // E
    A = mdp.action_space()
    cols = list(sampler.required_columns)
    if buffer is not None:
        cols = [k for k in buffer.keys() if k not in ("s", "a", "sp", "r", "done", "episode_end")]
    data = mdp_data(sampler.S, A, Nsteps, cols)                                   # :140
    sum_r, n_ee = 0.0, 0
    for t in range(T):
        cfg.i0 = int(i) + t * E
        a_all, lp_all = policy_explore(pi_on, cfg, sampler.svec, mdp.seed, sampler.steps_taken)      # :73 for all copies
        for e in range(E):
            j = e * T + t
            sampler.was_reset[e] = False                                            # :72
            a = int(np.argmax(a_all[:, e])) if mdp.discrete else a_all[:, e].copy()
            out = mdp.gen(e, sampler.s[e], a, int(sampler.steps_taken[e]))          # :93-94  sp, r = @gen(:sp,:r)(mdp, s, a; info)
            sp, r = out[0], out[1]; info = out[2] if len(out) > 2 else {}
            spvec = sampler._tovec(mdp.observation(sp))                             # :95-96
            done = bool(mdp.isterminal(sp))                                         # :97
            data["s"][:, j] = sampler.svec[:, e]; data["a"][:, j] = a_all[:, e]; data["sp"][:, j] = spvec      # :100-102
            data["r"][0, j] = np.float32(r); data["done"][0, j] = done                                            # :103-104
            if "logprob" in data:
                data["logprob"][0, j] = lp_all[e]                                   # :107
            if "t" in data:
                data["t"][0, j] = sampler.episode_length[e] + 1                     # :112
            if "i" in data:
                data["i"][0, j] = int(i) + t * E + e + 1                            # :113
            if "cost" in data:
                data["cost"][0, j] = np.float32(info["cost"])                       # :114
            sum_r += float(np.float32(r)); sampler.steps_taken[e] += 1
            sampler.episode_length[e] += 1                                          # :130
            if done or sampler.episode_length[e] >= sampler.max_steps:              # :131-132  terminate_episode!: the cut here, the fills after the push (crux_steps_push)
                data["episode_end"][0, j] = True; n_ee += 1
                sampler._reset_one(e)
            else:
                sampler.s[e] = sp; sampler.svec[:, e] = spvec                       # :134-135
    if reset:                                                                       # :148  reset && terminate_episode!(sampler, data, Nsteps), per copy
        for e in range(E):
            j = e * T + T - 1
            if not data["episode_end"][0, j]:
                data["episode_end"][0, j] = True; n_ee += 1
            sampler._reset_one(e)
    info_out = {"sum_r": sum_r, "n_episode_end": n_ee, "avg_r": sum_r / n_ee if n_ee else float("nan")}
    if buffer is None:
        return data
    first = buffer.next_ind - 1
    pa = getattr(sampler.agent, "pa", None) if buffer.haskey("importance_weight") else None
    if buffer.haskey("importance_weight") and pa is None:
        raise L.CruxError(L.EINVAL, "steps!: the buffer has an :importance_weight column but the agent has no nominal action policy `pa` (sampler.jl:109)")
    if pa is not None and float(getattr(pa, "logit_div", 0.0) or 0.0) != 0.0:
        raise L.CruxError(L.EUNSUP, "steps!: :importance_weight with a nominal DiscreteNetwork whose logit conversion is not the plain softmax is not supported")
    ptrs = (C.c_void_p * L.NCOLS)(); keep = []
    for k, v in data.items():
        arr = np.asfortranarray(v); keep.append(arr); ptrs[L.COL[k]] = arr.ctypes.data
    cr = critic(sampler.agent.pi) if buffer.haskey("advantage") else None
    fr = C.c_int64()
    buffer.ctx.check(buffer.ctx.lib.crux_steps_push(buffer.h, int(Nsteps), ptrs, int(T), 1 if reset else 0, cr.h if cr is not None else None, float(sampler.lam), float(sampler.gamma),
                                                    sampler.Vc.h if sampler.Vc is not None else None, pa.h if pa is not None else None,
                                                    (L.HEAD["categorical"] if isinstance(pa, DiscreteNetwork) else L.HEAD["gaussian"]) if pa is not None else 0, C.byref(fr)))
    assert fr.value == first
    if buffer.haskey("traj_importance_weight"):
        _fill_traj_weights(sampler, buffer, first, Nsteps, reset)
    if cb:
        cb(buffer, info_out)
    if store is not None:
        store.append(buffer.minibatch((first + np.arange(Nsteps)) % buffer.capacity + 1))
    return info_out


def _fill_block(sampler, buffer, first, Nsteps, reset):
    """terminate_episode!'s fill_gae! / fill_returns! (src/sampler.jl:56-57) on the rows [first, first + Nsteps) mod capacity of `buffer`."""
    if Nsteps > buffer.capacity:
        filled = ("advantage", "return", "cost_advantage", "cost_return", "importance_weight", "fwd_importance_weight", "cum_importance_weight", "rev_importance_weight", "traj_importance_weight")
        if any(buffer.haskey(k) for k in filled):
            raise L.CruxError(L.EINVAL, "steps!: a block of %d transitions does not fit the buffer (capacity %d) whose %s column(s) it must fill" % (Nsteps, buffer.capacity, " / ".join(":" + k for k in filled if buffer.haskey(k))))
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
        if float(getattr(pa, "logit_div", 0.0) or 0.0) != 0.0:      # the reference evaluates logpdf(pa, s, a) through pa's own logit_conversion (policies.jl:128-135); the kernel knows the plain softmax only
            raise L.CruxError(L.EUNSUP, "steps!: :importance_weight with a nominal DiscreteNetwork whose logit conversion is not the plain softmax (logit_div = %g) is not supported" % pa.logit_div)
        head = L.HEAD["categorical"] if isinstance(pa, DiscreteNetwork) else L.HEAD["gaussian"]
        C_ = buffer.capacity; n1 = min(Nsteps, C_ - first)
        ctx.check(lib.crux_importance_weight_rows(buffer.h, pa.h, head, int(first), int(n1)))
        if n1 < Nsteps:
            ctx.check(lib.crux_importance_weight_rows(buffer.h, pa.h, head, 0, int(Nsteps - n1)))
    if any(buffer.haskey(k) for k in ("fwd_importance_weight", "cum_importance_weight", "rev_importance_weight")):
        ctx.check(lib.crux_fill_importance_weights_rows(buffer.h, int(first), int(Nsteps), int(Nsteps // sampler.n_envs), 1 if reset else 0))
    if buffer.haskey("traj_importance_weight"):
        _fill_traj_weights(sampler, buffer, first, Nsteps, reset)


def _fill_traj_weights(sampler, buffer, first, Nsteps, reset):
    if True:
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


def copy_buffer(b):
    """deepcopy(b::ExperienceBuffer): same columns, same rows, same order."""
    out = buffer_like(b, capacity=b.capacity)
    if len(b):
        out.push_(b, ids=np.arange(1, len(b) + 1))
    return out


def shuffle_device_(b, seed, counter):
    """shuffle!(b) with the library's permutation stream (crux_rng.h), composed and applied on the device."""
    b.ctx.check(b.ctx.lib.crux_buffer_shuffle(b.h, int(seed), int(counter))); return b


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
