"""Pins the CPU oracle (oracle/crux_oracle.c) against the reference's own known answers, its recorded transitions and
independent float64 autograd. Runs without a GPU. Citations are into /root/reference (sisl/Crux.jl)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from crux_jl_amd import _lib as L

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
lib = O.lib()


# ---- test/experience_buffer_tests.jl:23-28  circular indices ---------------------------------------------------
def circ(start1, n, C_):
    out = np.empty(n, np.int64); lib.orc_circ_inds(start1 - 1, n, C_, O.vpz(out)); return out + 1


def test_circular_indices_reference_kats():
    assert list(circ(4, 60, 100)) == list(range(4, 64))
    assert list(circ(1, 100, 100)) == list(range(1, 101))
    assert list(circ(1, 101, 100)) == list(range(1, 101)) + [1]
    assert list(circ(1, 120, 100)) == list(range(1, 101)) + list(range(1, 21))
    assert list(circ(90, 20, 100)) == list(range(90, 101)) + list(range(1, 10))


def _data(n, s=2.0, obs=2, act=1, a=1, sp=1.0, r=1.0, done=0):
    return {"s": np.full((obs, n), s, np.float32), "a": np.full((act, n), a, np.bool_), "sp": np.full((obs, n), sp, np.float32),
            "r": np.full((1, n), r, np.float32), "done": np.full((1, n), done, np.bool_)}


def last_n(b, N):
    out = np.empty(max(N, 1), np.int64); n = lib.orc_buffer_last_n_indices(b.h, N, O.vpz(out)); return list(out[:n] + 1)


# ---- test/experience_buffer_tests.jl:32-51  get_last_N_indices -------------------------------------------------
def test_get_last_n_indices_reference_kats():
    b = O.OBuffer(2, 1, L.ACTION_CONTINUOUS, 100)
    d = _data(50); d["a"] = d["a"].astype(np.float32)
    b.push(d)
    assert last_n(b, 10) == list(range(41, 51)) and last_n(b, 1) == [50]
    assert last_n(b, 50) == list(range(1, 51)) == last_n(b, 51) == last_n(b, 1000)
    b.push(d); b.push(d)
    assert last_n(b, 10) == list(range(41, 51)) and last_n(b, 1) == [50] and last_n(b, 50) == list(range(1, 51))
    assert last_n(b, 51) == [100] + list(range(1, 51))
    assert last_n(b, 100) == list(range(51, 101)) + list(range(1, 51)) == last_n(b, 1000)


# ---- test/experience_buffer_tests.jl:121-147  push! ------------------------------------------------------------
def test_push_semantics_reference_kats():
    b = O.OBuffer(2, 4, L.ACTION_DISCRETE, 100)
    assert len(b) == 0
    b.push(_data(1, act=4))
    assert len(b) == 1 and (b["s"] == 2).all() and b["a"].all() and (b["sp"] == 1).all() and (b["r"] == 1).all() and not b["done"].any()
    rng = np.random.default_rng(0)
    d = {"s": np.full((2, 3), 3, np.float32), "a": rng.random((4, 3)) < 0.5, "sp": np.full((2, 3), 5, np.float32),
         "r": np.full((1, 3), 6, np.float32), "done": np.ones((1, 3), np.bool_)}
    b.push(d)
    assert len(b) == 4 and (b["s"][:, 1:] == 3).all() and (b["a"][:, 1:] == d["a"]).all() and (b["r"][:, 1:] == 6).all() and b["done"][:, 1:].all()
    b.push_buffer(b)                       # push!(b, b): self-push doubles (collect before copyto!, :250-252)
    assert len(b) == 8
    for k in b.keys():
        assert (b[k][:, :4] == b[k][:, 4:8]).all()


def test_mdp_data_initial_values():      # test/experience_buffer_tests.jl:8-20
    b = O.OBuffer(3, 4, L.ACTION_DISCRETE, 100, ["weight", "t", "advantage", "return", "logprob"])
    assert (b.col("s") == 0).all() and b.col("s").shape == (3, 100) and b.col("a").shape == (4, 100) and b.col("a").dtype == np.bool_
    assert (b.col("weight") == 1).all() and (b.col("return") == 0).all() and (b.col("t") == 0).all()
    assert not O.OBuffer(3, 4, L.ACTION_DISCRETE, 10).haskey("return")


def test_split_batches_reference_kats():  # test/experience_buffer_tests.jl:177-180
    def sb(N, fr):
        fr = np.asarray(fr, np.float64); out = np.empty(len(fr), np.int64); lib.orc_split_batches(N, O.vpz(fr), len(fr), O.vpz(out)); return list(out)
    assert sb(100, [0.5, 0.5]) == [50, 50] and sb(100, [1.0]) == [100] and sb(100, [1 / 3, 1 / 3, 1 / 3]) == [34, 33, 33]


# ---- test/experience_buffer_tests.jl:193-205  update_priorities! -----------------------------------------------
def test_update_priorities_reference_kats():
    b = O.OBuffer(2, 4, L.ACTION_DISCRETE, 50, prioritized=True, alpha=np.float32(0.6))
    assert b.haskey("weight") and (b.col("weight") == 1).all()
    I = np.array([0, 1, 2], np.int64); v = np.array([1.0, 2.0, 3.0])
    O.chk(lib.orc_per_update(b.h, O.vpz(I), O.vpz(v), 1, 3))
    pr = np.empty(50, np.float32); mx, mn = C.c_float(), C.c_float()
    O.chk(lib.orc_per_get(b.h, O.vpz(pr), C.byref(mx), C.byref(mn), None))
    assert mx.value == 3.0
    # exact Float32 values of (k + eps32)^0.6f0
    assert [float(x) for x in pr[:3]] == [1.0000001192092896, 1.5157166719436646, 1.9331821203231812]
    d = _data(3, act=4)
    b.push(d); b.push(d)
    O.chk(lib.orc_per_get(b.h, O.vpz(pr), C.byref(mx), C.byref(mn), None))
    assert mx.value == 3.0 and np.allclose(pr[:6], np.float32(3.0) ** np.float32(0.6), rtol=1e-6)


def test_pairwise_cumsum_matches_float64_and_leaf_form():
    rng = np.random.default_rng(1)
    v = (np.abs(rng.standard_normal(100_000)) + 1e-3).astype(np.float32); out = np.empty_like(v)
    lib.orc_pairwise_cumsum_f32(O.vpz(v), v.size, O.vpz(out))
    assert np.allclose(out, np.cumsum(v.astype(np.float64)), rtol=3e-6)
    v2 = v[:100].copy(); out2 = np.empty_like(v2); lib.orc_pairwise_cumsum_f32(O.vpz(v2), 100, O.vpz(out2))
    seq = np.empty_like(v2); seq[0] = v2[0]; acc = np.float32(0)
    for i in range(1, 100):                    # n < 128: c[i] = v[1] + (v[2] + ... + v[i]) -- the leaf form of accumulate_pairwise!
        acc = np.float32(acc + v2[i]) if i > 1 else v2[1]; seq[i] = np.float32(v2[0] + acc)
    assert (out2 == seq).all()


# ---- GAE / returns KAT derived from src/sampler.jl:262-281 ----------------------------------------
def _jl_mapreduce(a, f=lambda x: x):
    """Base.mapreduce_impl(f, +, A, ifirst, ilast, 1024) (reduce.jl) spelled out independently in numpy Float32 scalars: at most 1024 elements from the left, otherwise split at
    ifirst + (ilast - ifirst) >> 1"""
    def rec(i0, i1):
        if i0 == i1:
            return np.float32(f(a[i0]))
        if i1 - i0 < 1024:
            v = np.float32(f(a[i0])) + np.float32(f(a[i0 + 1]))
            for i in range(i0 + 2, i1 + 1):
                v = np.float32(v + np.float32(f(a[i])))
            return v
        mid = i0 + ((i1 - i0) >> 1)
        return np.float32(rec(i0, mid) + rec(mid + 1, i1))
    return rec(0, len(a) - 1)


@pytest.mark.parametrize("n", [1, 2, 3, 1000, 1024, 1025, 2049, 3000, 65536, 100003])
def test_julia_order_reductions(n):
    """[3P] the oracle's mean / std are Julia's: Statistics.mean = sum(A) / length(A) with Base's pairwise Float32 sum (block 1024), Statistics.std =
    sqrt(centralize_sumabs2(A, m) / (n - 1)) with the same scheme -- not a Float64 accumulation rounded at the end"""
    rng = np.random.default_rng(n); a = (rng.standard_normal(n) * 3 + 0.7).astype(np.float32)
    s = _jl_mapreduce(a)
    assert np.float32(O.lib().orc_jl_sum_f32(O.vpz(a), n)) == s
    m = np.float32(s / np.float32(n))
    assert np.float32(O.lib().orc_jl_mean_f32(O.vpz(a), n)) == m
    if n >= 2:
        ss = _jl_mapreduce(a, lambda x: np.float32(np.float32(x - m) * np.float32(x - m)))
        assert np.float32(O.lib().orc_jl_std_f32(O.vpz(a), n)) == np.float32(np.sqrt(np.float32(ss / np.float32(n - 1))))
    if n >= 65536:      # and it is not the left-to-right Float32 sum, nor the Float64 sum rounded (what the oracle did before round 5) -- the three differ at this length
        seq = np.float32(0)
        for x in a[:4096]:
            seq = np.float32(seq + x)
        assert abs(float(s) - float(a.astype(np.float64).sum())) < 1e-3 * np.sqrt(n)


def test_gae_and_returns_kat():
    r = np.full(5, 6, np.float32); d = np.ones(5, np.uint8); V = np.zeros(5, np.float32); adv = np.zeros(5, np.float32); ret = np.zeros(5, np.float32)
    lib.orc_gae_range(O.vpz(r), O.vpz(d), O.vpz(V), O.vpz(V), 0, 4, 0.9, 0.7, O.vpz(adv))
    lib.orc_returns_range(O.vpz(r), 0, 4, 0.7, O.vpz(ret))
    assert [float(x) for x in adv[::-1]] == [6.0, 9.779999732971191, 12.161399841308594, 13.66168212890625, 14.60685920715332]
    assert [float(x) for x in ret[::-1]] == [6.0, 10.199999809265137, 13.139999389648438, 15.197999000549316, 16.638599395751953]


def test_gae_bootstraps_through_truncation():       # done=false => + gamma*V(sp)
    r = np.array([1, 1], np.float32); d = np.zeros(2, np.uint8); Vs = np.array([0.5, 0.25], np.float32); Vsp = np.array([0.25, 2.0], np.float32)
    adv = np.zeros(2, np.float32); lib.orc_gae_range(O.vpz(r), O.vpz(d), O.vpz(Vs), O.vpz(Vsp), 0, 1, 0.95, 0.99, O.vpz(adv))
    a1 = np.float32(np.float32(1 + np.float32(0.99) * np.float32(2.0)) - np.float32(0.25))
    assert adv[1] == a1
    c = np.float32(np.float32(0.95) * np.float32(0.99))
    a0 = np.float32(np.float32(np.float32(c * a1) + np.float32(1)) + np.float32(np.float32(0.99) * np.float32(0.25))) - np.float32(0.5)
    assert adv[0] == np.float32(a0)


# ---- recorded transitions shipped with the reference (examples/il/expert_data/*.bson) ---------------------------
def test_cartpole_dynamics_match_recordings():
    z = np.load(os.path.join(GOLD, "cartpole_transitions.npz"))
    s = np.ascontiguousarray(z["s"].T.astype(np.float64)); a = np.ascontiguousarray(z["a"].T.astype(np.uint8)); n = s.shape[0]
    ns = np.zeros_like(s); obs = np.zeros((n, 4), np.float32); rr = np.zeros(n, np.float32); dd = np.zeros(n, np.uint8)
    O.chk(lib.orc_env_step_host(L.ENV["cartpole"], n, O.vpz(s), O.vpz(a), None, O.vpz(ns), O.vpz(obs), O.vpz(rr), O.vpz(dd)))
    assert np.abs(obs - z["sp"].T).max() <= 1e-6 and (rr == z["r"][0]).all() and (dd == z["done"][0]).all()
    # within an episode the recording's next state is the following row's state
    t = z["t"][0]; same = t[1:] == t[:-1] + 1
    assert (z["sp"].T[:-1][same] == z["s"].T[1:][same]).all()


def test_pendulum_dynamics_match_recordings():
    z = np.load(os.path.join(GOLD, "pendulum_transitions.npz"))
    s = np.ascontiguousarray(z["s"].T.astype(np.float64)); a = np.ascontiguousarray(z["a"].T.astype(np.float32)); n = s.shape[0]
    ns = np.zeros_like(s); obs = np.zeros((n, 3), np.float32); rr = np.zeros(n, np.float32); dd = np.zeros(n, np.uint8)
    O.chk(lib.orc_env_step_host(L.ENV["pendulum"], n, O.vpz(s), O.vpz(a), None, O.vpz(ns), O.vpz(obs), O.vpz(rr), O.vpz(dd)))
    sp = z["sp"].T
    assert np.abs(ns[:, 1] - sp[:, 1]).max() < 2e-6                      # angular velocity
    assert np.abs(rr - z["r"][0]).max() < 5e-6                           # reward -(th_norm^2 + .1 thdot^2 + .001 u^2)
    wrap = lambda x: (x + np.pi) % (2 * np.pi) - np.pi
    ok = np.abs(wrap(ns[:, 0] - sp[:, 0])) < 2e-6                        # angle, modulo the recorder's wrap convention
    assert ok.mean() > 0.97
    assert not dd.any()


# ---- schedules: test/util_tests.jl:56-86 -------------------------------------------------------------------------
def test_linear_decay_schedule():
    f = lambda i: lib.orc_linear_decay(1.0, 0.1, 10, i)
    assert f(0) == 1.0 and abs(f(5) - 0.55) < 1e-12 and f(10) == pytest.approx(0.1) and f(11) == 0.1 and f(1000) == 0.1


# ---- randomness spec -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 17, 128, 1000, 65536])
def test_feistel_permutation_is_a_bijection(n):
    out = np.empty(n, np.int64); lib.orc_perm(7, 3, n, O.vpz(out))
    assert sorted(out.tolist()) == list(range(n))
    out2 = np.empty(n, np.int64); lib.orc_perm(7, 4, n, O.vpz(out2))
    assert n < 17 or (out != out2).any()


def test_philox_known_answer():
    # Philox4x32-10 KAT from the Random123 distribution: counter = key = 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    out = np.empty(4, np.uint32); lib.orc_philox(0, 0, 0, 0, O.vpz(out))
    assert [hex(int(x)) for x in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


# ---- gradients / optimiser against independent float64 autograd -------------------------------------------------
torch = pytest.importorskip("torch")


def _torch_net(o, dtype=torch.float64):
    p = torch.tensor(o.params.copy(), dtype=dtype, requires_grad=True)
    return p


def _fwd(p, dims, acts, x):
    off = 0; h = x
    for l in range(len(acts)):
        i, o = dims[l], dims[l + 1]
        W = p[off:off + i * o].reshape(i, o).T; off += i * o      # column-major (out,in)
        b = p[off:off + o]; off += o
        h = W @ h + b[:, None]
        h = torch.relu(h) if acts[l] == "relu" else torch.tanh(h) if acts[l] == "tanh" else h
    return h, off


def _fill(ob, n, rng, act_kind, ad):
    d = {"s": rng.standard_normal((ob.obs_dim, n)).astype(np.float32), "sp": rng.standard_normal((ob.obs_dim, n)).astype(np.float32),
         "r": rng.standard_normal((1, n)).astype(np.float32), "done": rng.random((1, n)) < 0.1,
         "return": rng.standard_normal((1, n)).astype(np.float32), "logprob": (-0.7 + 0.1 * rng.standard_normal((1, n))).astype(np.float32),
         "advantage": rng.standard_normal((1, n)).astype(np.float32)}
    if act_kind == L.ACTION_DISCRETE:
        a = np.zeros((ad, n), np.bool_); a[rng.integers(0, ad, n), np.arange(n)] = True; d["a"] = a
    else:
        d["a"] = rng.standard_normal((ad, n)).astype(np.float32)
    ob.push(d); return d


def _cfg(loss, head, eps=0.2, lp=1.0, le=0.1):
    c = L.TrainCfg(); c.loss, c.head, c.batch_size, c.epochs = L.LOSS[loss], L.HEAD[head], 128, 1
    c.eps_clip, c.lambda_p, c.lambda_e, c.target_kl = eps, lp, le, -1.0
    return c


@pytest.mark.parametrize("acts", [["relu", "relu", "identity"], ["tanh", "tanh", "identity"]])
def test_ppo_categorical_gradient_vs_float64_autograd(acts):
    rng = np.random.default_rng(3); dims = [4, 16, 16, 3]; n = 40
    o = O.OMlp(dims, acts).init_glorot(5); o.params[:] += 0.05 * rng.standard_normal(o.n).astype(np.float32)
    ob = O.OBuffer(4, 3, L.ACTION_DISCRETE, n, ["return", "logprob", "advantage"]); d = _fill(ob, n, rng, L.ACTION_DISCRETE, 3)
    ids = np.arange(n, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32); cfg = _cfg("ppo", "categorical")
    O.chk(lib.orc_loss_grad(o.h, ob.h, C.byref(cfg), O.vpz(ids), n, O.vpz(info)))
    p = _torch_net(o); z, _ = _fwd(p, dims, acts, torch.tensor(d["s"], dtype=torch.float64))
    pr = torch.softmax(z, 0); a = torch.tensor(d["a"], dtype=torch.float64)
    newlp = torch.log((pr * a).sum(0)); r = torch.exp(newlp - torch.tensor(d["logprob"][0], dtype=torch.float64))
    A = torch.tensor(d["advantage"][0], dtype=torch.float64)
    p_loss = -torch.minimum(r * A, torch.clamp(r, 0.8, 1.2) * A).mean()
    ent = -(pr * torch.log(pr + float(np.finfo(np.float32).eps))).sum(0)
    loss = 1.0 * p_loss + 0.1 * (-ent.mean()); loss.backward()
    g = p.grad.numpy()
    assert abs(info[L.INFO["loss"]] - loss.item()) < 1e-5
    assert np.abs(o.grads - g).max() < 1e-5 * max(1.0, np.abs(g).max())
    assert abs(info[L.INFO["grad_norm"]] - np.linalg.norm(g)) < 1e-5
    assert abs(info[L.INFO["kl"]] - (torch.tensor(d["logprob"][0], dtype=torch.float64) - newlp).mean().item()) < 1e-6
    assert abs(info[L.INFO["entropy"]] - ent.mean().item()) < 1e-6
    assert info[L.INFO["clip_fraction"]] == pytest.approx(((r > 1.2) | (r < 0.8)).double().mean().item())


def test_ppo_gaussian_gradient_vs_float64_autograd():
    rng = np.random.default_rng(4); dims = [5, 16, 16, 2]; acts = ["relu", "relu", "identity"]; n = 32
    o = O.OMlp(dims, acts, 2).init_glorot(6, 0, -0.3)
    ob = O.OBuffer(5, 2, L.ACTION_CONTINUOUS, n, ["return", "logprob", "advantage"]); d = _fill(ob, n, rng, L.ACTION_CONTINUOUS, 2)
    ob.col("logprob")[:] = -2.5 + 0.2 * rng.standard_normal((1, n)).astype(np.float32); d["logprob"] = ob["logprob"]
    ids = np.arange(n, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32); cfg = _cfg("ppo", "gaussian")
    O.chk(lib.orc_loss_grad(o.h, ob.h, C.byref(cfg), O.vpz(ids), n, O.vpz(info)))
    p = _torch_net(o); mu, off = _fwd(p, dims, acts, torch.tensor(d["s"], dtype=torch.float64)); ls = p[off:off + 2]
    a = torch.tensor(d["a"], dtype=torch.float64); s2 = torch.exp(ls) ** 2
    newlp = (-((a - mu) ** 2) / (2 * s2[:, None]) - 0.9189385332046727 - ls[:, None]).sum(0)
    r = torch.exp(newlp - torch.tensor(d["logprob"][0], dtype=torch.float64)); A = torch.tensor(d["advantage"][0], dtype=torch.float64)
    p_loss = -torch.minimum(r * A, torch.clamp(r, 0.8, 1.2) * A).mean(); H = 1.4189385332046727 + ls.sum()     # scalar entropy (policies.jl:348)
    loss = p_loss + 0.1 * (-H); loss.backward(); g = p.grad.numpy()
    assert abs(info[L.INFO["loss"]] - loss.item()) < 2e-5 and np.abs(o.grads - g).max() < 2e-5 * max(1.0, np.abs(g).max())


def test_value_mse_and_td_gradients_vs_float64_autograd():
    rng = np.random.default_rng(5); dims = [4, 16, 16, 1]; acts = ["relu", "relu", "identity"]; n = 48
    o = O.OMlp(dims, acts).init_glorot(7)
    ob = O.OBuffer(4, 3, L.ACTION_DISCRETE, n, ["return", "logprob", "advantage", "weight"]); d = _fill(ob, n, rng, L.ACTION_DISCRETE, 3)
    ids = np.arange(n, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32); cfg = _cfg("value_mse", "deterministic")
    O.chk(lib.orc_loss_grad(o.h, ob.h, C.byref(cfg), O.vpz(ids), n, O.vpz(info)))
    p = _torch_net(o); v, _ = _fwd(p, dims, acts, torch.tensor(d["s"], dtype=torch.float64))
    loss = ((v[0] - torch.tensor(d["return"][0], dtype=torch.float64)) ** 2).mean(); loss.backward()
    assert abs(info[0] - loss.item()) < 1e-5 and np.abs(o.grads - p.grad.numpy()).max() < 1e-5 * max(1, np.abs(p.grad.numpy()).max())
    # td_loss: Q(s,a) = sum(Q .* onehot); mse against y (src/utils.jl:76-87), one Adam step compared with a float64 restatement
    dq = [4, 16, 16, 3]; q = O.OMlp(dq, acts).init_glorot(8).adam_init(float(np.float32(3e-4)))
    y = rng.standard_normal(n).astype(np.float32); p0 = q.params.copy(); info2 = np.zeros(L.INFO_N, np.float32)
    O.chk(lib.orc_td_step(q.h, ob.h, O.vpz(y), 0, O.vpz(info2)))
    pt = torch.tensor(p0, dtype=torch.float64, requires_grad=True); Q, _ = _fwd(pt, dq, acts, torch.tensor(d["s"], dtype=torch.float64))
    Qa = (Q * torch.tensor(d["a"], dtype=torch.float64)).sum(0); l2 = ((Qa - torch.tensor(y, dtype=torch.float64)) ** 2).mean(); l2.backward()
    g = pt.grad.numpy(); eta = float(np.float32(3e-4))
    m = 0.1 * g; v2 = 0.001 * g * g; step = m / (1 - 0.9) / (np.sqrt(v2 / (1 - 0.999)) + 1e-8) * eta     # Flux Adam, first step
    assert abs(info2[0] - l2.item()) < 1e-5 and np.abs(q.params - (p0 - step)).max() < 2e-7


def test_adam_matches_flux_semantics_over_many_steps():
    rng = np.random.default_rng(6); o = O.OMlp([3, 8, 2], ["relu", "identity"]).init_glorot(1).adam_init(float(np.float32(3e-4)))
    p = o.params.astype(np.float64).copy(); m = np.zeros_like(p); v = np.zeros_like(p); bp = [0.9, 0.999]; eta = float(np.float32(3e-4))
    p32 = o.params.copy()
    for _ in range(50):
        g = rng.standard_normal(o.n).astype(np.float32); o.grads[:] = g; O.chk(lib.orc_adam_apply(o.h, 1.0))
        g64 = g.astype(np.float64)
        m = (0.9 * m + (1 - 0.9) * g64).astype(np.float32).astype(np.float64); v = (0.999 * v + ((1 - 0.999) * g64) * g64).astype(np.float32).astype(np.float64)
        d = (m / (1 - bp[0]) / (np.sqrt(v / (1 - bp[1])) + 1e-8) * eta).astype(np.float32); p32 = (p32 - d).astype(np.float32)
        bp = [bp[0] * 0.9, bp[1] * 0.999]
    assert np.array_equal(o.params, p32)
    _, _, obp = o.adam_state(); assert obp[0] == pytest.approx(0.9 ** 51) and obp[1] == pytest.approx(0.999 ** 51)


def test_batch_train_early_stopping_uses_latest_minibatch_kl():
    """aggregate_info(minibatch_infos) aliases one dict, so the first minibatch whose KL exceeds target_kl
    ends training; batches_trained counts it (its update IS applied)."""
    rng = np.random.default_rng(7); dims = [4, 16, 16, 2]; acts = ["relu", "relu", "identity"]; n = 256
    o = O.OMlp(dims, acts).init_glorot(9).adam_init(0.05)       # large step so KL moves fast
    ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, n, ["return", "logprob", "advantage"]); _fill(ob, n, rng, L.ACTION_DISCRETE, 2)
    cfg = _cfg("ppo", "categorical"); cfg.batch_size, cfg.epochs, cfg.target_kl, cfg.shuffle_seed = 32, 5, 0.01, 3
    info = np.zeros(L.INFO_N, np.float32); ep = np.zeros((5, L.INFO_N), np.float32)
    O.chk(lib.orc_batch_train(o.h, ob.h, C.byref(cfg), None, O.vpz(info), O.vpz(ep)))
    nb, ne = int(info[L.INFO["batches_trained"]]), int(info[L.INFO["epochs_run"]])
    assert ne == 1 and 1 <= nb < 8 and ep[0, L.INFO["kl"]] > 0.01 and info[L.INFO["kl"]] == ep[0, L.INFO["kl"]]
    # max_batches (training.jl:45,50)
    o2 = O.OMlp(dims, acts).init_glorot(9).adam_init(1e-3); cfg.target_kl, cfg.max_batches = -1.0, 11
    O.chk(lib.orc_batch_train(o2.h, ob.h, C.byref(cfg), None, O.vpz(info), None))
    assert int(info[L.INFO["batches_trained"]]) == 11 and int(info[L.INFO["epochs_run"]]) == 2


def test_ragged_last_minibatch_and_permutation_semantics():
    rng = np.random.default_rng(8); n = 70
    ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, n, ["return", "logprob", "advantage"]); d = _fill(ob, n, rng, L.ACTION_DISCRETE, 2)
    perm = rng.permutation(n) + 1; ob.permute(perm)
    assert (ob["s"] == d["s"][:, perm - 1]).all() and (ob["a"] == d["a"][:, perm - 1]).all()      # new[:,j] = old[:,perm[j]]
    o = O.OMlp([4, 8, 8, 1], ["relu", "relu", "identity"]).init_glorot(2).adam_init(1e-3)
    cfg = _cfg("value_mse", "deterministic"); cfg.batch_size, cfg.epochs = 32, 2
    info = np.zeros(L.INFO_N, np.float32); O.chk(lib.orc_batch_train(o.h, ob.h, C.byref(cfg), None, O.vpz(info), None))
    assert int(info[L.INFO["batches_trained"]]) == 6           # partition(1:70, 32) -> 32, 32, 6 per epoch (training.jl:40)


# ---- SAC restatement: the analytic reverse pass of the oracle vs central differences of its own losses -----------------
def _sac_oracle_setup(rng, od=3, ad=2, B=24):
    adims, qdims = [od, 16, ad], [od + ad, 16, 1]
    oa = O.OMlp(adims, ["tanh", "identity"], ad).init_glorot(3, 0, -0.4)
    o1, o2 = O.OMlp(qdims, ["tanh", "identity"]).init_glorot(3, 1), O.OMlp(qdims, ["tanh", "identity"]).init_glorot(3, 2)
    ola = O.OMlp([0], [], 1); ola.params[:] = np.log(np.float32(0.6))
    ob = O.OBuffer(od, ad, L.ACTION_CONTINUOUS, B)
    ob.push({"s": rng.normal(0, 1, (od, B)).astype(np.float32), "a": rng.uniform(-1, 1, (ad, B)).astype(np.float32), "sp": rng.normal(0, 1, (od, B)).astype(np.float32),
             "r": rng.normal(0, 1, (1, B)).astype(np.float32), "done": rng.random((1, B)) < 0.2, "episode_end": np.zeros((1, B), bool)})
    for o in (oa, o1, o2, ola):
        o.adam_init(0.0)                                           # eta = 0: train! evaluates loss and gradient, the update is a no-op
    return oa, o1, o2, ola, ob, B


def _fd(loss_at, params, idx, h=2e-3):
    out = []
    for k in idx:
        keep = params[k]
        params[k] = keep + h; lp = loss_at()
        params[k] = keep - h; lm = loss_at()
        params[k] = keep; out.append((lp - lm) / (2 * h))
    return np.array(out)


def test_sac_oracle_gradients_match_finite_differences():
    rng = np.random.default_rng(8)
    oa, o1, o2, ola, ob, B = _sac_oracle_setup(rng)
    info, ol = np.zeros(L.INFO_N, np.float32), O.lib()
    # actor loss (sac.jl:34-40): gradient w.r.t. mean-network weights and logSigma
    def actor_loss():
        O.chk(ol.orc_sac_actor_step(oa.h, o1.h, o2.h, ola.h, ob.h, 5, 9, O.vpz(info))); return float(info[L.INFO["loss"]])
    actor_loss(); g = oa.grads.copy()
    idx = list(rng.choice(oa.n - 2, 8, replace=False)) + [oa.n - 2, oa.n - 1]
    assert np.allclose(g[idx], _fd(actor_loss, oa.params, idx), rtol=3e-2, atol=2e-3)
    # critic loss (utils.jl:89-96)
    y = rng.normal(0, 1, B).astype(np.float32)
    def critic_loss():
        O.chk(ol.orc_double_q_step(o1.h, o2.h, ob.h, O.vpz(y), 0, O.vpz(info))); return float(info[L.INFO["loss"]])
    critic_loss(); g1 = o1.grads.copy()
    idx = list(rng.choice(o1.n, 8, replace=False))
    assert np.allclose(g1[idx], _fd(critic_loss, o1.params, idx), rtol=3e-2, atol=2e-3)
    # temperature loss (sac.jl:45-52): d/dlog_alpha
    def temp_loss():
        O.chk(ol.orc_sac_temp_step(oa.h, ola.h, ob.h, -2.0, 5, 10, O.vpz(info))); return float(info[L.INFO["loss"]])
    temp_loss(); gl = ola.grads.copy()
    assert np.allclose(gl[:1], _fd(temp_loss, ola.params, [0]), rtol=2e-2, atol=1e-3)


def test_sac_target_restatement_against_numpy():
    """sac_target (sac.jl:4-9) recomputed in float64 numpy from the noise definition in include/crux_rng.h."""
    rng = np.random.default_rng(4)
    oa, o1, o2, ola, ob, B = _sac_oracle_setup(rng)
    y = np.empty(B, np.float32)
    O.chk(O.lib().orc_sac_target(oa.h, o1.h, o2.h, ola.h, ob.h, 0.9, 5, 3, O.vpz(y)))
    sp, r, done = ob["sp"].astype(np.float64), ob["r"][0].astype(np.float64), ob["done"][0]
    mu = oa.forward(ob["sp"]).astype(np.float64); ls = oa.params[-2:].astype(np.float64); sg = np.exp(ls)
    eps = np.empty_like(mu); out4 = (C.c_uint32 * 4)()
    for j in range(B):
        for d in range(2):
            O.lib().orc_philox(5, 3, j * 2 + d, 2, out4)                                  # purpose 2 = CRUX_RNG_NOISE
            u1 = (((out4[0] << 32) | out4[1]) >> 11) * 1.1102230246251565e-16
            u2 = (((out4[2] << 32) | out4[3]) >> 11) * 1.1102230246251565e-16
            eps[d, j] = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    a = eps * sg[:, None] + mu
    lp = (-((a - mu) ** 2) / (2 * sg[:, None] ** 2) - 0.9189385332046727 - ls[:, None]).sum(axis=0)
    sa = np.vstack([sp, a]).astype(np.float32)
    qmin = np.minimum(o1.forward(sa)[0], o2.forward(sa)[0]).astype(np.float64)
    yref = r + 0.9 * (1.0 - done) * (qmin - np.exp(np.float64(ola.params[0])) * lp)
    assert np.abs(y - yref).max() < 2e-5 * max(1.0, np.abs(yref).max())
    assert np.array_equal(y[done], ob["r"][0][done])


def _noise(seed, counter, ad, B):
    """the standard normal draws of exploration(pi, s) as include/crux_rng.h defines them (purpose 2 = CRUX_RNG_NOISE, Box-Muller on two 53-bit uniforms)"""
    eps = np.empty((ad, B)); out4 = (C.c_uint32 * 4)()
    for j in range(B):
        for d in range(ad):
            O.lib().orc_philox(seed, counter, j * ad + d, 2, out4)
            u1 = (((out4[0] << 32) | out4[1]) >> 11) * 1.1102230246251565e-16
            u2 = (((out4[2] << 32) | out4[3]) >> 11) * 1.1102230246251565e-16
            eps[d, j] = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return eps


def test_sac_losses_vs_float64_autograd():
    """The SAC twins of the oracle against torch float64 autograd of the reference formulas written out independently: sac_actor_loss (sac.jl:34-40: the
    reparameterised action and its logpdf both carry gradient), double_Q_loss (utils.jl:89-96) and sac_temp_loss (sac.jl:45-52). Pins R26 beyond the
    finite-difference self-check above."""
    rng = np.random.default_rng(8)
    oa, o1, o2, ola, ob, B = _sac_oracle_setup(rng)
    od, ad = 3, 2; adims, qdims, aacts, qacts = [od, 16, ad], [od + ad, 16, 1], ["tanh", "identity"], ["tanh", "identity"]
    info, ol = np.zeros(L.INFO_N, np.float32), O.lib()
    S = torch.tensor(ob["s"], dtype=torch.float64)
    # ---- actor
    O.chk(ol.orc_sac_actor_step(oa.h, o1.h, o2.h, ola.h, ob.h, 5, 9, O.vpz(info))); ga, la = oa.grads.copy(), float(info[L.INFO["loss"]])
    pa, p1, p2 = _torch_net(oa), _torch_net(o1), _torch_net(o2); alpha = float(np.exp(np.float64(ola.params[0])))
    mu, off = _fwd(pa, adims, aacts, S); ls = pa[off:off + ad]; eps = torch.tensor(_noise(5, 9, ad, B), dtype=torch.float64)
    a = mu + torch.exp(ls)[:, None] * eps                                                                   # exploration(pi, s): policies.jl:338-344
    lp = (-((a - mu) ** 2) / (2 * torch.exp(ls)[:, None] ** 2) - 0.9189385332046727 - ls[:, None]).sum(0)     # gaussian_logpdf (:333-336)
    sa = torch.cat([S, a], 0)
    q = torch.minimum(_fwd(p1, qdims, qacts, sa)[0][0], _fwd(p2, qdims, qacts, sa)[0][0])
    loss = (alpha * lp - q).mean(); loss.backward()                                                          # sac.jl:39
    assert abs(la - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
    assert np.abs(ga - pa.grad.numpy()).max() < 3e-6 * max(1.0, np.abs(pa.grad.numpy()).max())
    assert abs(float(info[L.INFO["entropy"]]) - (-lp.mean().item())) < 1e-5                                    # sac.jl:36-38 logs -mean(logpdf)
    # ---- critics
    y = rng.normal(0, 1, B).astype(np.float32)
    O.chk(ol.orc_double_q_step(o1.h, o2.h, ob.h, O.vpz(y), 0, O.vpz(info))); g1, g2, lc = o1.grads.copy(), o2.grads.copy(), float(info[L.INFO["loss"]])
    p1, p2 = _torch_net(o1), _torch_net(o2); sa0 = torch.tensor(np.vstack([ob["s"], ob["a"]]), dtype=torch.float64); yt = torch.tensor(y, dtype=torch.float64)
    l = 0.5 * (((_fwd(p1, qdims, qacts, sa0)[0][0] - yt) ** 2).mean() + ((_fwd(p2, qdims, qacts, sa0)[0][0] - yt) ** 2).mean()); l.backward()   # utils.jl:89-96
    assert abs(lc - l.item()) < 1e-5 * max(1.0, abs(l.item()))
    assert np.abs(g1 - p1.grad.numpy()).max() < 3e-6 * max(1.0, np.abs(p1.grad.numpy()).max()) and np.abs(g2 - p2.grad.numpy()).max() < 3e-6 * max(1.0, np.abs(p2.grad.numpy()).max())
    # ---- temperature (sac.jl:45-52): -mean(alpha * (logpdf + H_target)), alpha = exp(log_alpha), logpdf held constant
    O.chk(ol.orc_sac_temp_step(oa.h, ola.h, ob.h, -2.0, 5, 10, O.vpz(info))); gl, lt = float(ola.grads[0]), float(info[L.INFO["loss"]])
    pa = _torch_net(oa); mu, off = _fwd(pa, adims, aacts, S); ls = pa[off:off + ad].detach(); eps = torch.tensor(_noise(5, 10, ad, B), dtype=torch.float64)
    a = mu.detach() + torch.exp(ls)[:, None] * eps
    lp = (-((a - mu.detach()) ** 2) / (2 * torch.exp(ls)[:, None] ** 2) - 0.9189385332046727 - ls[:, None]).sum(0)
    log_alpha = torch.tensor(np.float64(ola.params[0]), requires_grad=True)
    lt_ref = -(torch.exp(log_alpha) * (lp + (-2.0))).mean(); lt_ref.backward()
    assert abs(lt - lt_ref.item()) < 1e-5 * max(1.0, abs(lt_ref.item())) and abs(gl - log_alpha.grad.item()) < 3e-6 * max(1.0, abs(log_alpha.grad.item()))


def test_dpg_losses_vs_float64_autograd():
    """ddpg_actor_loss (-mean(Q(s, mu(s))), ddpg.jl:26) and td_loss over vcat(s, a) (utils.jl:76-87) of the oracle against torch float64 autograd."""
    rng = np.random.default_rng(18); od, ad, B = 3, 2, 24
    adims, qdims, aacts, qacts = [od, 16, ad], [od + ad, 16, 1], ["relu", "tanh"], ["tanh", "identity"]
    oa = O.OMlp(adims, aacts).init_glorot(4, 0); q = O.OMlp(qdims, qacts).init_glorot(4, 1)
    oa.params[:] += 0.1 * rng.standard_normal(oa.n).astype(np.float32)
    ob = O.OBuffer(od, ad, L.ACTION_CONTINUOUS, B)
    ob.push({"s": rng.normal(0, 1, (od, B)).astype(np.float32), "a": rng.uniform(-1, 1, (ad, B)).astype(np.float32), "sp": rng.normal(0, 1, (od, B)).astype(np.float32),
             "r": rng.normal(0, 1, (1, B)).astype(np.float32), "done": rng.random((1, B)) < 0.2, "episode_end": np.zeros((1, B), bool)})
    oa.adam_init(0.0); q.adam_init(0.0)
    info, ol = np.zeros(L.INFO_N, np.float32), O.lib()
    O.chk(ol.orc_dpg_actor_step(oa.h, q.h, ob.h, O.vpz(info))); ga, la = oa.grads.copy(), float(info[0])
    pa, pq = _torch_net(oa), _torch_net(q); S = torch.tensor(ob["s"], dtype=torch.float64)
    l = -(_fwd(pq, qdims, qacts, torch.cat([S, _fwd(pa, adims, aacts, S)[0]], 0))[0][0]).mean(); l.backward()
    assert abs(la - l.item()) < 1e-5 * max(1.0, abs(l.item())) and np.abs(ga - pa.grad.numpy()).max() < 3e-6 * max(1.0, np.abs(pa.grad.numpy()).max())
    y = rng.normal(0, 1, B).astype(np.float32)
    O.chk(ol.orc_q_step(q.h, ob.h, O.vpz(y), 0, O.vpz(info))); gq, lq = q.grads.copy(), float(info[0])
    pq = _torch_net(q); l = ((_fwd(pq, qdims, qacts, torch.tensor(np.vstack([ob["s"], ob["a"]]), dtype=torch.float64))[0][0] - torch.tensor(y, dtype=torch.float64)) ** 2).mean(); l.backward()
    assert abs(lq - l.item()) < 1e-5 * max(1.0, abs(l.item())) and np.abs(gq - pq.grad.numpy()).max() < 3e-6 * max(1.0, np.abs(pq.grad.numpy()).max())


@pytest.mark.parametrize("loss", ["a2c", "reinforce"])
@pytest.mark.parametrize("head", ["categorical", "gaussian"])
def test_a2c_reinforce_gradients_vs_float64_autograd(loss, head):
    """a2c_loss (a2c.jl:4-15), reinforce_loss (reinforce.jl:4-13) of the oracle vs torch float64 autograd of the reference formulas."""
    rng = np.random.default_rng(12); n, acts = 48, ["tanh", "relu", "identity"]
    disc = head == "categorical"; ad = 3 if disc else 2; dims = [4, 16, 16, ad]
    o = O.OMlp(dims, acts, 0 if disc else ad).init_glorot(6, 0, -0.2); o.params[:] += 0.05 * rng.standard_normal(o.n).astype(np.float32)
    ob = O.OBuffer(4, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, n, ["return", "logprob", "advantage"])
    d = _fill(ob, n, rng, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, ad)
    ids = np.arange(n, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32); cfg = _cfg(loss, head, lp=0.7, le=0.2)
    O.chk(lib.orc_loss_grad(o.h, ob.h, C.byref(cfg), O.vpz(ids), n, O.vpz(info)))
    p = _torch_net(o); z, off = _fwd(p, dims, acts, torch.tensor(d["s"], dtype=torch.float64))
    a = torch.tensor(d["a"], dtype=torch.float64)
    if disc:
        pr = torch.softmax(z, 0); newlp = torch.log((pr * a).sum(0)); ent = (-(pr * torch.log(pr + float(np.finfo(np.float32).eps))).sum(0)).mean()
    else:
        ls = p[off:off + ad]; s2 = torch.exp(ls) ** 2
        newlp = (-((a - z) ** 2) / (2 * s2[:, None]) - 0.9189385332046727 - ls[:, None]).sum(0); ent = 1.4189385332046727 + ls.sum()
    w = torch.tensor(d["advantage" if loss == "a2c" else "return"][0], dtype=torch.float64)
    total = 0.7 * (-(newlp * w).mean()) + 0.2 * (-ent) if loss == "a2c" else -(newlp * w).mean()
    total.backward()
    assert abs(info[0] - total.item()) < 1e-5 * max(1, abs(total.item()))
    assert np.abs(o.grads - p.grad.numpy()).max() < 2e-6 * max(1, np.abs(p.grad.numpy()).max())
    assert abs(info[L.INFO["entropy"]] - ent.item()) < 1e-5 and abs(info[L.INFO["kl"]] - (torch.tensor(d["logprob"][0], dtype=torch.float64) - newlp).mean().item()) < 1e-5


def test_dpg_oracle_gradients_match_finite_differences():
    """ddpg_actor_loss (ddpg.jl:26) and td_loss over vcat(s, a) (utils.jl:76-87) of the oracle vs central differences of its own losses."""
    rng = np.random.default_rng(18); od, ad, B = 3, 2, 24
    oa = O.OMlp([od, 16, ad], ["relu", "tanh"]).init_glorot(4, 0); q = O.OMlp([od + ad, 16, 1], ["tanh", "identity"]).init_glorot(4, 1)
    oa.params[:] += 0.1 * rng.standard_normal(oa.n).astype(np.float32)
    ob = O.OBuffer(od, ad, L.ACTION_CONTINUOUS, B)
    ob.push({"s": rng.normal(0, 1, (od, B)).astype(np.float32), "a": rng.uniform(-1, 1, (ad, B)).astype(np.float32), "sp": rng.normal(0, 1, (od, B)).astype(np.float32),
             "r": rng.normal(0, 1, (1, B)).astype(np.float32), "done": rng.random((1, B)) < 0.2, "episode_end": np.zeros((1, B), bool)})
    oa.adam_init(0.0); q.adam_init(0.0)
    info, ol = np.zeros(L.INFO_N, np.float32), O.lib()
    def actor_loss():
        O.chk(ol.orc_dpg_actor_step(oa.h, q.h, ob.h, O.vpz(info))); return float(info[0])
    actor_loss(); g = oa.grads.copy(); idx = list(rng.choice(oa.n, 10, replace=False))
    assert np.allclose(g[idx], _fd(actor_loss, oa.params, idx), rtol=3e-2, atol=2e-3)
    y = rng.normal(0, 1, B).astype(np.float32)
    def critic_loss():
        O.chk(ol.orc_q_step(q.h, ob.h, O.vpz(y), 0, O.vpz(info))); return float(info[0])
    critic_loss(); gq = q.grads.copy(); idx = list(rng.choice(q.n, 10, replace=False))
    assert np.allclose(gq[idx], _fd(critic_loss, q.params, idx), rtol=3e-2, atol=2e-3)
    # ddpg_target on terminal rows is the reward; td3 smoothing respects the action clamp
    yt = np.empty(B, np.float32); O.chk(ol.orc_dpg_target(oa.h, q.h, None, ob.h, 0.9, -1.0, 0.0, 0.0, 0.0, 0.0, 1, 2, O.vpz(yt)))
    done = ob["done"][0]; assert np.array_equal(yt[done], ob["r"][0][done])
    sa = np.vstack([ob["sp"], oa.forward(ob["sp"])]).astype(np.float32)
    assert np.abs(yt - (ob["r"][0] + np.float32(0.9) * (1 - done) * q.forward(sa)[0])).max() < 1e-5


@pytest.mark.parametrize("case", ["logpdf_categorical", "logpdf_gaussian", "mse_action"])
def test_bc_losses_vs_float64_autograd(case):
    """logpdf_bc_loss / mse_action_loss (src/model_free/il/bc.jl:1-18) of the oracle vs torch float64 autograd of the reference formulas."""
    rng = np.random.default_rng(21); n, acts = 40, ["tanh", "relu", "identity"]
    disc = case == "logpdf_categorical"; ad = 3 if disc else 2; dims = [4, 16, 16, ad]
    nx = ad if case == "logpdf_gaussian" else 0
    o = O.OMlp(dims, acts, nx).init_glorot(6, 0, -0.2); o.params[:] += 0.05 * rng.standard_normal(o.n).astype(np.float32)
    ob = O.OBuffer(4, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, n)          # no logprob / advantage / return columns: BC reads only s and a
    d = {"s": rng.standard_normal((4, n)).astype(np.float32), "sp": rng.standard_normal((4, n)).astype(np.float32), "r": np.zeros((1, n), np.float32), "done": np.zeros((1, n), bool)}
    if disc:
        a = np.zeros((ad, n), np.bool_); a[rng.integers(0, ad, n), np.arange(n)] = True; d["a"] = a
    else:
        d["a"] = rng.standard_normal((ad, n)).astype(np.float32)
    ob.push(d)
    ids = np.arange(n, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32)
    cfg = _cfg("mse_action" if case == "mse_action" else "logpdf_bc", "categorical" if disc else ("gaussian" if nx else "deterministic"), lp=0.3, le=0.05)
    O.chk(lib.orc_loss_grad(o.h, ob.h, C.byref(cfg), O.vpz(ids), n, O.vpz(info)))
    p = _torch_net(o); z, off = _fwd(p, dims, acts, torch.tensor(d["s"], dtype=torch.float64)); a = torch.tensor(d["a"], dtype=torch.float64)
    if case == "mse_action":
        total = ((z - a) ** 2).mean()
    else:
        if disc:
            pr = torch.softmax(z, 0); lp = torch.log((pr * a).sum(0)); ent = (-(pr * torch.log(pr + float(np.finfo(np.float32).eps))).sum(0)).mean()
        else:
            ls = p[off:off + ad]; s2 = torch.exp(ls) ** 2
            lp = (-((a - z) ** 2) / (2 * s2[:, None]) - 0.9189385332046727 - ls[:, None]).sum(0); ent = 1.4189385332046727 + ls.sum()
        total = 0.05 * (-ent) + (-lp.mean())                                          # lambda_p is not part of logpdf_bc_loss
    total.backward()
    assert abs(info[0] - total.item()) < 1e-5 * max(1, abs(total.item()))
    assert np.abs(o.grads - p.grad.numpy()).max() < 2e-6 * max(1, np.abs(p.grad.numpy()).max())


@pytest.mark.parametrize("disc", [False, True])
def test_gail_discriminator_loss_and_reward_vs_float64_autograd(disc):
    """gail_d_loss(GAN_BCELoss()) (on_policy_gail.jl:1-5, extras/gans.jl:7-9) and the GAIL reward (:50-55) of the oracle vs torch float64:
    L = mean(-logsigmoid(D(a_ex, s_ex))) + mean(D(a_pi, s_pi) - logsigmoid(D(a_pi, s_pi))), r = ar*logsigmoid(z) - (1-ar)*(logsigmoid(z) - z)."""
    rng = np.random.default_rng(21); od, ad, n_ex, n_pi = 3, 2, 20, 28
    kind = L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS
    dims, acts = [ad + od, 16, 16, 1], ["tanh", "relu", "identity"]
    o = O.OMlp(dims, acts).init_glorot(8, 0); o.params[:] += 0.1 * rng.standard_normal(o.n).astype(np.float32); o.adam_init(1e-3)
    def mk(n):
        b = O.OBuffer(od, ad, kind, n + 4)
        a = np.eye(ad, dtype=bool)[:, rng.integers(0, ad, n + 4)] if disc else rng.uniform(-1, 1, (ad, n + 4)).astype(np.float32)
        d = {"s": rng.normal(0, 1, (od, n + 4)).astype(np.float32), "a": a, "sp": rng.normal(0, 1, (od, n + 4)).astype(np.float32), "r": np.zeros((1, n + 4), np.float32),
             "done": np.zeros((1, n + 4), bool), "episode_end": np.zeros((1, n + 4), bool)}
        b.push(d); return b, d
    bex, dex = mk(n_ex); bpi, dpi = mk(n_pi)
    info = np.zeros(L.INFO_N, np.float32); p0 = o.params.copy()
    O.chk(lib.orc_gail_d_step(o.h, bex.h, 2, n_ex, bpi.h, 3, n_pi, O.vpz(info)))
    pt = torch.tensor(p0, dtype=torch.float64, requires_grad=True)
    xe = torch.tensor(np.vstack([dex["a"][:, 2:2 + n_ex].astype(np.float64), dex["s"][:, 2:2 + n_ex]]), dtype=torch.float64)
    xp = torch.tensor(np.vstack([dpi["a"][:, 3:3 + n_pi].astype(np.float64), dpi["s"][:, 3:3 + n_pi]]), dtype=torch.float64)
    ze, _ = _fwd(pt, dims, acts, xe); zp, _ = _fwd(pt, dims, acts, xp)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(ze[0], torch.ones(n_ex, dtype=torch.float64)) + \
        torch.nn.functional.binary_cross_entropy_with_logits(zp[0], torch.zeros(n_pi, dtype=torch.float64))
    loss.backward()
    assert abs(info[0] - loss.item()) < 1e-5 * max(1, abs(loss.item()))
    assert np.abs(o.grads - pt.grad.numpy()).max() < 2e-6 * max(1, np.abs(pt.grad.numpy()).max())
    assert abs(info[L.INFO["grad_norm"]] - float(np.linalg.norm(pt.grad.numpy()))) < 1e-5
    assert not np.array_equal(o.params, p0)                       # Adam applied
    # reward on the (updated) discriminator
    m = np.zeros(1, np.float32); O.chk(lib.orc_gail_reward(o.h, bpi.h, 0.3, 2.0, O.vpz(m)))
    z = torch.tensor(o.forward(np.vstack([dpi["a"].astype(np.float32), dpi["s"]]))[0].astype(np.float64))
    ls = torch.nn.functional.logsigmoid(z); r = 0.3 * ls - 0.7 * (ls - z)
    assert np.abs(bpi["r"][0] - 2.0 * r.numpy()).max() < 1e-5 and abs(m[0] - r.mean().item()) < 1e-5


@pytest.mark.parametrize("loss", ["ppo", "a2c"])
def test_squashed_gaussian_gradient_vs_float64_autograd(loss):
    """SquashedGaussianPolicy (policies.jl:353-400) in the oracle's policy-gradient losses vs torch float64 autograd of the reference formulas:
    logpdf(pi, s, a) = squashed_gaussian_logprob(mu(s), logSigma, atanh(clamp(a/ascale, -1+1f-5, 1-1f-5))), sigma = exp(clamp(logSigma, -5, 2)).
    One log-std sits outside the clamp range so that its zero derivative through clamp is exercised."""
    rng = np.random.default_rng(31); n, ad, asc = 40, 3, 2.0; dims, acts = [4, 16, 16, ad], ["tanh", "relu", "identity"]
    o = O.OMlp(dims, acts, ad).init_glorot(9, 0, -0.3); o.params[:] += 0.05 * rng.standard_normal(o.n).astype(np.float32)
    o.params[-ad:] = np.array([-0.3, 2.5, -1.0], np.float32)                       # 2.5 > LOG_STD_MAX
    O.chk(lib.orc_mlp_set_squash(o.h, asc))
    ob = O.OBuffer(4, ad, L.ACTION_CONTINUOUS, n, ["return", "logprob", "advantage"])
    d = _fill(ob, n, rng, L.ACTION_CONTINUOUS, ad)
    a_sq = (asc * np.tanh(rng.normal(0, 1.2, (ad, n)))).astype(np.float32); a_sq[0, 0] = asc        # one action exactly on the bound (clamped before atanh)
    d["a"] = a_sq; ob2 = O.OBuffer(4, ad, L.ACTION_CONTINUOUS, n, ["return", "logprob", "advantage"]); ob2.push(d)
    ids = np.arange(n, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32); cfg = _cfg(loss, "gaussian", lp=0.8, le=0.05)
    O.chk(lib.orc_loss_grad(o.h, ob2.h, C.byref(cfg), O.vpz(ids), n, O.vpz(info)))
    p = _torch_net(o); mu, off = _fwd(p, dims, acts, torch.tensor(d["s"], dtype=torch.float64)); ls = p[off:off + ad]
    t = torch.clamp(torch.tensor((d["a"] / np.float32(asc)).astype(np.float64)), float(np.float32(-1.0) + np.float32(1e-5)), float(np.float32(1.0) - np.float32(1e-5)))
    u = torch.atanh(t); s2 = torch.exp(torch.clamp(ls, -5.0, 2.0)) ** 2
    corr = 2 * (float(np.log(2.0)) - u - torch.nn.functional.softplus(-2 * u))
    newlp = (-((u - mu) ** 2) / (2 * s2[:, None]) - 0.9189385332046727 - ls[:, None] - corr).sum(0)
    A = torch.tensor(d["advantage"][0], dtype=torch.float64); old = torch.tensor(d["logprob"][0], dtype=torch.float64)
    ent = 1.4189385332046727 + ls.sum()
    if loss == "ppo":
        r = torch.exp(newlp - old); pl = -torch.minimum(r * A, torch.clamp(r, 0.8, 1.2) * A).mean()
    else:
        pl = -(newlp * A).mean()
    total = 0.8 * pl + 0.05 * (-ent); total.backward()
    assert abs(info[0] - total.item()) < 2e-5 * max(1, abs(total.item()))
    g = p.grad.numpy(); assert np.abs(o.grads - g).max() < 5e-6 * max(1, np.abs(g).max())
    assert abs(info[L.INFO["kl"]] - (old - newlp).mean().item()) < 2e-5 and abs(info[L.INFO["entropy"]] - ent.item()) < 1e-6


def test_squashed_gaussian_exploration_is_consistent_with_logpdf():
    """test/policy_tests.jl:305-308: `a, logprob = exploration(p, s); logpdf(p, s, a) ≈ logprob` for SquashedGaussianPolicy, through the oracle's rollout
    (a = ascale*tanh(mu + sigma*eps), logprob on the un-tanh'd action) and its learner head (which un-tanh's the stored action again); actions stay inside ±ascale."""
    E, T, asc = 4, 40, 2.0
    oa = O.OMlp([3, 64, 64, 1], ["relu", "relu", "identity"], 1).init_glorot(3, 0, -0.5); O.chk(lib.orc_mlp_set_squash(oa.h, asc))
    ob = O.OBuffer(3, 1, L.ACTION_CONTINUOUS, E * T, ["return", "logprob", "advantage"])
    oe = O.OEnv("pendulum", E, 30, 0.99, 5)
    import parity
    oe.rollout(oa, parity.rollout_cfg(True, True, "gaussian"), ob, T)
    a, lp = ob["a"], ob["logprob"][0]
    assert np.abs(a).max() <= asc and np.abs(a).max() > 0.5 and np.isfinite(lp).all()
    # learner-side logpdf of the stored (s, a): PPO ratio r = exp(new - old) must be ~1 and kl ~0 with unchanged parameters
    cfg = _cfg("ppo", "gaussian"); ids = np.arange(E * T, dtype=np.int64); info = np.zeros(L.INFO_N, np.float32)
    ob.col("advantage")[...] = 1.0
    O.chk(lib.orc_loss_grad(oa.h, ob.h, C.byref(cfg), O.vpz(ids), E * T, O.vpz(info)))
    inner = np.abs(a[0]) < asc * (1 - 1e-3)                     # where the clamp before atanh is inactive, atanh(tanh(u)) returns u up to float rounding
    assert inner.mean() > 0.8 and abs(info[L.INFO["kl"]]) < 5e-3 and info[L.INFO["clip_fraction"]] < 0.05
