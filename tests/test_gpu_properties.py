"""Size-independent properties of the hot path AT BASELINE's full sizes -- no oracle in the loop, so they run at sizes where the scalar oracle would need minutes, and they hold
for ANY correct implementation of the reference functions (the full-size oracle REPLAYS are tests/test_gpu_fullsize.py):

  shuffle!            (experience_buffer.jl:118-124)  a permutation of whole rows: every column moves with the same bijection, nothing is lost or duplicated
  push! on the ring   (:232-259)                      one push of N rows == two pushes of its halves (ring, wrap-around and counters included)
  fill_returns!       (sampler.jl:275-281)            linear in r: doubling the rewards doubles the returns BIT FOR BIT (a power of two commutes with every rounding)
  fill_gae!           (sampler.jl:262-273)            affine in r for a fixed critic: A(r1 + r2) - A(r1) - A(r2) + A(0) == 0 up to float rounding
  whiten              (utils.jl:41-42)                mean 0, n-1-corrected std 1; whitening a whitened column changes nothing beyond rounding
  prioritized_sample! (experience_buffer.jl:324-349)  stratified keys are increasing, so the sampled indices are SORTED; every index carries weight <= 1; positive priority
  polyak_average!     (policies.jl:48-59)             tau = 1 copies, tau = 0 leaves the target untouched, bit for bit
"""
import numpy as np
import pytest

import parity
from parity import crux

pytestmark = pytest.mark.gpu
EXTRAS = ["return", "logprob", "advantage"]


def _block(rng, od, ad, N, T, disc=True):
    """an env-major block of N = E x T transitions with episode cuts every T rows (steps!(...; reset=true)) and a few terminal states inside"""
    a = np.zeros((ad, N), np.bool_) if disc else rng.standard_normal((ad, N)).astype(np.float32)
    if disc:
        a[rng.integers(0, ad, N), np.arange(N)] = True
    done = rng.random((1, N)) < 0.004
    ee = done.copy(); ee[0, T - 1::T] = True
    return {"s": rng.standard_normal((od, N)).astype(np.float32), "a": a, "sp": rng.standard_normal((od, N)).astype(np.float32), "r": rng.standard_normal((1, N)).astype(np.float32),
            "done": done, "episode_end": ee, "return": np.zeros((1, N), np.float32), "logprob": rng.standard_normal((1, N)).astype(np.float32), "advantage": np.zeros((1, N), np.float32)}


def test_shuffle_is_one_bijection_of_whole_rows_at_c2_size(gpu_ctx):
    rng = np.random.default_rng(11); N = 32 * 2048
    b = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), N, EXTRAS, ctx=gpu_ctx)
    d = _block(rng, 4, 2, N, 2048); d["logprob"][0] = np.arange(N, dtype=np.float32)          # a row id that survives the gather exactly (N < 2^24)
    b.push_(d)
    for ep in range(3):
        crux.shuffle_device_(b, seed=5, counter=ep)
    ids = b["logprob"][0].astype(np.int64)
    assert np.array_equal(np.sort(ids), np.arange(N))                                         # a bijection: no row lost, none duplicated
    assert not np.array_equal(ids, np.arange(N))
    for k in ("s", "sp", "a", "r", "done", "episode_end"):
        assert np.array_equal(b[k], d[k][:, ids]), k                                          # every column moved with the SAME permutation


def test_one_push_equals_two_half_pushes_on_a_wrapping_ring(gpu_ctx):
    rng = np.random.default_rng(12); N, cap = 50_000, 65_536
    d0 = _block(rng, 8, 4, 40_000, 1000); d = _block(rng, 8, 4, N, 1000)
    bs = []
    for halves in (False, True):
        b = crux.ExperienceBuffer(crux.ContinuousSpace(8), crux.DiscreteSpace(4), cap, EXTRAS, ctx=gpu_ctx)
        b.push_(d0)                                                                           # the second push wraps: 40 000 + 50 000 > 65 536
        if halves:
            h = N // 2 + 7
            b.push_({k: v[:, :h] for k, v in d.items()}); b.push_({k: v[:, h:] for k, v in d.items()})
        else:
            b.push_(d)
        bs.append(b)
    assert len(bs[0]) == len(bs[1]) == cap and bs[0].next_ind == bs[1].next_ind
    for k in bs[0].keys():
        assert np.array_equal(bs[0][k], bs[1][k]), k


def test_returns_are_linear_and_gae_is_affine_in_the_rewards_at_c5_size(gpu_ctx):
    rng = np.random.default_rng(13); E, T = 128, 2048; N = E * T
    critic = crux.ContinuousNetwork(parity.chain([17, 64, 64, 1], ["tanh", "tanh", "identity"]), ctx=gpu_ctx, seed=3, stream=1)
    d = _block(rng, 17, 6, N, T, disc=False)
    r1 = d["r"].copy(); r2 = rng.standard_normal((1, N)).astype(np.float32)

    def run(r):
        b = crux.ExperienceBuffer(crux.ContinuousSpace(17), crux.ContinuousSpace(6), N, EXTRAS, ctx=gpu_ctx)
        dd = dict(d); dd["r"] = r; b.push_(dd)
        crux.fill_gae_(b, critic, 0.95, 0.99); crux.fill_returns_(b, 0.99)
        return b["advantage"][0].astype(np.float64), b["return"][0]
    A1, R1 = run(r1); A2, _ = run(r2); A12, _ = run((r1.astype(np.float64) + r2).astype(np.float32)); A0, _ = run(np.zeros((1, N), np.float32))
    _, R2x = run(np.float32(2.0) * r1)
    assert np.array_equal(R2x.view(np.uint32), (np.float32(2.0) * R1).view(np.uint32))        # scaling by a power of two commutes with every rounding of the recurrence
    # r1 + r2 is rounded to Float32 once per element: the affine identity holds to the rounding of sums of ~1/(1 - 0.95 * 0.99) ~ 17 terms of O(1)
    resid = np.abs(A12 - A1 - A2 + A0)
    print("GAE affinity residual at %d transitions: max %.3g (|A| up to %.3g)" % (N, resid.max(), np.abs(A1).max()))
    assert resid.max() < 3e-5 and np.isfinite(A1).all()


def test_whiten_gives_mean_zero_std_one_and_is_idempotent_at_c2_size(gpu_ctx):
    rng = np.random.default_rng(14); N = 32 * 2048
    b = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), N, EXTRAS, ctx=gpu_ctx)
    d = _block(rng, 4, 2, N, 2048); d["advantage"] = (3.0 + 7.0 * rng.standard_normal((1, N))).astype(np.float32); b.push_(d)
    crux.whiten_(b, "advantage"); w1 = b["advantage"][0].copy()
    assert abs(float(w1.astype(np.float64).mean())) < 1e-6 and abs(float(w1.astype(np.float64).std(ddof=1)) - 1.0) < 1e-6
    crux.whiten_(b, "advantage"); w2 = b["advantage"][0]
    assert np.abs(w2 - w1).max() < 2e-6                                                       # (x - 0) / 1 up to the rounding of mean and std
    # order is preserved: whiten is an increasing affine map
    o = np.argsort(d["advantage"][0], kind="stable"); assert np.all(np.diff(w1[o]) >= 0)


def test_prioritized_sample_is_sorted_and_weighted_at_one_million_rows(gpu_ctx):
    rng = np.random.default_rng(15); N, B = 1_000_000, 128
    src = crux.ExperienceBuffer(crux.ContinuousSpace(8), crux.DiscreteSpace(4), N, prioritized=True, ctx=gpu_ctx)
    a = np.zeros((4, N), np.bool_); a[rng.integers(0, 4, N), np.arange(N)] = True
    src.push_({"s": rng.standard_normal((8, N)).astype(np.float32), "a": a, "sp": rng.standard_normal((8, N)).astype(np.float32), "r": rng.standard_normal((1, N)).astype(np.float32),
               "done": np.zeros((1, N), np.bool_), "episode_end": np.zeros((1, N), np.bool_)})
    I = rng.choice(N, 300_000, replace=False).astype(np.int64); v = (np.abs(rng.standard_normal(I.size)) + 1e-3)
    src.update_priorities_(I + 1, v)
    tgt = crux.ExperienceBuffer(crux.ContinuousSpace(8), crux.DiscreteSpace(4), B, ["weight"], ctx=gpu_ctx)
    pr = src.priority_params()["priorities"][:N]
    for k in range(6):
        ids = crux.prioritized_sample_(tgt, src, i=1000 + k, counter=50 + k) - 1
        assert ids.min() >= 0 and ids.max() < N
        assert np.all(np.diff(ids) >= 0), "stratum j's key lies in [j, j + 1) * total / B: the sampled indices are sorted"
        w = tgt["weight"][0]
        assert np.all(w > 0) and np.all(w <= 1.0 + 1e-6)                                      # w = (N p / total)^beta / max_w <= 1 (:343-347)
        assert np.all(pr[ids] > 0)
        assert np.array_equal(tgt["s"], src_rows(src, ids, "s"))                              # the gathered rows are the sampled rows


def src_rows(src, ids, key):
    t = crux.ExperienceBuffer(crux.ContinuousSpace(8), crux.DiscreteSpace(4), len(ids), ctx=src.ctx)
    crux.uniform_sample_(t, src, B=len(ids), ids=ids + 1)
    return t[key]


def test_polyak_with_tau_one_copies_and_tau_zero_keeps_at_c4_size(gpu_ctx):
    dims, acts = [4, 256, 256, 1], ["relu", "relu", "identity"]
    a = crux.ContinuousNetwork(parity.chain(dims, acts), ctx=gpu_ctx, seed=1, stream=0); b = crux.ContinuousNetwork(parity.chain(dims, acts), ctx=gpu_ctx, seed=2, stream=0)
    pa, pb = a.get_params().copy(), b.get_params().copy(); assert not np.array_equal(pa, pb)
    crux.polyak_average_(b, a, 0.0); assert np.array_equal(b.get_params().view(np.uint32), pb.view(np.uint32))
    crux.polyak_average_(b, a, 1.0); assert np.array_equal(b.get_params().view(np.uint32), pa.view(np.uint32))
    crux.polyak_average_(b, a, 0.005); assert np.abs(b.get_params() - pa).max() <= 1.2e-7 * np.abs(pa).max()      # averaging equal networks: tau x + (1 - tau) x = x up to one rounding
