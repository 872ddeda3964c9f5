"""CPU: the oracle's twin of crux_policy_explore (orc_policy_explore) against the oracle's own rollout -- stepping the restated CartPole from the HOST, one step! at a time
(src/sampler.jl:71-137: action from the policy head, @gen on the host, the Sampler's episode bookkeeping), reproduces orc_rollout's buffer bit for bit. Pins the counter
conventions the caller-stepped seam relies on: draws keyed by (seed, steps the sampler has taken, stream = sampler index), the interaction counter env-minor, blocks env-major."""
import ctypes as C

import numpy as np
import pytest

import host_env as H
import oracle as O
from crux_jl_amd import _lib as L


def _cfg(head, explore=1, reset=1, i0=0, eps=None):
    cfg = L.RolloutCfg()
    cfg.explore, cfg.reset_at_end, cfg.head, cfg.i0 = explore, reset, L.HEAD[head], i0
    cfg.eps_steps, cfg.noise_sigma = 0, -1.0
    cfg.noise_eps_min, cfg.noise_eps_max, cfg.a_min, cfg.a_max = -np.inf, np.inf, -np.inf, np.inf
    if eps:
        cfg.eps_start, cfg.eps_stop, cfg.eps_steps = eps
    return cfg


@pytest.mark.parametrize("head,eps", [("categorical", None), ("greedy_q", (1.0, 0.1, 300))])
def test_host_stepped_cartpole_reproduces_the_oracle_rollout(head, eps):
    E, T, max_steps, seed, i0 = 3, 40, 17, 12345, 1000
    extras = ["logprob", "t", "i"]
    pol = O.OMlp([4, 16, 2], ["relu", "identity"]).init_glorot(5, 0)
    cfg = _cfg(head, i0=i0, eps=eps)
    ref = O.OBuffer(4, 2, L.ACTION_DISCRETE, E * T, extras)
    O.OEnv("cartpole", E, max_steps, 0.99, seed).rollout(pol, cfg, ref, T)
    # ---- the same block, stepped from the host
    init, gen = H.cartpole_initialstate(seed), H.cartpole_gen(H.oracle_step)
    s = [init(e, 0) for e in range(E)]; n_resets = np.ones(E, np.int64); ep_len = np.zeros(E, np.int64); steps = np.zeros(E, np.int64)
    svec = np.stack([H.cartpole_observation(x) for x in s], axis=1).astype(np.float32, order="F")
    d = {"s": np.zeros((4, E * T), np.float32), "a": np.zeros((2, E * T), np.bool_), "sp": np.zeros((4, E * T), np.float32), "r": np.zeros((1, E * T), np.float32),
         "done": np.zeros((1, E * T), np.bool_), "episode_end": np.zeros((1, E * T), np.bool_), "logprob": np.zeros((1, E * T), np.float32),
         "t": np.zeros((1, E * T), np.int64), "i": np.zeros((1, E * T), np.int64)}
    for t in range(T):
        c = _cfg(head, i0=i0 + t * E, eps=eps)
        a = np.zeros((2, E), np.uint8, order="F"); lp = np.empty(E, np.float32)
        O.chk(O.lib().orc_policy_explore(pol.h, C.byref(c), E, O.vpz(np.asfortranarray(svec)), seed, O.vpz(steps), O.vpz(a), O.vpz(lp)))
        for e in range(E):
            j = e * T + t
            sn, r, _ = gen(e, s[e], int(np.argmax(a[:, e])), int(steps[e]))
            done = H.cartpole_isterminal(sn); spv = H.cartpole_observation(sn)
            d["s"][:, j] = svec[:, e]; d["a"][:, j] = a[:, e] != 0; d["sp"][:, j] = spv; d["r"][0, j] = r; d["done"][0, j] = done
            d["logprob"][0, j] = lp[e]; d["t"][0, j] = ep_len[e] + 1; d["i"][0, j] = i0 + t * E + e + 1
            steps[e] += 1; ep_len[e] += 1
            if done or ep_len[e] >= max_steps or t == T - 1:
                d["episode_end"][0, j] = True
                s[e] = init(e, int(n_resets[e])); n_resets[e] += 1; ep_len[e] = 0; svec[:, e] = H.cartpole_observation(s[e])
            else:
                s[e] = sn; svec[:, e] = spv
    for k in d:
        got, want = d[k], ref[k]
        if got.dtype.kind == "f":
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k      # bit for bit, NaNs included
        else:
            assert np.array_equal(got, want), k
    assert d["episode_end"].sum() > E      # episodes were cut by max_steps / termination inside the block, not only at its end
