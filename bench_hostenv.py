"""The caller-stepped environment seam, timed (/root/reference/src/sampler.jl:71-137): what `step!` on a user `mdp` costs when the policy
forward, the exploration draws and the buffer write live on the device and the environment on the host -- the way C3 (LunarLander) and C5 (HalfCheetah) users run, since
those simulators cannot be restated on the device.

Per configuration: a TRIVIAL vectorised numpy environment (x' = 0.9 x + 0.1 sin(roll(x) + mean(a)), r = 1, never terminal: ~2 us per step for all copies, so that the
seam -- not the environment -- is what is timed) is driven through the C ABI exactly as `HostMDP` / `steps!` drive it:

    for t in 1:T   a, logprob = crux_policy_explore(pi, cfg, E observations)      one launch + the read-back of E actions
                   sp, r = env(s, a)                                              host
    crux_steps_push(buffer, block of E*T transitions, critic)                     ring write + fill_gae! / fill_returns! on the pushed rows

Reported: us per crux_policy_explore call (median and p90 over the run), us per crux_steps_push, env-steps/s end to end, and beside them the DEVICE-environment rollout
(`crux_rollout`, same policy shape, same E and T) in env-steps/s. `python bench_hostenv.py` prints the three configurations as one JSON line; bench.py's default run
carries them under other_configs.host_env."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

# name: (obs, act, discrete, actor dims, acts, head, E, T per block, blocks timed, device env kind)
CONFIGS = {
    "c2_e32": (4, 2, True, [4, 64, 64, 2], ["relu", "relu", "identity"], "categorical", 32, 64, 8, "cartpole"),
    "c3_e1": (8, 4, True, [8, 256, 256, 4], ["relu", "relu", "identity"], "greedy_q", 1, 4, 256, "synth_discrete"),
    "c5_e128": (17, 6, False, [17, 64, 64, 6], ["tanh", "tanh", "identity"], "gaussian", 128, 16, 8, "synth"),
}


def _chain(crux, dims, acts):
    return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])


def _env_step(s, a_mean):
    return (0.9 * s + 0.1 * np.sin(np.roll(s, 1, axis=0) + a_mean)).astype(np.float32)


def one(crux, ctx, name):
    from crux_jl_amd import _lib as L
    od, ad, disc, dims, acts, head, E, T, blocks, kind = CONFIGS[name]
    S = crux.ContinuousSpace(od); A = crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad)
    if head == "categorical":
        pi = crux.DiscreteNetwork(_chain(crux, dims, acts), list(range(1, ad + 1)), ctx=ctx, seed=5, stream=0)
    elif head == "greedy_q":
        pi = crux.DiscreteNetwork(_chain(crux, dims, acts), list(range(1, ad + 1)), ctx=ctx, seed=5, stream=0)
    else:
        pi = crux.GaussianPolicy(_chain(crux, dims, acts), np.full(ad, -0.5, np.float32), ctx=ctx, seed=5, stream=0)
    on_policy = head != "greedy_q"
    critic = crux.ContinuousNetwork(_chain(crux, [od, 64, 64, 1], acts), ctx=ctx, seed=5, stream=1) if on_policy else None
    extras = ["return", "logprob", "advantage"] if on_policy else []
    N = E * T
    buf = crux.ExperienceBuffer(S, A, max(N, 1 << 16), extras, ctx=ctx)
    cfg = L.RolloutCfg(); cfg.explore, cfg.reset_at_end, cfg.i0 = 1, 0, 0
    cfg.eps_steps, cfg.noise_sigma = 0, -1.0
    cfg.noise_eps_min, cfg.noise_eps_max, cfg.a_min, cfg.a_max = -np.inf, np.inf, -np.inf, np.inf
    cfg.head = L.HEAD[head]
    if head == "greedy_q":
        cfg.eps_start, cfg.eps_stop, cfg.eps_steps = 1.0, 0.1, 50000
    lib = ctx.lib
    svec = np.asfortranarray(np.random.default_rng(1).standard_normal((od, E)).astype(np.float32))
    steps_taken = np.zeros(E, np.int64)
    a_out = np.zeros((ad, E), np.bool_ if disc else np.float32, order="F"); lp = np.empty(E, np.float32)
    cols = {"s": np.empty((od, N), np.float32, order="F"), "sp": np.empty((od, N), np.float32, order="F"), "a": np.zeros((ad, N), np.bool_ if disc else np.float32, order="F"),
            "r": np.ones((1, N), np.float32, order="F"), "done": np.zeros((1, N), np.bool_, order="F"), "episode_end": np.zeros((1, N), np.bool_, order="F")}
    if on_policy:
        cols["logprob"] = np.empty((1, N), np.float32, order="F")
    for e in range(E):
        cols["episode_end"][0, e * T + T - 1] = True          # steps!(...; reset=true): every copy's block is cut at its end (sampler.jl:148)
    ptrs = (C.c_void_p * L.NCOLS)()
    for k, v in cols.items():
        ptrs[L.COL[k]] = v.ctypes.data
    rows = np.arange(E) * T
    t_explore, t_push, t_env = [], [], []

    def block(i0):
        nonlocal svec, steps_taken
        for t in range(T):
            cfg.i0 = i0 + t * E
            t0 = time.perf_counter()
            ctx.check(lib.crux_policy_explore(pi.h, C.byref(cfg), E, svec.ctypes.data_as(C.c_void_p), 7, steps_taken.ctypes.data_as(C.c_void_p), a_out.ctypes.data_as(C.c_void_p),
                                              lp.ctypes.data_as(C.c_void_p)))
            t1 = time.perf_counter()
            am = a_out.argmax(axis=0).astype(np.float32)[None, :] if disc else a_out.mean(axis=0, keepdims=True)
            sp = _env_step(svec, am)
            cols["s"][:, rows + t] = svec; cols["a"][:, rows + t] = a_out; cols["sp"][:, rows + t] = sp
            if on_policy:
                cols["logprob"][0, rows + t] = lp
            svec = np.asfortranarray(sp); steps_taken = steps_taken + 1
            t2 = time.perf_counter()
            t_explore.append(t1 - t0); t_env.append(t2 - t1)
        t0 = time.perf_counter()
        fr = C.c_int64()
        ctx.check(lib.crux_steps_push(buf.h, N, ptrs, T, 1, critic.h if critic is not None else None, 0.95, 0.99, None, None, 0, C.byref(fr)))
        t_push.append(time.perf_counter() - t0)

    block(0); block(N)                                           # warm-up (scratch blocks, pinned staging, kernel load)
    del t_explore[:], t_push[:], t_env[:]
    ctx.sync(); w0 = time.perf_counter()
    for b in range(blocks):
        block((2 + b) * N)
    ctx.sync(); wall = time.perf_counter() - w0
    te, tp = np.array(t_explore) * 1e6, np.array(t_push) * 1e6
    out = {"policy": "%s %s" % ("->".join(map(str, dims)), head), "n_envs": E, "steps_per_block": T, "blocks_timed": blocks,
           "us_per_policy_explore_call": {"median": float(np.median(te)), "p90": float(np.percentile(te, 90)), "min": float(te.min())},
           "us_per_steps_push": {"median": float(np.median(tp)), "rows": N, "fills": "fill_gae! + fill_returns!" if on_policy else "none (off-policy ring)"},
           "us_per_env_step_host_side": float(np.median(np.array(t_env)) * 1e6),
           "env_steps_per_s_end_to_end": E * T * blocks / wall,
           "share_of_wall": {"policy_explore": float(te.sum() * 1e-6 / wall), "steps_push": float(tp.sum() * 1e-6 / wall), "numpy_env_and_block_fill": float(np.sum(t_env) / wall)}}
    # the device-environment rollout of the same policy shape, E and T (crux_rollout: environment, policy and buffer write in one launch)
    try:
        if kind == "cartpole":
            mdp = crux.CartPoleMDP(n_envs=E, seed=3)
        else:
            mdp = crux.SynthMDP(od, ad, n_envs=E, seed=3, discrete=disc)
        agent = crux.PolicyParams(crux.ActorCritic(pi, critic)) if on_policy else crux.PolicyParams(pi, pi_explore=crux.EpsGreedyPolicy(crux.LinearDecaySchedule(1.0, 0.1, 50000), list(range(1, ad + 1))))
        smp = crux.Sampler(mdp, agent, max_steps=1000, required_columns=extras, lam=0.95 if on_policy else float("nan"), ctx=ctx)
        dbuf = crux.ExperienceBuffer(S, A, max(N, 1 << 16), extras, ctx=ctx)
        for _ in range(2):
            crux.steps_(smp, dbuf, Nsteps=N, explore=True, i=0, reset=True, want_info=False)
        ctx.sync(); d0 = time.perf_counter()
        for b in range(blocks):
            crux.steps_(smp, dbuf, Nsteps=N, explore=True, i=(b + 1) * N, reset=True, want_info=False)
        ctx.sync(); dt = time.perf_counter() - d0
        out["device_environment_env_steps_per_s"] = E * T * blocks / dt
        out["device_environment_note"] = "crux_rollout + the GAE / return fills on the same E x T blocks (%s dynamics on the device)" % kind
    except Exception as e:      # noqa: BLE001
        out["device_environment_error"] = repr(e)[:300]
    return out


def run(crux, ctx, ab=True):
    """ab: also time the round-5 form of the two calls (CRUX_HOST_ZEROCOPY=0: staged uploads and read-backs instead of the pinned, device-mapped block) beside the default"""
    out = {"note": "the caller-stepped environment seam (crux_policy_explore + crux_steps_push) under a trivial vectorised numpy environment; reference seam: src/sampler.jl:71-137"}
    for name in CONFIGS:
        try:
            out[name] = one(crux, ctx, name)
            if ab:
                os.environ["CRUX_HOST_ZEROCOPY"] = "0"; crux.reload_switches()
                try:
                    old = one(crux, ctx, name)
                    out[name]["round5_form_staged_copies"] = {k: old[k] for k in ("us_per_policy_explore_call", "us_per_steps_push", "env_steps_per_s_end_to_end")}
                finally:
                    os.environ.pop("CRUX_HOST_ZEROCOPY", None); crux.reload_switches()
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": repr(e)[:400]}
    return out


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    import crux_jl_amd as crux
    ctx = crux.default_context()
    print(json.dumps(run(crux, ctx)))
