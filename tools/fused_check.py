"""Development check for dense_fused.h: the same chained epochs with CRUX_DENSE_FUSED=0 (one Gemm16 launch per layer) and with the fused block kernels must give the
same bits. usage: python tools/fused_check.py            (spawns itself twice per algorithm and compares)"""
import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_dqn(dims=(8, 256, 256, 4), B=128):
    import crux_jl_amd as crux, parity
    from crux_jl_amd import _lib as L
    rng = np.random.default_rng(3); N = 20_000; no = dims[0]
    S, A = crux.ContinuousSpace(no), crux.DiscreteSpace(4)
    buf = crux.ExperienceBuffer(S, A, N, prioritized=True); D = crux.buffer_like(buf, capacity=B)
    a = np.zeros((4, N), bool); a[rng.integers(0, 4, N), np.arange(N)] = True
    buf.push_({"s": rng.normal(0, 1, (no, N)).astype(np.float32), "a": a, "sp": rng.normal(0, 1, (no, N)).astype(np.float32), "r": rng.normal(0, 1, (1, N)).astype(np.float32),
               "done": rng.random((1, N)) < 0.02, "episode_end": np.zeros((1, N), bool)})
    buf.update_priorities_(np.arange(1, N + 1), (np.abs(rng.normal(0, 1, N)) + 1e-3).astype(np.float32))
    q = crux.DiscreteNetwork(parity.chain(list(dims), ["relu"] * (len(dims) - 2) + ["identity"]), [1, 2, 3, 4], seed=5)
    qm = crux.clone_policy(q); q.attach_optimizer(crux.Adam(np.float32(1e-3)))
    ctx = q.ctx; infos = np.zeros((12, L.INFO_N), np.float32)
    ctx.check(ctx.lib.crux_dqn_epochs(q.h, qm.h, buf.h, D.h, 0.99, 1, 0.6, 40, 12, infos.ctypes.data_as(L.vp)))
    return {"p": q.get_params(), "pr": buf.priority_params()["priorities"], "ids": D.indices.copy(), "s": D["s"], "infos": infos}


def run_sac(B=256):
    import crux_jl_amd as crux, parity
    S = crux.ContinuousSpace(3); acts = ["relu", "relu", "identity"]
    pi = crux.ActorCritic(crux.GaussianPolicy(parity.chain([3, 256, 256, 1], acts), np.zeros(1, np.float32), seed=2),
                          crux.DoubleNetwork(crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=3), crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], acts), seed=4)))
    sv = crux.SAC(pi, S, N=420, dN=6, buffer_size=1000, buffer_init=300, max_steps=50, c_opt={"batch_size": B}, a_opt={"batch_size": B}, SAC_alpha_opt={"batch_size": B})
    crux.solve(sv, crux.PendulumMDP(n_envs=1, seed=8))
    nets = (pi.A, pi.C.N1, pi.C.N2, sv.agent.pi_minus.C.N1, sv.P["SAC_log_alpha"])
    out = {"n%d" % k: n.get_params() for k, n in enumerate(nets)}
    out["loss"] = np.array([sv.history[-1][k] for k in ("critic_loss", "actor_loss", "temp_loss")], np.float64)
    return out


def run_dpg(algo="td3", B=256):
    import crux_jl_amd as crux, parity
    S = crux.ContinuousSpace(3)
    A = crux.ContinuousNetwork(parity.chain([3, 256, 256, 1], ["relu", "relu", "tanh"]), seed=2)
    mk = lambda sd: crux.ContinuousNetwork(parity.chain([4, 256, 256, 1], ["relu", "relu", "identity"]), seed=sd)
    C = crux.DoubleNetwork(mk(3), mk(4)) if algo == "td3" else mk(3)
    pi = crux.ActorCritic(A, C)
    mkr = crux.TD3 if algo == "td3" else crux.DDPG
    sv = mkr(pi, S, N=420, dN=6, buffer_size=1000, buffer_init=300, max_steps=50, c_opt={"batch_size": B}, a_opt={"batch_size": B})
    crux.solve(sv, crux.PendulumMDP(n_envs=1, seed=8))
    nets = (pi.A,) + ((pi.C.N1, pi.C.N2) if algo == "td3" else (pi.C,))
    return {"n%d" % k: n.get_params() for k, n in enumerate(nets)}


if __name__ == "__main__":
    if len(sys.argv) > 2:
        what, path = sys.argv[1], sys.argv[2]
        out = run_dqn() if what == "dqn" else run_dqn((8, 128, 128, 4), 64) if what == "dqn128" else run_sac() if what == "sac" else run_dpg(what)
        np.savez(path, **out); sys.exit(0)
    bad = 0
    for what in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["dqn", "dqn128", "sac", "td3", "ddpg"]):
        res = []
        for f in ("0", "1"):
            env = dict(os.environ, CRUX_DENSE_FUSED=f, CRUX_PER_FUSED_GATHER=f, CRUX_SAC_TILE_OPS=f); path = "/tmp/fc_%s_%s.npz" % (what, f)      # "0": the round-3 chains (per-layer launches, search and gather apart)
            r = subprocess.run([sys.executable, __file__, what, path], env=env, capture_output=True, text=True)
            if r.returncode: print(what, "fused=" + f, "FAILED\n", r.stderr[-1500:]); bad += 1; res = None; break
            res.append(np.load(path))
        if res is None: continue
        for k in res[0].files:
            same = np.array_equal(res[0][k], res[1][k], equal_nan=True)
            d = float(np.nanmax(np.abs(res[0][k].astype(np.float64) - res[1][k].astype(np.float64)))) if res[0][k].size else 0.0
            print("%-7s %-6s %s  max|d| %.3g" % (what, k, "identical" if same else "DIFFERENT", d)); bad += 0 if same else 1
    print("fused_check:", "OK" if not bad else "%d differences" % bad)
    sys.exit(1 if bad else 0)
