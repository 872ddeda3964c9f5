import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import crux_jl_amd as crux
def chain(dims, acts): return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])
for name, mdp, S, A, pol in (
    ("cartpole 4-32-32-2", crux.CartPoleMDP(n_envs=32, seed=5), crux.ContinuousSpace(4), crux.DiscreteSpace(2), lambda: crux.DiscreteNetwork(chain([4, 32, 32, 2], ["relu", "relu", "identity"]), [1, 2], seed=1)),
    ("pendulum 3-256-256-1", crux.PendulumMDP(n_envs=32, seed=5), crux.ContinuousSpace(3), crux.ContinuousSpace(1), lambda: crux.GaussianPolicy(chain([3, 256, 256, 1], ["relu", "relu", "identity"]), np.zeros(1, np.float32), seed=1))):
    E, T = 32, 1024
    buf = crux.ExperienceBuffer(S, A, E * T, ["logprob"])
    smp = crux.Sampler(mdp, pol(), max_steps=200, required_columns=["logprob"])
    ctx = buf.ctx
    crux.steps_(smp, buf, Nsteps=E * T, explore=True, i=0, reset=True); ctx.sync()
    t0 = time.perf_counter()
    for k in range(3): crux.steps_(smp, buf, Nsteps=E * T, explore=True, i=(k + 1) * E * T, reset=True)
    ctx.sync(); dt = (time.perf_counter() - t0) / 3
    print("%s: %.2f us per rollout step (32 envs x %d steps in %.2f ms)" % (name, 1e6 * dt / T, T, 1e3 * dt))
