"""solve(::OffPolicySolver) end to end on wide networks: DQN + PER (8-256-256-4, SYNTH discrete env, 4 env steps + 4 epochs per iteration) and SAC (Pendulum,
3-256-256-1 + twin Q, 50 + 50): microseconds per iteration, against the value_training share measured by bench_offpolicy.py."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import crux_jl_amd as crux

def chain(dims, acts): return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])
acts = ["relu", "relu", "identity"]

def dqn(iters):
    S = crux.ContinuousSpace(8)
    q = crux.DiscreteNetwork(chain([8, 256, 256, 4], acts), [1, 2, 3, 4], seed=3)
    sv = crux.DQN(q, S, N=1200 + 4 * iters, dN=4, buffer_size=100000, prioritized=True, buffer_init=1200, max_steps=200, c_opt={"batch_size": 128})
    mdp = crux.SynthMDP(8, 4, discrete=True, n_envs=1, seed=5, discount=0.99)
    sv.N = 1200 + 4 * 50; crux.solve(sv, mdp); sv.buffer.ctx.sync()
    t0 = time.perf_counter(); i0 = sv.i; sv.N = sv.i + 4 * iters; crux.solve(sv, mdp); sv.buffer.ctx.sync()
    return 1e6 * (time.perf_counter() - t0) / ((sv.i - i0) // 4)

def sac(iters):
    S = crux.ContinuousSpace(3)
    pi = crux.ActorCritic(crux.GaussianPolicy(chain([3, 256, 256, 1], acts), np.zeros(1, np.float32), seed=2),
                          crux.DoubleNetwork(crux.ContinuousNetwork(chain([4, 256, 256, 1], acts), seed=3), crux.ContinuousNetwork(chain([4, 256, 256, 1], acts), seed=4)))
    sv = crux.SAC(pi, S, N=1000 + 50 * iters, dN=50, buffer_size=100000, buffer_init=1000, max_steps=200, c_opt={"batch_size": 256}, a_opt={"batch_size": 256}, SAC_alpha_opt={"batch_size": 256})
    mdp = crux.PendulumMDP(n_envs=1, seed=8)
    sv.N = 1000 + 50 * 2; crux.solve(sv, mdp); sv.buffer.ctx.sync()
    t0 = time.perf_counter(); i0 = sv.i; sv.N = sv.i + 50 * iters; crux.solve(sv, mdp); sv.buffer.ctx.sync()
    return 1e6 * (time.perf_counter() - t0) / ((sv.i - i0) // 50)

print("DQN + PER solve: %.0f us per iteration (4 env steps + 4 epochs)" % dqn(400))
print("SAC solve: %.0f us per iteration (50 env steps + 50 epochs)" % sac(20))
