"""Whole solve(DQN + prioritized replay) iterations on the C3 shapes (8-256-256-4, B = 128, dN = 4, synthetic 8-observation / 4-action environment): wall time per iteration
against the time of its four value_training epochs -- how much of an iteration the host costs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import crux_jl_amd as crux

def chain(dims, acts): return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])
N_BUF = int(os.environ.get("C3S_BUF", "200000"))
mdp = crux.SynthMDP(8, 4, discrete=True, n_envs=1, seed=3)
S = crux.ContinuousSpace(8)
q = crux.DiscreteNetwork(chain([8, 256, 256, 4], ["relu", "relu", "identity"]), [1, 2, 3, 4], seed=1)
iters = int(os.environ.get("C3S_ITERS", "600"))
buf = None
if os.environ.get("C3S_PREFILL"):      # fill the ring with synthetic transitions instead of a rollout of N_BUF steps
    rng = np.random.default_rng(5)
    d = {"s": rng.standard_normal((8, N_BUF)).astype(np.float32), "sp": rng.standard_normal((8, N_BUF)).astype(np.float32), "r": rng.standard_normal((1, N_BUF)).astype(np.float32),
         "done": rng.random((1, N_BUF)) < 0.05, "episode_end": rng.random((1, N_BUF)) < 0.05}
    a = np.zeros((4, N_BUF), np.bool_); a[rng.integers(0, 4, N_BUF), np.arange(N_BUF)] = True; d["a"] = a
    buf = crux.ExperienceBuffer(S, crux.DiscreteSpace(4), N_BUF, prioritized=True); buf.push_(d)
sv = crux.DQN(q, S, N=4 * iters, dN=4, buffer_size=N_BUF, buffer=buf, buffer_init=N_BUF, prioritized=True, weighted_loss=True, max_steps=200,
              c_opt={"batch_size": 128, "optimizer": crux.Adam(np.float32(1e-3))})
ctx = q.ctx
t0 = time.perf_counter(); crux.solve(sv, mdp); ctx.sync(); t1 = time.perf_counter()          # first call: fills the ring (buffer_init) + iterations
sv.N = 4 * iters
t2 = time.perf_counter(); crux.solve(sv, mdp); ctx.sync(); t3 = time.perf_counter()
print("solve(DQN + PER, 8-256-256-4, B 128, dN 4, ring %d): %.1f us per iteration over %d iterations (first call incl. ring fill: %.2f s)" % (N_BUF, 1e6 * (t3 - t2) / iters, iters, t1 - t0))
