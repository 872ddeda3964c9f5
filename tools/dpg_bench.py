"""TD3 / DDPG value_training epochs (3-256-256-1 actor, 4-256-256-1 critics, B = 256, 50 epochs per call) through crux_dpg_epochs against the call-by-call loop: us per epoch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import crux_jl_amd as crux

def chain(dims, acts): return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])

def run(algo, fused, reps=6):
    ctx = crux.default_context()
    rng = np.random.default_rng(1); B, n = 256, 100_000
    S, A = crux.ContinuousSpace(3), crux.ContinuousSpace(1)
    buf = crux.ExperienceBuffer(S, A, n); D = crux.buffer_like(buf, capacity=B)
    buf.push_({"s": rng.normal(0, 1, (3, n)).astype(np.float32), "a": rng.uniform(-1, 1, (1, n)).astype(np.float32), "sp": rng.normal(0, 1, (3, n)).astype(np.float32),
               "r": rng.normal(-1, 1, (1, n)).astype(np.float32), "done": np.zeros((1, n), bool), "episode_end": np.zeros((1, n), bool)})
    twin = algo == "td3"; acts = ["relu", "relu", "identity"]
    q = lambda sd: crux.ContinuousNetwork(chain([4, 256, 256, 1], acts), seed=sd)
    pi = crux.ActorCritic(crux.ContinuousNetwork(chain([3, 256, 256, 1], ["relu", "relu", "tanh"]), seed=2), crux.DoubleNetwork(q(3), q(4)) if twin else q(3))
    opt = {"batch_size": B, "epochs": 50}
    sv = (crux.TD3 if twin else crux.DDPG)(pi, S, N=10**9, dN=50, c_opt=dict(opt), a_opt=dict(opt, **({"update_every": 2} if twin else {})), buffer=buf, noise_seed=5)
    sv.fused_epochs = fused; sv.batch = D
    def it():
        sv.i += 50; crux.value_training(sv, D, np.float32(0.99))
    it(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): it()
    ctx.sync(); return 1e6 * (time.perf_counter() - t0) / (50 * reps)

for algo in ("ddpg", "td3"):
    print(algo, "chained %.1f us/epoch" % run(algo, True), "| call-by-call %.1f us/epoch" % run(algo, False))
