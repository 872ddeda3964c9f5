"""batch_train!(actor, lagrange_ppo_loss) on a 32 x 2048 CartPole rollout with the cost channel (4-64-64-2, batch 128): microseconds per minibatch step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import crux_jl_amd as crux

def chain(dims, acts): return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])
EXTRAS = ["return", "logprob", "advantage", "cost", "cost_advantage", "cost_return"]
E, T = 32, 2048
acts = ["relu", "relu", "identity"]
A = crux.DiscreteNetwork(chain([4, 64, 64, 2], acts), [1, 2], seed=1); Cn = crux.ContinuousNetwork(chain([4, 64, 64, 1], acts), seed=2); Vc = crux.ContinuousNetwork(chain([4, 64, 64, 1], acts), seed=3)
mdp = crux.CartPoleMDP(n_envs=E, seed=5)
buf = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), E * T, EXTRAS)
smp = crux.Sampler(mdp, crux.ActorCritic(A, Cn), max_steps=12, required_columns=EXTRAS, lam=0.95, Vc=Vc)
crux.steps_(smp, buf, Nsteps=E * T, explore=True, i=0, reset=True); crux.whiten_(buf, "advantage")
from crux_jl_amd import _lib as L
lag = L.Lagrange(); lag.target_cost, lag.penalty_max, lag.Ki_max, lag.Ki, lag.Kp, lag.Kd, lag.ema_alpha = 0.05, float("inf"), 10.0, 1e-3, 1.0, 0.5, 0.95
P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1, "lagrange": lag}
ctx = buf.ctx
for epochs in (2, 10):
    p = crux.TrainingParams(loss=crux.lagrange_ppo_loss, batch_size=128, epochs=epochs, name="actor_", shuffle_seed=3)
    ctx.sync(); t0 = time.perf_counter(); info = crux.batch_train_(A, p, P, buf); ctx.sync(); dt = time.perf_counter() - t0
    print("epochs %d: %.2f us per minibatch step (%d steps), penalty %.4f" % (epochs, 1e6 * dt / info["actor_batches_trained"], info["actor_batches_trained"], info.get("penalty", float("nan"))))
