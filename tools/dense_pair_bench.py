"""policy_gradient_training for learners of the dense engine (8-128-128-4 actor, 8-128-128-1 critic, 65 536 rows, minibatches of 128): the two chains on two streams from two host
threads (default) against one after the other (CRUX_DENSE_PAIR=0) -- microseconds per minibatch step of a learner."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import crux_jl_amd as crux

def chain(dims, acts): return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])
W = int(os.environ.get("DPB_WIDTH", "128")); N, bs, epochs = 65536, 128, 4
rng = np.random.default_rng(0); acts = ["relu", "relu", "identity"]
A = crux.DiscreteNetwork(chain([8, W, W, 4], acts), [1, 2, 3, 4], seed=1); Cn = crux.ContinuousNetwork(chain([8, W, W, 1], acts), seed=2)
buf = crux.ExperienceBuffer(crux.ContinuousSpace(8), crux.DiscreteSpace(4), N, ["return", "logprob", "advantage"])
ai = rng.integers(0, 4, N)
buf.push_({"s": rng.normal(0, 1, (8, N)).astype(np.float32), "a": np.eye(4, dtype=bool)[:, ai], "sp": rng.normal(0, 1, (8, N)).astype(np.float32), "r": np.ones((1, N), np.float32),
           "done": np.zeros((1, N), bool), "episode_end": np.zeros((1, N), bool), "return": rng.normal(0, 1, (1, N)).astype(np.float32),
           "logprob": rng.normal(-1.4, 0.05, (1, N)).astype(np.float32), "advantage": rng.normal(0, 1, (1, N)).astype(np.float32)})
P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=epochs, target_kl=None, name="actor_", shuffle_seed=1)
c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=epochs, name="critic_", shuffle_seed=2)
class _Agent:
    pass
class _Solver:          # the fields policy_gradient_training reads
    pass
sv = _Solver(); sv.agent = _Agent(); sv.agent.pi = crux.ActorCritic(A, Cn); sv.a_opt, sv.c_opt, sv.P, sv.param_optimizers = a_opt, c_opt, P, []
ctx = buf.ctx
for mode in ("1", "0", "1", "0"):
    os.environ["CRUX_DENSE_PAIR"] = mode; crux.reload_switches()
    ctx.sync(); t0 = time.perf_counter(); info = crux.policy_gradient_training(sv, buf); ctx.sync(); dt = time.perf_counter() - t0
    steps = info["actor_batches_trained"]
    print("CRUX_DENSE_PAIR=%s width %d: %.1f us per (actor + critic) minibatch step, %d steps each, actor loss %.6f critic loss %.6f" % (mode, W, 1e6 * dt / steps, steps, info["actor_loss"], info["critic_loss"]), flush=True)
