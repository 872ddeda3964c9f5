"""(test infrastructure: it builds its shards with the oracle, like the tests) stress of the in-kernel replica exchange on ONE GPU: R contexts, actor || critic persistent learners per replica, thousands of minibatch steps;
the replicas must end bit-identical (any lost or torn slot read shows up as a divergence)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import parity
from parity import crux
import test_gpu_peer as TP

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
EPOCHS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctxs = [crux.Context(0) for _ in range(R)]
crux.set_default_context(ctxs[0])
crux.peer_attach_local(ctxs)
shards = [TP._shard(900 + r, E=8, T=128) for r in range(R)]
N = shards[0]["s"].shape[1]; extras = ["return", "logprob", "advantage"]
sv, bufs = [], []
for r, ctx in enumerate(ctxs):
    a = crux.DiscreteNetwork(parity.chain(parity.ACTOR_DIMS, parity.ACTS), [1, 2], ctx=ctx, seed=9, stream=0)
    c = crux.ContinuousNetwork(parity.chain(parity.CRITIC_DIMS, parity.ACTS), ctx=ctx, seed=9, stream=1)
    b = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), N, extras, ctx=ctx); b.push_(shards[r])
    class _S: pass
    s = _S(); s.agent = crux.PolicyParams(crux.ActorCritic(a, c)); s.P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    s.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=EPOCHS, target_kl=None, name="actor_", shuffle_seed=40 + r)
    s.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=EPOCHS, name="critic_", shuffle_seed=60 + r)
    sv.append(s); bufs.append(b)
infos = [None] * R
def make(r):
    def f(): infos[r] = crux.policy_gradient_training(sv[r], bufs[r])
    return f
t0 = time.time(); TP._run_threads([make(r) for r in range(R)]); dt = time.time() - t0
steps = infos[0]["actor_batches_trained"]
ok = all(np.array_equal(sv[0].agent.pi.A.get_params(), sv[r].agent.pi.A.get_params()) and np.array_equal(sv[0].agent.pi.C.get_params(), sv[r].agent.pi.C.get_params()) for r in range(1, R))
print("R=%d: %d actor + %d critic steps per replica in %.2f s (%.1f us per step), replicas bit-identical: %s" % (R, steps, infos[0]["critic_batches_trained"], dt, 1e6 * dt / steps, ok))
sys.exit(0 if ok else 1)
