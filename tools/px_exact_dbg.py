import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity, test_gpu_peer as T
from parity import crux
which = sys.argv[1] if len(sys.argv) > 1 else "actor"
c0 = crux.default_context(); c1 = crux.Context(0); crux.peer_attach_local([c0, c1])
shard = T._shard(210); N = shard["s"].shape[1]; extras = ["return", "logprob", "advantage"]; bs = 128
dims = parity.ACTOR_DIMS if which == "actor" else parity.CRITIC_DIMS
perms = np.stack([np.random.default_rng(6).permutation(N)])
P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
def mk(ctx):
    ch = parity.chain(dims, parity.ACTS)
    g = crux.DiscreteNetwork(ch, [1, 2], ctx=ctx, seed=79, stream=3) if which == "actor" else crux.ContinuousNetwork(ch, ctx=ctx, seed=79, stream=3)
    b = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), N, extras, ctx=ctx); b.push_(shard); return g, b
pairs = [mk(c0), mk(c1)]
opt = lambda: crux.TrainingParams(loss=crux.ppo_loss if which == "actor" else crux.value_mse_loss, batch_size=bs, epochs=1, name="n_", max_batches=1)
T._run_threads([lambda r=r: crux.batch_train_(pairs[r][0], opt(), P, pairs[r][1], perms=perms + 1) for r in range(2)])
c3 = crux.Context(0); g, b = mk(c3); crux.batch_train_(g, opt(), P, b, perms=perms + 1)
m0, v0, _ = pairs[0][0].adam_state(); m1, v1, _ = g.adam_state()
d = np.abs(m0 - m1); idx = np.nonzero(d)[0]
print(which, "after ONE step: params differ at", int((pairs[0][0].get_params() != g.get_params()).sum()), "of", m0.size, "; m differs at", idx.size, "first", idx[:12], "max rel", float((d / (np.abs(m1) + 1e-30)).max()) if idx.size else 0)
nd = dims; offs = []; o = 0
for l in range(3):
    offs.append(("W%d" % l, o, o + nd[l] * nd[l + 1])); o += nd[l] * nd[l + 1]; offs.append(("b%d" % l, o, o + nd[l + 1])); o += nd[l + 1]
for name, a, e in offs:
    print("  ", name, int((d[a:e] != 0).sum()), "of", e - a)
import ctypes as C
from parity import L, O
ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, N, extras); ob.push(shard)
o = O.OMlp(dims, parity.ACTS).init_glorot(79, 3).adam_init(float(np.float32(3e-4)))
cfg = parity.train_cfg("ppo" if which == "actor" else "value_mse", "categorical" if which == "actor" else "deterministic", bs, 1, -1.0, 0, max_batches=1); oi = np.zeros(L.INFO_N, np.float32)
O.chk(O.lib().orc_batch_train(o.h, ob.h, C.byref(cfg), O.vpz(perms.astype(np.int64).copy()), O.vpz(oi), None))
om, ov, _ = o.adam_state()
print("vs oracle m: group max rel %.3g | ungrouped max rel %.3g" % (float((np.abs(m0 - om) / (np.abs(om) + 1e-12)).max()), float((np.abs(m1 - om) / (np.abs(om) + 1e-12)).max())))
print("abs: group %.3g ungrouped %.3g ; max|m| %.3g" % (float(np.abs(m0 - om).max()), float(np.abs(m1 - om).max()), float(np.abs(om).max())))
