"""Development check of the persistent DQN kernels (dqn_persist.h): crux_dqn_epochs with CRUX_DQN_PERSIST=1 against the phase launches (=0) from the same state."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import crux_jl_amd as crux
from crux_jl_amd import _lib as L
import ctypes as C

def chain(dims, acts): return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])
def run(persist, n_ep, per=True, B=128, N=20_000, no=8, na=4):
    os.environ["CRUX_DQN_PERSIST"] = "1" if persist else "0"; crux.reload_switches()
    rng = np.random.default_rng(3)
    S, A = crux.ContinuousSpace(no), crux.DiscreteSpace(na)
    buf = crux.ExperienceBuffer(S, A, N, prioritized=per); D = crux.buffer_like(buf, capacity=B)
    a = np.zeros((na, N), bool); a[rng.integers(0, na, N), np.arange(N)] = True
    buf.push_({"s": rng.normal(0, 1, (no, N)).astype(np.float32), "a": a, "sp": rng.normal(0, 1, (no, N)).astype(np.float32), "r": rng.normal(0, 1, (1, N)).astype(np.float32),
               "done": rng.random((1, N)) < 0.02, "episode_end": np.zeros((1, N), bool)})
    if per: buf.update_priorities_(np.arange(1, N + 1), (np.abs(rng.normal(0, 1, N)) + 1e-3).astype(np.float32))
    q = crux.DiscreteNetwork(chain([no, 256, 256, na], ["relu", "relu", "identity"]), list(range(1, na + 1)), seed=5)
    qm = crux.clone_policy(q); q.attach_optimizer(crux.Adam(np.float32(1e-3)))
    ctx = q.ctx; infos = np.zeros((n_ep, L.INFO_N), np.float32)
    ctx.check(ctx.lib.crux_dqn_epochs(q.h, qm.h, buf.h, D.h, 0.99, 1 if per else 0, 0.6, 40, n_ep, infos.ctypes.data_as(L.vp)))
    pr = buf.priority_params()["priorities"] if per else np.zeros(1)
    ids = D.indices.copy()
    return q.get_params(), pr, ids, D["s"], infos

import sys as _s
for per in ((True,) if len(_s.argv) > 1 else (True, False)):
    for B in (128, 64):
        for n_ep in ((2,) if len(_s.argv) > 1 else (1, 2, 6)):
            a = run(True, n_ep, per, B); b = run(False, n_ep, per, B)
            dp = np.abs(a[0] - b[0]).max(); dpr = np.abs(a[1] - b[1]).max(); ids = np.array_equal(a[2], b[2]); ds = np.abs(a[3] - b[3]).max()
            print("per=%d B=%d epochs=%d: |dtheta| %.3g (|theta| max %.3g, moved %.3g)  |dprio| %.3g  ids equal %s  |drows| %.3g  loss %s / %s  gn %s / %s" % (
                per, B, n_ep, dp, np.abs(b[0]).max(), 0.0, dpr, ids, ds, a[4][:, 0], b[4][:, 0], a[4][:, 1], b[4][:, 1]), flush=True)
