"""Stand-alone launches of the dense engine's kernels (forward with cached activations, pullback) for rocprofv3 --kernel-trace --stats: per-kernel durations of
k_fwd12 / k_wgrad2 / k_dgrad2w1 / k_gemm16 outside the phase kernel. usage: python tools/dense_micro.py [B] [in] [iters]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import crux_jl_amd as crux, parity
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128; no = int(sys.argv[2]) if len(sys.argv) > 2 else 8; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
out = 4 if no == 8 else 1
q = crux.DiscreteNetwork(parity.chain([no, 256, 256, out], ["relu", "relu", "identity"]), list(range(1, out + 1)), seed=5) if out > 1 else crux.ContinuousNetwork(parity.chain([no, 256, 256, out], ["relu", "relu", "identity"]), seed=5)
q.attach_optimizer(crux.Adam(np.float32(1e-3)))
ctx = q.ctx; lib = ctx.lib
x = np.asfortranarray(np.random.default_rng(0).normal(0, 1, (no, B)).astype(np.float32)); dy = np.asfortranarray(np.random.default_rng(1).normal(0, 1, (out, B)).astype(np.float32))
d_x, d_y, d_dy = ctx.alloc(x.nbytes), ctx.alloc(4 * out * B), ctx.alloc(dy.nbytes); ctx.h2d(d_x, x); ctx.h2d(d_dy, dy)
import time
for rep in range(2):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(iters):
        ctx.check(lib.crux_mlp_forward_cached(q.h, d_x, B, None))
        ctx.check(lib.crux_mlp_backward(q.h, d_x, B, d_dy, 1.0, 1, None))
    ctx.sync(); t = time.perf_counter() - t0
print("B %d in %d: %.1f us per forward + pullback (host-timed, %d iterations)" % (B, no, 1e6 * t / iters, iters))
