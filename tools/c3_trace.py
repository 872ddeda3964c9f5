"""C3 (DQN + PER, 8-256-256-4, buffer 1 M, B = 128): a few hundred value_training epochs for rocprofv3 --kernel-trace (tools: phase durations and gaps)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import crux_jl_amd as crux
import bench_offpolicy
ctx = crux.default_context()
out = bench_offpolicy.c3(crux, ctx, cpu=False) if hasattr(bench_offpolicy, "c3") else bench_offpolicy.run(crux, ctx, False)
print({k: v for k, v in out.items() if k in ("us_per_epoch", "grad_steps_per_s")} if isinstance(out, dict) else out)
