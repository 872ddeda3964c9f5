"""C4 (SAC, 3-256-256-1 + twin Q, B = 256): value_training epochs for rocprofv3 --kernel-trace (phase durations and gaps)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import crux_jl_amd as crux
import bench_offpolicy
ctx = crux.default_context()
out = bench_offpolicy.c4(crux, ctx, cpu=False)
print({k: v for k, v in out.items() if k in ("us_per_epoch", "epochs_per_s")})
