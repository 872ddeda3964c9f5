"""debug: the two-replica test after a SAC solve on the same context"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import parity
from parity import crux
import test_gpu_sac as TS
import test_gpu_peer as TP
mode = sys.argv[1] if len(sys.argv) > 1 else "sac"
ctx = crux.Context(0)
if mode != "none":
    TS.test_sac_solve_matches_oracle_loop(ctx)
if mode == "sync":
    ctx.sync()
c1 = crux.Context(0)
order = [ctx, c1] if mode != "swap" else [c1, ctx]
crux.peer_attach_local(order)
try:
    TP.test_two_replicas_equal_the_single_learner_on_the_concatenated_batch(tuple(order), "actor")
    print("PASS", mode)
except AssertionError as e:
    print("FAIL", mode, str(e)[:100])
