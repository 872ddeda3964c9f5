#!/usr/bin/env python
"""Where the time of one multi-seed PPO iteration goes (S problems): rollout / whiten / training call, wall-clock with a sync after each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import crux_jl_amd as crux
from crux_jl_amd import dist as cdist
S = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ctx = crux.Context(0); crux.set_default_context(ctx)
ctx.set_learner_cus(int(os.environ.get("LEARNER_CUS", "0")))
probs = [bench.build_problem(crux, cdist.shard_seed(1000, r)) for r in range(S)]
P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
am = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bench.BATCH, epochs=bench.EPOCHS, target_kl=None, name="actor_", shuffle_seed=5000)
cm = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bench.BATCH, epochs=bench.EPOCHS, name="critic_", shuffle_seed=6000)
if len(sys.argv) > 2 and sys.argv[2] == "single-first":      # the order bench.py runs things in
    a1 = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bench.BATCH, epochs=2, target_kl=None, name="actor_", shuffle_seed=1)
    c1 = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bench.BATCH, epochs=2, name="critic_", shuffle_seed=2)
    bench.ppo_iteration(crux, probs[0][0], probs[0][1], probs[0][2], a1, c1, P, 0)
if len(sys.argv) > 2 and sys.argv[2] == "null-first":        # a null-stream copy before the second learner stream exists: used to make it share a hardware queue
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so"); d = ctx.alloc(1024); h = ctypes.create_string_buffer(1024)
    assert hip.hipMemcpy(d if isinstance(d, ctypes.c_void_p) else ctypes.c_void_p(int(d)), h, ctypes.c_size_t(1024), 1) == 0
for k in range(3):
    ctx.sync(); t0 = time.perf_counter()
    crux.steps_multi_([q[2] for q in probs], [q[1] for q in probs], Nsteps=probs[0][1].capacity, explore=True, i=k * probs[0][1].capacity, reset=True)
    ctx.sync(); t1 = time.perf_counter()
    crux.whiten_multi_([q[1] for q in probs], "advantage")
    ctx.sync(); t2 = time.perf_counter()
    ctx.prof_reset(); ctx.prof_enable(True)
    crux.policy_gradient_training_multi([q[0] for q in probs], am, cm, P, [q[1] for q in probs])
    ctx.sync(); t3 = time.perf_counter(); ctx.prof_enable(False)
    print("S=%d iter %d: rollout+gae %.1f ms, whiten %.1f ms, training call %.1f ms (actor launch %.1f ms)" % (S, k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), ctx.prof_get("train_actor")[0]))
