"""Supplementary lines of bench.py's JSON ("other_configs"): the other BASELINE.json configs on ONE MI355X, each with its own roofline and a CPU
baseline from the oracle timed on a bounded sample of the same workload, so that the driver's default `python bench.py` run records them.

  c5  configs[4], one GPU's shard: PPO 128 envs x 2048 steps on the 17-obs / 6-act synthetic env, tanh GaussianPolicy 17-64-64-6 + critic
  c3  configs[2]: DQN + prioritized replay, 8-256-256-4, buffer 1 M, B = 128
  c4  configs[3]: SAC, actor 3-256-256-1 + twin Q 4-256-256-1, B = 256
  c1  configs[0]: DQN on SimpleGridWorld, 2-8-4, N = 100 000, dN = 4 (the README example)
  host_env  the caller-stepped environment seam at C2 / C3 / C5 shapes (bench_hostenv.py)

Every entry is wrapped: a failure here never takes the headline measurement down."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS = 157.3, 8000.0


def _chain(crux, dims, acts):
    return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])


def _timed(ctx, fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.sync(); return (time.perf_counter() - t0) / steps


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    import parity
    from crux_jl_amd import _lib as L
    return O, parity, L


def c5(crux, ctx, cpu=True):
    import bench
    E, T, B, EP = 128, 2048, 128, bench.EPOCHS
    pi, buf, sampler = bench.build_problem(crux, 12345, workload="c5")
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.0}
    a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=B, epochs=EP, target_kl=None, name="actor_", shuffle_seed=71)
    c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=B, epochs=EP, name="critic_", shuffle_seed=72)
    it = [0]

    def one():
        bench.ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, P, it[0]); it[0] += 1
    one(); ctx.sync(); ctx.prof_reset(); ctx.prof_enable(True)
    t = _timed(ctx, one, 1, warmup=0)
    ctx.prof_enable(False)
    ms_a, n_a = ctx.prof_get("train_actor"); steps_l = EP * (E * T // B)
    fa = bench.flop_step(bench.WORKLOADS["c5"]["actor"]); ach = fa * steps_l / (ms_a / max(1, n_a) * 1e-3) / 1e12
    out = {"workload": bench.WORKLOADS["c5"]["name"] + ", batch 128, 80 + 80 epochs (%d Adam steps/iter)" % (2 * steps_l), "n_gpus": 1,
           "env_steps_per_s": E * T / t, "grad_steps_per_s": 2 * steps_l / t, "ms_per_iteration": 1e3 * t,
           "phase_ms": {k: ctx.prof_get(k)[0] for k in ("rollout", "values", "gae", "whiten", "train_actor")},
           "roofline": {"kernel": "batch_train! actor, k_train_fs2<17,6,GAUSSIAN,tanh> (4 workgroups x (4 compute + 4 helper waves))", "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TFLOPS, "frac_of_occupied_cus": ach / (PEAK_F32_MFMA_TFLOPS * 4.0 / 256.0), "occupied_cus": 4, "us_per_grad_step": ms_a / max(1, n_a) * 1e3 / steps_l, "allreduce_payload_bytes": 4 * pi.A.n_params,
                        "traffic": 2.832e9, "traffic_note": "RECORDED constant (profiles/r05_pmc_traffic_c5.txt: 2 x FETCH_SIZE + WRITE_SIZE per 163 840-step actor launch = 17.3 KB per step against 13.3 KB of algorithmic minibatch bytes; `python bench.py --workload c5` measures it in the run). Round 3 read 60.7 KB per step: a gathered 68-byte observation row, 24-byte action row and 4-byte scalars cost a sector each; the learners now fetch ONE 128-byte packed line per sample (s | a | logprob | advantage | return, written once per batch_train! call)",
                        "note": "one serially dependent learner on four CUs of one XCD (feature-split wave pairs + helper waves); %.2f MFLOP per step" % (fa / 1e6)}}
    if cpu:
        out["cpu_baseline"] = bench.cpu_baseline("c5")
    return out


def run(crux, ctx, cpu=True):
    out = {}
    for name, fn in (("c5_shard", c5),):
        try:
            t0 = time.perf_counter(); out[name] = fn(crux, ctx, cpu); out[name]["bench_seconds"] = time.perf_counter() - t0
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": repr(e)}
    try:      # the caller-stepped environment seam: us per crux_policy_explore call at E = 32 / 1 / 128, env-steps/s end to end, the device environment beside it
        import bench_hostenv
        t0 = time.perf_counter(); out["host_env"] = bench_hostenv.run(crux, ctx); out["host_env"]["bench_seconds"] = time.perf_counter() - t0
    except Exception as e:          # noqa: BLE001
        out["host_env"] = {"error": repr(e)}
    try:
        import bench_offpolicy
        out.update(bench_offpolicy.run(crux, ctx, cpu))
    except Exception as e:          # noqa: BLE001
        out["offpolicy"] = {"error": repr(e)}
    return out
