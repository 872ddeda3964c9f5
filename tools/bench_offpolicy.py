#!/usr/bin/env python
"""Secondary measurements for the off-policy configs of BASELINE.json (not the bench.py headline):
  C3  DQN + PER, 8-256-256-4 critic, buffer 1 M, B = 128: grad-steps/s and PER-samples/s
  C4  SAC, actor 3-256-256-1 + twin Q 4-256-256-1, B = 256: epochs/s (target + temperature + critic + actor + polyak)
Prints one JSON object. Run on the GPU box: python tools/bench_offpolicy.py [--steps K]."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crux_jl_amd as crux
from crux_jl_amd import _lib as L


def chain(dims, acts):
    return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])


def timed(ctx, fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.sync(); return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=200); a = ap.parse_args()
    ctx, rng, out = crux.default_context(), np.random.default_rng(0), {}
    # ---- C3
    N, B = 1_000_000, 128
    S, A = crux.ContinuousSpace(8), crux.DiscreteSpace(4)
    buf = crux.ExperienceBuffer(S, A, N, prioritized=True); D = crux.buffer_like(buf, capacity=B)
    chunk = 100_000
    for _ in range(N // chunk):
        a_id = rng.integers(0, 4, chunk)
        buf.push_({"s": rng.normal(0, 1, (8, chunk)).astype(np.float32), "a": np.eye(4, dtype=bool)[:, a_id], "sp": rng.normal(0, 1, (8, chunk)).astype(np.float32),
                   "r": rng.normal(0, 1, (1, chunk)).astype(np.float32), "done": rng.random((1, chunk)) < 0.01, "episode_end": np.zeros((1, chunk), bool)})
    buf.update_priorities_(np.arange(1, N + 1), (np.abs(rng.normal(0, 1, N)) + 1e-3).astype(np.float32))
    q = crux.DiscreteNetwork(chain([8, 256, 256, 4], ["relu", "relu", "identity"]), [1, 2, 3, 4], seed=1)
    qm = crux.clone_policy(q); q.attach_optimizer(crux.Adam(np.float32(1e-3)))
    dy, de = ctx.alloc(4 * B), ctx.alloc(4 * B); raw = np.zeros(L.INFO_N, np.float32); k = [0]
    def per_only():
        k[0] += 1; crux.prioritized_sample_(D, buf, i=k[0])
    def dqn_epoch():
        k[0] += 1; crux.prioritized_sample_(D, buf, i=k[0])
        ctx.check(ctx.lib.crux_dqn_target(qm.h, D.h, 0.99, dy))
        ctx.check(ctx.lib.crux_td_step_with_error(q.h, D.h, dy, 1, de, raw.ctypes.data_as(L.vp)))      # td_error + train! share the forward pass
        ctx.check(ctx.lib.crux_per_update_device(buf.h, ctx.lib.crux_buffer_indices_ptr(D.h), de, B))
    def td_only():
        ctx.check(ctx.lib.crux_td_step(q.h, D.h, dy, 1, raw.ctypes.data_as(L.vp)))
    t_per, t_epoch, t_td = timed(ctx, per_only, a.steps), timed(ctx, dqn_epoch, a.steps), timed(ctx, td_only, a.steps)
    out["C3"] = {"workload": "DQN+PER 8-256-256-4, buffer 1M, B=128", "per_sample_ms": 1e3 * t_per, "per_samples_per_s": B / t_per,
                 "epoch_ms (sample+target+td_error+update_priorities+train!)": 1e3 * t_epoch, "grad_steps_per_s": 1.0 / t_epoch, "train_step_only_ms": 1e3 * t_td,
                 "algorithmic_MFLOP_per_step": 87.8, "achieved_TFLOPs_train_step_only": 6 * 128 * 68608 * 2 / 2 / t_td / 1e12}
    # ---- C4
    B = 256
    S, A = crux.ContinuousSpace(3), crux.ContinuousSpace(1)
    buf = crux.ExperienceBuffer(S, A, 100_000); D = crux.buffer_like(buf, capacity=B)
    n = 100_000
    buf.push_({"s": rng.normal(0, 1, (3, n)).astype(np.float32), "a": rng.uniform(-2, 2, (1, n)).astype(np.float32), "sp": rng.normal(0, 1, (3, n)).astype(np.float32),
               "r": rng.normal(-1, 1, (1, n)).astype(np.float32), "done": np.zeros((1, n), bool), "episode_end": np.zeros((1, n), bool)})
    acts = ["relu", "relu", "identity"]
    pi = crux.ActorCritic(crux.GaussianPolicy(chain([3, 256, 256, 1], acts), np.zeros(1, np.float32), seed=2),
                          crux.DoubleNetwork(crux.ContinuousNetwork(chain([4, 256, 256, 1], acts), seed=3), crux.ContinuousNetwork(chain([4, 256, 256, 1], acts), seed=4)))
    opt = {"batch_size": B, "optimizer": crux.Adam(np.float32(3e-4))}
    solver = crux.SAC(pi, S, N=10**9, dN=1, c_opt=dict(opt), a_opt=dict(opt), SAC_alpha_opt=dict(opt), buffer=buf)
    solver.batch = D
    def sac_epoch():
        solver.i += 1; crux.value_training(solver, D, np.float32(0.99))
    t_sac = timed(ctx, sac_epoch, a.steps)
    out["C4"] = {"workload": "SAC actor 3-256-256-1 + twin Q 4-256-256-1, B=256", "epoch_ms": 1e3 * t_sac, "epochs_per_s": 1.0 / t_sac,
                 "algorithmic_GFLOP_per_epoch": 0.58, "achieved_TFLOPs": 0.58e9 / t_sac / 1e12}
    # ---- C5 learner only: PPO 17-64-64-6 Gaussian actor + 17-64-64-1 critic on 262144 synthetic rows (128 envs x 2048), batch 128, 4 epochs
    n = 128 * 2048
    S, A = crux.ContinuousSpace(17), crux.ContinuousSpace(6)
    buf = crux.ExperienceBuffer(S, A, n, ["return", "logprob", "advantage"])
    chunk = 65536
    for _ in range(n // chunk):
        buf.push_({"s": rng.normal(0, 1, (17, chunk)).astype(np.float32), "a": rng.uniform(-1, 1, (6, chunk)).astype(np.float32), "sp": rng.normal(0, 1, (17, chunk)).astype(np.float32),
                   "r": rng.normal(0, 1, (1, chunk)).astype(np.float32), "done": np.zeros((1, chunk), bool), "episode_end": np.zeros((1, chunk), bool),
                   "return": rng.normal(0, 1, (1, chunk)).astype(np.float32), "logprob": rng.normal(-6, 0.3, (1, chunk)).astype(np.float32), "advantage": rng.normal(0, 1, (1, chunk)).astype(np.float32)})
    acts5 = ["tanh", "tanh", "identity"]
    pi5 = crux.ActorCritic(crux.GaussianPolicy(chain([17, 64, 64, 6], acts5), np.full(6, -0.5, np.float32), seed=5), crux.ContinuousNetwork(chain([17, 64, 64, 1], acts5), seed=6))
    class _S:
        pass
    sv = _S(); sv.agent = crux.PolicyParams(pi5); sv.P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.0}
    sv.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=4, name="actor_", shuffle_seed=1)
    sv.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=4, name="critic_", shuffle_seed=2)
    def c5_train():
        crux.policy_gradient_training(sv, buf)
    t5 = timed(ctx, c5_train, 3, warmup=1)
    steps5 = 2 * 4 * (n // 128)
    out["C5_learner"] = {"workload": "PPO 17-64-64-6 tanh Gaussian actor || 17-64-64-1 critic, 262144 rows, B=128, 4+4 epochs (synthetic rows)", "ms_per_call": 1e3 * t5,
                         "grad_steps_per_s": steps5 / t5, "us_per_actor_step": 1e6 * t5 / (steps5 / 2)}
    # ---- C5 end to end: PPO on the synthetic 17-obs / 6-action environment, 128 envs x 2048 steps, tanh 17-64-64-6 Gaussian actor + critic, 10 + 10 epochs
    mdp = crux.SynthMDP(17, 6, n_envs=128, seed=3, discount=0.99)
    pi6 = crux.ActorCritic(crux.GaussianPolicy(chain([17, 64, 64, 6], acts5), np.full(6, -0.5, np.float32), seed=7), crux.ContinuousNetwork(chain([17, 64, 64, 1], acts5), seed=8))
    cols = ["return", "logprob", "advantage"]
    buf6 = crux.ExperienceBuffer(S, A, 128 * 2048, cols)
    smp6 = crux.Sampler(mdp, pi6, max_steps=1000, required_columns=cols, lam=0.95)
    sv6 = _S(); sv6.agent = crux.PolicyParams(pi6); sv6.P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.0}
    sv6.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=10, name="actor_", shuffle_seed=11)
    sv6.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=10, name="critic_", shuffle_seed=12)
    it = [0]
    def c5_iter():
        it[0] += 1
        crux.steps_(smp6, buf6, Nsteps=128 * 2048, explore=True, i=it[0] * 128 * 2048, reset=True); crux.whiten_(buf6, "advantage")
        crux.policy_gradient_training(sv6, buf6)
    ctx.prof_reset(); ctx.prof_enable(True)
    t6 = timed(ctx, c5_iter, 2, warmup=1)
    ctx.prof_enable(False)
    out["C5_end_to_end"] = {"workload": "PPO synthetic 17 obs / 6 act, 128 envs x 2048 steps, 10 + 10 epochs of 2048 minibatches", "ms_per_iteration": 1e3 * t6,
                            "env_steps_per_s": 128 * 2048 / t6, "grad_steps_per_s": 2 * 10 * 2048 / t6, "rollout_ms": ctx.prof_get("rollout")[0] / 3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
