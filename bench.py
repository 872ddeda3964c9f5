#!/usr/bin/env python3
"""bench.py -- PPO actor-learner hot path on MI355X (BASELINE.json configs[1]).

One "step" = one full PPO iteration on synthetic CartPole shards: 32 envs x 2048-step rollout (65 536 transitions,
policy forward + env dynamics + SoA column writes in one kernel), GAE/returns, advantage whitening, then
batch_train! for the actor (ppo_loss) and the critic (mse): 80 epochs x 512 minibatches of 128 each, i.e. exactly
81 920 sequential Adam steps per iteration (KL early stopping is OFF in the timed configuration so no work is skipped;
the early-stopping variant is reported separately under "early_stop").

    python bench.py --gpus N --steps K --warmup W

N>1: launched by torch.distributed.run, one rank per GPU; every rank owns an independent-seed shard of 32 envs (weak
scaling) and the replicas exchange parameters + Adam moments with an all-reduce (RCCL over xGMI) after every epoch.
Rank 0 prints ONE JSON line.
"""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts: keep the two learner streams on separate hardware queues next to torch/RCCL streams
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS, T, BATCH, EPOCHS, MAX_STEPS = 32, 2048, 128, 80, 500
GAMMA, LAM = 0.99, 0.95
ACTOR, CRITIC, ACTS = [4, 64, 64, 2], [4, 64, 64, 1], ["relu", "relu", "identity"]
# algorithmic work per unit (SURVEY.md 8d / DESIGN.md)
FLOP_ACTOR_STEP = 6 * BATCH * (4 * 64 + 64 * 64 + 64 * 2)     # 3.44 MFLOP: fwd 2*B*P + bwd 4*B*P (P = weight count)
FLOP_CRITIC_STEP = 6 * BATCH * (4 * 64 + 64 * 64 + 64 * 1)    # 3.39 MFLOP
PEAK_F32_MFMA_TFLOPS = 157.3                                  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak


def chain(crux, dims):
    return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], ACTS[i]) for i in range(3)])


def build_problem(crux, seed, n_envs=N_ENVS, T_=T):
    S, A = crux.ContinuousSpace(4), crux.DiscreteSpace(2)
    actor = crux.DiscreteNetwork(chain(crux, ACTOR), [1, 2], seed=1, stream=0)
    critic = crux.ContinuousNetwork(chain(crux, CRITIC), seed=1, stream=1)
    pi = crux.ActorCritic(actor, critic)
    extras = ["return", "logprob", "advantage"]
    buf = crux.ExperienceBuffer(S, A, n_envs * T_, extras)
    mdp = crux.CartPoleMDP(n_envs=n_envs, seed=seed, discount=GAMMA)
    sampler = crux.Sampler(mdp, pi, max_steps=MAX_STEPS, required_columns=extras, lam=LAM)
    return pi, buf, sampler


def ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, P, it, sync=None):
    n = len(buf) if len(buf) else buf.capacity
    info = crux.steps_(sampler, buf, Nsteps=buf.capacity, explore=True, i=it * buf.capacity, reset=True)
    crux.whiten_(buf, "advantage")
    if sync is None:
        class _S:       # the fields policy_gradient_training reads from an OnPolicySolver
            pass
        sv = _S(); sv.agent = crux.PolicyParams(pi); sv.a_opt, sv.c_opt, sv.P = a_opt, c_opt, P
        ti = crux.policy_gradient_training(sv, buf)            # on_policy.jl:56-78 (actor then critic; overlapped when exact)
        return ti["actor_batches_trained"] + ti["critic_batches_trained"], info
    # multi-GPU: one persistent launch per learner per `sync_every` epochs (actor || critic concurrently, as above), then the replicas'
    # parameters and Adam moments are averaged with ONE all-reduce ("periodic" exchange of north_star; every epoch by default)
    class _S:
        pass
    sv = _S(); sv.agent = crux.PolicyParams(pi); sv.a_opt, sv.c_opt, sv.P = a_opt, c_opt, P
    if sync == "native":   # the library's own RCCL path: chunks of SYNC_EVERY epochs + grouped all-reduce, all enqueued without a host sync
        ti = crux.policy_gradient_training_synced(sv, buf, SYNC_EVERY)
        return ti["actor_batches_trained"] + ti["critic_batches_trained"], info
    nb, e_total, k = 0, a_opt.epochs, SYNC_EVERY
    try:
        done = 0
        while done < e_total:
            a_opt.epochs = c_opt.epochs = min(k, e_total - done)
            ti = crux.policy_gradient_training(sv, buf)
            nb += ti["actor_batches_trained"] + ti["critic_batches_trained"]
            done += a_opt.epochs
            sync((pi.A, pi.C))
    finally:
        a_opt.epochs = c_opt.epochs = e_total
    return nb, info


SYNC_EVERY = int(os.environ.get("CRUX_SYNC_EVERY", "1"))   # epochs between parameter exchanges in the multi-GPU path


def cpu_baseline():
    """Oracle (CPU port of the reference algorithm, single thread) on a bounded sample of the same workload (~10 s of CPU work):
    one full 32-env x 2048-step rollout + GAE, and 2 x 8192 of the 2 x 40960 minibatch Adam steps (B=128); extrapolated to one full iteration."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from crux_jl_amd import _lib as L
    import parity
    Ts = T
    oa = O.OMlp(ACTOR, ACTS).init_glorot(1, 0); oc = O.OMlp(CRITIC, ACTS).init_glorot(1, 1)
    oa.adam_init(float(np.float32(3e-4))); oc.adam_init(float(np.float32(3e-4)))
    extras = ["return", "logprob", "advantage"]
    ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, N_ENVS * Ts, extras)
    oe = O.OEnv("cartpole", N_ENVS, MAX_STEPS, GAMMA, 0)
    t0 = time.perf_counter()
    oe.rollout(oa, parity.rollout_cfg(), ob, Ts)
    O.chk(O.lib().orc_fill_gae(ob.h, oc.h, LAM, GAMMA)); O.chk(O.lib().orc_fill_returns(ob.h, GAMMA)); O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"]))
    t_roll = time.perf_counter() - t0
    n_mb = 8192
    info = np.zeros(L.INFO_N, np.float32)
    t0 = time.perf_counter()
    for loss, head, net in (("ppo", "categorical", oa), ("value_mse", "deterministic", oc)):
        cfg = parity.train_cfg(loss, head, BATCH, 16, -1.0, 7, 0, max_batches=n_mb)
        O.chk(O.lib().orc_batch_train(net.h, ob.h, C.byref(cfg), None, O.vpz(info), None))
    t_train = time.perf_counter() - t0
    per_env_step = t_roll / (N_ENVS * Ts)
    per_grad_step = t_train / (2 * n_mb)
    t_iter = per_env_step * N_ENVS * T + per_grad_step * 2 * EPOCHS * (N_ENVS * T // BATCH)
    model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {"value": N_ENVS * T / t_iter, "unit": "env-steps/s", "cores": 1, "kind": "port", "host_cpu": model, "host_cores": os.cpu_count(),
            "sample": "oracle/ (C restatement of Crux.jl, 1 thread): 32x%d-step rollout+GAE (%.2fs) and %d Adam steps at B=128 (%.2fs), extrapolated to one 65536-transition / 81920-step iteration"
                      % (Ts, t_roll, 2 * n_mb, t_train),
            "grad_steps_per_s": 1.0 / per_grad_step, "rollout_env_steps_per_s": 1.0 / per_env_step}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sync", action="store_true", help="exercise the multi-GPU parameter exchange even at world size 1 (testing)")
    ap.add_argument("--replicas", type=int, default=64, help="also time S independent PPO learners (multi-seed) trained by two batched launches per iteration: the chip-level utilisation line (0 = skip)")
    ap.add_argument("--replicas-wide", type=int, default=128, help="second multi-seed line with one CU per learner (population > 64): the highest chip utilisation (0 = skip)")
    ap.add_argument("--early-stop", action="store_true", help="also time the KL-early-stopping variant (target_kl=0.012)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner on stdout) are pointed at stderr instead
    sys.stdout.flush(); json_fd = os.dup(1); os.dup2(2, 1)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    dist = torch = None
    if world > 1 or args.force_sync:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import crux_jl_amd as crux
    from crux_jl_amd import dist as cdist
    ctx = crux.Context(local)
    crux.set_default_context(ctx)
    pi, buf, sampler = build_problem(crux, cdist.shard_seed(0, rank))
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=BATCH, epochs=EPOCHS, target_kl=None, name="actor_", shuffle_seed=100 + rank)
    c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=BATCH, epochs=EPOCHS, name="critic_", shuffle_seed=200 + rank)

    sync = None
    if world > 1 or args.force_sync:
        wrapped = {}

        def sync(nets):   # average parameters and Adam moments of all networks across ranks: one RCCL all-reduce over xGMI
            key = tuple(id(n) for n in nets)
            if key not in wrapped:
                lib, ts = ctx.lib, []

                def wrap(ptr, n):
                    class _I:
                        __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}
                    return torch.as_tensor(_I(), device="cuda:%d" % local)
                for net in nets:
                    pm, pv = C.c_void_p(), C.c_void_p()
                    ctx.check(lib.crux_adam_state_ptrs(net.h, C.byref(pm), C.byref(pv)))
                    ts += [wrap(lib.crux_mlp_params_ptr(net.h), net.n_params), wrap(pm.value, net.n_params), wrap(pv.value, net.n_params)]
                wrapped[key] = (ts, [t.numel() for t in ts])
            ts, sizes = wrapped[key]
            ctx.sync()                                   # the library's streams produced the values
            flat = torch.cat(ts)                         # 6 x ~18 KB -> one 110 KB message (latency-bound either way on xGMI)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / world)
            torch._foreach_copy_(ts, list(flat.split(sizes)))   # tensors alias library memory
            torch.cuda.current_stream().synchronize()

        # preferred: the library's native communicator (RCCL dlopen'ed by libcruxhip, id shipped through torch.distributed);
        # the torch.distributed exchange above stays as the fallback when RCCL cannot be initialised from the library
        comm_kind = "torch.distributed all_reduce (host-synchronised per exchange)"; torch_sync = sync
        if os.environ.get("CRUX_NATIVE_COMM", "1") != "0":
            uid_np = np.zeros(129, np.uint8)
            if rank == 0:
                try:
                    uid_np[:128] = ctx.comm_unique_id(); uid_np[128] = 1
                except Exception as e:
                    print("bench.py: native RCCL unavailable (%r)" % (e,), file=sys.stderr)
            uid = torch.from_numpy(uid_np).cuda(); dist.broadcast(uid, 0); uid_np = uid.cpu().numpy()     # every rank takes part, whatever happened on rank 0
            if uid_np[128]:
                try:
                    ctx.comm_init(rank, world, uid_np[:128].copy())
                    sync = "native"; comm_kind = "libcruxhip RCCL communicator (stream-ordered grouped all-reduce)"
                except Exception as e:
                    print("bench.py: native RCCL communicator failed on rank %d (%r)" % (rank, e), file=sys.stderr)
            ok = torch.tensor([1 if sync == "native" else 0], device="cuda"); dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # all ranks or none
            if int(ok.item()) == 0 and sync == "native":
                ctx.comm_destroy()
            if int(ok.item()) == 0:
                sync = torch_sync; comm_kind = "torch.distributed all_reduce (host-synchronised per exchange)"

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier(); torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, P, it, sync); it += 1
    barrier()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    grad_steps = 0
    for _ in range(args.steps):
        nb, info = ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, P, it, sync); it += 1
        grad_steps += nb
    barrier()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dt = float(tmax.item())
        gs = torch.tensor([grad_steps], dtype=torch.float64, device="cuda"); dist.all_reduce(gs); grad_steps = int(gs.item())

    prof = {k: ctx.prof_get(k) for k in ("rollout", "values", "gae", "whiten", "train_actor", "train_critic")}
    early = None
    if args.early_stop and world == 1:
        a_es = crux.TrainingParams(loss=crux.ppo_loss, batch_size=BATCH, epochs=EPOCHS, target_kl=0.012, name="actor_", shuffle_seed=300)
        ctx.sync(); t1 = time.perf_counter(); nbs = 0
        for _ in range(args.steps):
            nb, _i = ppo_iteration(crux, pi, buf, sampler, a_es, c_opt, P, it); it += 1; nbs += nb
        ctx.sync(); d1 = time.perf_counter() - t1
        early = {"env_steps_per_s": args.steps * N_ENVS * T / d1, "grad_steps_per_s": nbs / d1, "grad_steps_per_iter": nbs / args.steps}

    multi = None
    def multi_seed_line(Sn):
        try:
            probs = [build_problem(crux, cdist.shard_seed(1000, r)) for r in range(Sn)]
            am = crux.TrainingParams(loss=crux.ppo_loss, batch_size=BATCH, epochs=EPOCHS, target_kl=None, name="actor_", shuffle_seed=5000)
            cm = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=BATCH, epochs=EPOCHS, name="critic_", shuffle_seed=6000)

            def multi_iteration(k):
                crux.steps_multi_([q[2] for q in probs], [q[1] for q in probs], Nsteps=probs[0][1].capacity, explore=True, i=k * probs[0][1].capacity, reset=True)   # one rollout launch for all S problems
                crux.whiten_multi_([q[1] for q in probs], "advantage")
                infos = crux.policy_gradient_training_multi([q[0] for q in probs], am, cm, P, [q[1] for q in probs])
                return sum(i["actor_batches_trained"] + i["critic_batches_trained"] for i in infos)
            multi_iteration(0); ctx.sync(); ctx.prof_reset(); ctx.prof_enable(True); t1 = time.perf_counter(); gsm = 0
            n_it = max(1, min(args.steps, 2))
            for k in range(n_it):
                gsm += multi_iteration(1 + k)
            ctx.sync(); d1 = time.perf_counter() - t1; ctx.prof_enable(False)
            ms_a, n_a = ctx.prof_get("train_actor")
            flops_a = Sn * EPOCHS * (N_ENVS * T // BATCH) * FLOP_ACTOR_STEP            # per batched actor launch
            cus = 2 if Sn <= 64 else 1
            return {"replicas": Sn, "cus_per_learner": cus, "env_steps_per_s": n_it * Sn * N_ENVS * T / d1, "grad_steps_per_s": gsm / d1, "ms_per_iteration": 1e3 * d1 / n_it,
                    "phase_ms_per_iteration": {k: ctx.prof_get(k)[0] / n_it for k in ("rollout", "values", "gae", "whiten", "train_actor")},
                    "batched_actor_launch_ms": ms_a / max(1, n_a), "actor_launch_TFLOPs": flops_a / (ms_a / max(1, n_a) * 1e-3) / 1e12,
                    "actor_launch_frac_of_f32_mfma_peak": flops_a / (ms_a / max(1, n_a) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "actor_plus_critic_frac_of_f32_mfma_peak": (flops_a + Sn * EPOCHS * (N_ENVS * T // BATCH) * FLOP_CRITIC_STEP) / (ms_a / max(1, n_a) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "note": "S independent PPO problems (own envs, buffers, seeds); all S actors in one launch, all S critics in a concurrent one: %d S CUs busy" % (2 * cus)}
        except Exception as e:      # supplementary line: never let it take the headline measurement down
            return {"replicas": Sn, "error": repr(e)}

    multi = multi_wide = None
    if args.replicas > 1 and world == 1:
        multi = multi_seed_line(args.replicas)
        if args.replicas_wide > 1:
            multi_wide = multi_seed_line(args.replicas_wide)

    if rank == 0:
        env_steps = args.steps * N_ENVS * T * world
        ms_actor, n_actor = prof["train_actor"]
        launches_per_iter = max(1, n_actor // max(1, args.steps))
        steps_per_launch = EPOCHS * (N_ENVS * T // BATCH) / launches_per_iter
        avg_launch_s = (ms_actor / max(1, n_actor)) * 1e-3
        achieved = FLOP_ACTOR_STEP * steps_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        out = {
            "metric": "env-steps/sec + grad-steps/sec, PPO 2048x32 rollout",
            "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PPO CartPole-v1 (device dynamics), DiscreteNetwork 4-64-64-2 + critic 4-64-64-1, 32 envs x 2048-step rollout per GPU, "
                                   "batch 128, 80 epochs actor + 80 epochs critic (81920 Adam steps/iter, KL early-stop off), Adam 3e-4",
                       "envs_per_gpu": N_ENVS, "rollout_T": T, "batch_size": BATCH, "epochs": EPOCHS,
                       "parallelism": ("env-shards x%d, parameter+Adam-moment all-reduce every %d epoch(s), %s" % (world, SYNC_EVERY, comm_kind)) if sync is not None else "single GPU"},
            "grad_steps_per_s": grad_steps / dt,
            "phase_ms_per_iter": {k: v[0] / args.steps for k, v in prof.items()},
            "rollout_env_steps_per_s": (N_ENVS * T * args.steps) / (prof["rollout"][0] * 1e-3) if prof["rollout"][0] > 0 else None,
            "roofline": {"kernel": "batch_train! actor (persistent fwd+ppo_loss+bwd+Adam)", "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": 16.6e6 if world == 1 else None,
                         "traffic_note": "HBM-side bytes per actor launch from rocprofv3 --pmc FETCH_SIZE (13.86 MB) + WRITE_SIZE (2.70 MB), separate passes of this command (tools/pmc_traffic.sh -> profiles/r01_pmc_traffic.txt); a recorded constant, not re-measured in this run. Algorithmic minibatch bytes per launch are 136 MB (40960 x 3328 B): the 3.4 MB buffer is re-read from L2/MALL, and the 1.6 GB/launch of gradient exchange between the two workgroups stays inside one XCD's L2",
                         "avg_launch_ms": avg_launch_s * 1e3, "grad_steps_per_launch": steps_per_launch,
                         "us_per_grad_step": avg_launch_s * 1e6 / steps_per_launch if steps_per_launch else None,
                         "note": "81920 serially dependent 3.4-MFLOP steps: each learner step is split over two CUs of one XCD (gradient exchange through the shared L2), actor and critic run concurrently -> 4 CUs busy; per-CU f32 MFMA peak is 0.614 TFLOP/s"},
        }
        if early:
            out["early_stop"] = early
        if multi is not None:
            out["multi_seed"] = multi
        if multi_wide is not None:
            out["multi_seed_one_cu_per_learner"] = multi_wide
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush(); os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        if sync == "native":
            ctx.comm_destroy()                 # the library's communicator goes first, then torch's
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
