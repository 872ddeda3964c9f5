#!/usr/bin/env python3
"""bench.py -- the PPO actor-learner hot path on MI355X (BASELINE.json configs[1]; `--workload c5` = configs[4]'s per-GPU shard).

One "step" = one full PPO iteration on one GPU's environment shard: E envs x 2048-step rollout (policy forward + env dynamics + SoA column
writes in one kernel), GAE / returns, advantage whitening, then batch_train! for the actor (ppo_loss) and the critic (mse): 80 epochs x
(E x 2048 / 128) minibatches of 128 each -- for C2 (E = 32) exactly 81 920 sequential Adam steps per iteration. KL early stopping is OFF in the
timed configuration so no work is skipped (the early-stopping variant is reported separately under "early_stop").

    python bench.py --gpus N --steps K --warmup W [--workload c2|c5] [--sync grad|params]

N > 1: one rank per GPU (the script re-launches itself under torch.distributed.run when it was started plainly); every rank owns an
independent-seed shard of E envs (weak scaling) and the replicas exchange GRADIENTS: every minibatch step SUM-all-reduces the flattened local
gradient over the N GPUs inside the persistent learner kernel (peer slots over xGMI, include/cruxhip.h "replica group"), Adam runs on the mean
-- N replicas with minibatches of 128 are one learner with minibatches of N x 128, parameters stay bit-identical on all ranks (checked after
the timed region). `--sync params` selects the older periodic form (parameters + Adam moments averaged by RCCL every epoch).
Rank 0 prints ONE JSON line.
"""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts: keep the two learner streams on separate hardware queues next to torch/RCCL streams
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, BATCH, EPOCHS = 2048, 128, 80
GAMMA, LAM = 0.99, 0.95
PEAK_F32_MFMA_TFLOPS = 157.3                                  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# workloads: BASELINE.json configs[1] (the metric's config) and configs[4] (per-GPU shard; SYNTH 17/6 dynamics stand in for HalfCheetah, cruxhip.h)
WORKLOADS = {
    "c2": dict(n_envs=32, obs=4, act=2, discrete=True, actor=[4, 64, 64, 2], critic=[4, 64, 64, 1], acts=["relu", "relu", "identity"], max_steps=500, lambda_e=0.1,
               name="PPO CartPole-v1 (device dynamics), DiscreteNetwork 4-64-64-2 + critic 4-64-64-1, 32 envs x 2048-step rollout per GPU"),
    "c5": dict(n_envs=128, obs=17, act=6, discrete=False, actor=[17, 64, 64, 6], critic=[17, 64, 64, 1], acts=["tanh", "tanh", "identity"], max_steps=1000, lambda_e=0.0,
               name="PPO on the HalfCheetah-shaped synthetic env (17 obs / 6 act), tanh GaussianPolicy 17-64-64-6 + critic 17-64-64-1, 128 envs x 2048-step rollout per GPU"),
}
N_ENVS, MAX_STEPS, ACTOR, CRITIC, ACTS = 32, 500, WORKLOADS["c2"]["actor"], WORKLOADS["c2"]["critic"], WORKLOADS["c2"]["acts"]   # C2 names used by tests/


def flop_step(dims, batch=BATCH):
    """algorithmic flops of one learner step: forward 2 B P + backward 4 B P, P = weight count."""
    return 6 * batch * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))


FLOP_ACTOR_STEP, FLOP_CRITIC_STEP = flop_step(ACTOR), flop_step(CRITIC)     # 3.44 / 3.39 MFLOP


def chain(crux, dims, acts=ACTS):
    return crux.Chain(*[crux.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(acts))])


def build_problem(crux, seed, n_envs=None, T_=T, workload="c2"):
    w = WORKLOADS[workload]; n_envs = n_envs or w["n_envs"]
    S = crux.ContinuousSpace(w["obs"]); A = crux.DiscreteSpace(w["act"]) if w["discrete"] else crux.ContinuousSpace(w["act"])
    if w["discrete"]:
        actor = crux.DiscreteNetwork(chain(crux, w["actor"], w["acts"]), list(range(1, w["act"] + 1)), seed=1, stream=0)
        mdp = crux.CartPoleMDP(n_envs=n_envs, seed=seed, discount=GAMMA)
    else:
        actor = crux.GaussianPolicy(chain(crux, w["actor"], w["acts"]), np.full(w["act"], -0.5, np.float32), seed=1, stream=0)
        mdp = crux.SynthMDP(w["obs"], w["act"], n_envs=n_envs, seed=seed, discount=GAMMA)
    critic = crux.ContinuousNetwork(chain(crux, w["critic"], w["acts"]), seed=1, stream=1)
    pi = crux.ActorCritic(actor, critic)
    extras = ["return", "logprob", "advantage"]
    buf = crux.ExperienceBuffer(S, A, n_envs * T_, extras)
    sampler = crux.Sampler(mdp, pi, max_steps=w["max_steps"], required_columns=extras, lam=LAM)
    return pi, buf, sampler


class _Solver:      # the fields policy_gradient_training reads from an OnPolicySolver
    def __init__(self, crux, pi, a_opt, c_opt, P):
        self.agent, self.a_opt, self.c_opt, self.P = crux.PolicyParams(pi), a_opt, c_opt, P


def ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, P, it, sync=None):
    info = crux.steps_(sampler, buf, Nsteps=buf.capacity, explore=True, i=it * buf.capacity, reset=True)
    crux.whiten_(buf, "advantage")
    sv = _Solver(crux, pi, a_opt, c_opt, P)
    if sync is None or sync == "grad":
        # single GPU, or a replica group with peer slots attached to the context: the SAME call -- with a group every minibatch step of the two
        # persistent kernels all-reduces its gradient over the GPUs (on_policy.jl:56-78 / training.jl:13-25 with the exchange between :18 and :21)
        ti = crux.policy_gradient_training(sv, buf)
        return ti["actor_batches_trained"] + ti["critic_batches_trained"], info
    if sync == "native":   # periodic parameter averaging by the library's RCCL communicator: chunks of SYNC_EVERY epochs + grouped all-reduce, enqueued without a host sync
        ti = crux.policy_gradient_training_synced(sv, buf, SYNC_EVERY)
        return ti["actor_batches_trained"] + ti["critic_batches_trained"], info
    nb, e_total, k = 0, a_opt.epochs, SYNC_EVERY       # torch.distributed fallback of the periodic form (host-synchronised per exchange)
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


SYNC_EVERY = int(os.environ.get("CRUX_SYNC_EVERY", "1"))   # epochs between parameter exchanges in the `--sync params` path


def host_cpu():
    model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(workload="c2"):
    """Oracle (CPU restatement of the reference algorithm, the faithful analogue of the single-threaded reference) on a bounded sample of the same
    workload (~10-20 s of CPU work): a rollout + GAE of all E envs over T_s steps and n_mb Adam steps per learner at B = 128; scaled to one full
    iteration. When oracle/libcruxoracle_omp.so exists (the OpenMP env-parallel / batch-parallel variant on all host cores), it is timed too."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from crux_jl_amd import _lib as L
    import parity
    w = WORKLOADS[workload]; E = w["n_envs"]
    Ts = T if workload == "c2" else 256
    n_mb = 8192 if workload == "c2" else 2048
    kind = L.ACTION_DISCRETE if w["discrete"] else L.ACTION_CONTINUOUS
    head = "categorical" if w["discrete"] else "gaussian"
    extras = ["return", "logprob", "advantage"]

    def run(lib_sel):
        O.select(lib_sel)
        oa = O.OMlp(w["actor"], w["acts"], 0 if w["discrete"] else w["act"]).init_glorot(1, 0, -0.5); oc = O.OMlp(w["critic"], w["acts"]).init_glorot(1, 1)
        oa.adam_init(float(np.float32(3e-4))); oc.adam_init(float(np.float32(3e-4)))
        ob = O.OBuffer(w["obs"], w["act"], kind, E * Ts, extras)
        oe = O.OEnv("cartpole" if workload == "c2" else "synth", E, w["max_steps"], GAMMA, 0, so=w["obs"] if workload != "c2" else 0, sa=w["act"] if workload != "c2" else 0)
        t0 = time.perf_counter()
        oe.rollout(oa, parity.rollout_cfg(head=head), ob, Ts)
        O.chk(O.lib().orc_fill_gae(ob.h, oc.h, LAM, GAMMA)); O.chk(O.lib().orc_fill_returns(ob.h, GAMMA)); O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"]))
        t_roll = time.perf_counter() - t0
        info = np.zeros(L.INFO_N, np.float32)
        t0 = time.perf_counter()
        for loss, hd, net in (("ppo", head, oa), ("value_mse", "deterministic", oc)):
            cfg = parity.train_cfg(loss, hd, BATCH, 64, -1.0, 7, 0, max_batches=n_mb, le=w["lambda_e"])
            O.chk(O.lib().orc_batch_train(net.h, ob.h, C.byref(cfg), None, O.vpz(info), None))
        t_train = time.perf_counter() - t0
        return t_roll / (E * Ts), t_train / (2 * n_mb), t_roll, t_train
    per_env_step, per_grad_step, t_roll, t_train = run("scalar")
    steps_iter = 2 * EPOCHS * (E * T // BATCH)
    t_iter = per_env_step * E * T + per_grad_step * steps_iter
    model, ncpu = host_cpu()
    out = {"value": E * T / t_iter, "unit": "env-steps/s", "cores": 1, "kind": "port", "host_cpu": model, "host_cores": ncpu,
           "sample": "oracle/ (C restatement of Crux.jl, 1 thread): %dx%d-step rollout+GAE (%.2fs) and %d Adam steps at B=128 (%.2fs), scaled to one %d-transition / %d-step iteration"
                     % (E, Ts, t_roll, 2 * n_mb, t_train, E * T, steps_iter),
           "grad_steps_per_s": 1.0 / per_grad_step, "rollout_env_steps_per_s": 1.0 / per_env_step}
    if O.have("omp"):
        try:
            pe, pg, tr, tt = run("omp")
            nthr = int(O.lib().orc_omp_threads())
            used = max(min(nthr, E), min(nthr, BATCH // 8))
            out["openmp_all_cores"] = {"value": E * T / (pe * E * T + pg * steps_iter), "unit": "env-steps/s", "cores": used, "host_threads_available": nthr,
                                       "grad_steps_per_s": 1.0 / pg, "rollout_env_steps_per_s": 1.0 / pe,
                                       "sample": "oracle built with -fopenmp: one thread per environment in the rollout (%d), %d threads over the samples of a minibatch inside a learner step "
                                                 "(the steps themselves are serially dependent; more threads than B/8 only add reduction cost): same sample, %.2fs + %.2fs"
                                                 % (min(nthr, E), min(nthr, BATCH // 8), tr, tt)}
        except Exception as e:      # noqa: BLE001  (supplementary)
            out["openmp_all_cores"] = {"error": repr(e)}
        O.select("scalar")
    try:      # what a BLAS-backed host (Flux -> OpenBLAS sgemm, the reference's arithmetic) does with the same matrix products: a lower bound on its step time
        import bench_offpolicy
        bl = bench_offpolicy.blas_leg([(list(w["actor"]), 1, 1), (list(w["critic"]), 1, 1)], BATCH, None)      # one actor + one critic step: forward + pullback each
        pair = 2.0
        out["blas_gemm_leg"] = {"gemm_s_per_step_one_thread": (bl["one_thread_s"] / pair) if bl["one_thread_s"] else None, "gemm_s_per_step_all_threads": bl["all_threads_s"] / pair,
                                "grad_steps_per_s_upper_bound_one_thread": (pair / bl["one_thread_s"]) if bl["one_thread_s"] else None,
                                "grad_steps_per_s_upper_bound_all_threads": pair / bl["all_threads_s"],
                                "env_steps_per_s_upper_bound_one_thread": (E * T / (per_env_step * E * T + steps_iter * bl["one_thread_s"] / pair)) if bl["one_thread_s"] else None,
                                "note": "the Dense products of one actor + one critic minibatch step (forward + pullback, B = 128) through numpy / OpenBLAS sgemm; elementwise work, the loss head, Zygote's tape and Adam excluded: "
                                        "an UPPER bound on a BLAS-backed host's Adam-step rate, next to the scalar port so that the port's number is not read as the speed of the reference's arithmetic "
                                        "(the env-steps bound takes the port's rollout time and this step time)"}
    except Exception as e:      # noqa: BLE001
        out["blas_gemm_leg"] = {"error": repr(e)}
    ref = julia_reference_probe()
    if ref:
        out["julia_reference"] = ref
    return out


def julia_reference_probe():
    """when a `julia` with Crux installed exists on the box, time the reference itself (bench/crux_ref.jl); otherwise say why not."""
    import shutil
    jl = shutil.which("julia")
    if not jl:
        return {"available": False, "why": "no `julia` binary on this box (the reference is pure Julia; it cannot be timed here)"}
    script = os.path.join(ROOT, "julia", "crux_ref_bench.jl")
    try:
        r = subprocess.run([jl, "--project=" + os.path.join(ROOT, "julia"), script], capture_output=True, text=True, timeout=900)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                d = json.loads(ln); d["available"] = True; return d
        return {"available": False, "why": "julia ran but printed no result (Crux.jl not installed?): " + (r.stderr or r.stdout)[-300:]}
    except Exception as e:      # noqa: BLE001
        return {"available": False, "why": repr(e)}


TRAFFIC_RECORDED = {"c2": 39.97e6, "c5": 2.832e9}      # bytes per actor launch: profiles/r05_pmc_traffic.txt (2 x 16 828 KB FETCH_SIZE + 5 374 KB WRITE_SIZE), r05_pmc_traffic_c5.txt (2 x 1 372 560 KB + 20 809 KB: 17.3 KB per step with the packed learner rows)


def measure_traffic(workload):
    """HBM-side bytes of ONE actor launch, measured now: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes of a one-iteration child run of this script
    (MI355X_MICROARCH.md: no trace domains next to --pmc). FETCH_SIZE tallies 128-byte requests at 64 B (calibrated on a 2 GiB stream, profiles/r02_fetch_calibration.txt),
    both counters are reported in KB: traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024."""
    import csv, glob, shutil, tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="crux_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0",
               "--no-cpu-baseline", "--replicas", "0", "--replicas-wide", "0", "--no-extra", "--no-measure-traffic", "--no-early-stop", "--workload", workload]
        try:
            subprocess.run(cmd, capture_output=True, timeout=600, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            per = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    kn = r["Kernel_Name"].split("(")[0]
                    if r["Counter_Name"] != ctr or "k_train" not in kn or "<" not in kn:
                        continue
                    targs = kn[kn.index("<") + 1:].split(",")
                    if len(targs) > 2 and targs[2].strip() != "2":       # template arguments <IN, OUT, KIND, ...>: KIND 2 is the critic (value head)
                        per.append(float(r["Counter_Value"]))
            if not per:
                return None, "no learner kernel in the %s pass" % ctr
            vals[ctr] = sum(per) / len(per)
        except Exception as e:      # noqa: BLE001
            return None, repr(e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "measured in this run: FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB per actor launch" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"])


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` started plainly: become the launcher (one rank per GPU on this node, rendezvous on 127.0.0.1)."""
    port = os.environ.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


PROBE = {}      # the rendezvous probe's waits of this rank (setup_group), reported in the JSON line


def setup_group(args, crux, ctx, rank, world, local):
    """Attach this rank to the replica group. Preference order, each level agreed by ALL ranks (a MIN all-reduce) before it is used:
    (1) peer slots: in-kernel per-minibatch gradient all-reduce over xGMI (exact data parallelism)   [--sync grad]
    (2) the library's RCCL communicator: parameters + Adam moments averaged every SYNC_EVERY epochs   [--sync params, or fallback]
    (3) torch.distributed all-reduce of the same tensors (host-synchronised)                         [last resort]"""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", local) if dist.get_backend() == "nccl" else torch.device("cpu")

    def all_agree(flag):
        t = torch.tensor([1 if flag else 0], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MIN); return int(t.item()) == 1

    if args.sync == "grad":
        ok, handles = True, None
        try:
            mine = torch.from_numpy(ctx.peer_export().copy()).to(dev)
        except Exception as e:      # noqa: BLE001
            print("bench.py: rank %d cannot export a peer region (%r)" % (rank, e), file=sys.stderr); ok = False; mine = torch.zeros(64, dtype=torch.uint8, device=dev)
        gathered = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(gathered, mine)
        if all_agree(ok):
            try:
                handles = np.stack([g.cpu().numpy() for g in gathered])
                if args.inject_attach_failure:      # testing the ladder below: the peers' handles are corrupted, hipIpcOpenMemHandle refuses them on every rank
                    handles = handles ^ np.uint8(0x5A)
                ctx.peer_attach(rank, world, handles)
            except Exception as e:      # noqa: BLE001
                print("bench.py: rank %d cannot attach the peer regions (%r)" % (rank, e), file=sys.stderr); ok = False
            if all_agree(ok):
                dist.barrier()      # every rank has attached before any of them trains (cruxhip.h)
                # the rendezvous probe (crux_peer_probe, collective): do the ranks' kernels answer each other at the speed the in-kernel exchange assumes? A few us per
                # rendezvous on one device, ~10 us over xGMI; milliseconds when the device's hardware queues are time-sliced (too many ranks on ONE GPU: 100 x slower
                # iterations, profiles/r06_same_device_oversubscription.txt) -- such a group is refused here instead of being timed.
                limit_us = 250.0 if args.same_device else 1000.0
                try:
                    ctx.peer_set_budget_ms(60000)
                    PROBE["us"] = ctx.peer_probe(rounds=256, first_bound_ms=5000, round_bound_ms=50); ok = max(PROBE["us"][1], PROBE["us"][3]) <= limit_us
                    if not ok:
                        print("bench.py: rank %d: rendezvous probe too slow: %r us (limit %.0f)" % (rank, PROBE["us"], limit_us), file=sys.stderr)
                except Exception as e:      # noqa: BLE001
                    print("bench.py: rank %d: rendezvous probe failed (%r)" % (rank, e), file=sys.stderr); ok = False
                if all_agree(ok):
                    return "grad", "per-minibatch SUM all-reduce of the gradient inside the persistent learner kernel (peer slots over xGMI, hipIpc-mapped)"
                if args.same_device:
                    raise SystemExit("bench.py --same-device: the %d ranks' learner kernels do not run side by side on this one device (rendezvous probe, us: %r; co-resident kernels "
                                     "meet in a few us): its hardware queues are time-sliced. Refusing to time a group that waits a scheduling quantum per exchange; use fewer ranks." % (world, PROBE.get("us")))
            try:
                ctx.peer_detach()
            except Exception:       # noqa: BLE001
                pass
        print("bench.py: peer-slot gradient exchange unavailable, falling back to periodic parameter averaging", file=sys.stderr)

    wrapped = {}

    def torch_sync(nets):   # average parameters and Adam moments of all networks across ranks
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
        flat = torch.cat(ts)
        if dist.get_backend() != "nccl":
            h = flat.cpu(); dist.all_reduce(h, op=dist.ReduceOp.SUM); flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world)
        torch._foreach_copy_(ts, list(flat.split(sizes)))   # tensors alias library memory
        torch.cuda.current_stream().synchronize()

    if os.environ.get("CRUX_NATIVE_COMM", "1") != "0" and dist.get_backend() == "nccl":
        uid_np = np.zeros(129, np.uint8)
        if rank == 0:
            try:
                uid_np[:128] = ctx.comm_unique_id(); uid_np[128] = 1
            except Exception as e:      # noqa: BLE001
                print("bench.py: native RCCL unavailable (%r)" % (e,), file=sys.stderr)
        uid = torch.from_numpy(uid_np).to(dev); dist.broadcast(uid, 0); uid_np = uid.cpu().numpy()     # every rank takes part, whatever happened on rank 0
        ok = False
        if uid_np[128]:
            try:
                ctx.comm_init(rank, world, uid_np[:128].copy()); ok = True
            except Exception as e:      # noqa: BLE001
                print("bench.py: native RCCL communicator failed on rank %d (%r)" % (rank, e), file=sys.stderr)
        if all_agree(ok):
            return "native", "parameters + Adam moments averaged every %d epoch(s) by the library's RCCL communicator (stream-ordered grouped all-reduce)" % SYNC_EVERY
        if ok:
            ctx.comm_destroy()
    return torch_sync, "parameters + Adam moments averaged every %d epoch(s) by a torch.distributed all_reduce (host-synchronised)" % SYNC_EVERY


def run_selftest(crux, cdist, ctx, dist, torch, rank, world, local, args, P):
    """The first thing a multi-GPU run should do on new hardware: exercise the in-kernel exchange across the REAL devices before anything is timed.
    (1) identical shards: every rank trains on the same rows, so the SUM over ranks / N equals each rank's own gradient up to the rounding of (N - 1) additions -- the
        group must reproduce a GROUP OF ONE on a second context of rank 0 (crux_peer_attach with nranks = 1: the same instantiation of the learner kernel, no peer) BIT FOR BIT
        at N = 2, where g + g and the halving are exact (1e-5 for N > 2), and an un-grouped learner -- another compilation of the step -- to 1e-6; a lost, torn or stale slot
        read shows up as a different sum;
    (2) distinct shards: 64 minibatch steps per learner, the replicas' parameters and Adam state must be bit-identical afterwards;
    (3) flag-wait histograms of (2), one per rank: how long each learner workgroup waited for the slowest peer per exchange;
    (4) the library's own RCCL communicator (crux_comm_init, crux_allreduce_grads -- ncclAllReduce over xGMI): a known vector summed over the ranks.
    Returns the dict that goes into the JSON line under "selftest" (rank 0 gathers)."""
    tdev = torch.device("cuda", local) if dist.get_backend() == "nccl" else torch.device("cpu")
    res = {"world": world}

    def gather_obj(o):
        out = [None] * world; dist.all_gather_object(out, o); return out

    def short_opts(seed, nb):
        return (crux.TrainingParams(loss=crux.ppo_loss, batch_size=BATCH, epochs=1, target_kl=None, name="actor_", shuffle_seed=seed, max_batches=nb),
                crux.TrainingParams(loss=crux.value_mse_loss, batch_size=BATCH, epochs=1, name="critic_", shuffle_seed=seed + 50, max_batches=nb))

    # (1) identical shards, same shuffle on every rank
    pi1, buf1, smp1 = build_problem(crux, cdist.shard_seed(7, 0), workload=args.workload)
    pa, pc = short_opts(700, 16)
    ppo_iteration(crux, pi1, buf1, smp1, pa, pc, P, 0, "grad"); ctx.sync()
    mine = np.concatenate([pi1.A.get_params(), pi1.C.get_params()])
    allp = gather_obj(mine)
    res["identical_shards_replicas_equal"] = bool(all(np.array_equal(allp[0], x) for x in allp))
    if rank == 0:
        ctx2 = crux.Context(local); prev = crux.default_context(); crux.set_default_context(ctx2)      # a second context on the same device
        try:
            refs = {}
            for form in ("group_of_one", "ungrouped"):
                if form == "group_of_one":
                    ctx2.peer_attach(0, 1, ctx2.peer_export()[None, :])      # the replica-group instantiation of the learner kernel with no peer
                pi2, buf2, smp2 = build_problem(crux, cdist.shard_seed(7, 0), workload=args.workload)
                pa2, pc2 = short_opts(700, 16)
                ppo_iteration(crux, pi2, buf2, smp2, pa2, pc2, P, 0, None); ctx2.sync()
                refs[form] = np.concatenate([pi2.A.get_params(), pi2.C.get_params()])
                if form == "group_of_one":
                    ctx2.peer_detach()
            res["identical_shards_max_abs_diff_vs_single_learner"] = float(np.abs(refs["ungrouped"] - mine).max())
            res["identical_shards_max_abs_diff_vs_group_of_one"] = float(np.abs(refs["group_of_one"] - mine).max())
            # (N = 2: g + g and the division by two are exact, so the group of one -- same kernel instantiation -- must be reproduced bit for bit; the un-grouped learner is another
            #  instantiation of k_train_fs2 (compiled without FMA contraction, so it too matches to the bit in practice). A lost, torn or stale slot read is orders larger than either.)
            res["identical_shards_bit_identical_to_group_of_one"] = bool(np.array_equal(refs["group_of_one"], mine))
            res["identical_shards_bit_identical_to_single_learner"] = bool(np.array_equal(refs["ungrouped"], mine))
            res["identical_shards_ok"] = bool((res["identical_shards_bit_identical_to_group_of_one"] if world == 2 else res["identical_shards_max_abs_diff_vs_group_of_one"] < 1e-5)
                                              and res["identical_shards_max_abs_diff_vs_single_learner"] < (1e-6 if world == 2 else 1e-5))
        finally:
            crux.set_default_context(prev)
    # (2) + (3) distinct shards, histograms on
    ctx.peer_wait_hist(reset=True); ctx.peer_hist_enable(True)
    pi3, buf3, smp3 = build_problem(crux, cdist.shard_seed(11, rank), workload=args.workload)
    pa, pc = short_opts(800 + rank, 64)
    ppo_iteration(crux, pi3, buf3, smp3, pa, pc, P, 0, "grad"); ctx.sync()
    ctx.peer_hist_enable(False)
    dg = gather_obj(int(params_digest((pi3.A, pi3.C))))
    res["distinct_shards_replicas_bit_identical"] = bool(len(set(dg)) == 1)
    hist = ctx.peer_wait_hist(reset=True)

    def fmt(h):      # {"<= 0.64 us": n, ...} for the non-empty log2 bins of 10 ns ticks
        return {"<%.2fus" % (10e-3 * 2 ** (b + 1)): int(n) for b, n in enumerate(h) if n}
    # (the short max_batches runs of this selftest train actor and critic back to back on learner stream 0; the timed iterations run them concurrently on streams 0 and 1)
    res["flag_wait_hist_per_rank"] = gather_obj({"stream0_wg0": fmt(hist[0, 0]), "stream0_wg1": fmt(hist[0, 1]), "stream1_wg0": fmt(hist[1, 0]), "stream1_wg1": fmt(hist[1, 1])})
    # (4) RCCL through the library's communicator
    rc = rccl_check(ctx, dist, torch, rank, world, tdev, args, pi3.A)
    res["rccl_allreduce_per_rank"] = gather_obj(rc)
    ok = res["identical_shards_replicas_equal"] and res["distinct_shards_replicas_bit_identical"] and (args.same_device or all(r.get("ok") for r in res["rccl_allreduce_per_rank"]))
    flag = torch.tensor([1 if (ok and res.get("identical_shards_ok", True)) else 0], device=tdev); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    res["passed"] = bool(int(flag.item()) == 1)
    if rank == 0:
        print("bench.py selftest: %s" % json.dumps(res), file=sys.stderr)
    return res


def rccl_check(ctx, dist, torch, rank, world, tdev, args, net):
    """the library's own RCCL communicator (crux_comm_init, crux_allreduce_grads -- ncclAllReduce over xGMI): a known vector summed over the ranks, checked exactly."""
    rc = {"ok": False}
    try:
        if args.same_device:
            raise RuntimeError("skipped: all ranks share one device (--same-device), RCCL needs one device per rank")
        uid = np.zeros(128, np.uint8)
        if rank == 0:
            uid[:] = ctx.comm_unique_id()
        t = torch.from_numpy(uid).to(tdev); dist.broadcast(t, 0); uid = t.cpu().numpy().copy()
        ctx.comm_init(rank, world, uid)
        n = net.n_params
        g = (np.arange(n, dtype=np.float32) % 97 + 1.0) * np.float32(rank + 1)
        ctx.h2d(ctx.lib.crux_mlp_grads_ptr(net.h), g)
        ctx.check(ctx.lib.crux_allreduce_grads(net.h)); ctx.sync()
        out = np.empty(n, np.float32); ctx.d2h(ctx.lib.crux_mlp_grads_ptr(net.h), out)
        want = (np.arange(n, dtype=np.float32) % 97 + 1.0) * np.float32(world * (world + 1) / 2)
        rc = {"ok": bool(np.array_equal(out, want)), "floats": int(n), "backend": "RCCL ncclAllReduce(sum, f32) issued by libcruxhip on its own stream"}
        ctx.comm_destroy()
    except Exception as e:      # noqa: BLE001
        rc = {"ok": False, "error": repr(e)}
    return rc


def guarded(fn, seconds):
    """fn() in a daemon thread: (result or None, still running?) -- for checks that must not be able to take the measurement down with them"""
    import threading
    box = {}

    def run():
        try:
            box["r"] = fn()
        except Exception as e:      # noqa: BLE001
            box["e"] = repr(e)
    t = threading.Thread(target=run, daemon=True); t.start(); t.join(seconds)
    return box.get("r"), t.is_alive()


def wait_percentiles(hist):
    """per-rank summary of the in-kernel flag waits (crux_peer_wait_hist: log2 bins of 10 ns ticks, [2 learner streams][2 workgroups][32]): the UPPER edge of the bin that holds the
    p-th percentile, in microseconds, over both streams and workgroups of this rank."""
    h = np.asarray(hist, np.int64).reshape(-1, 32).sum(axis=0); n = int(h.sum())
    if n == 0:
        return {"n": 0}
    cum = np.cumsum(h); out = {"n": n}
    for name, q in (("p50", 0.50), ("p90", 0.90), ("p99", 0.99), ("max", 1.0)):
        b = int(np.searchsorted(cum, q * n)) if q < 1.0 else int(np.nonzero(h)[0][-1])
        out[name + "_us"] = 10e-3 * 2 ** (min(b, 31) + 1)
    return out


def params_digest(nets):
    """order-independent exact digest of the replicated state (parameters + Adam moments): equal on all ranks iff the replicas are bit-identical."""
    import hashlib
    h = hashlib.sha256()
    for n in nets:
        h.update(n.get_params().tobytes()); m, v, bp = n.adam_state(); h.update(m.tobytes()); h.update(v.tobytes()); h.update(bp.tobytes())
    return np.frombuffer(h.digest()[:8], np.int64)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2", help="c2 = BASELINE configs[1] (the metric's config, default); c5 = configs[4]'s per-GPU shard")
    ap.add_argument("--sync", choices=["grad", "params"], default="grad", help="multi-GPU exchange: grad = per-minibatch gradient all-reduce inside the learner kernel (exact); params = periodic parameter averaging (RCCL)")
    ap.add_argument("--sync-every", type=int, default=1, help="--sync grad: 1 = the gradient is exchanged every minibatch inside the learner kernel (exact: equals one learner on the concatenated batch); "
                    "k > 1 = the in-kernel PERIODIC form: local Adam steps, theta / m / v averaged through the peer slots after every k-th step (crux_peer_set_sync_every; k must divide the minibatches of an epoch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sync", action="store_true", help="exercise the multi-GPU path even at world size 1 (testing)")
    ap.add_argument("--same-device", action="store_true", help="testing on a 1-GPU box: all ranks use device 0 (gloo rendezvous; the peer regions are still exchanged through hipIpc between the processes)")
    ap.add_argument("--replicas", type=int, default=64, help="also time S independent PPO learners (multi-seed) trained by two batched launches per iteration: the chip-level utilisation line (0 = skip)")
    ap.add_argument("--replicas-wide", type=int, default=128, help="second multi-seed line with one CU per learner (population > 64): the highest chip utilisation (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the supplementary configs (C5 shard, off-policy lines)")
    ap.add_argument("--early-stop", dest="early_stop", action="store_true", default=True, help="also time the KL-early-stopping variant (target_kl = 0.012, the reference's default PPO, rl/ppo.jl:40-65); on by default at N = 1 (BASELINE.md section 3: the metric is reported with KL early stopping on, and with it off)")
    ap.add_argument("--no-early-stop", dest="early_stop", action="store_false")
    ap.add_argument("--no-measure-traffic", dest="measure_traffic", action="store_false", help="keep the recorded roofline.traffic constant instead of re-measuring it (the default run measures when rocprofv3 is on PATH)")
    ap.add_argument("--measure-traffic", action="store_true", default=True, help="N = 1: re-measure roofline.traffic in this run -- two child passes of this script under rocprofv3 --pmc FETCH_SIZE / "
                    "--pmc WRITE_SIZE (separate passes, no trace domains), corrected as the microarchitecture guide prescribes; otherwise the recorded constant of profiles/ is reported")
    ap.add_argument("--inject-attach-failure", action="store_true", help="testing: corrupt the peers' IPC handles so that crux_peer_attach fails on every rank and the fallback ladder is taken")
    ap.add_argument("--selftest", action="store_true", help="N > 1: before timing, check the in-kernel gradient exchange across the real devices (identical shards on every rank must "
                    "reproduce an un-grouped learner; distinct shards must leave the replicas bit-identical), print per-rank flag-wait histograms, and run a 2..N-rank RCCL all-reduce "
                    "through the library's communicator (crux_comm_init / crux_allreduce_grads)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local = 0
    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner on stdout) are pointed at stderr instead
    sys.stdout.flush(); json_fd = os.dup(1); os.dup2(2, 1)
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = torch = None
    if world > 1 or args.force_sync:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        if args.same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import crux_jl_amd as crux
    from crux_jl_amd import dist as cdist
    ctx = crux.Context(local)
    crux.set_default_context(ctx)
    wl = WORKLOADS[args.workload]; E = wl["n_envs"]
    pi, buf, sampler = build_problem(crux, cdist.shard_seed(0, rank), workload=args.workload)
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": wl["lambda_e"]}
    a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=BATCH, epochs=EPOCHS, target_kl=None, name="actor_", shuffle_seed=100 + rank)
    c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=BATCH, epochs=EPOCHS, name="critic_", shuffle_seed=200 + rank)

    sync, comm_kind = None, "single GPU"
    requested_sync = args.sync
    if world > 1 or args.force_sync:
        sync, comm_kind = setup_group(args, crux, ctx, rank, world, local)
        if sync == "grad":
            # probe: a few minibatches through the in-kernel exchange before anything is timed. A rank whose peers never answer gets CRUX_EHIP from the kernel's own
            # timeout (and raises the abort word for the others); if ANY rank failed, all ranks drop to periodic parameter averaging with fresh learners.
            ok = True
            try:
                pa = crux.TrainingParams(loss=crux.ppo_loss, batch_size=BATCH, epochs=1, target_kl=None, name="actor_", shuffle_seed=900 + rank, max_batches=8)
                pc = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=BATCH, epochs=1, name="critic_", shuffle_seed=950 + rank, max_batches=8)
                ppo_iteration(crux, pi, buf, sampler, pa, pc, P, 0, sync); ctx.sync()
            except Exception as e:      # noqa: BLE001
                ok = False; print("bench.py: rank %d: the peer-slot exchange failed its probe (%r)" % (rank, e), file=sys.stderr)
            tdev0 = torch.device("cuda", local) if dist.get_backend() == "nccl" else torch.device("cpu")
            flag = torch.tensor([1 if ok else 0], device=tdev0); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                try:
                    ctx.peer_detach()
                except Exception:       # noqa: BLE001
                    pass
                args.sync = "params"
                pi, buf, sampler = build_problem(crux, cdist.shard_seed(0, rank), workload=args.workload)
                sync, comm_kind = setup_group(args, crux, ctx, rank, world, local)
                comm_kind += " [fallback: the in-kernel peer-slot gradient exchange failed its probe iteration on at least one rank]"

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    selftest = None
    if args.selftest and dist is not None and sync == "grad":
        selftest = run_selftest(crux, cdist, ctx, dist, torch, rank, world, local, args, P)

    if sync == "grad" and args.sync_every > 1:
        ctx.peer_set_sync_every(args.sync_every)      # (after the probe / selftest: their max_batches runs need the per-step form)
        comm_kind = "in-kernel PERIODIC exchange: local Adam steps, theta / m / v averaged through the peer slots after every %d-th minibatch inside the persistent learner kernel" % args.sync_every
    it = 0
    for _ in range(args.warmup):
        ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, P, it, sync); it += 1
    if sync == "grad":
        ctx.peer_wait_hist(reset=True); ctx.peer_hist_enable(True)      # lane 0 of two workgroups reads the wall clock around its flag wait: no measurable cost
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
    replicas_identical = None
    if dist is not None:
        tdev = torch.device("cuda", local) if dist.get_backend() == "nccl" else torch.device("cpu")
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev); dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dt = float(tmax.item())
        gs = torch.tensor([grad_steps], dtype=torch.float64, device=tdev); dist.all_reduce(gs); grad_steps = int(gs.item())
        dg = torch.tensor([int(params_digest((pi.A, pi.C)))], dtype=torch.int64, device=tdev)
        lo, hi = dg.clone(), dg.clone(); dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(int(lo.item()) == int(hi.item()))

    exchange = None; rccl_hung = False
    if dist is not None:
        tdev = torch.device("cuda", local) if dist.get_backend() == "nccl" else torch.device("cpu")
        mine = {"rank": rank}
        if sync == "grad":
            ctx.peer_hist_enable(False); mine.update(wait_percentiles(ctx.peer_wait_hist(reset=True)))
        waits = [None] * world; dist.all_gather_object(waits, mine)
        # RCCL as the library drives it: the params/native form used it in the timed region; otherwise one checked all-reduce through crux_comm_init / crux_allreduce_grads now
        if sync == "native":
            rccl = [{"ok": True, "how": "the timed region ran on the library's RCCL communicator"}] * world
        else:
            # under a watchdog: the timed numbers above must reach the JSON line whatever a second communicator does on a machine this was never run on
            def _rccl():
                r1 = rccl_check(ctx, dist, torch, rank, world, tdev, args, pi.A); got = [None] * world; dist.all_gather_object(got, r1); return got
            rccl, rccl_hung = guarded(_rccl, 120.0)
            if rccl is None:
                rccl = [{"ok": False, "error": "the RCCL check did not return within 120 s" if rccl_hung else "the RCCL check raised"}] * world
        exchange = {"kind": ("peer_slots_grad_every_step" if args.sync_every <= 1 else "peer_slots_periodic_k%d" % args.sync_every) if sync == "grad" else ("rccl_params_every_%d_epochs" % SYNC_EVERY if sync == "native" else "torch_allreduce_params_every_%d_epochs" % SYNC_EVERY),
                    "requested": "grad" if requested_sync == "grad" else "params", "fell_back": bool(requested_sync == "grad" and sync != "grad"),
                    "sync_every_minibatches": args.sync_every if sync == "grad" else None,
                    "flag_wait_per_rank": waits if sync == "grad" else None,
                    "rendezvous_probe_us": {"first_round_stream0": PROBE["us"][0], "slowest_later_round_stream0": PROBE["us"][1], "first_round_stream1": PROBE["us"][2], "slowest_later_round_stream1": PROBE["us"][3],
                                            "note": "crux_peer_probe on rank 0: 256 rendezvous through the peer regions before anything is timed; the first round absorbs the launch skew between the ranks"} if PROBE.get("us") else None,
                    "flag_wait_note": "upper edge (us) of the log2 bin holding the percentile of the per-exchange wait for the slowest peer's flag, both learner streams, workgroups 0 and 1 of every learner",
                    "rccl_ranks": int(sum(1 for r in rccl if r and r.get("ok"))), "rccl_detail": rccl[0] if rccl else None}

    prof = {k: ctx.prof_get(k) for k in ("rollout", "values", "gae", "whiten", "train_actor", "train_critic")}
    early = None
    if args.early_stop and world == 1:
        a_es = crux.TrainingParams(loss=crux.ppo_loss, batch_size=BATCH, epochs=EPOCHS, target_kl=0.012, name="actor_", shuffle_seed=300)
        ctx.sync(); t1 = time.perf_counter(); nbs = 0
        for _ in range(args.steps):
            nb, _i = ppo_iteration(crux, pi, buf, sampler, a_es, c_opt, P, it); it += 1; nbs += nb
        ctx.sync(); d1 = time.perf_counter() - t1
        early = {"env_steps_per_s": args.steps * E * T / d1, "grad_steps_per_s": nbs / d1, "grad_steps_per_iter": nbs / args.steps, "ms_per_iteration": 1e3 * d1 / args.steps, "target_kl": 0.012,
                 "note": "the reference's default PPO (KL early stopping, rl/ppo.jl:59): the actor stops after the epoch whose last minibatch has kl > target_kl; the critic learner is started speculatively beside the actor "
                         "and re-run on the right shuffle order when the actor stopped early (csrc/train.hip) -- bit-identical to actor-then-critic either way. The policy of this leg has already been trained by the timed iterations above."}

    def multi_seed_line(Sn):
        try:
            probs = [build_problem(crux, cdist.shard_seed(1000, r)) for r in range(Sn)]
            am = crux.TrainingParams(loss=crux.ppo_loss, batch_size=BATCH, epochs=EPOCHS, target_kl=None, name="actor_", shuffle_seed=5000)
            cm = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=BATCH, epochs=EPOCHS, name="critic_", shuffle_seed=6000)
            P2 = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}

            def multi_iteration(k):
                crux.steps_multi_([q[2] for q in probs], [q[1] for q in probs], Nsteps=probs[0][1].capacity, explore=True, i=k * probs[0][1].capacity, reset=True)   # one rollout launch for all S problems
                crux.whiten_multi_([q[1] for q in probs], "advantage")
                infos = crux.policy_gradient_training_multi([q[0] for q in probs], am, cm, P2, [q[1] for q in probs])
                return sum(i["actor_batches_trained"] + i["critic_batches_trained"] for i in infos)
            multi_iteration(0); ctx.sync(); ctx.prof_reset(); ctx.prof_enable(True); t1 = time.perf_counter(); gsm = 0
            n_it = max(1, min(args.steps, 2))
            for k in range(n_it):
                gsm += multi_iteration(1 + k)
            ctx.sync(); d1 = time.perf_counter() - t1; ctx.prof_enable(False)
            ms_a, n_a = ctx.prof_get("train_actor")
            steps_l = EPOCHS * (N_ENVS * T // BATCH)
            flops_all = Sn * steps_l * (FLOP_ACTOR_STEP + FLOP_CRITIC_STEP) * n_it          # every learner step of every replica in the timed region
            cus = 2 if Sn <= 64 else 1
            return {"replicas": Sn, "cus_per_learner": cus, "env_steps_per_s": n_it * Sn * N_ENVS * T / d1, "grad_steps_per_s": gsm / d1, "ms_per_iteration": 1e3 * d1 / n_it,
                    "phase_ms_per_iteration": {k: ctx.prof_get(k)[0] / n_it for k in ("rollout", "values", "gae", "whiten", "train_actor")},
                    "batched_actor_launch_ms": ms_a / max(1, n_a),
                    "end_to_end_TFLOPs": flops_all / d1 / 1e12, "end_to_end_frac_of_f32_mfma_peak": flops_all / d1 / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "learner_launches_frac_of_f32_mfma_peak": Sn * steps_l * (FLOP_ACTOR_STEP + FLOP_CRITIC_STEP) / (ms_a / max(1, n_a) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "note": "S independent PPO problems (own envs, buffers, seeds); all S actors in one launch, all S critics in a concurrent one: %d S CUs busy. end_to_end = all learner flops / wall time of whole iterations (rollout, GAE, order composition included); learner_launches = the same flops / duration of the concurrent actor||critic launches" % (2 * cus)}
        except Exception as e:      # noqa: BLE001  supplementary line: never let it take the headline measurement down
            return {"replicas": Sn, "error": repr(e)}

    multi = multi_wide = extra = None
    if world == 1 and not args.force_sync and args.workload == "c2":
        if args.replicas > 1:
            multi = multi_seed_line(args.replicas)
            if args.replicas_wide > 1:
                multi_wide = multi_seed_line(args.replicas_wide)
        if not args.no_extra:
            try:
                import bench_extra
                extra = bench_extra.run(crux, ctx, cpu=not args.no_cpu_baseline)
            except Exception as e:      # noqa: BLE001
                extra = {"error": repr(e)}

    traffic, traffic_how = (TRAFFIC_RECORDED.get(args.workload) if world == 1 else None), "RECORDED constant (profiles/r05_pmc_traffic*.txt), not measured in this run"
    if rank == 0 and world == 1 and args.measure_traffic:
        tv, how = measure_traffic(args.workload)
        if tv is not None:
            traffic, traffic_how = tv, how
        else:
            traffic_how += " [the in-run measurement (two child passes under rocprofv3 --pmc) was not possible: %s]" % how
    if rank == 0:
        env_steps = args.steps * E * T * world
        ms_actor, n_actor = prof["train_actor"]
        launches_per_iter = max(1, n_actor // max(1, args.steps))
        steps_per_launch = EPOCHS * (E * T // BATCH) / launches_per_iter
        avg_launch_s = (ms_actor / max(1, n_actor)) * 1e-3
        fa = flop_step(wl["actor"])
        achieved = fa * steps_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        out = {
            "metric": "env-steps/sec + grad-steps/sec, PPO 2048x32 rollout" if args.workload == "c2" else "env-steps/sec + grad-steps/sec, PPO 2048x128 rollout (C5 shard)",
            "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"] + ", batch 128, 80 epochs actor + 80 epochs critic (%d Adam steps/iter, KL early-stop off), Adam 3e-4" % (2 * EPOCHS * (E * T // BATCH)),
                       "envs_per_gpu": E, "rollout_T": T, "batch_size": BATCH, "epochs": EPOCHS,
                       "parallelism": ("env-shards x%d (global minibatch %d): %s" % (world, world * BATCH, comm_kind)) if sync is not None else "single GPU"},
            "grad_steps_per_s": grad_steps / dt,
            "phase_ms_per_iter": {k: v[0] / args.steps for k, v in prof.items()},
            "rollout_env_steps_per_s": (E * T * args.steps) / (prof["rollout"][0] * 1e-3) if prof["rollout"][0] > 0 else None,
            "roofline": {"kernel": "batch_train! actor (k_train_fs2, the role-specialised form of the feature-split learner: persistent fwd + ppo_loss + bwd + gradient exchange + Adam, 40 960 steps per launch; N > 1: its replica-group form with the gradient all-reduce over the peer slots inside the step; lagrange_ppo_loss is an instantiation of the same kernel)", "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "frac_of_occupied_cus": achieved / (PEAK_F32_MFMA_TFLOPS * 4.0 / 256.0), "occupied_cus": 4, "traffic": traffic,
                         "traffic_note": "HBM-side bytes per actor launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes; FETCH_SIZE tallies 128-byte requests at 64 B, calibrated in profiles/r02_fetch_calibration.txt). " + traffic_how + ". Algorithmic minibatch bytes per launch are %.0f MB (%d steps x %d B): the buffer is re-read from L2/MALL, and the per-step gradient exchange between the learner's workgroups (4 x 18 KB written, 3 x 18 KB read per workgroup and step) stays inside one XCD's L2" % (steps_per_launch * BATCH * (4 * wl["obs"] + (wl["act"] if wl["discrete"] else 4 * wl["act"]) + 8) / 1e6, int(steps_per_launch), BATCH * (4 * wl["obs"] + (wl["act"] if wl["discrete"] else 4 * wl["act"]) + 8)),
                         "avg_launch_ms": avg_launch_s * 1e3, "grad_steps_per_launch": steps_per_launch,
                         "us_per_grad_step": avg_launch_s * 1e6 / steps_per_launch if steps_per_launch else None,
                         "note": "serially dependent %.2f-MFLOP steps: each learner step is split over FOUR CUs of one XCD (k_train_fs2: four compute waves -- feature-split pairs per 16-sample tile -- and four helper waves per workgroup with code paths of their own; gradient exchange through the shared L2, the W2 partials handed over by the helper leader while the compute waves finish the first-layer pullback), actor and critic run concurrently -> 8 CUs busy; per-CU f32 MFMA peak is 0.614 TFLOP/s, so frac is structurally <= 4/256" % (fa / 1e6)},
        }
        if replicas_identical is not None:
            out["replicas_bit_identical_after_run"] = replicas_identical
        if exchange is not None:
            out["exchange"] = exchange; out["rccl_ranks"] = exchange["rccl_ranks"]; out["exchange_kind"] = exchange["kind"]
        if selftest is not None:
            out["selftest"] = selftest
        if early:
            out["early_stop"] = early; out["early_stop_env_steps_per_s"] = early["env_steps_per_s"]; out["early_stop_grad_steps_per_s"] = early["grad_steps_per_s"]
        if multi is not None:
            out["multi_seed"] = multi
        if multi_wide is not None:
            out["multi_seed_one_cu_per_learner"] = multi_wide
        if extra is not None:
            out["other_configs"] = extra
            # the other BASELINE configs' headline numbers as short top-level keys (the driver's parsed record carries them)
            def _g(name, key):
                v = extra.get(name) if isinstance(extra, dict) else None
                return v.get(key) if isinstance(v, dict) else None
            out["c1_dqn_gridworld_env_steps_per_s"] = _g("c1_dqn_gridworld", "env_steps_per_s"); out["c1_dqn_gridworld_seconds_N100k"] = _g("c1_dqn_gridworld", "seconds")
            out["c3_dqn_per_us_per_epoch"] = _g("c3_dqn_per", "us_per_epoch"); out["c3_dqn_per_grad_steps_per_s"] = _g("c3_dqn_per", "grad_steps_per_s")
            out["c3_solve_us_per_iteration"] = (_g("c3_dqn_per", "solve") or {}).get("us_per_iteration")
            out["c3_dqn_per_us_per_epoch_async_chains"] = _g("c3_dqn_per", "us_per_epoch_async_chains")
            out["c4_sac_us_per_epoch"] = _g("c4_sac", "us_per_epoch")
            out["c4_sac_us_per_epoch_async_chains"] = _g("c4_sac", "us_per_epoch_async_chains")
            out["c5_shard_env_steps_per_s"] = _g("c5_shard", "env_steps_per_s")
            out["c5_shard_us_per_grad_step"] = (extra.get("c5_shard", {}).get("roofline", {}) or {}).get("us_per_grad_step") if isinstance(extra, dict) else None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        sys.stdout.flush(); os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None and rccl_hung:
        os._exit(0)            # a thread of this process sits in a collective that never returned: the line is out, leave without the group's tear-down
    if dist is not None:
        barrier()
        if sync == "native":
            ctx.comm_destroy()                 # the library's communicator goes first, then torch's
        if sync == "grad":
            ctx.peer_detach()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
