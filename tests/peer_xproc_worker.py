"""One RANK of a cross-process replica group on one device (driven by tests/test_gpu_peer_xproc.py; not a test module itself).

The path every real multi-GPU run takes (crux.jl_amd/csrc/comm.hip): crux_peer_export -> the 64-byte hipIpc handles travel between the PROCESSES -> crux_peer_attach
maps the peers' fine-grained regions -> the persistent learner kernels of the ranks exchange through them. Here the ranks are processes on device 0 and the handles travel through
files in --dir (atomic renames; no torch, no network): what is exercised is the library, not a launcher. bench.py --gpus N is the launcher form (torch.distributed rendezvous) and
is run by the same test module.

    python tests/peer_xproc_worker.py --dir D --rank r --world n

D/cfg.json: {"family", "which": ["actor", "critic"] | "pair", "bs", "epochs", "k", "seed", "stream", "P", "timeout_ms", "die_rank", "le"}
D/shard_<r>.npz: the rank's buffer columns; D/perms_<r>.npy: int64 [n_nets][epochs][N] 0-based epoch permutations (actor's, then critic's).
Writes D/out_<r>.npz: parameters / Adam state / infos per network, or D/err_<r>.json when the training call raised (the "a peer died" scenario expects exactly that)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _atomic_save(path, arr):
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        np.save(f, arr)
    os.rename(tmp, path)


def _wait_for(paths, seconds, what):
    t0 = time.time()
    while not all(os.path.exists(p) for p in paths):
        if time.time() - t0 > seconds:
            raise SystemExit("peer_xproc_worker: timed out waiting for %s (%s)" % (what, [p for p in paths if not os.path.exists(p)]))
        time.sleep(0.01)


def barrier(d, name, rank, world, seconds=60.0):
    open(os.path.join(d, "bar_%s_%d" % (name, rank)), "w").close()
    _wait_for([os.path.join(d, "bar_%s_%d" % (name, r)) for r in range(world)], seconds, "barrier " + name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True); ap.add_argument("--rank", type=int, required=True); ap.add_argument("--world", type=int, required=True)
    a = ap.parse_args()
    d, rank, world = a.dir, a.rank, a.world
    cfg = json.load(open(os.path.join(d, "cfg.json")))
    import crux_jl_amd as crux
    from crux_jl_amd import _lib as L
    import parity
    ctx = crux.Context(0); crux.set_default_context(ctx)
    if cfg.get("timeout_ms"):
        ctx.peer_set_timeout_ms(int(cfg["timeout_ms"]))
    # ---- the exchange of the IPC handles between the processes, then the attach ------------------------------------------------------------
    _atomic_save(os.path.join(d, "handle_%d.npy" % rank), ctx.peer_export())
    hp = [os.path.join(d, "handle_%d.npy" % r) for r in range(world)]
    _wait_for(hp, 60.0, "the peers' handles")
    handles = np.stack([np.load(p) for p in hp])
    ctx.peer_attach(rank, world, handles)
    assert ctx.peer_size() == world
    barrier(d, "attached", rank, world)            # every rank has attached before any of them trains (cruxhip.h)
    if cfg.get("die_rank") == rank:               # this rank dies with the group attached and never trains: the survivors' kernels must time out, not hang
        os._exit(0)
    ctx.peer_set_budget_ms(int(cfg.get("budget_ms", 20000)))      # all waits of one launch together (csrc/peer_wait.h): a group that answers, but slowly, ends inside the test's limits
    probe_us = None
    if cfg.get("die_rank") is None and cfg.get("probe", True):      # the collective rendezvous probe: both processes' kernels answer each other at exchange speed
        probe_us = ctx.peer_probe(rounds=128, first_bound_ms=10000, round_bound_ms=50)
    if int(cfg.get("k", 1)) > 1:
        ctx.peer_set_sync_every(int(cfg["k"]))
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES[cfg["family"]]
    cacts = parity.CRITIC_ACTS.get(cfg["family"], acts)
    shard = dict(np.load(os.path.join(d, "shard_%d.npz" % rank)))
    perms = np.load(os.path.join(d, "perms_%d.npy" % rank))
    N = shard["s"].shape[1]; extras = ["return", "logprob", "advantage"]
    S, A = crux.ContinuousSpace(od), (crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad))
    buf = crux.ExperienceBuffer(S, A, N, extras); buf.push_(shard)
    seed, stream = int(cfg["seed"]), int(cfg["stream"])
    if disc:
        actor = crux.DiscreteNetwork(parity.chain(adims, acts), list(range(1, ad + 1)), seed=seed, stream=stream)
    else:
        actor = crux.GaussianPolicy(parity.chain(adims, acts), np.full(ad, -0.5, np.float32), seed=seed, stream=stream)
    critic = crux.ContinuousNetwork(parity.chain(cdims, cacts), seed=seed, stream=stream + 1)
    P = cfg["P"]; bs, epochs = int(cfg["bs"]), int(cfg["epochs"])
    out, t0 = {}, time.time()
    try:
        if cfg["which"] == "pair":               # policy_gradient_training: actor || critic, two persistent kernels and two exchange streams per rank
            class _S:
                pass
            s = _S(); s.agent = crux.PolicyParams(crux.ActorCritic(actor, critic)); s.P = P
            s.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=epochs, target_kl=cfg.get("target_kl"), name="actor_", shuffle_seed=int(cfg.get("shuffle_seed", 40)) + rank)
            s.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=epochs, name="critic_", shuffle_seed=int(cfg.get("shuffle_seed", 40)) + 20 + rank)
            info = crux.policy_gradient_training(s, buf)
            out["info_json"] = np.frombuffer(json.dumps({k: float(v) for k, v in info.items()}).encode(), np.uint8)
            nets = {"actor": actor, "critic": critic}
        else:
            nets = {}
            for i, which in enumerate(cfg["which"]):
                net = actor if which == "actor" else critic
                opt = crux.TrainingParams(loss=crux.ppo_loss if which == "actor" else crux.value_mse_loss, batch_size=bs, epochs=epochs, name="n_")
                info = crux.batch_train_(net, opt, P, buf, perms=perms[i] + 1)
                out[which + "_info"] = np.array([info["n_loss"], info["n_grad_norm"], info["n_batches_trained"]], np.float64)
                nets[which] = net
                buf.clear_(); buf.push_(shard)      # the critic trains on the un-shuffled shard again (each learner is compared with its own oracle run)
        ctx.sync()
    except crux.CruxError as e:
        json.dump({"code": int(e.code), "message": str(e), "seconds": time.time() - t0, "EHIP": int(L.EHIP)}, open(os.path.join(d, "err_%d.json.tmp" % rank), "w"))
        os.rename(os.path.join(d, "err_%d.json.tmp" % rank), os.path.join(d, "err_%d.json" % rank))
        os._exit(0)                              # (no detach / destroy: the group is broken, the process image goes away as a whole)
    for name, net in nets.items():
        m, v, bp = net.adam_state()
        out[name + "_params"], out[name + "_m"], out[name + "_v"], out[name + "_bp"] = net.get_params(), m, v, bp
    out["seconds"] = np.array([time.time() - t0])
    if probe_us is not None:
        out["probe_us"] = np.array(probe_us, np.float64)
    np.savez(os.path.join(d, "out_%d.tmp.npz" % rank), **out)
    os.rename(os.path.join(d, "out_%d.tmp.npz" % rank), os.path.join(d, "out_%d.npz" % rank))
    barrier(d, "done", rank, world)                # nobody unmaps a region a peer's kernel may still write to
    ctx.peer_detach()


if __name__ == "__main__":
    main()
