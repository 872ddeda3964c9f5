"""GPU parity at BASELINE.json's full workload sizes: oracle REPLAYS (not only size-independent properties) of configs[1] (C2) and configs[4]
(C5, one GPU's 128-env shard), and a long-horizon drift test of the persistent learner kernel against the oracle's Float64-Adam restatement.
Precedent for an end-to-end tolerance: /root/reference/test/gym/solver_tests.jl:20-52 (same-seed reruns identical, CPU vs GPU within 1e-2);
the bounds here are the ones stated in tests/parity.py (param_tol) -- four orders tighter."""
import ctypes as C

import numpy as np
import pytest

import parity
from parity import L, O, crux

pytestmark = pytest.mark.gpu


def test_c2_full_size_iteration_replays_oracle(gpu_ctx):
    """configs[1]: PPO CartPole 4->64->64->2 (+ critic), 32 envs x 2048 steps, batch 128: rollout + GAE + returns + whiten + 2 full epochs of
    actor and critic batch_train! (1 024 Adam steps each) through crux_policy_gradient_training (the two concurrent persistent kernels)."""
    res = parity.ppo_iteration_parity(n_envs=32, T=2048, batch_size=128, epochs=2, seed=1234, max_steps=500, pair=True)
    print({k: v for k, v in res.items() if k != "ok"})
    assert res["actor_batches"] == (1024, 1024) and res["critic_batches"] == (1024, 1024)
    assert res["ok"], res


def test_c5_full_size_shard_iteration_replays_oracle(gpu_ctx):
    """configs[4], one GPU's shard: 128 SYNTH 17/6 environments x 2048 steps, tanh 17->64->64->6 GaussianPolicy + 17->64->64->1 critic, batch 128:
    rollout + GAE + returns + whiten + 1 full epoch (2 048 Adam steps) of both learners."""
    res = parity.ppo_iteration_parity(n_envs=128, T=2048, batch_size=128, epochs=1, seed=77, max_steps=1000, pair=True, family="synth_c5", gamma=0.99)
    print({k: v for k, v in res.items() if k != "ok"})
    assert res["actor_batches"] == (2048, 2048) and res["critic_batches"] == (2048, 2048)
    assert res["ok"], res


@pytest.mark.parametrize("which", ["actor", "critic"])
def test_persistent_learner_drift_over_4096_steps(gpu_ctx, which):
    """8 epochs x 512 minibatches = 4 096 consecutive Adam steps in ONE launch of the two-CU persistent kernel (v_rcp/v_sqrt Adam in f32) vs the
    oracle (Flux's per-element Float64 Adam): the difference must stay inside the stated growth law at every checkpoint."""
    E, T, bs, seed = 32, 2048, 128, 4321
    N = E * T
    extras = ["return", "logprob", "advantage"]
    ga, oa = parity.make_pair(parity.ACTOR_DIMS, parity.ACTS, seed, 0, "discrete")
    gc, oc = parity.make_pair(parity.CRITIC_DIMS, parity.ACTS, seed, 1)
    ob = O.OBuffer(4, 2, L.ACTION_DISCRETE, N, extras)
    oe = O.OEnv("cartpole", E, 500, 0.99, seed)
    oe.rollout(oa, parity.rollout_cfg(), ob, T)
    O.chk(O.lib().orc_fill_gae(ob.h, oc.h, 0.95, 0.99)); O.chk(O.lib().orc_fill_returns(ob.h, 0.99)); O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"]))
    data0 = {k: ob[k] for k in ob.keys()}                      # identical inputs on both sides: the test isolates the learner
    g, o = (ga, oa) if which == "actor" else (gc, oc)
    loss, head = ("ppo", "categorical") if which == "actor" else ("value_mse", "deterministic")
    o.adam_init(float(np.float32(3e-4)))
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    p0 = g.get_params().copy()
    # the oracle advances epoch by epoch (shuffle counter e, like one 8-epoch batch_train!) and is snapshotted after 1, 2, 4, 8 epochs;
    # the GPU runs 1, 2, 4 and 8 epochs from the same start, each as ONE persistent launch
    oinfo = np.zeros(L.INFO_N, np.float32); o_snap = {}
    for e in range(8):
        cfg = parity.train_cfg(loss, head, bs, 1, -1.0, 900, counter=e)
        O.chk(O.lib().orc_batch_train(o.h, ob.h, C.byref(cfg), None, O.vpz(oinfo), None))
        if e + 1 in (1, 2, 4, 8):
            o_snap[e + 1] = o.params.copy()
    worst = []
    for n_ep in (1, 2, 4, 8):
        g.set_params(p0)
        src = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), N, extras); src.push_(data0)
        opt = crux.TrainingParams(loss=crux.ppo_loss if which == "actor" else crux.value_mse_loss, batch_size=bs, epochs=n_ep, name=which + "_", shuffle_seed=900)
        inf = crux.batch_train_(g, opt, P, src)                 # a fresh TrainingParams attaches a fresh Adam state (m = v = 0, beta powers reset)
        steps = n_ep * (N // bs)
        assert inf[which + "_batches_trained"] == steps
        worst.append((steps, float(np.abs(g.get_params() - o_snap[n_ep]).max()), parity.param_tol(steps)))
    print(which, "drift (steps, max |dtheta|, bound):", worst)
    for steps, d, tol in worst:
        assert d < tol, worst
