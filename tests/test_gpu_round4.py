"""Round-4 GPU tests: speculative actor || critic learners under KL early stopping, the fused block kernels of the dense engine against the per-layer launches,
prioritized sampling as one launch."""
import os
import subprocess
import sys

import numpy as np
import pytest

import parity
from parity import L, O, crux

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pg_pair(target_kl, seed, epochs=6, n_envs=8, T=128, bs=128):
    """rollout -> GAE -> whiten -> policy_gradient_training (one call for actor + critic) at the headline shapes; returns what the call left behind"""
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES["cartpole"]
    N = n_envs * T; extras = ["return", "logprob", "advantage"]
    ga, _ = parity.make_pair(adims, acts, seed, 0, kind); gc, _ = parity.make_pair(cdims, acts, seed, 1)
    gb = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad), N, extras)
    pi = crux.ActorCritic(ga, gc)
    gs = crux.Sampler(crux.CartPoleMDP(n_envs=n_envs, seed=seed), pi, max_steps=50, required_columns=extras, lam=0.95)
    crux.steps_(gs, gb, Nsteps=N, explore=True, i=0, reset=True); crux.whiten_(gb, "advantage")
    a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=bs, epochs=epochs, target_kl=target_kl, name="actor_", shuffle_seed=seed + 100)
    c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=bs, epochs=epochs, name="critic_", shuffle_seed=seed + 200)
    class _S: pass
    sv = _S(); sv.agent = crux.PolicyParams(pi); sv.a_opt, sv.c_opt, sv.P = a_opt, c_opt, {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    info = crux.policy_gradient_training(sv, gb)
    return ga.get_params(), gc.get_params(), gb["s"].copy(), gb["advantage"].copy(), info


@pytest.mark.parametrize("target_kl,stops", [(1e-4, True), (10.0, False)])
def test_speculative_critic_start_equals_actor_then_critic(gpu_ctx, monkeypatch, target_kl, stops):
    """policy_gradient_training with KL early stopping (the reference's default PPO, rl/ppo.jl:59): the critic learner starts beside the actor on the row order of
    "the actor runs every epoch" and is restarted on the right order when the actor stops early (csrc/train.hip). Both ways the result is the sequential
    batch_train!(actor) ; batch_train!(critic) of on_policy.jl:65-69, bit for bit -- parameters, Adam-trained critic, buffer row order, infos -- for a seed that stops
    and for one that does not; and the sequential form is the one the oracle is compared with in test_gpu_ppo_parity.py."""
    monkeypatch.setenv("CRUX_SPEC_PAIR", "0"); ref = _pg_pair(target_kl, seed=11)
    monkeypatch.delenv("CRUX_SPEC_PAIR"); got = _pg_pair(target_kl, seed=11)
    assert (ref[4]["actor_batches_trained"] < 6 * 8) == stops, ref[4]["actor_batches_trained"]      # 1024 rows / 128 = 8 minibatches per epoch, 6 epochs
    for x, y in zip(ref[:4], got[:4]):
        assert np.array_equal(x, y), float(np.abs(x - y).max())
    for k in ref[4]:
        assert ref[4][k] == got[4][k] or (np.isnan(ref[4][k]) and np.isnan(got[4][k])), k


def test_speculative_pair_matches_the_oracle(gpu_ctx):
    """the same call against the oracle's sequential loop, with a stop after the first epochs (batch 128: the feature-split kernels, the speculative path)"""
    res = parity.ppo_iteration_parity(n_envs=8, T=128, batch_size=128, epochs=4, seed=11, target_kl=1e-4, pair=True)
    assert res["ok"], res


def test_fused_dense_kernels_equal_the_per_layer_launches(gpu_ctx):
    """dense_fused.h (layers 0 + 1 forward in registers, LDS-staged weight gradient, quarter-split data gradient with the layer-0 partials completed by the norm op, the
    output layer's data gradient folded in) and the one-launch prioritized sampling against CRUX_DENSE_FUSED=0 / CRUX_PER_FUSED_GATHER=0 (one Gemm16 launch per layer,
    search and gather apart): chained DQN + PER epochs at 256 and 128 wide, SAC, TD3 and DDPG solves -- parameters, priorities, sampled ids, infos bit for bit.
    The switches are read once per process, so the two forms run in child processes (tools/fused_check.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fused_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fused_check: OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
