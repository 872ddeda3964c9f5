"""GPU parity: one PPO iteration through the C ABI vs the CPU oracle (same Philox-defined randomness)."""
import pytest

import parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_envs,T,bs,epochs", [(4, 64, 32, 2), (8, 128, 128, 2), (3, 50, 64, 1)])
def test_ppo_iteration_matches_oracle(gpu_ctx, n_envs, T, bs, epochs):
    res = parity.ppo_iteration_parity(n_envs=n_envs, T=T, batch_size=bs, epochs=epochs, seed=5 + n_envs)
    assert res["ok"], res


@pytest.mark.parametrize("target_kl", [-1.0, 0.002])
def test_policy_gradient_training_pair_matches_sequential_oracle(gpu_ctx, target_kl):
    """actor || critic concurrent learners (exact when no early stopping) and the sequential fallback (KL stop) both equal the oracle."""
    res = parity.ppo_iteration_parity(n_envs=8, T=64, batch_size=64, epochs=3, seed=21, target_kl=target_kl, pair=True)
    assert res["ok"], res
