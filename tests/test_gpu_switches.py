"""The CRUX_* switches that select between two forms of the same computation, each against the default form: whatever a run leaves behind must be the same bits.
(The other switches have tests of their own: CRUX_FS / CRUX_MFMA_X2 / CRUX_FORCE_GENERIC in test_gpu_fs2.py, test_gpu_ppo_parity.py, test_gpu_round3.py;
CRUX_SPEC_PAIR, CRUX_DENSE_FUSED, CRUX_PER_FUSED_GATHER, CRUX_SAC_TILE_OPS in test_gpu_round4.py; CRUX_PUSH_FUSED in test_gpu_round5.py; CRUX_DQN_PERSIST, CRUX_EXEC_NO_KERNARG,
CRUX_SMALL_SOLVE_GENERIC in test_gpu_round3.py / test_gpu_round2.py. DESIGN.md section 8.2 has the table.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crux_jl_amd as crux  # noqa: E402
import bench  # noqa: E402
from test_gpu_round5 import _small_per_solve  # noqa: E402


def _ppo_iteration_state(workload, hidden=None, n_envs=4, T=128, epochs=2, seed=7):
    """one PPO iteration (steps!, whiten, actor || critic batch_train!) on a small problem of the workload's shapes; hidden = other hidden widths (the dense-engine learner)"""
    w = dict(bench.WORKLOADS[workload])
    if hidden:
        w["actor"] = [w["actor"][0]] + hidden + [w["actor"][-1]]; w["critic"] = [w["critic"][0]] + hidden + [1]
        bench.WORKLOADS["_switch_test"] = w; workload = "_switch_test"
    try:
        pi, buf, sampler = bench.build_problem(crux, seed, n_envs=n_envs, T_=T, workload=workload)
    finally:
        bench.WORKLOADS.pop("_switch_test", None)
    a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=epochs, target_kl=None, name="actor_", shuffle_seed=300)
    c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=epochs, name="critic_", shuffle_seed=400)
    nb, _ = bench.ppo_iteration(crux, pi, buf, sampler, a_opt, c_opt, {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}, 0)
    return nb, pi.A.get_params(), pi.C.get_params(), buf["s"].copy(), buf["advantage"].copy()


def _same(ref, got, names):
    for x, y, n in zip(ref, got, names):
        assert np.array_equal(np.asarray(x), np.asarray(y)), n


@pytest.mark.parametrize("switch,value,kw", [
    ("CRUX_PACK_ROWS", "0", dict(workload="c5")),                       # the learners gather the minibatch column by column instead of from the packed 128-byte rows
    ("CRUX_DENSE_PAIR", "0", dict(workload="c2", hidden=[128, 128])),   # dense-engine actor and critic one after the other instead of side by side
], ids=["pack_rows_off", "dense_pair_off"])
def test_on_policy_switch_forms_are_bit_identical(gpu_ctx, monkeypatch, switch, value, kw):
    monkeypatch.setenv(switch, value); ref = _ppo_iteration_state(**kw)
    monkeypatch.delenv(switch); got = _ppo_iteration_state(**kw)
    assert ref[0] == got[0] and ref[0] > 0
    _same(ref[1:], got[1:], ("actor", "critic", "s", "advantage"))


@pytest.mark.parametrize("switch,value", [("CRUX_NO_CHAINED_EPOCHS", "1"), ("CRUX_NO_FUSED_EPOCH", "1"), ("CRUX_SYNC_CHAINS", "1"), ("CRUX_EXEC_PERSISTENT", "1")],
                         ids=["no_chained_epochs", "no_fused_epoch", "sync_chains", "exec_persistent_is_ignored_by_asynchronous_chains"])
def test_off_policy_switch_forms_are_bit_identical(gpu_ctx, monkeypatch, switch, value):
    """a DQN + prioritized-replay solve on a full ring (8-256-256-4, 24 iterations of 4 steps + 4 epochs): the epochs call by call / one recording per epoch / chains with a
    read-back after each, against the default chained phase launches. CRUX_EXEC_PERSISTENT (the one-XCD persistent executor, development) applies to synchronous calls only: an
    asynchronous chain cannot read its status word back, so solve's chains must ignore it."""
    monkeypatch.setenv(switch, value); ref = _small_per_solve()
    monkeypatch.delenv(switch); got = _small_per_solve()
    _same(ref, got, ("params", "priorities", "cumsum", "max_priority", "min_priority", "indices", "s"))
