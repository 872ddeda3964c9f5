"""GPU parity: OnPolicyGAIL pieces vs the oracle.

Reference seams: gail_d_loss / OnPolicyGAIL / GAIL_callback (src/model_free/il/on_policy_gail.jl:1-69), GAN_BCELoss Lᴰ (src/extras/gans.jl:7-9),
batch_train! over two buffers (src/training.jl:28-44), logcompσ (src/utils.jl:140-143), shuffle! (src/experience_buffer.jl:118-124).
Tolerances as in test_gpu_sac.py (dense engine: gradients 1e-4 of their scale, parameters 2e-5 after a step of ~1e-3)."""
import numpy as np
import pytest

import parity
from parity import crux, L, O
from test_gpu_sac import _step_close

pytestmark = pytest.mark.gpu


def _bufs(rng, od, ad, n, disc, ctx, extras=()):
    S = crux.ContinuousSpace(od); A = crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad)
    gb = crux.ExperienceBuffer(S, A, n, list(extras), ctx=ctx); ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, n, list(extras))
    a = np.eye(ad, dtype=bool)[:, rng.integers(0, ad, n)] if disc else rng.uniform(-1, 1, (ad, n)).astype(np.float32)
    ee = np.zeros((1, n), bool); ee[0, n // 3] = True; ee[0, -1] = True
    d = {"s": rng.normal(0, 1, (od, n)).astype(np.float32), "a": a, "sp": rng.normal(0, 1, (od, n)).astype(np.float32), "r": rng.normal(0, 1, (1, n)).astype(np.float32),
         "done": ee.copy(), "episode_end": ee}
    for k in extras:
        d[k] = rng.normal(0, 1, (1, n)).astype(np.float32)
    gb.push_(d); ob.push(d)
    return gb, ob, d


@pytest.mark.parametrize("od,ad,disc,hidden,n_ex,n_pi", [(3, 1, False, [32], 64, 64), (4, 2, True, [64, 64], 100, 128), (17, 6, False, [256, 256], 128, 77)])
def test_gail_discriminator_step_and_reward_match_oracle(gpu_ctx, od, ad, disc, hidden, n_ex, n_pi):
    ctx, rng = gpu_ctx, np.random.default_rng(5)
    dims, acts = [ad + od] + hidden + [1], ["relu"] * len(hidden) + ["identity"]
    g, o = parity.make_pair(dims, acts, 13, 0)
    g.attach_optimizer(crux.Adam(np.float32(1e-3))); o.adam_init(float(np.float32(1e-3)))
    gex, oex, _ = _bufs(rng, od, ad, n_ex + 9, disc, ctx); gpi, opi, _ = _bufs(rng, od, ad, n_pi + 5, disc, ctx)
    gi, oi = np.zeros(L.INFO_N, np.float32), np.zeros(L.INFO_N, np.float32)
    for step in range(3):
        ctx.check(ctx.lib.crux_gail_d_step(g.h, gex.h, 4, n_ex, gpi.h, 2, n_pi, gi.ctypes.data_as(L.vp)))
        O.chk(O.lib().orc_gail_d_step(o.h, oex.h, 4, n_ex, opi.h, 2, n_pi, O.vpz(oi)))
        assert abs(gi[0] - oi[0]) < 1e-4 * max(1, abs(oi[0])) and abs(gi[1] - oi[1]) < 1e-4 * max(1, abs(oi[1])), step
        assert _step_close(g, o, ctx), step
        g.set_params(o.params.copy())            # keep the two trajectories on the same point (Adam moments stay within tolerance)
    gm, om = np.zeros(1, np.float32), np.zeros(1, np.float32)
    ctx.check(ctx.lib.crux_gail_reward(g.h, gpi.h, 0.5, 1.5, gm.ctypes.data_as(L.vp))); O.chk(O.lib().orc_gail_reward(o.h, opi.h, 0.5, 1.5, O.vpz(om)))
    assert np.abs(gpi["r"] - opi["r"]).max() < 2e-5 * max(1, np.abs(opi["r"]).max()) and abs(gm[0] - om[0]) < 2e-5 * max(1, abs(om[0]))


def test_device_shuffle_is_the_spec_permutation(gpu_ctx):
    """crux_buffer_shuffle(seed, counter) == shuffle! with orc_perm(seed, counter, n) (include/crux_rng.h), every column, bit for bit."""
    rng = np.random.default_rng(2)
    for n in (2, 77, 1000):
        gb, ob, d = _bufs(rng, 3, 2, n, False, gpu_ctx, extras=["return", "advantage"])
        perm = np.zeros(n, np.int64); O.lib().orc_perm(41, 7, n, O.vpz(perm))
        crux.shuffle_device_(gb, 41, 7); ob.permute(perm + 1)
        for k in gb.keys():
            assert np.array_equal(gb[k], ob[k]), (n, k)
        assert np.array_equal(gb["s"], d["s"][:, perm])


def test_gail_solve_runs_and_trains_the_discriminator(gpu_ctx):
    """OnPolicyGAIL on Pendulum with demonstrations from the committed recording: PPO + GAIL_callback (discriminator batch_train! over two buffers,
    reward replacement, GAE/returns/whiten) end to end, like test/gym/solver_tests.jl:96."""
    import os
    ctx = gpu_ctx
    d = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pendulum_transitions.npz")))
    n = d["s"].shape[1]; S, A = crux.ContinuousSpace(3), crux.ContinuousSpace(1)
    demo = crux.ExperienceBuffer(S, A, n, ctx=ctx)
    obs3 = lambda x: np.vstack([np.cos(x[0]), np.sin(x[0]), x[1]]).astype(np.float32)      # the recording stores (theta, omega); the device env observes (cos, sin, omega)
    demo.push_({"s": obs3(d["s"]), "sp": obs3(d["sp"]), "a": d["a"], "r": d["r"], "done": d["done"], "episode_end": np.zeros((1, n), bool)})
    acts = ["relu", "relu", "identity"]
    pi = crux.ActorCritic(crux.GaussianPolicy(parity.chain([3, 64, 64, 1], acts), np.zeros(1, np.float32), seed=1, ctx=ctx), crux.ContinuousNetwork(parity.chain([3, 64, 64, 1], acts), seed=2, ctx=ctx))
    Dn = crux.ContinuousNetwork(parity.chain([4, 64, 64, 1], acts), seed=3, ctx=ctx)
    p0 = Dn.get_params().copy()
    mdp = crux.PendulumMDP(n_envs=4, seed=0)
    sv = crux.OnPolicyGAIL(pi, S, gamma=0.99, D=Dn, demo=demo, N=3 * 256, dN=256, max_steps=64, normalize_demo=False,
                           a_opt={"epochs": 2, "batch_size": 128}, c_opt={"epochs": 2, "batch_size": 128}, d_opt={"epochs": 2, "batch_size": 128}, target_kl=None)
    crux.solve(sv, mdp)
    assert len(sv.history) == 3
    assert not np.array_equal(Dn.get_params(), p0) and np.isfinite(Dn.get_params()).all() and np.isfinite(pi.A.get_params()).all()
    r = sv.buffer["r"][0]; Dout = Dn.forward(np.vstack([sv.buffer["a"], sv.buffer["s"]]))[0].astype(np.float64)
    # the rewards in the buffer are the discriminator's (computed BEFORE the actor/critic step, with the discriminator as it was then: only sanity-check the range)
    assert np.isfinite(r).all() and sv.d_opt.shuffle_counter == 3 * 2
    ls = -np.logaddexp(0, -Dout); assert np.abs(r - (0.5 * ls - 0.5 * (ls - Dout))).max() < 1e-5      # D did not change after the callback
