"""GPU parity: behavioural cloning on the reference's recorded demonstrations.

Reference seams: mse_action_loss / logpdf_bc_loss / BC (src/model_free/il/bc.jl:1-70), BatchSolver solve (src/model_free/batch.jl:38-85),
normalize! / split (src/experience_buffer.jl:133-148), stop_on_validation_increase (src/utils.jl:59-72), BSON dumps of ExperienceBuffer
(examples/il/expert_data/*.bson; 512-row slices are committed under tests/golden/)."""
import ctypes as C
import os

import numpy as np
import pytest

import parity
from parity import crux, L, O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _demo(name, tmp_path, ctx):
    """golden slice -> BSON file in the reference's dump layout -> load_buffer (exercises the on-disk format both ways)."""
    from crux_jl_amd import bson
    d = dict(np.load(os.path.join(GOLD, name + "_transitions.npz")))
    n = d["s"].shape[1]

    class Host:
        next_ind = 1
        def __len__(self): return n
        def keys(self): return list(d)
        def __getitem__(self, k): return d[k]
    path = str(tmp_path / (name + ".bson")); bson.save_buffer(Host(), path)
    return bson.load_buffer(path, ctx=ctx), d


@pytest.mark.parametrize("case", ["cartpole_logpdf", "pendulum_logpdf", "pendulum_mse"])
def test_bc_losses_and_solve_match_oracle(gpu_ctx, tmp_path, case):
    name = case.split("_")[0]
    demo, d = _demo(name, tmp_path, gpu_ctx)
    n = len(demo); assert n == 512 and np.array_equal(demo["s"], d["s"]) and np.array_equal(demo["a"], d["a"])
    od, ad = d["s"].shape[0], d["a"].shape[0]
    if case == "cartpole_logpdf":
        dims, acts, kind, n_extra, loss, head = [od, 64, 64, ad], ["relu", "relu", "identity"], "discrete", 0, "logpdf_bc", "categorical"
    elif case == "pendulum_logpdf":
        dims, acts, kind, n_extra, loss, head = [od, 32, ad], ["tanh", "identity"], "gaussian", ad, "logpdf_bc", "gaussian"
    else:
        dims, acts, kind, n_extra, loss, head = [od, 32, ad], ["relu", "tanh"], "continuous", 0, "mse_action", "deterministic"
    g, o = parity.make_pair(dims, acts, 27, 0, kind, n_extra=n_extra, extra_init=-0.2)
    S = crux.ContinuousSpace(od, mu=np.full(od, 0.1, np.float32), sigma=np.full(od, 2.0, np.float32))
    perm = np.random.default_rng(5).permutation(n).astype(np.int64) + 1
    epochs, bs, window = 3, 64, 2
    solver = crux.BC(g, S, demo, opt={"batch_size": bs, "epochs": epochs, "optimizer": crux.Adam(np.float32(1e-3)), "shuffle_seed": 77}, window=window,
                     lambda_e=np.float32(1e-3), shuffle_perm=perm)
    assert (solver.a_opt.loss.name == loss)
    crux.solve(solver)
    # ---- the same on the oracle: normalize!, shuffle!, split, (epochs + 1) x batch_train!(epochs = 1), validation error per epoch
    disc = kind == "discrete"
    ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, n)
    nd = {k: v.copy() for k, v in d.items() if k in ("s", "a", "sp", "r", "done")}; nd["episode_end"] = np.zeros((1, n), bool)
    for k in ("s", "sp"):
        nd[k] = ((nd[k] - np.float32(0.1)) / np.float32(2.0)).astype(np.float32)
    ob.push(nd); O.chk(O.lib().orc_buffer_permute(ob.h, O.vpz(np.ascontiguousarray(perm - 1))))
    ntr = int(crux.split_batches(n, [1 - 0.3, 0.3])[0])
    otr = O.OBuffer(od, ad, ob.act_kind, ntr); ova = O.OBuffer(od, ad, ob.act_kind, n - ntr)
    full = {k: ob[k] for k in ("s", "a", "sp", "r", "done", "episode_end")}
    otr.push({k: v[:, :ntr] for k, v in full.items()}); ova.push({k: v[:, ntr:] for k, v in full.items()})
    o.adam_init(float(np.float32(1e-3)))
    info = np.zeros(L.INFO_N, np.float32); ves = []
    for ep in range(epochs + 1):
        cfg = parity.train_cfg(loss, head, bs, 1, -1.0, 77, counter=ep, lp=1.0, le=np.float32(1e-3))
        O.chk(O.lib().orc_batch_train(o.h, otr.h, C.byref(cfg), None, O.vpz(info), None))
        ids = np.arange(n - ntr, dtype=np.int64); vi = np.zeros(L.INFO_N, np.float32)
        O.chk(O.lib().orc_loss_grad(o.h, ova.h, C.byref(cfg), O.vpz(ids), ids.size, O.vpz(vi))); ves.append(float(vi[0]))
        if len(ves) >= 2 * window and np.mean(ves[-window:]) >= np.mean(ves[-2 * window:-window]):
            break
    assert len(solver.history) == len(ves)
    gv = [h["validation_error"] for h in solver.history]
    assert np.allclose(gv, ves, rtol=2e-4, atol=1e-6), (gv, ves)
    assert np.abs(g.get_params() - o.params).max() < 5e-5
    assert abs(solver.history[-1]["loss"] - info[0]) < 2e-6 * max(1.0, abs(info[0]))
    if loss == "logpdf_bc":
        assert "logpdf" in solver.history[-1] and "entropy" in solver.history[-1]
