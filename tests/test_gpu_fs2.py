"""k_train_fs2 (csrc/train_fs2_kernel.h), the role-specialised form of the register-resident learner, against k_train_fs (CRUX_FS2=0): the same arithmetic in the same order, so
parameters, Adam moments, beta powers, the buffer's row order and every statistic except the gradient norm (whose per-wave partial sums are grouped differently) must come out
BIT-IDENTICAL -- full epochs, ragged last minibatch, KL early stopping, max_batches, a NaN step (training.jl:20: error, no update), actor || critic through
crux_policy_gradient_training. The oracle comparisons of the other test modules run through k_train_fs2 by default."""
import numpy as np
import pytest

import parity
import replica_group as RG
from parity import L, O, crux

pytestmark = pytest.mark.gpu


def _shard(family, seed, E, T):
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES[family]
    _, oa = parity.make_pair(adims, acts, 50, 0, kind, n_extra=0 if disc else ad, extra_init=-0.5); _, oc = parity.make_pair(cdims, parity.CRITIC_ACTS.get(family, acts), 50, 1)
    extras = ["return", "logprob", "advantage"]
    ob = O.OBuffer(od, ad, L.ACTION_DISCRETE if disc else L.ACTION_CONTINUOUS, E * T, extras)
    env = O.OEnv("cartpole", E, 60, 0.99, seed) if family == "cartpole" else O.OEnv(okind, E, 60, 0.99, seed, so=od, sa=ad)
    env.rollout(oa, parity.rollout_cfg(head=head), ob, T)
    O.chk(O.lib().orc_fill_gae(ob.h, oc.h, 0.95, 0.99)); O.chk(O.lib().orc_fill_returns(ob.h, 0.99)); O.chk(O.lib().orc_whiten(ob.h, L.COL["advantage"]))
    return {k: ob[k] for k in ob.keys()}


def _nets(family, seed=31):
    od, ad, disc, adims, cdims, acts, kind, head, okind = parity.FAMILIES[family]
    if disc:
        a = crux.DiscreteNetwork(parity.chain(adims, acts), list(range(1, ad + 1)), seed=seed, stream=0)
    else:
        a = crux.GaussianPolicy(parity.chain(adims, acts), np.full(ad, -0.5, np.float32), seed=seed, stream=0)
    return a, crux.ContinuousNetwork(parity.chain(cdims, parity.CRITIC_ACTS.get(family, acts)), seed=seed, stream=1)


def _buffer(family, data):
    od, ad, disc = parity.FAMILIES[family][:3]
    N = data["s"].shape[1]
    b = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad), N, ["return", "logprob", "advantage"]); b.push_(data)
    return b


def _state(net):
    m, v, bp = net.adam_state()
    return [net.get_params(), m, v, bp]


def _same_bits(x, y):
    return all(np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b) for a, b in zip(x, y))


def _info_equal(i1, i2):
    for k in i1:
        if k == "_epoch_infos":          # rows of CRUX_INFO_*: everything but the gradient norm (column 1) bit for bit
            e1, e2 = np.asarray(i1[k]), np.asarray(i2[k]); keep = [j for j in range(e1.shape[1]) if j != L.INFO["grad_norm"]]
            assert np.array_equal(e1[:, keep], e2[:, keep], equal_nan=True) and np.allclose(e1[:, 1], e2[:, 1], rtol=2e-6, atol=2e-6, equal_nan=True)
        elif k.endswith("grad_norm"):
            assert abs(i1[k] - i2[k]) <= 2e-6 * max(1.0, abs(i2[k])), (k, i1[k], i2[k])
        else:
            assert i1[k] == i2[k] or (np.isnan(i1[k]) and np.isnan(i2[k])), (k, i1[k], i2[k])


CASES = [("cartpole", 8, 128, {}), ("synth_c5", 8, 128, {}), ("cheetah_ref", 4, 128, {}), ("synth_8_4", 4, 128, {}), ("synth_2_1", 4, 128, {}),
         ("cartpole", 5, 200, {}),                                   # 1000 rows: a ragged last minibatch of 104
         ("cartpole", 8, 128, {"max_batches": 11}), ("synth_c5", 8, 128, {"max_batches": 5})]


@pytest.mark.parametrize("family,E,T,kw", CASES, ids=["%s-%dx%d%s" % (c[0], c[1], c[2], "-" + "-".join(c[3]) if c[3] else "") for c in CASES])
def test_fs2_is_bit_identical_to_fs(gpu_ctx, monkeypatch, family, E, T, kw):
    data = _shard(family, 900, E, T)
    le = 0.1 if parity.FAMILIES[family][2] else 0.0
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": le}
    out = {}
    for form in ("fs2", "fs"):
        monkeypatch.setenv("CRUX_FS2", "1" if form == "fs2" else "0")
        a, c = _nets(family); res = []
        for net, loss in ((a, crux.ppo_loss), (c, crux.value_mse_loss)):
            b = _buffer(family, data)
            info = crux.batch_train_(net, crux.TrainingParams(loss=loss, batch_size=128, epochs=3, name="n_", shuffle_seed=17, **kw), P, b)
            res.append((_state(net), info, {k: b[k] for k in ("s", "advantage")}))
        out[form] = res
    for (s2, i2, b2), (s1, i1, b1) in zip(out["fs2"], out["fs"]):
        assert i2["n_batches_trained"] == i1["n_batches_trained"] > 0
        assert _same_bits(s2, s1), "max |dtheta| = %.3g" % float(np.abs(s2[0] - s1[0]).max())
        _info_equal(i2, i1)
        assert _same_bits([b2["s"], b2["advantage"]], [b1["s"], b1["advantage"]])          # the buffer's row order after the call


def test_fs2_kl_early_stopping_and_the_pair_call(gpu_ctx, monkeypatch):
    """crux_policy_gradient_training: actor || critic (two k_train_fs2 launches on the two learner streams); with target_kl the critic starts speculatively and may be re-run"""
    family = "cartpole"; data = _shard(family, 901, 8, 128)
    out = {}
    for form in ("fs2", "fs"):
        monkeypatch.setenv("CRUX_FS2", "1" if form == "fs2" else "0")
        res = []
        for tk in (None, 2e-4, 5.0):
            a, c = _nets(family); b = _buffer(family, data)
            class _S:
                pass
            s = _S(); s.agent = crux.PolicyParams(crux.ActorCritic(a, c)); s.P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
            s.a_opt = crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=6, target_kl=tk, name="actor_", shuffle_seed=3)
            s.c_opt = crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=4, name="critic_", shuffle_seed=4)
            info = crux.policy_gradient_training(s, b)
            res.append((_state(a), _state(c), info, b["s"]))
        out[form] = res
    for (a2, c2, i2, b2), (a1, c1, i1, b1) in zip(out["fs2"], out["fs"]):
        assert _same_bits(a2, a1) and _same_bits(c2, c1) and np.array_equal(b2, b1)
        _info_equal(i2, i1)
    assert out["fs2"][1][2]["actor_batches_trained"] < 6 * 8          # the KL stop fired with target_kl = 2e-4
    assert out["fs2"][0][2]["actor_batches_trained"] == out["fs2"][2][2]["actor_batches_trained"] == 6 * 8


@pytest.mark.parametrize("family,col", [("cartpole", "s"), ("synth_c5", "s"), ("cartpole", "return")])
def test_fs2_nan_step_is_an_error_and_leaves_the_parameters_of_the_step_before(gpu_ctx, monkeypatch, family, col):
    """training.jl:20: a NaN gradient norm throws BEFORE Flux.update!. One poisoned row that the third minibatch of the first epoch picks up: the call must fail with CRUX_ENAN and
    leave exactly the state after two steps -- also in W2, which the helper waves update before the small parameters' news is in."""
    data = _shard(family, 902, 8, 128); N = data["s"].shape[1]
    perm = np.random.default_rng(1).permutation(N)
    data[col] = data[col].copy(); data[col][0, perm[2 * 128 + 5]] = np.nan
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1 if family == "cartpole" else 0.0}
    which = 0 if col == "s" else 1; loss = crux.ppo_loss if which == 0 else crux.value_mse_loss
    got = {}
    for form in ("fs2", "fs"):
        monkeypatch.setenv("CRUX_FS2", "1" if form == "fs2" else "0")
        net = _nets(family)[which]; b = _buffer(family, data)
        with pytest.raises(crux.CruxError) as e:
            crux.batch_train_(net, crux.TrainingParams(loss=loss, batch_size=128, epochs=2, name="n_"), P, b, perms=np.stack([perm, perm]) + 1)
        assert e.value.code == L.ENAN
        got[form] = _state(net)
    # the reference state: two clean steps
    monkeypatch.setenv("CRUX_FS2", "1")
    net = _nets(family)[which]; clean = _shard(family, 902, 8, 128); b = _buffer(family, clean)
    crux.batch_train_(net, crux.TrainingParams(loss=loss, batch_size=128, epochs=1, max_batches=2, name="n_"), P, b, perms=perm[None, :] + 1)
    assert _same_bits(got["fs2"], got["fs"]) and _same_bits(got["fs2"][:3], _state(net)[:3])


@pytest.mark.parametrize("family,k", [("cartpole", 1), ("cartpole", 4), ("synth_c5", 1), ("synth_c5", 4)])
def test_fs2_replica_group_forms_against_those_of_fs(gpu_ctx, monkeypatch, family, k):
    """k_train_fs2<..., PX> / <..., PX, PXK> (round 5: the replica-group exchange inside the role-specialised kernel, C2 and C5 shapes) against the same forms of k_train_fs
    (CRUX_FS2=0): two contexts on one device wired into a group of two (crux_peer_attach_local), distinct shards, per-step all-reduce (k = 1) and the periodic form (k = 4).
    Same sums in the same order: the two replicas of a group and the two kernels must leave the same bits (the learner kernels are compiled without FMA contraction: before
    that the PX instantiations of k_train_fs differed from its plain form in the last place)."""
    import threading
    od, ad, disc = parity.FAMILIES[family][:3]
    shards = [_shard(family, 910, 8, 128), _shard(family, 911, 8, 128)]
    N = shards[0]["s"].shape[1]; epochs = 2
    rng = np.random.default_rng(12); perms = [np.stack([rng.permutation(N) for _ in range(epochs)]) for _ in range(2)]
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1 if disc else 0.0}
    c1 = crux.Context(0); ctxs = [gpu_ctx, c1]
    out = {}
    try:
        RG.attach_or_skip(ctxs)
        for c in ctxs:
            c.peer_set_sync_every(k)
        for form in ("fs2", "fs"):
            monkeypatch.setenv("CRUX_FS2", "1" if form == "fs2" else "0")
            res = []
            for which in (0, 1):
                nets, bufs = [], []
                for r, ctx in enumerate(ctxs):
                    ch = parity.chain(parity.FAMILIES[family][3 + which], parity.FAMILIES[family][5] if which == 0 else parity.CRITIC_ACTS.get(family, parity.FAMILIES[family][5]))
                    if which == 1:
                        g = crux.ContinuousNetwork(ch, ctx=ctx, seed=83, stream=3)
                    elif disc:
                        g = crux.DiscreteNetwork(ch, list(range(1, ad + 1)), ctx=ctx, seed=83, stream=3)
                    else:
                        g = crux.GaussianPolicy(ch, np.full(ad, -0.5, np.float32), ctx=ctx, seed=83, stream=3)
                    b = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad), N, ["return", "logprob", "advantage"], ctx=ctx); b.push_(shards[r])
                    nets.append(g); bufs.append(b)
                errs = [None, None]; infos = [None, None]
                def run(r):
                    try:
                        opt = crux.TrainingParams(loss=crux.ppo_loss if which == 0 else crux.value_mse_loss, batch_size=128, epochs=epochs, name="n_")
                        infos[r] = crux.batch_train_(nets[r], opt, P, bufs[r], perms=perms[r] + 1)
                    except Exception as e:      # noqa: BLE001
                        errs[r] = e
                RG.run_threads([(lambda r=r: run(r)) for r in range(2)])
                for e in errs:
                    if e is not None:
                        raise e
                st = [_state(n) for n in nets]
                assert _same_bits(st[0], st[1])                       # the replicas of a group never diverge
                res.append((st[0], infos[0]))
            out[form] = res
        for (s2, i2), (s1, i1) in zip(out["fs2"], out["fs"]):
            assert i2["n_batches_trained"] == i1["n_batches_trained"] == epochs * (N // 128)
            d = [float(np.abs(x - y).max() / max(1.0, float(np.abs(y).max()))) for x, y in zip(s2[:3], s1[:3])]
            print("fs2 vs fs replica-group form, %s k=%d: max |dtheta| %.3g |dm| %.3g |dv| %.3g (relative to the largest entry)" % (family, k, *d))
            assert _same_bits(s2, s1)
            _info_equal(i2, i1)
    finally:
        for c in ctxs:
            try:
                c.peer_set_sync_every(1); c.peer_detach()
            except Exception:       # noqa: BLE001
                pass
        c1.close()


@pytest.mark.parametrize("family,k", [("synth_8_4", 1), ("synth_8_4", 4), ("cheetah_ref", 1), ("cheetah_ref", 4), ("synth_2_1", 1)])
def test_fs2_group_of_two_on_identical_shards_equals_a_group_of_one(gpu_ctx, family, k):
    """The other shapes of the family as members of a replica group (k_train_fs2<..., PX / PXK>; k_train_fs has no group forms for them any more): two replicas with the SAME
    rows and shuffles -- g + g and the division by two are exact, the average of two identical theta / m / v is that theta / m / v -- must leave exactly the bits a group of
    ONE leaves (same instantiation, same code path)."""
    import threading
    od, ad, disc = parity.FAMILIES[family][:3]
    shard = _shard(family, 920, 8, 128); N = shard["s"].shape[1]; epochs = 2
    rng = np.random.default_rng(13); perms = np.stack([rng.permutation(N) for _ in range(epochs)])
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1 if disc else 0.0}
    def run_group(ctxs):
        RG.attach_or_skip(ctxs)
        try:
            for c in ctxs:
                c.peer_set_sync_every(k)
            res = []
            for which in (0, 1):
                nets, bufs = [], []
                for ctx in ctxs:
                    ch = parity.chain(parity.FAMILIES[family][3 + which], parity.FAMILIES[family][5] if which == 0 else parity.CRITIC_ACTS.get(family, parity.FAMILIES[family][5]))
                    if which == 1:
                        g = crux.ContinuousNetwork(ch, ctx=ctx, seed=85, stream=3)
                    elif disc:
                        g = crux.DiscreteNetwork(ch, list(range(1, ad + 1)), ctx=ctx, seed=85, stream=3)
                    else:
                        g = crux.GaussianPolicy(ch, np.full(ad, -0.5, np.float32), ctx=ctx, seed=85, stream=3)
                    b = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad), N, ["return", "logprob", "advantage"], ctx=ctx); b.push_(shard)
                    nets.append(g); bufs.append(b)
                errs = [None] * len(ctxs)
                def run(r):
                    try:
                        crux.batch_train_(nets[r], crux.TrainingParams(loss=crux.ppo_loss if which == 0 else crux.value_mse_loss, batch_size=128, epochs=epochs, name="n_"), P, bufs[r], perms=perms + 1)
                    except Exception as e:      # noqa: BLE001
                        errs[r] = e
                RG.run_threads([(lambda r=r: run(r)) for r in range(len(ctxs))])
                for e in errs:
                    if e is not None:
                        raise e
                st = [_state(n) for n in nets]
                for x in st[1:]:
                    assert _same_bits(st[0], x)
                res.append(st[0])
            return res
        finally:
            for c in ctxs:
                try:
                    c.peer_set_sync_every(1); c.peer_detach()
                except Exception:       # noqa: BLE001
                    pass
    c1 = crux.Context(0)
    try:
        two = run_group([gpu_ctx, c1]); one = run_group([gpu_ctx])
    finally:
        c1.close()
    for s2, s1 in zip(two, one):
        assert _same_bits(s2, s1), "max |dtheta| = %.3g" % float(np.abs(s2[0] - s1[0]).max())


@pytest.mark.parametrize("form,k", [("fs2", 1), ("fs2", 4), ("fs", 4)])
def test_fs2_replica_group_nan_step(gpu_ctx, monkeypatch, form, k):
    """training.jl:20 inside a replica group (k_train_fs2<..., PX / PXK>). Per-step form: the NaN arrives in every replica's group mean at the same step, so BOTH fail with
    CRUX_ENAN and keep the identical state of the step before. Periodic form: the gradients are local between exchanges -- the replica that met the NaN fails with CRUX_ENAN
    (state of its step before), skips the exchange and raises its peers' abort words, so the other replica leaves with CRUX_EHIP instead of waiting for the timeout."""
    import threading
    monkeypatch.setenv("CRUX_FS2", "1" if form == "fs2" else "0")
    family = "cartpole"; shards = [_shard(family, 930, 8, 128), _shard(family, 931, 8, 128)]
    N = shards[0]["s"].shape[1]; perm = np.random.default_rng(2).permutation(N)
    shards[1]["s"] = shards[1]["s"].copy(); shards[1]["s"][0, perm[2 * 128 + 5]] = np.nan          # replica 1's third minibatch picks the poisoned row up
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    c1 = crux.Context(0); ctxs = [gpu_ctx, c1]
    try:
        RG.attach_or_skip(ctxs)
        for c in ctxs:
            c.peer_set_sync_every(k)
        nets, bufs = [], []
        for r, ctx in enumerate(ctxs):
            g = crux.DiscreteNetwork(parity.chain(parity.FAMILIES[family][3], parity.FAMILIES[family][5]), [1, 2], ctx=ctx, seed=87, stream=3)
            b = crux.ExperienceBuffer(crux.ContinuousSpace(4), crux.DiscreteSpace(2), N, ["return", "logprob", "advantage"], ctx=ctx); b.push_(shards[r])
            nets.append(g); bufs.append(b)
        errs = [None, None]
        def run(r):
            try:
                crux.batch_train_(nets[r], crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=1, name="n_"), P, bufs[r], perms=perm[None, :] + 1)
            except crux.CruxError as e:
                errs[r] = e
        RG.run_threads([(lambda r=r: run(r)) for r in range(2)])
        assert errs[1] is not None and errs[1].code == L.ENAN
        st = [_state(n) for n in nets]
        if k == 1:
            assert errs[0] is not None and errs[0].code == L.ENAN
            assert _same_bits(st[0], st[1])
            assert not np.isnan(st[0][0]).any() and float(st[0][3][0]) == pytest.approx(0.9 ** 3)      # two clean steps taken: beta1^(t+1) with t = 2
        else:
            assert errs[0] is not None and errs[0].code == L.EHIP
            assert "NaN step" in str(errs[0]) and ctxs[0].peer_abort_reason()[0] == 5      # the peer learns WHY the group ended (round 6: it used to read "did not answer")
            assert not np.isnan(st[1][0]).any() and float(st[1][3][0]) == pytest.approx(0.9 ** 3)
    finally:
        for c in ctxs:
            try:
                c.peer_set_sync_every(1); c.peer_detach()
            except Exception:       # noqa: BLE001
                pass
        c1.close()
