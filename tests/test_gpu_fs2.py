"""k_train_fs2 (csrc/train_fs2_kernel.h), the feature-split, role-specialised learner of the register-resident family. Until round 6 this module compared it bit for bit with
a second kernel of the same arithmetic (k_train_fs, retired); what remains is what does not need a twin:
  * against the sample-split two-CU kernel / the dense engine (CRUX_FS=0: another summation order) to fp32 tolerance -- full epochs, a ragged last minibatch, max_batches --, with
    the buffer's final row order bit for bit; the ORACLE comparisons of the other modules (whole PPO iterations, teacher-forced windows, full sizes) run through k_train_fs2;
  * KL early stopping through crux_policy_gradient_training (actor || critic);
  * a NaN step (training.jl:20: error, no update) leaves exactly the bits of the steps before it -- plain and inside a replica group;
  * replica groups: the replicas of a group never diverge; a group of two on identical shards leaves the bits of a group of one (every shape class, the 24- / 27-input ones
    included)."""
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
def test_fs2_against_the_sample_split_kernels(gpu_ctx, monkeypatch, family, E, T, kw):
    """the same call on k_train_fs2 and, with CRUX_FS=0, on the two-CU kernel (64-wide second layer) or the dense engine (cheetah_ref): another decomposition of the minibatch sum,
    so parameters / moments agree to 2e-5 after up to 24 steps; step counts and the buffer's final row order are exact"""
    data = _shard(family, 900, E, T)
    le = 0.1 if parity.FAMILIES[family][2] else 0.0
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": le}
    out = {}
    for form in ("fs2", "split"):
        monkeypatch.setenv("CRUX_FS", "1" if form == "fs2" else "0")
        a, c = _nets(family); res = []
        for net, loss in ((a, crux.ppo_loss), (c, crux.value_mse_loss)):
            b = _buffer(family, data)
            info = crux.batch_train_(net, crux.TrainingParams(loss=loss, batch_size=128, epochs=3, name="n_", shuffle_seed=17, **kw), P, b)
            res.append((_state(net), info, {k: b[k] for k in ("s", "advantage")}))
        out[form] = res
    for (s2, i2, b2), (s1, i1, b1) in zip(out["fs2"], out["split"]):
        steps = int(i2["n_batches_trained"])
        assert steps == i1["n_batches_trained"] > 0
        d = float(np.abs(s2[0] - s1[0]).max())
        assert d < 2e-5, d          # two kernels that each sit within parity.param_tol of the oracle (the one-CU against the two-CU form: the same bound, test_gpu_ppo_parity.py)
        assert np.allclose(s2[1], s1[1], rtol=0, atol=2e-5) and np.allclose(s2[2], s1[2], rtol=0, atol=2e-5) and np.array_equal(s2[3], s1[3])      # m, v; the beta powers exactly
        for k in i2:
            if k != "_epoch_infos" and isinstance(i2[k], float):
                assert abs(i2[k] - i1[k]) <= 2e-4 * max(1.0, abs(i1[k])), (k, i2[k], i1[k])
        assert _same_bits([b2["s"], b2["advantage"]], [b1["s"], b1["advantage"]])          # the buffer's row order after the call


def test_fs2_kl_early_stopping_and_the_pair_call(gpu_ctx, monkeypatch):
    """crux_policy_gradient_training: actor || critic (two k_train_fs2 launches on the two learner streams); with target_kl the critic starts speculatively and may be re-run.
    Without a stop (None, or a bound that never fires) the pair call leaves the bits of two separate batch_train! calls; a bound that fires ends the actor early, the critic
    still trains all its epochs from the buffer order the stopped actor left."""
    family = "cartpole"; data = _shard(family, 901, 8, 128)
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
    assert res[1][2]["actor_batches_trained"] < 6 * 8          # the KL stop fired with target_kl = 2e-4
    assert res[0][2]["actor_batches_trained"] == res[2][2]["actor_batches_trained"] == 6 * 8
    assert res[1][2]["critic_batches_trained"] == res[0][2]["critic_batches_trained"] == 4 * 8
    assert _same_bits(res[0][0], res[2][0]) and _same_bits(res[0][1], res[2][1]) and np.array_equal(res[0][3], res[2][3])      # a bound that never fires changes nothing
    # the same two loops as separate calls (actor, then critic on the buffer the actor left): the pair call's overlap is invisible in the results
    a, c = _nets(family); b = _buffer(family, data); P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1}
    crux.batch_train_(a, crux.TrainingParams(loss=crux.ppo_loss, batch_size=128, epochs=6, name="actor_", shuffle_seed=3), P, b)
    crux.batch_train_(c, crux.TrainingParams(loss=crux.value_mse_loss, batch_size=128, epochs=4, name="critic_", shuffle_seed=4), P, b)
    assert _same_bits(_state(a), res[0][0]) and _same_bits(_state(c), res[0][1]) and np.array_equal(b["s"], res[0][3])
    for st in res[1][:2]:
        assert all(np.isfinite(x).all() for x in st)


@pytest.mark.parametrize("family,col", [("cartpole", "s"), ("synth_c5", "s"), ("cartpole", "return")])
def test_fs2_nan_step_is_an_error_and_leaves_the_parameters_of_the_step_before(gpu_ctx, monkeypatch, family, col):
    """training.jl:20: a NaN gradient norm throws BEFORE Flux.update!. One poisoned row that the third minibatch of the first epoch picks up: the call must fail with CRUX_ENAN and
    leave exactly the state after two steps -- also in W2, which the helper waves update before the small parameters' news is in."""
    data = _shard(family, 902, 8, 128); N = data["s"].shape[1]
    perm = np.random.default_rng(1).permutation(N)
    data[col] = data[col].copy(); data[col][0, perm[2 * 128 + 5]] = np.nan
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1 if family == "cartpole" else 0.0}
    which = 0 if col == "s" else 1; loss = crux.ppo_loss if which == 0 else crux.value_mse_loss
    net = _nets(family)[which]; b = _buffer(family, data)
    with pytest.raises(crux.CruxError) as e:
        crux.batch_train_(net, crux.TrainingParams(loss=loss, batch_size=128, epochs=2, name="n_"), P, b, perms=np.stack([perm, perm]) + 1)
    assert e.value.code == L.ENAN
    got = _state(net)
    # the reference state: two clean steps
    net = _nets(family)[which]; clean = _shard(family, 902, 8, 128); b = _buffer(family, clean)
    crux.batch_train_(net, crux.TrainingParams(loss=loss, batch_size=128, epochs=1, max_batches=2, name="n_"), P, b, perms=perm[None, :] + 1)
    assert _same_bits(got[:3], _state(net)[:3])


@pytest.mark.parametrize("family,k", [("cartpole", 1), ("cartpole", 4), ("synth_c5", 1), ("synth_c5", 4)])
def test_fs2_replica_group_on_distinct_shards(gpu_ctx, family, k):
    """k_train_fs2<..., PX> / <..., PX, PXK> (the replica-group exchange inside the kernel, C2 and C5 shapes): two contexts on one device wired into a group of two
    (crux_peer_attach_local), DISTINCT shards, per-step all-reduce (k = 1) and the periodic form (k = 4). The two replicas add the same contributions in rank order: they must
    leave the same bits; every step ran; and the result is not what a replica alone would have trained (the peer's shard matters)."""
    od, ad, disc = parity.FAMILIES[family][:3]
    shards = [_shard(family, 910, 8, 128), _shard(family, 911, 8, 128)]
    N = shards[0]["s"].shape[1]; epochs = 2
    rng = np.random.default_rng(12); perms = [np.stack([rng.permutation(N) for _ in range(epochs)]) for _ in range(2)]
    P = {"eps": 0.2, "lambda_p": 1.0, "lambda_e": 0.1 if disc else 0.0}
    c1 = crux.Context(0); ctxs = [gpu_ctx, c1]

    def make(which, ctx, r):
        ch = parity.chain(parity.FAMILIES[family][3 + which], parity.FAMILIES[family][5] if which == 0 else parity.CRITIC_ACTS.get(family, parity.FAMILIES[family][5]))
        if which == 1:
            g = crux.ContinuousNetwork(ch, ctx=ctx, seed=83, stream=3)
        elif disc:
            g = crux.DiscreteNetwork(ch, list(range(1, ad + 1)), ctx=ctx, seed=83, stream=3)
        else:
            g = crux.GaussianPolicy(ch, np.full(ad, -0.5, np.float32), ctx=ctx, seed=83, stream=3)
        b = crux.ExperienceBuffer(crux.ContinuousSpace(od), crux.DiscreteSpace(ad) if disc else crux.ContinuousSpace(ad), N, ["return", "logprob", "advantage"], ctx=ctx); b.push_(shards[r])
        return g, b
    alone = []
    for which in (0, 1):      # replica 0 on its own shard, no group
        g, b = make(which, gpu_ctx, 0)
        crux.batch_train_(g, crux.TrainingParams(loss=crux.ppo_loss if which == 0 else crux.value_mse_loss, batch_size=128, epochs=epochs, name="n_"), P, b, perms=perms[0] + 1)
        alone.append(_state(g))
    try:
        RG.attach_or_skip(ctxs)
        for c in ctxs:
            c.peer_set_sync_every(k)
        for which in (0, 1):
            nets, bufs = zip(*[make(which, ctx, r) for r, ctx in enumerate(ctxs)])
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
            assert infos[0]["n_batches_trained"] == infos[1]["n_batches_trained"] == epochs * (N // 128)
            assert all(np.isfinite(x).all() for x in st[0])
            d = float(np.abs(st[0][0] - alone[which][0]).max())
            assert 1e-6 < d < 0.1, d                              # the group mean of two shards is neither replica 0's own gradient nor something wild
    finally:
        for c in ctxs:
            try:
                c.peer_set_sync_every(1); c.peer_detach()
            except Exception:       # noqa: BLE001
                pass
        c1.close()


@pytest.mark.parametrize("family,k", [("synth_8_4", 1), ("synth_8_4", 4), ("cheetah_ref", 1), ("cheetah_ref", 4), ("synth_2_1", 1), ("synth_24_4", 1), ("synth_27_8", 1), ("synth_27_8", 4)])
def test_fs2_group_of_two_on_identical_shards_equals_a_group_of_one(gpu_ctx, family, k):
    """The other shapes of the family as members of a replica group (k_train_fs2<..., PX / PXK>; the 24- / 27-input ones keep their W2 backups in registers): two replicas with the SAME
    rows and shuffles -- g + g and the division by two are exact, the average of two identical theta / m / v is that theta / m / v -- must leave exactly the bits a group of
    ONE leaves (same instantiation, same code path)."""
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


@pytest.mark.parametrize("k", [1, 4])
def test_fs2_replica_group_nan_step(gpu_ctx, k):
    """training.jl:20 inside a replica group (k_train_fs2<..., PX / PXK>). Per-step form: the NaN arrives in every replica's group mean at the same step, so BOTH fail with
    CRUX_ENAN and keep the identical state of the step before. Periodic form: the gradients are local between exchanges -- the replica that met the NaN fails with CRUX_ENAN
    (state of its step before), skips the exchange and raises its peers' abort words, so the other replica leaves with CRUX_EHIP instead of waiting for the timeout."""
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
