"""No-GPU checks: the C ABI library loads and exports every symbol include/cruxhip.h declares; host-side mirror logic."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import crux_jl_amd as crux
from crux_jl_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "cruxhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(crux_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(crux.LIB_PATH), "libcruxhip.so must be built (python -c 'import __graft_entry__ as g; g.build()')"
    lib = C.CDLL(crux.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 60
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s


def test_binding_table_covers_the_header():
    assert sorted(L.SIGNATURES) == header_symbols()


def test_struct_layouts_match_the_header_sizes():
    # crux_rollout_cfg / crux_train_cfg are passed by pointer: sizes must match the C definitions (natural alignment)
    assert C.sizeof(L.RolloutCfg) == 72 and C.sizeof(L.TrainCfg) == 64


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(crux.CruxError):
        crux.Context(0)


def test_spaces_and_mdp_data():        # src/spaces.jl, test/experience_buffer_tests.jl:8-20
    S, A = crux.ContinuousSpace(3), crux.DiscreteSpace(4)
    d1 = crux.mdp_data(S, A, 100); d2 = crux.mdp_data(S, A, 100, ["weight", "t", "advantage", "return", "logprob"])
    assert d1["s"].shape == (3, 100) and d1["s"].dtype == np.float32 and (d1["s"] == 0).all()
    assert d1["a"].shape == (4, 100) and d1["a"].dtype == np.bool_ and d1["done"].dtype == np.bool_
    assert (d2["weight"] == 1).all() and (d2["return"] == 0).all() and d2["t"].dtype == np.int64 and "return" not in d1
    with pytest.raises(KeyError):
        crux.mdp_data(S, A, 10, ["bad_key"])
    assert crux.DiscreteSpace([5, 6, 7]).N == 3 and crux.dim(crux.ContinuousSpace((2, 2))) == (2, 2)


def test_split_batches_and_schedule():  # test/experience_buffer_tests.jl:177-180, test/util_tests.jl:56-86
    assert crux.split_batches(100, [0.5, 0.5]) == [50, 50] and crux.split_batches(100, [1.0]) == [100]
    assert crux.split_batches(100, [1 / 3, 1 / 3, 1 / 3]) == [34, 33, 33]
    with pytest.raises(AssertionError):
        crux.split_batches(100, 0.4)
    s = crux.LinearDecaySchedule(1.0, 0.1, 10)
    assert s(0) == 1.0 and s(10) == pytest.approx(0.1) and s(100) == 0.1 and s(5) == pytest.approx(0.55)


def test_adam_constructor_keeps_float32_literal_as_float64():   # Adam(3f-4).eta == Float64(3f-4)
    assert crux.Adam(np.float32(3e-4)).eta == 0.0003000000142492354


def test_shard_helpers():
    from crux_jl_amd import dist
    assert dist.partition_envs(256, 8, 3) == (96, 128)
    with pytest.raises(ValueError):
        dist.partition_envs(10, 4, 0)
    assert len({dist.shard_seed(0, r) for r in range(8)}) == 8


def test_bson_buffer_round_trip_host_side(tmp_path):
    """crux.jl_amd/bson.py: writer -> reader round trip of the ExperienceBuffer dump layout (no GPU: a stand-in object with the buffer's host interface)."""
    from crux_jl_amd import bson
    rng = np.random.default_rng(0); n = 37

    class FakeBuf:
        next_ind = 1
        total_count = 2 * n - 3                       # not equal to elements: push_reservoir! counts what it has SEEN (experience_buffer.jl:262-288)
        cols = {"s": rng.normal(0, 1, (4, n)).astype(np.float32), "a": np.eye(2, dtype=bool)[:, rng.integers(0, 2, n)], "sp": rng.normal(0, 1, (4, n)).astype(np.float32),
                "r": np.ones((1, n), np.float32), "done": rng.random((1, n)) < 0.1, "t": np.arange(1, n + 1, dtype=np.int64)[None, :]}
        extra = {"expert_val": rng.normal(0, 1, (1, n)).astype(np.float32)}
        def __len__(self): return n
        def keys(self): return list(self.cols)
        def __getitem__(self, k): return self.cols[k]
    fb = FakeBuf(); path = str(tmp_path / "buf.bson")
    bson.save_buffer(fb, path)
    cols, meta = bson.read_columns(path)
    assert meta["elements"] == n and meta["next_ind"] == 1 and meta["priority_params"] is None and meta["total_count"] == 2 * n - 3
    # struct field count = the reference's ExperienceBuffer (data, elements, next_ind, indices, priority_params, total_count; experience_buffer.jl:53-60)
    assert len(bson._parse(open(path, "rb").read())["data"]["data"]) == 6
    for k, v in {**fb.cols, **fb.extra}.items():
        assert cols[k].dtype == v.dtype and np.array_equal(cols[k], v), k


def test_tensorboard_event_files_round_trip_and_log_cadence(tmp_path):
    """crux_jl_amd.logging: CRC-32C known answer, TFRecord/Event encoding read back by readtb (src/analysis.jl:2-13), run-directory increment,
    elapsed() for steps and ranges (src/logging.jl:1-2), aggregate_info (:60-66), Base.log cadence and log_episode_averages (:29-57, :99-111)."""
    from crux_jl_amd import logging as lg
    assert lg.crc32c(b"123456789") == 0xE3069283                      # the standard CRC-32C check value
    assert lg.elapsed(500, 500) and not lg.elapsed(501, 500) and lg.elapsed((499, 503), 500) and not lg.elapsed((501, 999), 500) and lg.elapsed((1, 1000), 500)
    assert lg.aggregate_info([{"a": 1.0, "b": 2.0}, {"a": 3.0}]) == {"a": 2.0, "b": 2.0}
    d = str(tmp_path / "log" / "ppo")
    tb = lg.TBLogger(d); tb2 = lg.TBLogger(d)
    assert tb.logdir == d and tb2.logdir == d + "_1"                  # tb_increment
    for i in range(1, 6):
        tb.log_value("actor_loss", 0.5 / i, step=100 * i)
    tb.log_value("returns", [1.0, 2.0], step=7)
    tb.close()
    h = lg.readtb(d)
    assert h["actor_loss"][0] == [100, 200, 300, 400, 500] and np.allclose(h["actor_loss"][1], [0.5 / i for i in range(1, 6)])
    assert h["returns/1"] == ([7], [1.0]) and h["returns/2"] == ([7], [2.0])
    raw = bytearray(open(tb.path, "rb").read()); raw[-6] ^= 0x40; open(tb.path, "wb").write(raw)
    with pytest.raises(ValueError):
        lg.readtb(d)

    class Buf:                                                         # the two buffer operations log_episode_averages uses
        cols = {"r": np.arange(10, dtype=np.float32)[None], "episode_end": np.array([[0, 0, 1, 0, 0, 0, 1, 0, 0, 1]], bool)}
        def get_last_N_indices(self, N): return list(range(10 - N + 1, 11))
        def __getitem__(self, k): return self.cols[k]
    class Sv:
        buffer = Buf()
    p = lg.LoggerParams(dir=str(tmp_path / "run"), period=4, fns=[lg.log_episode_averages(["r"], 4)])
    assert lg.log(p, 3, {"x": 1.0}, S=Sv()) is None                    # period not reached
    w = lg.log(p, (5, 8), {"x": 1.0}, lambda **kw: {"y": 2.0}, S=Sv())
    assert w == {"avg_r": (6 + 7 + 8 + 9) / 2.0, "x": 1.0, "y": 2.0}
    assert lg.readtb(p.logger.logdir)["avg_r"] == ([8], [15.0])


def test_multitask_decay_schedule_reference_kats():
    """test/util_tests.jl:55-86 verbatim: MultitaskDecaySchedule(10, [1,2,3]) restarts the linear decay per task; with task ids [1,2,1] the third
    block continues where task 1 stopped; m(31) == 0.1, m(0) == 1."""
    import crux_jl_amd as crux
    l = crux.LinearDecaySchedule(1.0, 0.1, 10)
    m = crux.MultitaskDecaySchedule(10, [1, 2, 3])
    for i in range(1, 11):
        assert m(i) == l(i)
    for i in range(11, 21):
        assert m(i) == l(i - 10)
    for i in range(21, 31):
        assert m(i) == l(i - 20)
    m = crux.MultitaskDecaySchedule(10, [1, 2, 1])
    for i in range(1, 11):
        assert m(i) == l(i)
    for i in range(11, 21):
        assert m(i) == l(i - 10)
    for i in range(21, 31):
        assert m(i) == l(i - 10)
    assert m(31) == 0.1 and m(0) == 1
