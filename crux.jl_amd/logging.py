"""Logging side of the solve loops: `LoggerParams`, `log`, `aggregate_info`, the built-in `log_*` closures, and a TensorBoard
event-file writer/reader so `plot_learning`/`readtb`-style tooling keeps working on runs of this library.

Reference: src/logging.jl:1-111 (elapsed, LoggerParams, Base.log, aggregate_info, log_undiscounted_return, log_episode_averages, ...),
src/analysis.jl:2-13 (readtb). The event files are the format TensorBoardLogger.jl writes (third-party, not under /root/reference; restated from
the public TFRecord / tensorflow.Event proto definitions): records of
    uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)
with data = Event{ double wall_time = 1; int64 step = 2; oneof { string file_version = 3; Summary summary = 5 } },
Summary{ repeated Value value = 1 }, Value{ string tag = 1; float simple_value = 2 }.
"""
import os
import struct
import time

import numpy as np


# ---------------------------------------------------------------------------------------------- crc32c (Castagnoli), table driven
def _crc_table():
    t = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_T = _crc_table()


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _T[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- minimal protobuf
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F; n >>= 7
        out.append(b | 0x80 if n else b)
        if not n:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _event(wall, step, file_version=None, scalars=None):
    ev = _varint((1 << 3) | 1) + struct.pack("<d", wall) + _varint((2 << 3) | 0) + _varint(step)
    if file_version is not None:
        ev += _ld(3, file_version.encode())
    if scalars:
        summ = b"".join(_ld(1, _ld(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(v))) for tag, v in scalars)
        ev += _ld(5, summ)
    return ev


def _read_varint(b, p):
    n, sh = 0, 0
    while True:
        x = b[p]; p += 1
        n |= (x & 0x7F) << sh; sh += 7
        if not x & 0x80:
            return n, p


def _fields(b):
    p = 0
    while p < len(b):
        key, p = _read_varint(b, p); f, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _read_varint(b, p)
        elif wt == 1:
            v = b[p:p + 8]; p += 8
        elif wt == 5:
            v = b[p:p + 4]; p += 4
        elif wt == 2:
            ln, p = _read_varint(b, p); v = b[p:p + ln]; p += ln
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield f, wt, v


class TBLogger:
    """TBLogger(dir, tb_increment): a fresh run directory `dir`, `dir_1`, `dir_2`, ... and one events.out.tfevents.* file in it."""

    def __init__(self, logdir, increment=True):
        d, k = logdir.rstrip("/"), 0
        if increment:
            while os.path.exists(d if k == 0 else "%s_%d" % (logdir.rstrip("/"), k)):
                k += 1
            d = d if k == 0 else "%s_%d" % (logdir.rstrip("/"), k)
        os.makedirs(d, exist_ok=True)
        self.logdir = d
        self.path = os.path.join(d, "events.out.tfevents.%d.cruxhip" % int(time.time()))
        self._f = open(self.path, "ab")
        self._record(_event(time.time(), 0, file_version="brain.Event:2"))

    def _record(self, data):
        hdr = struct.pack("<Q", len(data))
        self._f.write(hdr + struct.pack("<I", _masked(hdr)) + data + struct.pack("<I", _masked(data))); self._f.flush()

    def log_value(self, tag, value, step):
        """log_value(logger, tag, v, step=i) for scalars; vectors are written as tag/1, tag/2, ... like TensorBoardLogger's preprocess."""
        v = np.asarray(value)
        if v.ndim == 0:
            self._record(_event(time.time(), int(step), scalars=[(tag, float(v))]))
        else:
            self._record(_event(time.time(), int(step), scalars=[("%s/%d" % (tag, j + 1), float(x)) for j, x in enumerate(v.ravel())]))

    def close(self):
        self._f.close()


def readtb(logdir, key=None):
    """readtb(logdir[, key]) (src/analysis.jl:2-13): {tag: (iterations, values)} over every event file in the directory; checks both CRCs."""
    hist = {}
    for fn in sorted(os.listdir(logdir)):
        if "tfevents" not in fn:
            continue
        b = open(os.path.join(logdir, fn), "rb").read(); p = 0
        while p + 12 <= len(b):
            (ln,) = struct.unpack_from("<Q", b, p)
            if struct.unpack_from("<I", b, p + 8)[0] != _masked(b[p:p + 8]):
                raise ValueError("%s: corrupt record header at %d" % (fn, p))
            data = b[p + 12:p + 12 + ln]
            if struct.unpack_from("<I", b, p + 12 + ln)[0] != _masked(data):
                raise ValueError("%s: corrupt record payload at %d" % (fn, p))
            p += 16 + ln
            step, summ = 0, None
            for f, wt, v in _fields(data):
                if f == 2 and wt == 0:
                    step = v
                elif f == 5 and wt == 2:
                    summ = v
            if summ is None:
                continue
            for f, wt, val in _fields(summ):
                if f != 1:
                    continue
                tag, x = None, None
                for f2, wt2, v2 in _fields(val):
                    if f2 == 1 and wt2 == 2:
                        tag = v2.decode()
                    elif f2 == 2 and wt2 == 5:
                        (x,) = struct.unpack("<f", v2)
                if tag is not None and x is not None:
                    it, vals = hist.setdefault(tag, ([], []))
                    it.append(step); vals.append(x)
    return hist if key is None else hist[key]


# ---------------------------------------------------------------------------------------------- LoggerParams / log
def elapsed(i, N):
    """elapsed(i::Int, N) = i % N == 0; elapsed(i::UnitRange, N) = any step of the range hits the period (src/logging.jl:1-2). Ranges are (first, last) inclusive."""
    if isinstance(i, (tuple, range, list)):
        lo, hi = (i[0], i[-1])
        return (hi // N) * N >= lo and hi >= lo
    return i % N == 0


def aggregate_info(infos):
    """aggregate_info(infos) (src/logging.jl:60-66): per key, the mean over the infos that have it."""
    keys = []
    for d in infos:
        for k in d:
            if k not in keys:
                keys.append(k)
    return {k: float(np.mean([d[k] for d in infos if k in d])) for k in keys}


def log_performance(s, name, fn, **kw):
    return {"%s/T%d" % (name, j + 1): fn(x, **kw) for j, x in enumerate(s)} if isinstance(s, (list, tuple)) else {name: fn(s, **kw)}


def log_undiscounted_return(Neps, name="undiscounted_return"):
    from . import api
    return lambda s=None, **kw: log_performance(s, name, api.undiscounted_return, Neps=Neps)


def log_discounted_return(Neps):
    from . import api
    return lambda s=None, **kw: log_performance(s, "discounted_return", api.discounted_return, Neps=Neps)


def log_failure(Neps):
    from . import api
    return lambda s=None, **kw: log_performance(s, "failure_rate", api.failure, Neps=Neps)


def log_validation_error(p, P, D_val, name="validation_error"):
    """log_validation_error(loss, D_val) (src/logging.jl:82): the loss of TrainingParams `p` on the held-out buffer, no update."""
    from . import api
    return lambda s=None, **kw: {name: api.loss_value(s.agent.pi, p, P, D_val)}


def log_exploration(policy, name=None):
    """log_exploration (src/logging.jl:84-95): eps of an eps-greedy policy, noise_std of Gaussian-noise exploration, nothing otherwise."""
    from . import api
    if isinstance(policy, api.EpsGreedyPolicy):
        return lambda i=0, **kw: {name or "eps": float(policy.eps(i))}
    if isinstance(policy, api.GaussianNoiseExplorationPolicy):
        return lambda i=0, **kw: {name or "noise_std": float(policy.sigma(i) if callable(policy.sigma) else policy.sigma)}
    return lambda **kw: {}


def _last_period_sums(solver, keys, period):
    buf = getattr(solver, "buffer", None)
    if buf is None:
        return None
    idx = buf.get_last_N_indices(period)                  # 1-based, like the reference
    ee = buf["episode_end"][0][np.asarray(idx) - 1]
    return {k: float(buf[k][0][np.asarray(idx) - 1].sum()) for k in keys}, float(ee.sum())


def log_episode_averages(keys, period):
    """log_episode_averages (src/logging.jl:99-111): sum(buffer[k][last `period` rows]) / sum(episode_end[those rows]) as avg_<k>."""
    def fn(S=None, **kw):
        r = _last_period_sums(S, keys, period)
        if r is None:
            return {}
        sums, n_ee = r
        return {"avg_%s" % k: (v / n_ee if n_ee else float("nan")) for k, v in sums.items()}
    return fn


def log_experience_sums(keys, period):
    """log_experience_sums (src/logging.jl:113-125): the plain sums, under the reference's (avg_) key names."""
    def fn(S=None, **kw):
        r = _last_period_sums(S, keys, period)
        return {} if r is None else {"avg_%s" % k: v for k, v in r[0].items()}
    return fn


class LoggerParams:
    """LoggerParams(; dir="log/", period=500, fns=[log_undiscounted_return(10), log_episode_averages([:r], period)], writeout, verbose, sampler)
    (src/logging.jl:12-25). The W&B branch is not mirrored (it errors in the reference too unless an extra package is loaded)."""

    def __init__(self, dir="log/", period=500, fns=None, writeout=None, verbose=False, sampler=None, logger=None):
        self.dir, self.period, self.verbose, self.sampler = dir, int(period), verbose, sampler
        self.logger = logger if logger is not None else TBLogger(dir, increment=True)
        self.fns = list(fns) if fns is not None else [log_undiscounted_return(10), log_episode_averages(["r"], self.period)]
        self.writeout = dict(writeout or {})


def log(p, i, *data, S=None):
    """Base.log(p::LoggerParams, i, data...; S) (src/logging.jl:29-57). `i` is a step or an inclusive (first, last) range; returns what was written."""
    if p is None:
        return None
    last = i[-1] if isinstance(i, (tuple, range, list)) else i
    for period, fn in p.writeout.items():
        if elapsed(i, period):
            fn(i=last, s=p.sampler, dir=p.dir, logger=p.logger)
    if not elapsed(i, p.period):
        return None
    dicts = list(p.fns) + list(data)
    if p.sampler is not None:
        s0 = p.sampler[0] if isinstance(p.sampler, (list, tuple)) else p.sampler
        pe = getattr(s0.agent, "pi_explore", None)
        if pe is not None:
            dicts.append(log_exploration(pe))
    written = {}
    for d in dicts:
        d = d(s=p.sampler, i=last, S=S) if callable(d) else d
        for k, v in d.items():
            p.logger.log_value(str(k), v, step=last); written[str(k)] = v
    if p.verbose:
        print("Step: %d" % last + "".join(", %s: %s" % kv for kv in written.items()))
    return written
