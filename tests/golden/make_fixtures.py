#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own recorded-transition data files.

Run in the build container only (needs /root/reference):
    python tests/golden/make_fixtures.py

Inputs : /root/reference/examples/il/expert_data/{cartpole,pendulum,lunar_lander_discrete,
         half_cheetah_mujoco}.bson  -- BSON.jl dumps of Crux `ExperienceBuffer`s (data, not code).
Outputs: small .npz slices (<= 512 rows) holding the recorded columns. They pin the CartPole and
         Pendulum dynamics restated in oracle/  and give realistic obs/act
         distributions for the 8-obs/4-act and 17-obs/6-act synthetic configs.

The BSON reader below is a from-scratch ~60-line parser of the public BSON spec plus BSON.jl's
`tag` convention for arrays ({tag:"array", type:{name:[...]}, size:[...], data:<bin>}).
"""
import struct, sys, os
import numpy as np

SRC = "/root/reference/examples/il/expert_data"
OUT = os.path.dirname(os.path.abspath(__file__))


def _cstr(b, p):
    e = b.index(b"\x00", p)
    return b[p:e].decode("utf8"), e + 1


def parse_doc(b, p=0, as_list=False):
    (n,) = struct.unpack_from("<i", b, p)
    end = p + n - 1
    p += 4
    out = [] if as_list else {}
    while p < end:
        t = b[p]
        p += 1
        k, p = _cstr(b, p)
        if t == 0x01:
            (v,) = struct.unpack_from("<d", b, p); p += 8
        elif t == 0x02:
            (ln,) = struct.unpack_from("<i", b, p); v = b[p + 4:p + 4 + ln - 1].decode("utf8"); p += 4 + ln
        elif t == 0x03:
            (ln,) = struct.unpack_from("<i", b, p); v = parse_doc(b, p); p += ln
        elif t == 0x04:
            (ln,) = struct.unpack_from("<i", b, p); v = parse_doc(b, p, as_list=True); p += ln
        elif t == 0x05:
            (ln,) = struct.unpack_from("<i", b, p); v = bytes(b[p + 5:p + 5 + ln]); p += 5 + ln
        elif t == 0x08:
            v = bool(b[p]); p += 1
        elif t == 0x0A:
            v = None
        elif t == 0x10:
            (v,) = struct.unpack_from("<i", b, p); p += 4
        elif t == 0x12:
            (v,) = struct.unpack_from("<q", b, p); p += 8
        else:
            raise ValueError("bson type 0x%02x at %d" % (t, p))
        if as_list:
            out.append(v)
        else:
            out[k] = v
    return out


JL2NP = {"Float32": np.float32, "Float64": np.float64, "Bool": np.bool_, "Int64": np.int64,
         "Int32": np.int32, "UInt8": np.uint8}


def find_arrays(node, path, found):
    """Collect every BSON.jl tagged array as (path, ndarray [column-major -> numpy order='F'])."""
    if isinstance(node, dict):
        if node.get("tag") == "array" and isinstance(node.get("data"), (bytes, bytearray)):
            tname = node["type"]["name"][-1] if isinstance(node["type"], dict) else None
            if tname in JL2NP:
                dt = JL2NP[tname]
                size = [int(x) for x in node["size"]]
                a = np.frombuffer(node["data"], dtype=dt)
                if a.size == int(np.prod(size)):
                    found.append((path, a.reshape(size, order="F").copy()))
            return
        for k, v in node.items():
            find_arrays(v, path + [k], found)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            find_arrays(v, path + [i], found)


def find_symbols(node, acc):
    """Symbols appear as {tag:'symbol', name:'s'}; collect in document order."""
    if isinstance(node, dict):
        if node.get("tag") == "symbol":
            acc.append(node["name"]); return
        for v in node.values():
            find_symbols(v, acc)
    elif isinstance(node, list):
        for v in node:
            find_symbols(v, acc)


def load_buffer(fn):
    b = open(os.path.join(SRC, fn), "rb").read()
    doc = parse_doc(b)
    arrs, syms = [], []
    find_arrays(doc, [], arrs)
    find_symbols(doc, syms)
    return doc, arrs, syms


def columns(arrs):
    return {path[-1]: a for path, a in arrs if isinstance(path[-1], str) and a.ndim == 2}


def window_with_done(cols, n):
    """Pick n consecutive rows that contain at least one done=True row (if any exists)."""
    d = cols["done"][0]
    idx = np.flatnonzero(d)
    start = 0
    if idx.size:
        start = max(0, int(idx[0]) - n // 2)
    return slice(start, start + n)


if __name__ == "__main__":
    spec = {"cartpole.bson": 512, "pendulum.bson": 512, "lunar_lander_discrete.bson": 256,
            "half_cheetah_mujoco.bson": 256}
    for fn, n in spec.items():
        doc, arrs, syms = load_buffer(fn)
        cols = columns(arrs)
        sl = window_with_done(cols, n)
        out = {k: np.ascontiguousarray(v[..., sl]) for k, v in cols.items() if k != "expert_val"}
        name = fn.replace(".bson", "") + "_transitions.npz"
        np.savez_compressed(os.path.join(OUT, name), **out)
        print(name, {k: (v.dtype.str, v.shape) for k, v in out.items()},
              "done rows:", int(out["done"].sum()))
