"""On-disk format of the reference's ExperienceBuffer dumps: BSON.jl files such as examples/il/expert_data/*.bson
(`BSON.@save path data` of a `Crux.ExperienceBuffer`, read back by `BSON.load(path)[:data]`, e.g. test/gym/solver_tests.jl:93).

Layout (public BSON spec + BSON.jl's tagging convention, observed on the reference's own files):
    { data: {tag:"struct", type:<Crux.ExperienceBuffer datatype>, data:[ <Dict Symbol=>Array as a document>, elements::Int64,
              next_ind::Int64, indices::Array{Int64}, priority_params|nothing ]}, _backrefs:[...] }
    arrays: {tag:"array", type:{tag:"datatype", name:["Core", "Float32"], params:[]}, size:[d1, d2], data:<binary, column-major>}

`load_buffer` returns the columns as numpy arrays (features, N) and, given a context, a device ExperienceBuffer holding them;
`save_buffer` writes the same structure back (round trip with this reader; the type descriptors are the ones the reference files carry).
"""
import struct

import numpy as np

_JL2NP = {"Float32": np.float32, "Float64": np.float64, "Bool": np.bool_, "Int64": np.int64, "Int32": np.int32, "UInt8": np.uint8}
_NP2JL = {np.dtype(v): k for k, v in _JL2NP.items()}


# ---------------------------------------------------------------------------------------------- reader
def _cstr(b, p):
    e = b.index(b"\x00", p)
    return b[p:e].decode("utf8"), e + 1


def _parse(b, p=0, as_list=False):
    (n,) = struct.unpack_from("<i", b, p)
    end = p + n - 1
    p += 4
    out = [] if as_list else {}
    while p < end:
        t = b[p]; p += 1
        k, p = _cstr(b, p)
        if t == 0x01:
            (v,) = struct.unpack_from("<d", b, p); p += 8
        elif t == 0x02:
            (ln,) = struct.unpack_from("<i", b, p); v = b[p + 4:p + 4 + ln - 1].decode("utf8"); p += 4 + ln
        elif t in (0x03, 0x04):
            (ln,) = struct.unpack_from("<i", b, p); v = _parse(b, p, as_list=(t == 0x04)); p += ln
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
            raise ValueError("BSON element type 0x%02x at offset %d is not used by ExperienceBuffer dumps" % (t, p))
        if as_list:
            out.append(v)
        else:
            out[k] = v
    return out


def _array(node):
    if not (isinstance(node, dict) and node.get("tag") == "array" and isinstance(node.get("data"), (bytes, bytearray))):
        return None
    name = node["type"]["name"][-1] if isinstance(node.get("type"), dict) else None
    if name not in _JL2NP:
        raise ValueError("array element type %r is not supported" % (name,))
    size = [int(x) for x in node["size"]]
    a = np.frombuffer(node["data"], dtype=_JL2NP[name])
    if a.size != int(np.prod(size)):
        raise ValueError("array payload has %d elements, size says %s" % (a.size, size))
    return a.reshape(size, order="F").copy(order="F")


def read_columns(path):
    """-> (columns: {key: ndarray (features, N)}, meta: {elements, next_ind (1-based, as stored), indices})."""
    doc = _parse(open(path, "rb").read())
    node = doc.get("data", doc)
    if not (isinstance(node, dict) and node.get("tag") == "struct" and isinstance(node.get("data"), list)):
        raise ValueError("%s does not hold a tagged ExperienceBuffer struct under :data" % path)
    fields = node["data"]
    cols = {k: _array(v) for k, v in fields[0].items() if _array(v) is not None}
    meta = {"elements": int(fields[1]), "next_ind": int(fields[2]), "indices": _array(fields[3]) if len(fields) > 3 else None,
            "priority_params": fields[4] if len(fields) > 4 else None,
            "total_count": int(fields[5]) if len(fields) > 5 and isinstance(fields[5], (int, np.integer)) else None}   # 6th field of the current struct (experience_buffer.jl:59); the shipped expert dumps predate it (5 fields)
    return cols, meta


def load_buffer(path, S=None, A=None, ctx=None, capacity=None):
    """BSON.load(path)[:data] as a device ExperienceBuffer (extra Float32 1 x N columns the library knows are kept; others are returned in .extra)."""
    from . import api
    cols, meta = read_columns(path)
    n = meta["elements"]
    disc = cols["a"].dtype == np.bool_
    S = S or api.ContinuousSpace(cols["s"].shape[0])
    A = A or (api.DiscreteSpace(cols["a"].shape[0]) if disc else api.ContinuousSpace(cols["a"].shape[0]))
    known = [k for k in cols if k in api.L.COL and k not in ("s", "a", "sp", "r", "done", "episode_end")]
    buf = api.ExperienceBuffer(S, A, capacity or n, known, ctx=ctx)
    data = {k: v[:, :n] for k, v in cols.items() if k in api.L.COL}
    data.setdefault("episode_end", np.zeros((1, n), bool))
    buf.push_(data)
    buf.extra = {k: v[:, :n] for k, v in cols.items() if k not in api.L.COL}
    buf.loaded_total_count = meta["total_count"] if meta["total_count"] is not None else n      # push_reservoir! continues from the stored count
    return buf


# ---------------------------------------------------------------------------------------------- writer
def _e_cstr(s):
    return s.encode("utf8") + b"\x00"


def _emit(v, as_list=False):
    items = enumerate(v) if as_list else v.items()
    body = b""
    for k, x in items:
        key = _e_cstr(str(k))
        if x is None:
            body += b"\x0A" + key
        elif isinstance(x, bool):
            body += b"\x08" + key + (b"\x01" if x else b"\x00")
        elif isinstance(x, (int, np.integer)):
            body += b"\x12" + key + struct.pack("<q", int(x))
        elif isinstance(x, float):
            body += b"\x01" + key + struct.pack("<d", x)
        elif isinstance(x, str):
            e = x.encode("utf8") + b"\x00"; body += b"\x02" + key + struct.pack("<i", len(e)) + e
        elif isinstance(x, (bytes, bytearray)):
            body += b"\x05" + key + struct.pack("<i", len(x)) + b"\x00" + bytes(x)
        elif isinstance(x, dict):
            body += b"\x03" + key + _emit(x)
        elif isinstance(x, (list, tuple)):
            body += b"\x04" + key + _emit(list(x), as_list=True)
        else:
            raise TypeError("cannot encode %r" % type(x))
    return struct.pack("<i", len(body) + 5) + body + b"\x00"


def _dt(name):
    return {"tag": "datatype", "params": [], "name": ["Core", name]}


def _arr(a):
    a = np.asarray(a)
    return {"tag": "array", "type": _dt(_NP2JL[a.dtype]), "size": [int(x) for x in a.shape], "data": np.asfortranarray(a).tobytes(order="F")}


def _typevar(ref_name):
    return {"tag": "struct", "type": {"tag": "backref", "ref": 1}, "data": [{"tag": "symbol", "name": ref_name}, {"tag": "jl_bottom_type"}, _dt("Any")]}


def save_buffer(buf, path, extra=None):
    """BSON.@save path data for an ExperienceBuffer: the struct layout of the reference's own dumps."""
    n = len(buf)
    cols = {k: buf[k] for k in buf.keys()}
    for k, v in (extra or getattr(buf, "extra", None) or {}).items():
        cols[k] = np.asarray(v)
    ebtype = {"tag": "datatype", "name": ["Crux", "ExperienceBuffer"],
              "params": [{"tag": "unionall", "var": {"tag": "backref", "ref": 2},
                          "body": {"tag": "unionall", "var": {"tag": "backref", "ref": 3},
                                   "body": {"tag": "datatype", "name": ["Core", "Array"], "params": [{"tag": "backref", "ref": 2}, {"tag": "backref", "ref": 3}]}}}]}
    doc = {"data": {"tag": "struct", "type": ebtype,
                    "data": [{k: _arr(v) for k, v in cols.items()}, int(n), int(buf.next_ind), _arr(np.zeros(0, np.int64)), None,
                             int(getattr(buf, "total_count", n))]},      # data, elements, next_ind, indices, priority_params, total_count (experience_buffer.jl:53-60)
           "_backrefs": [_dt("TypeVar"), _typevar("T"), _typevar("N")]}
    with open(path, "wb") as f:
        f.write(_emit(doc))
