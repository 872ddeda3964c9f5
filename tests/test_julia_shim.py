"""Mechanical check of the Julia binding (julia/CruxHIP.jl) against the C ABI (include/cruxhip.h) -- VERDICT r4 next-round #2c. There is no Julia toolchain in the build image,
so the shim cannot be executed; what CAN be checked without one is everything a `ccall` gets wrong silently:

  * every `ccall((:crux_x, LIB), Ret, (T1, ..., Tn), a1, ..., an)` names an exported prototype, with n == the prototype's arity == the number of arguments passed,
    a return type and argument C types that match the prototype's (Int32 <-> int32_t, Ptr{Float32} <-> [const] float*, Ref{TrainCfg} <-> const crux_train_cfg*, ...);
  * the isbits struct mirrors (RolloutCfg, TrainCfg, Lagrange) have the header's fields in the header's order with the header's types, and the sizes the shim asserts;
    the same for the ctypes Structures of crux.jl_amd/_lib.py (the tested twin);
  * the constant tables (COL, HEAD_*, LOSS_*, INFO_N, NCOLS) carry the header's enum values;
  * the entry points bench.py times and the seams SURVEY 8(a)/(b) list are bound (the chained asynchronous epochs, the cost fills, episodes!' metrics, the caller-stepped seam).
"""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cruxhip.h")
SHIM = os.path.join(ROOT, "julia", "CruxHIP.jl")


# ---------------------------------------------------------------------------------------------------------------- the header
def _strip_c_comments(t):
    return re.sub(r"//[^\n]*", " ", re.sub(r"/\*.*?\*/", " ", t, flags=re.S))


def _split_top(s, sep=","):
    out, depth, cur, q = [], 0, "", None
    for ch in s:
        if q:
            cur += ch
            if ch == q:
                q = None
            continue
        if ch == '"':
            q = ch; cur += ch; continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _ctype(decl):
    """'const crux_mlp* const* actors' -> ('crux_mlp', 2); 'float gamma' -> ('float', 0); 'void' -> None"""
    d = decl.strip()
    if d == "void" or not d:
        return None
    d = re.sub(r"\[[^\]]*\]", "*", d)                  # T x[4] decays to T*
    depth = d.count("*")
    toks = [t for t in re.sub(r"[*]", " ", d).split() if t not in ("const", "struct", "volatile")]
    base_types = ("int32_t", "int64_t", "uint8_t", "uint32_t", "uint64_t", "float", "double", "char", "void", "int", "unsigned", "long")
    base = toks[0]
    if len(toks) >= 2 and toks[0] in ("unsigned", "long") and toks[1] in base_types:
        base = toks[0] + " " + toks[1]
    assert base in base_types or base.startswith("crux_"), decl
    return base, depth


def header_prototypes():
    t = _strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][A-Za-z_0-9]*\s*\**)\s*(crux_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", t, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ps = [_ctype(p) for p in _split_top(params)]
        protos[name] = (_ctype(ret + " r"), [p for p in ps if p is not None])
    return protos


def header_structs():
    t = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(crux_[a-z_0-9]+)\s*;", t, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            first, *rest = _split_top(stmt)
            base, depth = _ctype(first)
            assert depth == 0, stmt
            fields.append((first.split()[-1], base))
            for r in rest:
                fields.append((r.strip(), base))
        out[m.group(2)] = fields
    return out


def header_enums():
    t = _strip_c_comments(open(HEADER).read())
    vals = {}
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", t, flags=re.S):
        nxt = 0
        for item in _split_top(m.group(1)):
            if not item:
                continue
            if "=" in item:
                k, v = item.split("=", 1); nxt = int(eval(v.strip(), {}, dict(vals))); k = k.strip()
            else:
                k = item.strip()
            vals[k] = nxt; nxt += 1
    for m in re.finditer(r"#define\s+(CRUX_[A-Z_]+)\s+(-?\d+)", t):
        vals[m.group(1)] = int(m.group(2))
    return vals


C_SIZES = {"int32_t": (4, 4), "int64_t": (8, 8), "uint64_t": (8, 8), "uint32_t": (4, 4), "uint8_t": (1, 1), "float": (4, 4), "double": (8, 8)}


def c_layout(fields):
    off, al = 0, 1
    for _, b in fields:
        sz, a = C_SIZES[b]
        off = (off + a - 1) // a * a + sz; al = max(al, a)
    return (off + al - 1) // al * al


# ---------------------------------------------------------------------------------------------------------------- the shim
JL_SCALAR = {"Int32": "int32_t", "Int64": "int64_t", "UInt64": "uint64_t", "UInt32": "uint32_t", "UInt8": "uint8_t", "Float32": "float", "Float64": "double", "Bool": "uint8_t"}
JL_STRUCT = {"RolloutCfg": "crux_rollout_cfg", "TrainCfg": "crux_train_cfg", "Lagrange": "crux_lagrange"}


def _jtype(t):
    """Julia ccall type -> (C base or 'void' = any, pointer depth)"""
    t = t.strip()
    if t == "Cstring":
        return "char", 1
    if t in JL_SCALAR:
        return JL_SCALAR[t], 0
    m = re.fullmatch(r"(Ptr|Ref)\{(.*)\}", t)
    assert m, "unknown ccall type %r" % t
    inner = m.group(2).strip()
    if inner == "Cvoid":
        return "void", 1
    if inner in JL_STRUCT:
        return JL_STRUCT[inner], 1
    b, d = _jtype(inner)
    return b, d + 1


def shim_ccalls():
    src = open(SHIM).read()
    src = "\n".join(ln.split("#", 1)[0] if '"' not in ln.split("#", 1)[0] or ln.split("#", 1)[0].count('"') % 2 == 0 else ln for ln in src.splitlines())      # drop trailing comments
    calls = []
    for m in re.finditer(r"ccall\(\(:(crux_[a-z_0-9]+),\s*LIB\),", src):
        i = m.end(); depth = 1; j = i
        while depth:                                                   # the matching parenthesis of ccall(
            ch = src[j]
            depth += ch in "([{"; depth -= ch in ")]}"; j += 1
        parts = _split_top(src[i:j - 1])
        ret, types, args = parts[0], parts[1], parts[2:]
        assert types.startswith("(") and types.endswith(")"), (m.group(1), types)
        tl = [x for x in _split_top(types[1:-1]) if x]
        calls.append((m.group(1), ret.strip(), tl, args, src.count("\n", 0, m.start()) + 1))
    return calls


def shim_structs():
    src = open(SHIM).read()
    out = {}
    for name in JL_STRUCT:
        m = re.search(r"(?:mutable\s+)?struct\s+%s\s*\n(.*?)\nend" % name, src, flags=re.S)
        assert m, name
        fields = []
        for stmt in re.split(r"[;\n]", m.group(1)):
            stmt = stmt.split("#")[0].strip()
            if stmt:
                n, t = stmt.split("::"); fields.append((n.strip(), JL_SCALAR[t.strip()]))
        out[name] = fields
    return out


def _compatible(j, c):
    (jb, jd), (cb, cd) = j, c
    if jd != cd:
        return False
    if jd == 0:
        return jb == cb or (jb == "int32_t" and cb == "int")
    return jb == "void" or cb == "void" or jb == cb or (jd >= 2)      # Ptr{Ptr{Cvoid}} / Ref{Ptr{Cvoid}}: any pointer to pointers


# ---------------------------------------------------------------------------------------------------------------- tests
def test_every_ccall_matches_its_prototype():
    protos = header_prototypes(); calls = shim_ccalls()
    assert len(protos) >= 135 and len(calls) >= 90 and len({c[0] for c in calls}) >= 80
    bad = []
    for name, ret, types, args, line in calls:
        if name not in protos:
            bad.append("%s (line %d): not declared in include/cruxhip.h" % (name, line)); continue
        cret, cparams = protos[name]
        if len(types) != len(cparams) or len(args) != len(cparams):
            bad.append("%s (line %d): %d ccall types, %d arguments, the prototype has %d parameters" % (name, line, len(types), len(args), len(cparams))); continue
        if not _compatible(_jtype(ret), cret):
            bad.append("%s (line %d): returns %s, the prototype %r" % (name, line, ret, cret))
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            if not _compatible(_jtype(jt), ct):
                bad.append("%s (line %d): argument %d is %s, the prototype has %s%s" % (name, line, k + 1, jt, ct[0], "*" * ct[1]))
    assert not bad, "\n".join(bad)


def test_struct_mirrors_have_the_headers_layout():
    hs, js = header_structs(), shim_structs()
    from crux_jl_amd import _lib as L
    ct = {"crux_rollout_cfg": L.RolloutCfg, "crux_train_cfg": L.TrainCfg, "crux_lagrange": L.Lagrange}
    ctmap = {C.c_int32: "int32_t", C.c_int64: "int64_t", C.c_uint64: "uint64_t", C.c_uint32: "uint32_t", C.c_float: "float", C.c_double: "double"}
    src = open(SHIM).read()
    for jname, cname in JL_STRUCT.items():
        want = [b for _, b in hs[cname]]
        assert [b for _, b in js[jname]] == want, "%s: field types %r, header %r" % (jname, js[jname], hs[cname])
        assert len(js[jname]) == len(hs[cname])
        assert [ctmap[t] for _, t in ct[cname]._fields_] == want, cname
        assert [n for n, _ in ct[cname]._fields_] == [n for n, _ in hs[cname]], cname       # the ctypes twin carries the header's field names too
        assert C.sizeof(ct[cname]) == c_layout(hs[cname])
    m = re.search(r"@assert sizeof\(RolloutCfg\) == (\d+) && sizeof\(TrainCfg\) == (\d+)", src)
    assert (int(m.group(1)), int(m.group(2))) == (c_layout(hs["crux_rollout_cfg"]), c_layout(hs["crux_train_cfg"])) == (72, 64)
    assert c_layout(hs["crux_lagrange"]) == 64 and "crux_lagrange (64 bytes)" in src


def test_constant_tables_carry_the_headers_values():
    en = header_enums(); src = open(SHIM).read()
    col = dict(re.findall(r":(\w+) => (\d+)", re.search(r"const COL = Dict\((.*?)\)\n", src, flags=re.S).group(1)))
    assert len(col) == en["CRUX_NCOLS"] == int(re.search(r"const NCOLS = (\d+)", src).group(1))
    for k, v in col.items():
        assert en["CRUX_COL_" + k.upper()] == int(v), k
    assert int(re.search(r"const INFO_N = (\d+)", src).group(1)) == en["CRUX_INFO_N"]
    heads = re.search(r"const HEAD_CATEGORICAL, HEAD_GAUSSIAN, HEAD_GREEDY_Q, HEAD_DETERMINISTIC = (.*)", src).group(1)
    assert [int(x) for x in re.findall(r"Int32\((\d+)\)", heads)] == [en["CRUX_HEAD_CATEGORICAL"], en["CRUX_HEAD_GAUSSIAN"], en["CRUX_HEAD_GREEDY_Q"], en["CRUX_HEAD_DETERMINISTIC"]]
    losses = re.search(r"const LOSS_PPO, LOSS_VALUE_MSE, LOSS_A2C, LOSS_REINFORCE, LOSS_LOGPDF_BC, LOSS_MSE_ACTION = (.*)", src).group(1)
    assert [int(x) for x in re.findall(r"Int32\((\d+)\)", losses)] == [en[k] for k in ("CRUX_LOSS_PPO", "CRUX_LOSS_VALUE_MSE", "CRUX_LOSS_A2C", "CRUX_LOSS_REINFORCE", "CRUX_LOSS_LOGPDF_BC", "CRUX_LOSS_MSE_ACTION")]
    assert int(re.search(r"const LOSS_LAGRANGE_PPO = Int32\((\d+)\)", src).group(1)) == en["CRUX_LOSS_LAGRANGE_PPO"]
    assert int(re.search(r"const EUNSUP = Int32\((-?\d+)\)", src).group(1)) == en["CRUX_EUNSUP"]


def test_the_benchmarked_paths_and_the_seams_are_bound():
    bound = {c[0] for c in shim_ccalls()}
    need = ["crux_dqn_value_training_async", "crux_sac_epochs_async", "crux_dpg_epochs_async", "crux_peer_probe", "crux_peer_abort", "crux_peer_set_budget_ms", "crux_dqn_epochs", "crux_sac_epochs", "crux_softq_epochs", "crux_dpg_epochs",
            "crux_fill_gae_rows_keys", "crux_fill_returns_rows_keys", "crux_first_episode_metrics", "crux_buffer_shuffle", "crux_policy_gradient_training_multi", "crux_per_get",
            "crux_policy_explore", "crux_steps_push", "crux_rollout", "crux_policy_gradient_training", "crux_batch_train", "crux_batch_train_lagrange", "crux_peer_export", "crux_peer_attach",
            "crux_peer_set_sync_every", "crux_peer_set_timeout_ms", "crux_per_sample", "crux_uniform_sample", "crux_per_update", "crux_whiten", "crux_fill_gae_rows", "crux_fill_returns_rows"]
    assert not [n for n in need if n not in bound], [n for n in need if n not in bound]
    # value_training goes through the chained asynchronous calls (the 46 / 99 us path), not the per-epoch synchronising ones
    src = open(SHIM).read()
    body = src[src.index("function Crux.value_training(𝒮::Crux.OffPolicySolver, 𝒟::HipBuffer, γ"):]
    body = body[:body.index("\nend\n")]
    assert "crux_dqn_value_training_async" in body and "crux_sac_epochs_async" in body and "dqn_epoch!(" not in body and "sac_epoch!(" not in body


def test_the_shim_parses_as_balanced_julia():
    """no Julia here: at least every bracket closes and every block opener has its `end` (julia/check_syntax.jl is the real parser where a julia binary exists)"""
    src = open(SHIM).read()
    code = []
    for ln in src.splitlines():
        out, q, i = "", False, 0
        while i < len(ln):
            ch = ln[i]
            if ch == '"' and (i == 0 or ln[i - 1] != "\\"):
                q = not q
            elif ch == "#" and not q:
                break
            out += " " if q and ch != '"' else ch
            i += 1
        code.append(out)
    text = "\n".join(code)
    text = re.sub(r'"""(.|\n)*?"""', '""', "\n".join(src.splitlines())) if False else text
    depth = {"(": 0, "[": 0, "{": 0}
    pair = {")": "(", "]": "[", "}": "{"}
    for ch in text:
        if ch in depth:
            depth[ch] += 1
        elif ch in pair:
            depth[pair[ch]] -= 1
            assert depth[pair[ch]] >= 0
    assert depth == {"(": 0, "[": 0, "{": 0}, depth
