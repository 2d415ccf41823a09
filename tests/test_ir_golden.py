"""The expression IR pinned three ways: tests/golden/ir_golden.json holds byte strings hand-assembled from
include/dfdb_ir.h (tests/golden/make_ir_golden.py, which never imports dfdb/ir.py) with the answers Julia gives.
CPU: the header's numbers == dfdb/ir.py's == the Julia shim's tables; dfdb/ir.py emits exactly the golden bytes; the oracle evaluates
the golden bytes to the golden answers.  GPU: the engine evaluates the same bytes to the same answers."""
import ctypes as C
import json
import os
import re
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = json.load(open(os.path.join(HERE, "golden", "ir_golden.json")))
NP_OF = {"Int64": np.int64, "Int8": np.int8, "UInt8": np.uint8, "UInt64": np.uint64, "Float64": np.float64, "Float32": np.float32, "Bool": np.bool_}


def _fl(v):
    return {"NaN": float("nan"), "Inf": float("inf"), "-Inf": float("-inf")}.get(v, v) if isinstance(v, str) else v


def _columns():
    t = G["table"]
    return {"a": np.array(t["a"], np.int64), "x": np.array([_fl(v) for v in t["x"]], np.float64), "s": list(t["s"]),
            "m": np.ma.masked_array(np.array([0 if v is None else v for v in t["m"]], np.int64), mask=[v is None for v in t["m"]]),
            "u": np.array(t["u"], np.uint8),
            # round 3: the corner columns (ordinals 5..11)
            "i8": np.array(t["i8"], np.int8), "w": np.array(t["w"], np.uint64), "z": np.array([_fl(v) for v in t["z"]], np.float64),
            "f": np.array([_fl(v) for v in t["f"]], np.float32), "b": np.array(t["b"], np.bool_),
            "mb": np.ma.masked_array(np.array([bool(v) for v in t["mb"]], np.bool_), mask=[v is None for v in t["mb"]]),
            "big": np.array(t["big"], np.int64), "us": list(t["us"]), "ns": list(t["ns"])}


def _check(case, got):
    want = case["expect"]
    base = case["type"].replace("Missing(", "").replace(")", "")
    if case["type"].startswith("Missing("):
        assert isinstance(got, np.ma.MaskedArray), case["name"]
        miss = [v == "missing" for v in want]
        assert np.ma.getmaskarray(got).tolist() == miss, case["name"]
        assert [int(v) for v, m in zip(np.asarray(got.data), miss) if not m] == [v for v in want if v != "missing"], case["name"]
        return
    assert not isinstance(got, np.ma.MaskedArray), case["name"]
    assert got.dtype == np.dtype(NP_OF[base]), (case["name"], got.dtype)
    w = np.array([_fl(v) for v in want], NP_OF[base])
    assert np.array_equal(got, w, equal_nan=got.dtype.kind == "f"), (case["name"], got, w)
    if got.dtype.kind == "f":                      # -0.0 == 0.0 for array_equal: the sign of a zero is part of the answer (min / max)
        nn = ~np.isnan(w)
        assert np.array_equal(np.signbit(got[nn]), np.signbit(w[nn])), (case["name"], got, w)


def test_header_python_and_julia_tables_agree():
    from dfdb import ir
    ops = G["opcodes"]
    hdr = open(os.path.join(ROOT, "include", "dfdb_ir.h")).read()
    assert {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define\s+(DFIR_\w+)\s+(0x[0-9a-fA-F]+)", hdr)} == ops     # the JSON is current
    for name, val in ops.items():
        assert getattr(ir, name[len("DFIR_"):]) == val, name                                        # dfdb/ir.py carries the header's numbers
    for name, val in G["dtypes"].items():
        short = {"DFDB_STRING": "STRING", "DFDB_BOOL": "BOOL", "DFDB_NULLABLE": "NULLABLE", "DFDB_DTYPE_MASK": "DTYPE_MASK"}.get(name, name[len("DFDB_"):])
        assert getattr(ir, short) == val, name
    # the Julia shim: OPS / UNARY / the literal opcodes it writes
    jl = open(os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "DataFrameDBsAMD.jl")).read()
    want = {"(+)": "ADD", "(-)": "SUB", "(*)": "MUL", "(/)": "DIV", "div": "IDIV", "(÷)": "IDIV", "rem": "REM", "(%)": "REM", "mod": "MOD", "min": "MIN", "max": "MAX",
            "(==)": "EQ", "(!=)": "NE", "(<)": "LT", "(<=)": "LE", "(>)": "GT", "(>=)": "GE", "(&)": "AND", "(|)": "OR", "xor": "XOR",
            "startswith": "STARTSWITH", "endswith": "ENDSWITH", "coalesce": "COALESCE"}
    ops_src = jl[jl.index("const OPS = Dict"):jl.index("const UNARY")]
    found = dict(re.findall(r"(\([^()\s]+\)|\w+)\s*=>\s*(0x[0-9a-f]+)", ops_src))
    assert {k: int(v, 16) for k, v in found.items()} == {k: ops["DFIR_" + v] for k, v in want.items()}
    un_src = jl[jl.index("const UNARY"):jl.index("\n", jl.index("const UNARY"))]
    found = dict(re.findall(r"(\([^()\s]+\)|\w+)\s*=>\s*(0x[0-9a-f]+)", un_src))
    assert {k: int(v, 16) for k, v in found.items()} == {"(-)": ops["DFIR_NEG"], "abs": ops["DFIR_ABS"], "(!)": ops["DFIR_NOT"], "ismissing": ops["DFIR_ISMISSING"], "sizeof": ops["DFIR_SIZEOF"]}
    assert re.search(r"emit_col\(io, ord::Integer\) = \(write\(io, 0x01\)", jl) and "write(io, 0x02); write(io, DT[T])" in jl and "write(io, 0x03)" in jl
    assert "0x40))" in jl and "0x50, DT[Float64]" in jl                                            # in.() and Float64()
    dt_src = jl[jl.index("const DT = Dict"):]
    dt_src = dt_src[:dt_src.index(")\n") + 1]
    jdt = dict(re.findall(r"(\w+)\s*=>\s*(\d+)", dt_src))
    for jname, dname in (("Int8", "I8"), ("Int16", "I16"), ("Int32", "I32"), ("Int64", "I64"), ("UInt8", "U8"), ("UInt16", "U16"), ("UInt32", "U32"),
                         ("UInt64", "U64"), ("Float32", "F32"), ("Float64", "F64"), ("Bool", "BOOL"), ("String", "STRING")):
        assert int(jdt[jname]) == G["dtypes"]["DFDB_" + dname], jname


def test_ir_py_emits_the_golden_bytes():
    from dfdb import ir
    env = {"ir": ir, "A": ir.col(0), "Xf": ir.col(1), "St": ir.col(2), "Mi": ir.col(3), "Uc": ir.col(4),
           "I8c": ir.col(5), "Wc": ir.col(6), "Zf": ir.col(7), "Ff": ir.col(8), "Bc": ir.col(9), "Mb": ir.col(10), "Big": ir.col(11), "Us": ir.col(12), "Ns": ir.col(13)}
    for c in G["cases"]:
        e = eval(c["ir_py"], env)
        assert e.to_ir().hex() == c["hex"], (c["name"], e.to_ir().hex(), c["hex"])


def test_isin_over_strings_is_the_spelled_out_any():
    """in.(s, Ref(["a", "b"])) has no opcode of its own: both front ends lower it to (s == "a") | (s == "b") (three-valued like Base.in)"""
    from dfdb import ir
    St = ir.col(2)
    assert ir.isin(St, ["sony", "dell", "é"]).to_ir() == ((St == "sony") | (St == "dell") | (St == "é")).to_ir()
    assert ir.isin(St, ["sony"]).to_ir() == (St == "sony").to_ir()
    assert ir.isin(ir.col(0), [1, 2]).to_ir()[-1] == ir.IN_SET                        # numbers keep DFIR_IN_SET
    jl = open(os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "DataFrameDBsAMD.jl")).read()
    assert "Base.in(a::Tr, s::AbstractVector{<:AbstractString})" in jl and "[a == v for v in s]" in jl


def test_oracle_evaluates_golden_bytes(oracle):
    cols = _columns()
    t = oracle.Table(block_size=G["table"]["block_size"])
    for k, v in cols.items():
        if isinstance(v, np.ma.MaskedArray):
            t.add_column(k, np.ascontiguousarray(v.filled(0)), missing=np.ma.getmaskarray(v))
        else:
            t.add_column(k, v)
    for c in G["cases"]:
        v = t.view().set_projection([("k", bytes.fromhex(c["hex"]))])
        if c["expect"] == "DivideError":
            with pytest.raises(ZeroDivisionError):
                v.materialize()
        elif c["expect"] == "InexactError":
            with pytest.raises(ValueError, match="InexactError"):
                v.materialize()
        else:
            _check(c, v.materialize()[0])
        # a Bool-typed case is also a selection function: the selected rows are where the answer is true
        if c["type"] == "Bool" and isinstance(c["expect"], list):
            sel = t.view().add_predicate(bytes.fromhex(c["hex"])).select_indices()
            assert sel.tolist() == [i + 1 for i, b in enumerate(c["expect"]) if b], c["name"]


@pytest.mark.gpu
def test_engine_evaluates_golden_bytes(dfdb_mod, ctx):
    from dfdb import _native as N, api
    L = N.load()
    t = dfdb_mod.DFTable.from_columns(_columns(), block_size=G["table"]["block_size"])

    def query(proj_hex=None, pred_hex=None):
        q = api._Query.__new__(api._Query)
        q._h = C.c_void_p()
        N.check(L.dfdb_query_new(t._h, C.byref(q._h)))
        if pred_hex:
            b = bytes.fromhex(pred_hex)
            N.check(L.dfdb_query_add_predicate(q._h, b, len(b)))
        if proj_hex:
            b = bytes.fromhex(proj_hex)
            buf = C.create_string_buffer(b, len(b))
            N.check(L.dfdb_query_set_projection(q._h, 1, (C.c_char_p * 1)(b"k"), (C.c_void_p * 1)(C.cast(buf, C.c_void_p)), (C.c_size_t * 1)(len(b))))
            q.view = types.SimpleNamespace(projection=[0])
        else:
            q.view = types.SimpleNamespace(projection=[0] * t.ncols)
        return q

    for c in G["cases"]:
        q = query(proj_hex=c["hex"])
        if c["expect"] == "DivideError":
            with pytest.raises(ZeroDivisionError):
                q.materialize()
        elif c["expect"] == "InexactError":
            with pytest.raises(ValueError, match="InexactError"):
                q.materialize()
        else:
            _check(c, q.materialize()[0])
        if c["type"] == "Bool" and isinstance(c["expect"], list):
            assert query(pred_hex=c["hex"]).indices().tolist() == [i + 1 for i, b in enumerate(c["expect"]) if b], c["name"]


def _shim_emit_const(v, T):
    """Python transcription of `emit_const(io, v::T)` of julia/DataFrameDBsAMD.jl, branch for branch: opcode 0x02, DT[T], then Float64 as it is,
    Float32 followed by a zero UInt32, every integer and Bool as `v % Int64` (the low 64 bits: wraps, never throws)."""
    import struct
    DT = {"Int8": 1, "Int16": 2, "Int32": 3, "Int64": 4, "UInt8": 5, "UInt16": 6, "UInt32": 7, "UInt64": 8, "Float32": 9, "Float64": 10, "Bool": 11}
    out = bytes([0x02, DT[T]])
    if T == "Float64":
        return out + struct.pack("<d", v)
    if T == "Float32":
        return out + struct.pack("<f", v) + struct.pack("<I", 0)
    return out + struct.pack("<q", ((int(v) + 2 ** 63) % 2 ** 64) - 2 ** 63)


def test_shim_constant_emitter_transcription():
    """the shim emitted integer constants as `Int64(v) % Int64`, which throws InexactError for a UInt64 >= 2^63 before the `%`.
    The fixed form (`v % Int64`) is pinned here without a Julia: its transcription must give dfdb/ir.py's bytes and the hand-assembled golden bytes
    for UInt64 constants above typemax(Int64), negative integers of every width, Bool, Float32 and Float64."""
    from dfdb import ir
    jl = open(os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "DataFrameDBsAMD.jl")).read()
    body = jl[jl.index("function emit_const(io, v::T) where"):]
    body = body[:body.index("\nend") + 4]
    assert "write(io, v % Int64)" in body and "Int64(v)" not in body.split("#")[0].replace("# ", ""), body
    code = "\n".join(l.split("#")[0] for l in body.splitlines())
    assert "Int64(v)" not in code and "T == Float64 ? write(io, v)" in code and "write(io, UInt32(0))" in code
    DTN = {"Int8": ir.I8, "Int16": ir.I16, "Int32": ir.I32, "Int64": ir.I64, "UInt8": ir.U8, "UInt16": ir.U16, "UInt32": ir.U32, "UInt64": ir.U64,
           "Float32": ir.F32, "Float64": ir.F64, "Bool": ir.BOOL}
    samples = [(2 ** 63, "UInt64"), (2 ** 63 + 5, "UInt64"), (2 ** 64 - 1, "UInt64"), (0, "UInt64"), (255, "UInt8"), (65535, "UInt16"), (2 ** 32 - 1, "UInt32"),
               (-1, "Int8"), (-128, "Int8"), (-32768, "Int16"), (-2 ** 31, "Int32"), (-2 ** 63, "Int64"), (2 ** 63 - 1, "Int64"), (-1, "Int64"),
               (True, "Bool"), (False, "Bool"), (0.1, "Float64"), (-0.0, "Float64"), (float("inf"), "Float64"), (0.1, "Float32"), (-2.5, "Float32")]
    for v, T in samples:
        assert _shim_emit_const(v, T) == ir.const(v, DTN[T]).to_ir(), (v, T)
    # the hand-assembled golden bytes of the constants the cases use
    by_name = {c["name"]: bytes.fromhex(c["hex"]) for c in G["cases"]}
    assert _shim_emit_const(2 ** 63, "UInt64") in by_name["uint64_const_2p63"]
    assert _shim_emit_const(2 ** 63 - 1, "Int64") in by_name["uint64_above_typemax_int64"]
    assert _shim_emit_const(-1, "Int8") in by_name["int8_typemin_idiv_minus1_divide_error"]
    assert _shim_emit_const(-1, "Int64") in by_name["typemin_idiv_minus1_divide_error"]
    assert _shim_emit_const(True, "Bool") in by_name["xor_missing_is_missing"] and _shim_emit_const(-0.0, "Float64") in by_name["max_orders_plus_zero_last"]
    # integer sets are Int64 in the IR: the shim refuses unsigned members above typemax(Int64) with Unsupported (stock path) instead of throwing InexactError
    setfn = jl[jl.index("function emit_const(io, v::AbstractVector{T})"):]
    setfn = setfn[:setfn.index("\nend") + 4]
    assert "x > typemax(Int64)" in setfn and "throw(Unsupported(" in setfn and "convert(E, x)" in setfn


# ---------------------------------------------------------------- the shim's walk over a closure's lowered code (round 4)
def _shim_walk(code, args, ops):
    """Python transcription of DataFrameDBsAMD.walk / select_bool over a miniature of Core.CodeInfo: statements are ("call", fname, operands...),
    ("gotoifnot", cond, dest), ("goto", dest), ("return", value), ("assign", slot, stmt); operands are ("ssa", n), ("slot", n) or literals.  A traced value
    is `bytes` (postfix IR), exactly the shim's Tr."""
    G_ = G  # noqa: F841
    from dfdb import ir

    def const(v):
        return ir.const(v).to_ir()

    def tr(v):
        return v if isinstance(v, bytes) else const(v)

    def call(fn, av):
        if fn == "!":
            return av[0] + bytes([ops["!"]]) if isinstance(av[0], bytes) else (not av[0])
        a, b = av
        if not isinstance(a, bytes) and not isinstance(b, bytes):
            return {">": a > b, "<": a < b, "==": a == b, "|": a | b, "&": a & b}[fn]
        return tr(a) + tr(b) + bytes([ops[fn]])

    def select_bool(c, t, e):
        tb, eb = isinstance(t, bool), isinstance(e, bool)
        if tb and eb:
            return t if t == e else (c if t else call("!", [c]))
        if not tb and eb:
            return call("|", [call("!", [c]), t]) if e else call("&", [c, t])
        if tb and not eb:
            return call("|", [c, e]) if t else call("&", [call("!", [c]), e])
        return call("|", [call("&", [c, t]), call("&", [call("!", [c]), e])])

    def value(x, ssa, slots):
        if isinstance(x, tuple) and x[0] == "ssa":
            return ssa[x[1]]
        if isinstance(x, tuple) and x[0] == "slot":
            return slots[x[1]]
        if isinstance(x, tuple) and x[0] == "call":
            return call(x[1], [value(a, ssa, slots) for a in x[2:]])
        return x

    def walk(pc, ssa, slots):
        while True:
            st = code[pc - 1]
            if st[0] == "return":
                return value(st[1], ssa, slots)
            if st[0] == "goto":
                assert st[1] > pc
                pc = st[1]
            elif st[0] == "gotoifnot":
                c = value(st[1], ssa, slots)
                if isinstance(c, bool):
                    pc = pc + 1 if c else st[2]
                else:
                    t = walk(pc + 1, dict(ssa), dict(slots))
                    e = walk(st[2], dict(ssa), dict(slots))
                    return select_bool(c, t, e)
            elif st[0] == "assign":
                slots[st[1]] = value(st[2], ssa, slots)
                ssa[pc] = slots[st[1]]
                pc += 1
            else:
                ssa[pc] = value(st, ssa, slots)
                pc += 1
    return walk(1, {}, {i + 2: a for i, a in enumerate(args)})


def test_shim_lowered_code_walk_transcription():
    """the tracer could not lower the reference's OWN test closure `(a)->65>a>34` (test/selection.jl:53): a chained comparison is
    `&&`, which needs a real Bool out of a traced value.  The shim now walks the closure's lowered code (lower_closure / walk / select_bool).  No Julia here:
    the walk is transcribed into Python above and run over the lowered code Julia produces for three closures; it must emit exactly the hand-assembled
    golden bytes (which the oracle and the engine evaluate to Julia's answers), and the shim's source must hold the rules the transcription follows."""
    from dfdb import ir
    jl = open(os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "DataFrameDBsAMD.jl")).read()
    ops = {}
    for name, key in ((">", "(>)"), ("<", "(<)"), ("==", "(==)"), ("&", "(&)"), ("|", "(|)"), ("!", "(!)")):
        m = re.search(re.escape(key) + r"\s*=>\s*0x([0-9a-fA-F]{2})", jl)
        assert m, key
        ops[name] = int(m.group(1), 16)
    by_name = {c["name"]: bytes.fromhex(c["hex"]) for c in G["cases"]}
    a = ir.col(0).to_ir()
    # a -> 3 > a > -2        %1 = 3 > a ; goto #5 if not %1 ; %3 = a > -2 ; return %3 ; return false
    code = [("call", ">", 3, ("slot", 2)), ("gotoifnot", ("ssa", 1), 5), ("call", ">", ("slot", 2), -2), ("return", ("ssa", 3)), ("return", False)]
    assert _shim_walk(code, [a], ops) == by_name["closure_chained_comparison"]
    # a -> a < -2 || a > 5   %1 = a < -2 ; goto #4 if not %1 ; return true ; %4 = a > 5 ; return %4
    code = [("call", "<", ("slot", 2), -2), ("gotoifnot", ("ssa", 1), 4), ("return", True), ("call", ">", ("slot", 2), 5), ("return", ("ssa", 4))]
    assert _shim_walk(code, [a], ops) == by_name["closure_short_circuit_or"]
    # a -> (a > -2 && a < 5) | (a == 10)
    #   %1 = a > -2 ; goto #5 if not %1 ; @_3 = a < 5 ; goto #6 ; @_3 = false ; %6 = @_3 ; %7 = a == 10 ; %8 = %6 | %7 ; return %8
    code = [("call", ">", ("slot", 2), -2), ("gotoifnot", ("ssa", 1), 5), ("assign", 3, ("call", "<", ("slot", 2), 5)), ("goto", 6), ("assign", 3, False),
            ("slot", 3), ("call", "==", ("slot", 2), 10), ("call", "|", ("ssa", 6), ("ssa", 7)), ("return", ("ssa", 8))]
    assert _shim_walk(code, [a], ops) == by_name["closure_and_then_or_through_a_join"]
    # the reference's own closure: (a) -> 65 > a > 34
    code = [("call", ">", 65, ("slot", 2)), ("gotoifnot", ("ssa", 1), 5), ("call", ">", ("slot", 2), 34), ("return", ("ssa", 3)), ("return", False)]
    assert _shim_walk(code, [a], ops) == ((ir.const(65) > ir.col(0)) & (ir.col(0) > 34)).to_ir()
    # the rules, in the shim's own text
    for needle in ("function lower_closure(f, args::Vector{Any})", "Base.code_lowered(f)", "st isa Core.GotoIfNot", "select_bool(c::Tr, t::Tr, e::Bool) = e ? (!c | t) : (c & t)",
                   "select_bool(c::Tr, t::Bool, e::Tr) = t ? (c | e) : (!c & e)", "select_bool(c::Tr, t::Tr, e::Tr) = (c & t) | (!c & e)",
                   "const RAISING_OPS = (0x14, 0x15, 0x16, 0x50)", "(may_raise(t) || may_raise(e)) && throw(Unsupported(", "st.dest > pc || throw(Unsupported(",
                   "return lower_closure(f, collect(Any, args))", "@warn \"DataFrameDBsAMD: $(f) falls back to the stock CPU path"):
        assert needle in jl, needle
    hdr = open(os.path.join(ROOT, "include", "dfdb_ir.h")).read()
    for name, val in (("DFIR_IDIV", 0x14), ("DFIR_REM", 0x15), ("DFIR_MOD", 0x16), ("DFIR_CAST", 0x50)):      # what RAISING_OPS names
        assert re.search(r"#define\s+" + name + r"\s+0x%02x" % val, hdr), name
