"""CPU tests (no GPU): the oracle against the reference's known answers (golden file), against numpy, and
the codec round-trip properties of test/block_streams.jl."""
import os

import numpy as np
import pytest

import golden_cases as G
from helpers import is_str_col

CASES = G.load_cases()


def make_oracle_table(O, cols, block_size):
    t = O.Table(block_size=block_size)
    for k, v in cols.items():
        if isinstance(v, np.ma.MaskedArray):
            t.add_column(k, np.ascontiguousarray(v.filled(0)), missing=np.ma.getmaskarray(v))
        else:
            t.add_column(k, v)
    return t


def oracle_view(O, t, stages, proj):
    v = t.view()
    for st in stages:
        if st[0] == "range":
            v.add_range(st[1], st[2], st[3])
        elif st[0] == "int":
            v.add_integer(st[1])
        elif st[0] == "idx":
            v.add_indices(st[1])
        else:
            v.add_predicate(st[1].to_ir())
    if proj is not None:
        v.set_projection([(n, e.to_ir()) for n, e in proj])
    return v


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_known_answers(oracle, case):
    cols = G.build_columns(case["table"])
    names = list(cols.keys())
    for bs in case["block_sizes"]:
        t = make_oracle_table(oracle, cols, bs)
        v = oracle_view(oracle, t, G.stages_for(case, names), G.proj_for(case, names))
        assert v.select_indices().tolist() == case["expect_rows"], f"{case['name']} ({case['ref']}) block_size={bs}"
        assert v.nrow() == len(case["expect_rows"])
        G.check_columns(case, names, v.materialize(), oracle.flat_to_strings)
        G.check_columns(case, names, v.materialize(count_pass=False), oracle.flat_to_strings)


def test_queue_composition_rules(oracle):
    """test/selection.jl:5-37."""
    from dfdb import ir
    t = make_oracle_table(oracle, {"a": np.arange(1, 101, dtype=np.int64)}, 50)
    v = t.view()
    assert v.nstages == 0
    v.add_range(5, 1, 20)
    assert v.nstages == 1
    v.add_range(1, 1, 5)                       # (5:20)[1:5] == 5:9
    assert v.nstages == 1 and v.stage(0) == dict(kind="range", start=5, step=1, stop=9, n=5)
    tb = (ir.col(0) == 1).to_ir()
    v.add_predicate(tb)
    assert v.nstages == 2 and v.stage(1)["kind"] == "predicate"
    v2 = t.view().add_predicate(tb).add_predicate(tb)   # two predicates fuse
    assert v2.nstages == 1
    v2.add_range(1, 1, 3)
    assert v2.nstages == 2
    with pytest.raises(ValueError):            # non-Bool result: ArgumentError
        v2.add_predicate((ir.col(0) * 3).to_ir())
    with pytest.raises(IndexError):            # (5:20)[1:30] is a BoundsError
        t.view().add_range(5, 1, 20).add_range(1, 1, 30)
    # range[vector], vector[range], strided collapse
    v3 = t.view().add_range(10, 2, 30).add_range(2, 2, 6)
    assert v3.stage(0) == dict(kind="range", start=12, step=4, stop=20, n=3)
    v4 = t.view().add_range(10, 2, 30).add_indices([3, 1]).add_range(2, 1, 2)
    assert v4.nstages == 1 and v4.stage(0)["kind"] == "indices" and v4.select_indices().tolist() == [10]


def test_executor_state_across_blocks(oracle):
    """test/selection.jl:62-72: offsets carry over blocks, is_finished flips after the last needed row."""
    from dfdb import ir
    a = np.arange(1, 101, dtype=np.int64)
    t = make_oracle_table(oracle, {"a": a}, 50)
    v = t.view().add_range(10, 1, 60).add_predicate(((65 > ir.col(0)) & (ir.col(0) > 34)).to_ir()).add_range(15, 1, 18)
    e = oracle.SelExec(v)
    res = []
    for blk in range(2):
        assert not e.is_finished()
        part = a[blk * 50:(blk + 1) * 50]
        res += part[e.apply(50, {0: part})].tolist()
    assert res == [49, 50, 51, 52] and e.is_finished()
    # skip_if_can: only the first stage, nominal block size (selection.jl:177-190)
    e2 = oracle.SelExec(t.view().add_range(120, 1, 130))
    assert e2.skip_if_can(50) and e2.skip_if_can(50) and not e2.skip_if_can(50)


def test_required_columns_and_types(oracle):
    """test/broadcast.jl:15-44."""
    from dfdb import ir
    t = make_oracle_table(oracle, {"a": np.arange(1, 101, dtype=np.int64), "b": [str(i) for i in range(100)], "c": 0.5 * np.arange(1, 101)}, 100)
    a, c = ir.col(0), ir.col(2)
    assert oracle.expr_result_type(t, (a * 2).to_ir()) == oracle.I64
    assert oracle.expr_result_type(t, (a + c).to_ir()) == oracle.F64
    assert oracle.expr_result_type(t, (a + (a + c)).to_ir()) == oracle.F64
    assert oracle.expr_required_columns(t, (a + (a + c)).to_ir()) == [0, 2]
    assert oracle.expr_result_type(t, (a / 50).to_ir()) == oracle.F64
    assert oracle.expr_result_type(t, (a > c).to_ir()) == oracle.BOOL
    with pytest.raises(ValueError):            # arrays are rejected (test/broadcast.jl:73-81)
        ir.isin(a, [1]) + [1, 2, 3]
    with pytest.raises(ValueError):
        a + np.array([1, 2, 3])


@pytest.mark.parametrize("n", [64000, 74000])
def test_block_codec_roundtrip(oracle, n):
    """test/block_streams.jl:11-67: header + LZ4 round trip, skip_block."""
    rng = np.random.default_rng(n)
    a = rng.integers(1, 100000, n).astype(np.int64)
    blk = oracle.block_encode(a.tobytes(), n)
    rows, body, used = oracle.block_decode(blk)
    assert rows == n and used == len(blk) and np.array_equal(np.frombuffer(body, np.int64), a)
    b = rng.integers(1, 100000, 1000).astype(np.int64)
    two = blk + oracle.block_encode(b.tobytes(), 1000)
    rows2, body2, used2 = oracle.block_decode(two, offset=used)      # skip the first, read the second
    assert rows2 == 1000 and used + used2 == len(two) and np.array_equal(np.frombuffer(body2, np.int64), b)


def test_docs_compression_ratios(oracle):
    """docs/src/index.md:53,56: LZ4 ratio 2.0 for 1:3e6 and 2.55 for rand(1:1000) (COMPRESSION_LEVEL = 2)."""
    t = oracle.Table()
    t.add_column("a", np.arange(1, 3_000_001, dtype=np.int64))
    st = t.column_stats(0)
    assert st["uncompressed"] == 3_000_000 * 8 and round(st["uncompressed"] / st["compressed"], 2) == 2.0
    t.add_column("r", np.random.default_rng(1).integers(1, 1001, 3_000_000).astype(np.int64))
    st = t.column_stats(1)
    assert abs(st["uncompressed"] / st["compressed"] - 2.55) < 0.03
    assert st["blocks"] == 46 and st["rows"] == 3_000_000


def test_table_files_and_header_validation(oracle, tmp_path):
    """test/tables.jl:36-70, test/table_io.jl:4-9."""
    t = oracle.Table(block_size=1223)
    t.add_column("a", np.arange(10, dtype=np.int32))
    t.add_column("b", ["x", "yy", ""] + ["z"] * 7)
    t.add_column("c", np.arange(10, dtype=np.int64))
    p = str(tmp_path / "test_tb")
    t.save(p)
    assert sorted(os.listdir(p)) == ["1.bin", "2.bin", "3.bin", "meta.bin"]
    t2 = oracle.Table.open(p)
    assert [t2.colinfo(i) for i in range(3)] == [(1, "a", oracle.I32), (2, "b", oracle.STRING), (3, "c", oracle.I64)]
    assert t2.block_size == 1223
    got = t2.view().materialize()
    assert got[0].tolist() == list(range(10)) and oracle.flat_to_strings(*got[1]) == ["x", "yy", ""] + ["z"] * 7
    with pytest.raises(OSError):
        oracle.Table.open(str(tmp_path / "missing"))
    import struct
    raw = open(os.path.join(p, "3.bin"), "rb").read()
    open(os.path.join(p, "3.bin"), "wb").write(struct.pack("<q", 10) + raw[8:])          # wrong block size
    with pytest.raises(OSError):
        oracle.Table.open(p)
    open(os.path.join(p, "3.bin"), "wb").write(struct.pack("<qi", 1223, 5) + b"Int32")   # wrong type
    with pytest.raises(OSError):
        oracle.Table.open(p)


def test_oracle_vs_numpy_random_queries(oracle):
    """Third opinion: numpy boolean indexing (what DataFrames.jl does for the reference's own tests)."""
    from dfdb import ir
    rng = np.random.default_rng(5)
    n = 30_011
    a = rng.integers(-1000, 1000, n).astype(np.int64)
    c = rng.integers(1, 50, n).astype(np.int64)
    x = rng.normal(0, 100, n)
    t = make_oracle_table(oracle, {"a": a, "c": c, "x": x}, 4096)
    A, Cc, X = ir.col(0), ir.col(1), ir.col(2)
    # Julia rem: sign of the dividend == numpy fmod
    checks = [(A % Cc == 0, np.fmod(a, c) == 0), ((A * 2 + Cc) > X, (a * 2 + c) > x), (A / Cc > 1.5, a / c > 1.5),
              (ir.mod(A, Cc) == 3, np.mod(a, c) == 3), (ir.div(A, Cc) == -2, np.trunc(a / c) == -2), (abs(A) < 10, np.abs(a) < 10),
              (ir.isin(A, [1, 11, 21, -5]), np.isin(a, [1, 11, 21, -5])), ((A > 0) & ~(X < 0), (a > 0) & ~(x < 0)), (A == X, a == x)]
    for e, m in checks:
        v = t.view().add_predicate(e.to_ir())
        assert np.array_equal(v.select_indices(), np.nonzero(m)[0] + 1)
    # range after predicate indexes the survivor stream (quirk Q1)
    v = t.view().add_predicate((A > 0).to_ir()).add_range(5, 3, 400)
    assert np.array_equal(v.select_indices(), (np.nonzero(a > 0)[0] + 1)[4:400:3])
    # float sum is strictly left to right
    v = t.view().add_predicate((A > 0).to_ir()).set_projection([("x", X.to_ir())])
    acc = 0.0
    for val in x[a > 0]:
        acc += val
    assert v.sum_f64(0) == acc


def test_generator_formulas(oracle):
    """splitmix64-based columns; spot values computed by hand in Python ints."""
    M = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)
    seed = 0x9E3779B97F4A7C15
    assert oracle.splitmix64(0) == sm(0) and oracle.splitmix64(seed + 5) == sm(seed + 5)
    a = oracle.gen_i64(seed, 100, 50)
    assert a.tolist() == [sm((seed + 100 + i) & M) % 1_000_000 for i in range(50)]
    x = oracle.gen_f64(seed, 0, 50)
    assert x.tolist() == [(sm((seed + i) & M) >> 11) * 2.0 ** -53 * 2000.0 for i in range(50)]
    brands = ["apple", "samsung", "huawei", "microsoft", "dell", "xbox", "sony", "intel", "lenovo", "asus"]
    sz, by = oracle.gen_str(seed, 7, 40)
    assert oracle.flat_to_strings(sz, by) == [brands[sm((seed + 7 + i) & M) % 10] for i in range(40)]
    big = oracle.gen_i64(seed, 0, 1_000_000)
    assert abs((big > 899_999).mean() - 0.1) < 0.002


# ------------------------------------------------------------------ Union{T,Missing} inside expressions
MISSING_CASES = None


def missing_cases():
    """(name, expr builder, expected values, expected missing flags or None) over
         m = [1, missing, 3, missing, 0] :: Union{Int64,Missing}   c = [0, 0, 5, 5, 0] :: Int64
         sm = ["a", missing, "b", "ab", missing] :: Union{String,Missing}
    Known answers are Julia's: every Base method propagates missing; & and | are three-valued; ismissing / coalesce end it."""
    from dfdb import ir
    m, c, sm = ir.col(0), ir.col(1), ir.col(2)
    T, F = True, False
    return [
        ("m+c", m + c, [1, 0, 8, 0, 0], [0, 1, 0, 1, 0]),
        ("m>2", m > 2, [F, F, T, F, F], [0, 1, 0, 1, 0]),
        ("!(m>2)", ~(m > 2), [T, F, F, F, T], [0, 1, 0, 1, 0]),
        ("coalesce(m>2,false)", ir.coalesce(m > 2, False), [F, F, T, F, F], None),
        ("coalesce(m>2,true)", ir.coalesce(m > 2, True), [F, T, T, T, F], None),
        ("coalesce(m,-1)", ir.coalesce(m, -1), [1, -1, 3, -1, 0], None),
        ("(m>2)&(c>1)", (m > 2) & (c > 1), [F, F, T, F, F], [0, 0, 0, 1, 0]),     # missing & false == false
        ("(c>1)&(m>2)", (c > 1) & (m > 2), [F, F, T, F, F], [0, 0, 0, 1, 0]),
        ("(m>2)|(c>1)", (m > 2) | (c > 1), [F, F, T, T, F], [0, 1, 0, 0, 0]),     # missing | true == true
        ("xor", (m > 2) ^ (c > 1), [F, F, F, F, F], [0, 1, 0, 1, 0]),
        ("ismissing(m+c)", ir.ismissing(m + c), [F, T, F, T, F], None),
        ("ismissing(m)", ir.ismissing(m), [F, T, F, T, F], None),
        ("m*2.5", m * 2.5, [2.5, 0, 7.5, 0, 0.0], [0, 1, 0, 1, 0]),
        ("-m", -m, [-1, 0, -3, 0, 0], [0, 1, 0, 1, 0]),
        ("in(m,[1,3])", ir.isin(m, [1, 3]), [T, F, T, F, F], [0, 1, 0, 1, 0]),
        ("sm==a", sm == "a", [T, F, F, F, F], [0, 1, 0, 0, 1]),
        ("startswith(sm,a)", ir.startswith(sm, "a"), [T, F, F, T, F], [0, 1, 0, 0, 1]),
        ("coalesce(sm==a,false)|(c>1)", ir.coalesce(sm == "a", False) | (c > 1), [T, F, T, T, F], None),
        ("sizeof(sm)", ir.sizeof(sm), [1, 0, 1, 2, 0], [0, 1, 0, 0, 1]),
        ("div(c, coalesce(m,1))", ir.div(c, ir.coalesce(m, 1)), None, None),          # row 5 divides by zero: DivideError
        ("div(c+1, m)", ir.div(c + 1, ir.coalesce(m + 1, 7) * 0 + m), None, None),    # (m==0 in row 5) DivideError; rows 2,4 are missing, not errors
        ("rem(c, m+1)", ir.rem(c, m + 1), [0, 0, 1, 0, 0], [0, 1, 0, 1, 0]),          # a missing divisor is not a DivideError
    ]


def missing_table(oracle, block_size=2):
    t = oracle.Table(block_size=block_size)
    t.add_column("m", np.array([1, 99, 3, 77, 0], np.int64), missing=np.array([0, 1, 0, 1, 0], np.uint8))
    t.add_column("c", np.array([0, 0, 5, 5, 0], np.int64))
    t.add_column("sm", ["a", None, "b", "ab", None])
    return t


def test_missing_propagation_and_three_valued_logic(oracle):
    from dfdb import ir
    t = missing_table(oracle)
    for name, e, want, miss in missing_cases():
        v = t.view()
        v.set_projection([("k", e.to_ir())])
        if want is None:
            with pytest.raises(ZeroDivisionError):
                v.materialize()
            continue
        got = v.materialize()[0]
        if miss is None:
            assert not isinstance(got, np.ma.MaskedArray), name
            assert np.array_equal(got, np.array(want, got.dtype)), name
        else:
            assert isinstance(got, np.ma.MaskedArray), name
            assert np.array_equal(np.ma.getmaskarray(got), np.array(miss, bool)), name
            keep = ~np.array(miss, bool)
            assert np.array_equal(np.asarray(got.data)[keep], np.array(want, got.dtype)[keep]), name
    # as predicates: a Union{Missing,Bool} function is refused (selection.jl:52-55), its coalesce is a plain predicate
    with pytest.raises(ValueError):
        t.view().add_predicate((ir.col(0) > 2).to_ir())
    v = t.view().add_predicate(ir.coalesce((ir.col(0) > 2) | (ir.col(1) > 1), False).to_ir())
    assert v.select_indices().tolist() == [3, 4]
    v = t.view().add_predicate((~ir.ismissing(ir.col(0) + ir.col(1))).to_ir())
    assert v.select_indices().tolist() == [1, 3, 5]
    assert oracle.expr_result_type(t, (ir.col(0) + 1.5).to_ir()) == (ir.F64 | ir.NULLABLE)
    assert oracle.expr_result_type(t, ir.coalesce(ir.col(0), 0).to_ir()) == ir.I64
    with pytest.raises(NotImplementedError):     # coalesce(Int64?, Float64) would be Union{Int64,Float64}: outside the IR
        oracle.expr_result_type(t, ir.coalesce(ir.col(0), 0.5).to_ir())


def test_oracle_raises_the_error_the_block_iteration_meets_first(oracle):
    """Known answers for WHICH error a view with several faults raises (derived by hand from the reference's loops): predicates — the earliest
    erroring row, if the iteration reaches its block (skip_if_can / is_finished, selection.jl:177-196); computed projection columns — per block the
    columns in order (projection.jl:149-154): first erroring block, then first such column."""
    from dfdb import ir
    n, bs = 10_000, 1000
    a = np.arange(1, n + 1, dtype=np.int64)
    z = np.ones(n, np.int64); z[7_500] = 0                       # zero divisor in block 7
    f = np.full(n, 2.0); f[3_200] = 2.5                          # inexact conversion in block 3
    f2 = np.full(n, 2.0); f2[7_100] = 2.5                        # ... in block 7, before the zero divisor's row
    A, Z, F = ir.col(0), ir.col(1), ir.col(2)
    div, cast = A % Z, ir.cast(F, ir.I64)

    def table(fcol):
        t = oracle.Table(block_size=bs)
        for k, v in (("a", a), ("z", z), ("f", fcol)):
            t.add_column(k, v)
        return t

    def raised(fn):
        try:
            fn()
        except Exception as e:          # noqa: BLE001
            return type(e).__name__
        return None

    t1, t2 = table(f), table(f2)
    for t, proj, want in ((t1, [("d", div), ("c", cast)], "ValueError"), (t1, [("c", cast), ("d", div)], "ValueError"),
                          (t2, [("d", div), ("c", cast)], "ZeroDivisionError"), (t2, [("c", cast), ("d", div)], "ValueError")):
        v = t.view(); v.set_projection([(k, e.to_ir()) for k, e in proj])
        assert raised(v.materialize) == want, [k for k, _ in proj]
    for t, stages, want in ((t1, [("pred", (div == 0) & (cast == 2))], "ValueError"),                        # row 3200 comes before row 7500
                            (t2, [("pred", (div == 0) & (cast == 2))], "ValueError"),                        # row 7100 before row 7500
                            (t1, [("pred", div == 0), ("range", 1, 1, 7_000)], None),                         # done at the end of block 6
                            (t1, [("pred", div == 0), ("range", 1, 1, 7_001)], "ZeroDivisionError")):
        v = t.view()
        for st in stages:
            v.add_predicate(st[1].to_ir()) if st[0] == "pred" else v.add_range(st[1], st[2], st[3])
        assert raised(v.nrow) == want, stages


def _meta_type_strings(path):
    """the type strings of a table's meta.bin as written (table_io.jl:9-19, common_io.jl:1-4: Int32 length + bytes)"""
    import struct
    raw = open(os.path.join(path, "meta.bin"), "rb").read()
    _, _, ncols = struct.unpack_from("<qqq", raw, 0)
    pos, out = 24, []
    for _ in range(ncols):
        pos += 8
        (n,) = struct.unpack_from("<i", raw, pos); pos += 4 + n
        (n,) = struct.unpack_from("<i", raw, pos); out.append(raw[pos + 4:pos + 4 + n].decode()); pos += 4 + n
    return out


def test_type_strings_are_the_reference_s(oracle, tmp_path):
    """test/column_types.jl:31-43 (`deserialize("Int32") == Int32`, `deserialize("Missing(Int32)") == Union{Missing, Int32}`) and the names typestring writes
    (columntypes/base.jl:108-126,163-168): a table written here carries exactly those strings in meta.bin and in every column header, and a file whose strings
    were put there BY HAND (as the Julia package would write them) opens as the same types; a Tuple(...) column (column_types.jl:46-50) is refused by name."""
    import struct
    n = 5
    cols = [("i8", np.arange(n, dtype=np.int8), "Int8"), ("i16", np.arange(n, dtype=np.int16), "Int16"), ("i32", np.arange(n, dtype=np.int32), "Int32"),
            ("i64", np.arange(n, dtype=np.int64), "Int64"), ("u8", np.arange(n, dtype=np.uint8), "UInt8"), ("u16", np.arange(n, dtype=np.uint16), "UInt16"),
            ("u32", np.arange(n, dtype=np.uint32), "UInt32"), ("u64", np.arange(n, dtype=np.uint64), "UInt64"), ("f32", np.arange(n, dtype=np.float32), "Float32"),
            ("f64", np.arange(n, dtype=np.float64), "Float64"), ("b", np.array([True, False, True, True, False]), "Bool"), ("s", ["a", "bb", "", "d", "e"], "String"),
            ("m", np.ma.masked_array(np.arange(n, dtype=np.int32), mask=[0, 1, 0, 0, 1]), "Missing(Int32)"), ("ms", ["a", None, "", "d", None], "Missing(String)")]
    t = oracle.Table(block_size=3)
    for name, v, _ in cols:
        if isinstance(v, np.ma.MaskedArray):
            t.add_column(name, v.filled(0), missing=np.ma.getmaskarray(v))
        else:
            t.add_column(name, v)
    p = str(tmp_path / "types")
    t.save(p)
    assert _meta_type_strings(p) == [ts for _, _, ts in cols]
    for k, (_, _, ts) in enumerate(cols):                 # column header: Int64 block size + the same string (filesystem.jl:14-23)
        raw = open(os.path.join(p, f"{k + 1}.bin"), "rb").read(64)
        bs, ln = struct.unpack_from("<qi", raw, 0)
        assert bs == 3 and raw[12:12 + ln].decode() == ts
    t2 = oracle.Table.open(p)
    want = [oracle.I8, oracle.I16, oracle.I32, oracle.I64, oracle.U8, oracle.U16, oracle.U32, oracle.U64, oracle.F32, oracle.F64, oracle.BOOL, oracle.STRING,
            oracle.I32 | oracle.NULLABLE, oracle.STRING | oracle.NULLABLE]
    assert [t2.colinfo(i)[2] for i in range(len(cols))] == want
    # a meta.bin / column header written by hand with the reference's strings: two columns, Int32 and Missing(Int32), one block of two rows each
    import ctypes as C
    hand = str(tmp_path / "hand"); os.mkdir(hand)

    def jstr(s_):
        return struct.pack("<i", len(s_)) + s_.encode()

    def block(body):
        return oracle.block_encode(body, 2)              # Int32 rows, Int64 origin, Int64 compressed, LZ4 block (BlockStreams.jl:50-53)
    open(os.path.join(hand, "meta.bin"), "wb").write(struct.pack("<qqq", 1, 4, 2) + struct.pack("<q", 1) + jstr("x") + jstr("Int32") + struct.pack("<q", 2) + jstr("y") + jstr("Missing(Int32)"))
    open(os.path.join(hand, "1.bin"), "wb").write(struct.pack("<q", 4) + jstr("Int32") + block(struct.pack("<ii", 7, -8)))
    open(os.path.join(hand, "2.bin"), "wb").write(struct.pack("<q", 4) + jstr("Missing(Int32)") + block(struct.pack("<Q", 0b10) + struct.pack("<ii", 5, 99)))
    th = oracle.Table.open(hand)
    assert [th.colinfo(i)[2] for i in range(2)] == [oracle.I32, oracle.I32 | oracle.NULLABLE]
    got = th.view().materialize()
    assert got[0].tolist() == [7, -8] and got[1].tolist() == [5, None]
    open(os.path.join(hand, "meta.bin"), "wb").write(struct.pack("<qqq", 1, 4, 1) + struct.pack("<q", 1) + jstr("x") + jstr("Tuple(Int32, UInt64)"))
    with pytest.raises(NotImplementedError, match="Tuple"):
        oracle.Table.open(hand)


# ---------------------------------------------------------------- the file format pinned without either writer (round 6)
def _format_golden():
    import json
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = json.load(open(os.path.join(g, "format_v1.json")))
    files = {n: open(os.path.join(g, "format_v1", n), "rb").read() for n in ("meta.bin", "1.bin", "2.bin")}
    return spec, files, os.path.join(g, "format_v1")


def test_hand_assembled_format_fixture_is_what_its_script_makes(tmp_path):
    """tests/golden/format_v1 is the output of tests/golden/make_format_golden.py (struct.pack + system liblz4, one commented field per Julia line): the
    committed bytes are current, and every field the script recorded lies where it says."""
    import importlib.util
    spec, files, _ = _format_golden()
    assert sum(len(v) for v in files.values()) <= 4096
    for f in spec["fields"]:
        assert f["offset"] + f["length"] <= len(files[f["file"]]) and f["ref"], f
    # re-run the assembly into a scratch directory (liblz4 is in the build image; the GPU box only reads the committed bytes)
    try:
        import ctypes
        ctypes.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("no system liblz4 here: the committed bytes are checked by the other tests")
    sp = importlib.util.spec_from_file_location("make_format_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_format_golden.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    mod.OUT = str(tmp_path / "fmt")
    mod.HERE = str(tmp_path)
    mod.main()
    for n, want in files.items():
        assert open(os.path.join(mod.OUT, n), "rb").read() == want, n


def test_oracle_reads_and_rewrites_the_hand_assembled_format(oracle, tmp_path):
    """The oracle's READER decodes the fixture to the values its script started from; the oracle's WRITER, given those values, reproduces meta.bin and both
    column files byte for byte (it calls the same LZ4_compress_fast(…, 2) of liblz4 1.9.3 the script did)."""
    spec, files, path = _format_golden()
    t = oracle.Table.open(path)
    assert t.block_size == spec["block_size"] and t.names() == ["a", "s"]
    assert [t.colinfo(i)[0] for i in range(2)] == [1, 2]
    assert t.colinfo(0)[2] == oracle.I64 and t.colinfo(1)[2] == (oracle.STRING | oracle.NULLABLE)
    a, s = t.view().materialize()
    assert a.tolist() == spec["a"] and oracle.flat_to_strings(*s) == spec["s"]
    for i, n in ((0, "1.bin"), (1, "2.bin")):
        assert t.image(i) == files[n]
        st = t.column_stats(i)
        assert (st["rows"], st["blocks"]) == (10, 3)
    w = oracle.Table(block_size=spec["block_size"])
    w.add_column("a", np.array(spec["a"], np.int64))
    w.add_column("s", spec["s"], dtype=oracle.STRING | oracle.NULLABLE)
    out = str(tmp_path / "rewritten")
    w.save(out)
    for n, want in files.items():
        assert open(os.path.join(out, n), "rb").read() == want, n
