"""Round 4: k_scan_cmp_narrow (16 bytes per lane for 1-, 2- and 4-byte columns) against the oracle and numpy; see also tests/test_gpu_stream.py
(late materialization) and tests/test_gpu_round3.py (group counts bound to their exchange)."""
import operator
import os

import numpy as np
import pytest

from helpers import Pair, apply_stages, assert_same

pytestmark = pytest.mark.gpu

OPS = {"==": operator.eq, "!=": operator.ne, "<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 5 * 4096 + 1009, 300_001])
@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.bool_, np.int16, np.uint16, np.int32, np.uint32, np.float32])
def test_narrow_scan_every_size_and_type(oracle, dfdb_mod, ctx, dtype, n):
    """`col OP const` over narrow columns: a lane compares 16 bytes of rows, lane groups OR their pieces into 64-row words (k_scan.hip: k_scan_cmp_narrow).
    Row counts around every boundary of that layout — one vector, one wave load (256 / 512 / 1024 rows), one tile, one four-tile group — for every op,
    as a fresh mask and after a range stage (AND_EXISTING with dead tiles), with ctx option scan_narrow = 2 (every narrow type), 0 (the one-element-per-lane
    kernel) and 1 (the default: 1-byte types only):
    the oracle, numpy and both kernels agree bit for bit."""
    from dfdb import ir
    rng = np.random.default_rng(n * 31 + np.dtype(dtype).itemsize)
    kind = np.dtype(dtype).kind
    if kind == "b":
        x = rng.integers(0, 2, n).astype(bool); c = True
    elif kind == "f":
        x = (rng.integers(-50, 50, n) / 4).astype(dtype); x[::7] = np.nan; c = dtype(3.25)
    elif kind == "u":
        x = rng.integers(0, 100, n).astype(dtype); c = 40
    else:
        x = rng.integers(-60, 60, n).astype(dtype); c = -7
    p = Pair(oracle, dfdb_mod, {"x": x}, block_size=4096)
    for narrow in (2, 0, 1):
        ctx.set_option("scan_narrow", narrow)
        try:
            for name, f in OPS.items():
                if kind == "b" and name not in ("==", "!="):
                    continue
                ov, dv = apply_stages(p, [("pred", f(ir.col(0), ir.const(c)))])
                assert_same(p, ov, dv)
                assert np.array_equal(dv._query().indices(), np.nonzero(f(x, c))[0] + 1), (name, narrow)
            if kind == "b":
                ov, dv = apply_stages(p, [("pred", ir.col(0))])                   # a Bool column as the selection itself
                assert_same(p, ov, dv)
            # after a range stage: the scan ANDs into an existing mask and skips the tiles the range left empty
            lo, hi = max(1, n // 3), max(1, n // 3 + min(n, 2000))
            hi = min(hi, n)
            ov, dv = apply_stages(p, [("range", lo, 1, hi), ("pred", OPS[">="](ir.col(0), ir.const(c)) if kind != "b" else OPS["=="](ir.col(0), ir.const(c)))])
            assert_same(p, ov, dv)
        finally:
            ctx.set_option("scan_narrow", 1)


def test_narrow_scan_large_properties(dfdb_mod, ctx):
    """2e8 rows per type made on the device (casts of the generated column): the narrow kernel and the one-element-per-lane kernel produce the same bitmap,
    and the count equals what the Int64 column the values were cast from gives for the same threshold."""
    import torch
    from dfdb import ir
    n = 200_000_000
    t = dfdb_mod.DFTable.new(ctx=ctx)
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)

    def add(name, expr):
        t.add_column_from(name, dfdb_mod.DFView(t, dfdb_mod.Projection({name: expr}), dfdb_mod.DFView(t).selection))
    a = ir.col(0)
    add("i32", ir.cast(a, ir.I32)); add("i16", ir.cast(a % 30000, ir.I16)); add("u8", ir.cast(a % 200, ir.U8))
    cases = [("i32", t.i32 > 899_999, t.a > 899_999), ("i16", t.i16 >= 27_000, (t.a % 30000) >= 27_000), ("u8", t.u8 == 7, (t.a % 200) == 7)]
    for name, narrow_pred, wide_pred in cases:
        want = t[wide_pred, dfdb_mod.ALL]._query().count()
        maps = []
        for narrow in (2, 0):
            ctx.set_option("scan_narrow", narrow)
            try:
                q = t[narrow_pred, dfdb_mod.ALL]._query()
                assert q.count() == want, (name, narrow)
                bm = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
                from dfdb import _native as N
                N.check(N.load().dfdb_select_bitmap(q._h, bm.data_ptr(), N.MEM_DEVICE))
                ctx.synchronize()
                maps.append(bm)
            finally:
                ctx.set_option("scan_narrow", 1)
        assert torch.equal(maps[0], maps[1]), name
    t.close()


@pytest.mark.parametrize("n", [1, 63, 1023, 1025, 4097, 65_536 + 7, 300_017])
def test_every_projected_predicate_column_is_captured(oracle, dfdb_mod, ctx, n):
    """never gather a column the scan already held.  Under dfdb_query_hint_materialize the launch that produces the query's final mask
    keeps the selected values of up to TWO projected 8-byte predicate columns (k_scan_terms EXTRA = 5: the term before the last is parked in LDS) — also
    when string / dictionary / generic conjuncts or earlier stages ran before it and it only ANDs into their mask.  Every result equals the oracle's
    (materialization.jl:27-40, projection.jl:128-154), the capture path is the one that ran, and ctx option scan_capture = 1 / 0 give the same answers."""
    from dfdb import ir
    rng = np.random.default_rng(n)
    strs = oracle.flat_to_strings(*oracle.gen_str(0x77, 0, n))
    cols = {"a": oracle.gen_i64(0x9E3779B97F4A7C15, 0, n), "b": oracle.gen_i64(0x1111, 0, n), "x": oracle.gen_f64(0x2222, 0, n),
            "u": rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2), "s": strs, "i32": rng.integers(-100, 100, n).astype(np.int32)}
    p = Pair(oracle, dfdb_mod, cols, block_size=65536)
    a, b, x, u, s, i32 = (ir.col(k) for k in range(6))
    two = [("a", a), ("x", x)]
    cases = [
        ([("pred", (a > 300_000) & (x < 1200.0))], two, 2),                                        # two terms, both projected
        ([("pred", (a > 300_000) & (x < 1200.0) & (s != "sony"))], two, 2),                        # config 5: a flat string scan runs first
        ([("pred", (a > 300_000) & (x < 1200.0) & (b % 7 != 0))], two + [("b", b)], 2),            # a generic conjunct first; b is gathered
        ([("pred", (x < 1500.0) & (u >= 2**62) & (a > 100_000))], [("u", u), ("a", a), ("x", x)], 2),   # three candidates: two captured, one gathered
        ([("range", 5, 3, n), ("pred", (a > 300_000) & (x < 1200.0))], two, 2),                    # an earlier stage: the scan ANDs into its mask, dead tiles skipped
        ([("pred", b > 100_000), ("pred", (a > 300_000) & (x < 1200.0) & (i32 > -50))], two + [("i32", i32)], 2),
        ([("pred", (65 > a % 100) & (a > 300_000) & (x < 1200.0))], two, 2),                        # (a rem term of a is not a capture candidate; the plain one is)
        ([("pred", x < 1200.0)], [("x", x), ("x2", x)], 1),                                        # one term: k_scan_cmp's own capture
        ([("range", 1, 1, max(1, n // 2)), ("pred", x < 1200.0)], [("x", x)], 1),                  # one term over an existing mask: k_scan_terms<AND_EXISTING, 1>
        ([("pred", (a > 300_000) & (x < 1200.0)), ("range", 1, 2, n)], two, 0),                    # a range stage LAST: the final mask is not the scan's, nothing captured
        ([("pred", (a > 900_000) | (x < 100.0))], two, 0),                                         # a disjunction: no capture
    ]
    for cap in (2, 1, 0):
        ctx.set_option("scan_capture", cap)
        try:
            for stages, proj, want_caps in cases:
                ov, dv = apply_stages(p, stages, proj=proj)
                ctx.profile(True)
                assert_same(p, ov, dv)
                ncap, _ = ctx.profile_get("compact_captured")
                ngat, _ = ctx.profile_get("gather")
                ctx.profile(False)
                if ov.nrow() > 0:
                    # assert_same materialises once through the hinted path; the captured columns leave as copies, the others as gathers
                    expect = min(want_caps, cap)
                    assert (ncap >= 1) == (expect >= 1) and ncap >= expect, (stages, cap, ncap, ngat)
        finally:
            ctx.set_option("scan_capture", 2)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


@pytest.mark.parametrize("n", [1, 1023, 70_001])
def test_nullable_string_generator_and_three_valued_equality(oracle, dfdb_mod, ctx, n):
    """DFDB_GEN_STR_BRANDS10_MISSING (the bench's Union{String,Missing} column, like the docs' real data set: docs/src/index.md:264-272): row i is missing when
    (h >> 32) mod 8 == 7, else brands10[h mod 10] — rebuilt here in numpy from the oracle's plain generator — and `s == "sony"` over it selects what the oracle
    selects over the same column (a comparison with missing is missing; the selection needs coalesce(., false): selection.jl:52-55 wants plain Bool)."""
    from dfdb import ir
    seed = 0x9E3779B97F4A7C15
    t = dfdb_mod.DFTable.new(block_size=4096, ctx=ctx)
    t.add_generated("s", dfdb_mod.GEN_STR_BRANDS10_MISSING, seed, n)
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, seed, n)
    with np.errstate(over="ignore"):
        h = _splitmix64(np.uint64(seed) + np.arange(n, dtype=np.uint64))
    miss = ((h >> np.uint64(32)) & np.uint64(7)) == np.uint64(7)
    plain = oracle.flat_to_strings(*oracle.gen_str(seed, 0, n))
    want = [None if m else s for s, m in zip(plain, miss.tolist())]
    got = dfdb_mod.materialize(t[dfdb_mod.ALL, ["s"]])["s"].tolist()
    assert [g if isinstance(g, str) else None for g in got] == want
    p = Pair(oracle, dfdb_mod, {"s": want, "a": oracle.gen_i64(seed, 0, n)}, block_size=4096)
    pred = ir.coalesce(ir.col(0) == "sony", False)
    ov, dv = apply_stages(p, [("pred", pred)])
    assert_same(p, ov, dv)
    dq = t[pred, dfdb_mod.ALL]._query()
    assert np.array_equal(dq.indices(), ov.select_indices())
    # coalesce(<string term>, false) over a nullable String column is K5's own answer (a missing row selects nothing), not an interpreter program: every
    # term kind, the empty pattern (which every NON-missing row matches), a long pattern, and a conjunction with a numeric term
    ctx.profile(True)
    for term in (ir.col(0) != "sony", ir.startswith(ir.col(0), "s"), ir.endswith(ir.col(0), "y"), ir.col(0) == "", ir.col(0) != "", ir.startswith(ir.col(0), ""),
                 ir.col(0) == "a-pattern-that-is-longer-than-sixteen-bytes", ir.col(0) != "a-pattern-that-is-longer-than-sixteen-bytes"):
        ov, dv = apply_stages(p, [("pred", ir.coalesce(term, False))])
        assert_same(p, ov, dv)
        ov, dv = apply_stages(p, [("pred", ir.coalesce(term, False) & (ir.col(1) > 300_000))], proj=[("s", ir.col(0)), ("a", ir.col(1))])
        assert_same(p, ov, dv)
    assert ctx.profile_get("str_match")[0] >= 16 and ctx.profile_get("interp_predicate")[0] + ctx.profile_get("jit_predicate")[0] == 0
    ctx.profile(False)
    # ismissing counts (docs/src/index.md:326-328)
    assert t[ir.ismissing(ir.col(0)), dfdb_mod.ALL]._query().count() == int(miss.sum())
    t.close()


# ------------------------------------------------------------------ unique / groupreduce: the table sized by the distinct values, and the dense form
def _julia_unique(values, missing=None):
    from test_gpu_parity import julia_unique
    return julia_unique(values, missing)


@pytest.mark.parametrize("form", ["dense", "dense_one_tile_chunks", "hashed", "hashed_tiny_table"])
def test_unique_and_groupreduce_forms_agree_with_first_appearance(oracle, dfdb_mod, form):
    """unique / groupreduce (column.jl:102-126, aggregate.jl:1-36) through every form of the round-4 rewrite: integer keys of a small range without a hash
    table (presence bits in LDS, first rows found in row-ordered launches that stop early), and the hash table that starts small and MIGRATES as distinct
    values turn up (a table of 1024 slots and one-tile chunks force aborts, repeated chunks and several migrations).  Keys: every integer width, a value
    that first turns up in the very last rows (the early exit must not miss it), all-distinct keys, few keys, nullable keys, floats with NaN / -0.0,
    Strings; over a filtered view; against Base.unique's order of first appearance."""
    from test_gpu_parity import _np_group_ids, _np_groupreduce
    ctx = dfdb_mod.Context(0)
    opts = {"dense": {}, "dense_one_tile_chunks": {"unique_chunk_tiles": 1, "unique_dense_sample": 0}, "hashed": {"unique_dense": 0, "unique_test_collide": 2},
            "hashed_tiny_table": {"unique_dense": 0, "unique_cap0_log2": 10, "unique_chunk_tiles": 1}}[form]
    for k, v in opts.items():
        ctx.set_option(k, v)
    rng = np.random.default_rng(77)
    n = 600_011                # (586 tiles: the dense form samples every second tile for the range of a wide key, and `late` puts two keys outside what it sees)
    late = rng.integers(0, 900, n).astype(np.int64) + 5_000_000; late[-3] = 5_000_950; late[-1] = 4_999_990
    f = rng.integers(-3, 4, n).astype(np.float64); f[::97] = np.nan; f[5::101] = -0.0
    words = [f"w{k:05d}" for k in range(3000)]
    cols = {"i8": rng.integers(-128, 128, n).astype(np.int8), "u8": rng.integers(0, 256, n).astype(np.uint8), "flag": rng.integers(0, 2, n).astype(bool),
            "i16": rng.integers(-30000, 30000, n).astype(np.int16), "u16": rng.integers(0, 65536, n).astype(np.uint16),
            "i32": rng.integers(-70_000, 70_000, n).astype(np.int32), "u32": (rng.integers(0, 1000, n) + 4_000_000_000).astype(np.uint32),
            "late": late, "neg": rng.integers(-2**63, -2**63 + 5000, n, dtype=np.int64), "top": (rng.integers(0, 3000, n).astype(np.uint64) + np.uint64(2**64 - 3000)),
            "wide": rng.integers(-2**62, 2**62, n).astype(np.int64), "distinct": rng.permutation(n).astype(np.int64) * 7,
            "m": np.ma.masked_array(rng.integers(0, 5000, n).astype(np.int64), mask=rng.random(n) < 0.2), "f": f,
            "s": [words[i] for i in rng.integers(0, len(words), n)], "c": rng.integers(-1000, 1000, n).astype(np.int64)}
    t = dfdb_mod.DFTable.from_columns(cols, block_size=65536, ctx=ctx)
    sel = cols["c"] > -700
    v = t[t.c > -700, dfdb_mod.ALL]
    for name, src in cols.items():
        if name == "c":
            continue
        for view, keep in ((t, np.ones(n, bool)), (v, sel)):
            got = getattr(view, name).unique()
            if isinstance(src, np.ma.MaskedArray):
                want = _julia_unique(src.data[keep].tolist(), np.ma.getmaskarray(src)[keep].tolist())
                assert [None if mm else x for x, mm in zip(np.asarray(got.data).tolist(), np.ma.getmaskarray(got).tolist())] == want, (form, name)
            elif isinstance(src, list):
                assert list(got) == _julia_unique([x for x, k in zip(src, keep) if k]), (form, name)
            elif src.dtype.kind == "f":
                want = _julia_unique(src[keep].tolist())
                assert len(got) == len(want) and all((x != x and y != y) or (x == y and np.signbit(x) == np.signbit(y)) for x, y in zip(got.tolist(), want)), (form, name)
            else:
                assert got.tolist() == _julia_unique(src[keep].tolist()), (form, name)
    vals = cols["c"]
    for by in ("i8", "u16", "i32", "late", "neg", "top", "wide", "m", "s"):
        keys = cols[by] if not isinstance(cols[by], list) else np.array(cols[by], dtype=object)
        ids = _np_group_ids([keys[i] for i in np.nonzero(sel)[0]])
        for stat in ("sum", "min"):
            got = dfdb_mod.groupreduce(v, by, "c", stat)
            order, cnt, want = _np_groupreduce(ids, vals[sel], stat)
            import pandas as pd
            gk = [None if (not isinstance(k, str) and pd.isna(k)) else (k if isinstance(k, str) else int(k)) for k in got[by].tolist()]
            wk = [None if (k is np.ma.masked or k is None) else (k if isinstance(k, str) else int(k)) for k in order]
            assert gk == wk and got["count"].tolist() == cnt.tolist() and np.array_equal(got[stat].to_numpy().astype(np.int64), want.astype(np.int64)), (form, by, stat)
    assert dfdb_mod.nrow(v) == int(sel.sum())
    if form == "dense":          # which kernels ran: the dense form for the narrow ranges, the table for the wide ones
        ctx.profile(True)
        t.late.unique(); t.wide.unique()
        ctx.synchronize()
        assert ctx.profile_get("unique_presence")[0] == 1 and ctx.profile_get("unique_insert")[0] >= 1
        ctx.profile(False)
    t.close()
    ctx.close()


def test_nothing_reads_a_recycled_buffer_it_has_not_written():
    """DevPool (common.hpp) hands the device buffers of freed queries and of unique / groupreduce to the next allocation of their size class instead of to hipFree:
    with DFDB_POOL_POISON=1 every buffer is filled with 0xA5 as it enters the pool, so a kernel that counts on fresh memory being zero — or on a neighbour's old
    contents — gets garbage every time.  A second process runs the unique / groupreduce forms, the capture and the narrow-scan tests of this file that way."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DFDB_POOL_POISON="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_round4.py"), os.path.join(root, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
                        "-k", "(unique or groupreduce or capture or narrow or dictionary) and not recycled"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500, cwd=root)
    tail = p.stdout.decode(errors="replace")[-1500:]
    assert p.returncode == 0 and " passed" in tail and "failed" not in tail, tail
