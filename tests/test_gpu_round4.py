"""Round 4: k_scan_cmp_narrow (16 bytes per lane for 1-, 2- and 4-byte columns) against the oracle and numpy; see also tests/test_gpu_stream.py
(late materialization) and tests/test_gpu_round3.py (group counts bound to their exchange)."""
import operator

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
