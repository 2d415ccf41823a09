"""Scan kernel forms: the pipelined two-column kernel against the generic one, the narrow-column (1 / 2 / 4-byte) scans at every size and type.
(re-filed by component in round 6 from the round-named files; no test body changed)"""


import operator

import numpy as np
import pytest

from helpers import Pair, apply_stages, assert_same


pytestmark = pytest.mark.gpu


OPS = {"==": operator.eq, "!=": operator.ne, "<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}


@pytest.mark.parametrize("pair", [0, 1])
@pytest.mark.parametrize("kinds", ["ii", "if", "fi", "ff"])
def test_two_column_conjunctions_every_form(dfdb_mod, kinds, pair):
    """`(a OP c1) & (b OP c2)` over Int64 / Float64 columns (docs/src/index.md:503-517) through the pipelined pair kernel (ctx option scan_pair = 1)
    and through the generic term kernel (0): the selection, the projected last-term column (kept by the scan in an LDS-staged capture buffer: sparse,
    dense enough to flush in the middle of a tile, and all rows), and sum / minimum / maximum over it, on full 4096-row groups, a partial group and
    a ragged last tile."""
    n = 4096 * 5 + 1024 * 2 + 777
    rng = np.random.default_rng(hash(kinds) & 0xffff)
    def col(k):
        return rng.integers(-1000, 1000, n).astype(np.int64) if k == "i" else np.round(rng.normal(0, 500, n), 3)
    a, b = col(kinds[0]), col(kinds[1])
    if kinds[1] == "f":
        b[rng.integers(0, n, 5)] = np.nan                       # NaN rows never satisfy an ordered comparison and poison min / max when selected
    c = dfdb_mod.Context(0)
    try:
        c.set_option("scan_pair", pair)
        t = dfdb_mod.DFTable.from_columns({"a": a, "b": b}, ctx=c)
        c.profile(True)
        assert t[(t.a >= 0) & (t.b <= 0), dfdb_mod.ALL]._query().count() == int(((a >= 0) & (b <= 0)).sum())
        assert c.profile_get("scan_terms")[0] == 1 and c.profile_get("scan_terms.pair")[0] == pair      # which kernel took it
        c.profile(False)
        for lo_a, hi_b in ((900, 900), (0, 0), (-2000, 2000), (-2000, -2000)):      # ~0.3 %, 25 %, every row, none
            for form in ("plain", "interval"):
                if form == "plain":
                    v = t[(t.a >= lo_a) & (t.b <= hi_b), ["b"]]
                    want = (a >= lo_a) & (b <= hi_b)
                else:
                    v = t[(t.a >= lo_a) & (t.a < 1500) & (t.b <= hi_b) & (t.b > -1500), ["b"]]
                    want = (a >= lo_a) & (a < 1500) & (b <= hi_b) & (b > -1500)
                rows = np.flatnonzero(want)
                q = v._query()
                assert np.array_equal(q.indices(), rows.astype(np.int64) + 1), (kinds, pair, lo_a, hi_b, form)
                got = dfdb_mod.materialize(v)["b"].to_numpy()
                assert np.array_equal(got, b[rows], equal_nan=True), (kinds, pair, lo_a, hi_b, form)
                if rows.size:
                    sel = b[rows]
                    q2 = v._query()
                    got_sum = q2.aggregate(dfdb_mod.AGG_SUM, 0)
                    if kinds[1] == "i":
                        assert got_sum == int(sel.sum()), (kinds, pair, form)
                    elif np.isnan(sel).any():
                        assert np.isnan(got_sum)
                    else:
                        assert abs(got_sum - float(np.sum(sel))) <= 64 * np.finfo(np.float64).eps * float(np.abs(sel).sum()) + 1e-300
                    for op, f in ((dfdb_mod.AGG_MIN, np.min), (dfdb_mod.AGG_MAX, np.max)):
                        q3 = v._query()
                        r = q3.aggregate(op, 0)
                        w = f(sel)
                        assert (np.isnan(r) and np.isnan(w)) or r == w, (kinds, pair, form, op)
        # the single-term scan keeps its own column the same way
        for thr in (900, 0, -2000):
            v = t[t.a >= thr, ["a"]]
            assert np.array_equal(dfdb_mod.materialize(v)["a"].to_numpy(), a[a >= thr])
        t.close()
    finally:
        c.close()


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
