"""Out of core BEHIND the C ABI (round 6): a query over a table whose required columns are not resident is answered by the ordinary entry
points — dfdb_count / dfdb_select_indices / dfdb_result_string_bytes / dfdb_materialize / dfdb_aggregate / dfdb_query_unique / dfdb_query_groupreduce —
which stream inside the library (csrc/ooc.cpp).  Every test below drives those C functions through `dfdb._native` (the `_Query` helpers are 1:1 ctypes
calls); no chunk loop runs in Python.  Reference behaviour: blocksiterator.jl:20-33,98-145, view.jl:183-206, materialization.jl:27-40, column.jl:102-126."""
import ctypes as C

import numpy as np
import pytest

from helpers import Pair, apply_stages

pytestmark = pytest.mark.gpu
SEED = 0x9E3779B97F4A7C15
CHUNKINGS = (1, 3, 7, 16, 64)


def col_seed(k):
    return (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF


@pytest.fixture(scope="module")
def pair(oracle, dfdb_mod, tmp_path_factory):
    n = 41_333                      # 41 full blocks of 1000 rows + 333
    rng = np.random.default_rng(11)
    strs = oracle.flat_to_strings(*oracle.gen_str(col_seed(3), 0, n))
    f = rng.integers(0, 40, n).astype(np.float64) / 4.0
    f[rng.random(n) < 0.01] = np.nan
    f[rng.random(n) < 0.01] = -0.0
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "x": oracle.gen_f64(col_seed(1), 0, n), "s": strs,
            "m": np.ma.masked_array(rng.integers(0, 50, n).astype(np.int64), mask=rng.random(n) < 0.2),
            "k": (rng.integers(0, 3000, n) * 7).astype(np.int64), "f": f, "iota": np.arange(1, n + 1, dtype=np.int64)}
    path = str(tmp_path_factory.mktemp("ooc") / "tb")
    p = Pair(oracle, dfdb_mod, cols, block_size=1000, via_files=path)
    p.d.close()
    p.d = dfdb_mod.open_table(path, load=False)      # metadata only: nothing resident
    p.cols = cols
    p.path = path
    return p


def views(ir):
    a, x, s, m, k = ir.col(0), ir.col(1), ir.col(2), ir.col(3), ir.col(4)
    return [
        ([("pred", a > 899_999)], None),
        ([("pred", (a > 500_000) & (x < 700.0))], [("a", a), ("s", s)]),
        ([("pred", s == "apple")], [("s", s), ("x", x)]),
        ([("range", 2_500, 3, 39_000), ("pred", a > 300_000)], [("iota", ir.col(6)), ("m", m)]),
        ([("pred", a > 700_000), ("range", 5, 2, 4001)], [("x", x), ("k", k)]),
        ([("idx", [41_000, 7, 20_500, 999, 1000, 1001])], None),
        ([("pred", a < 0)], None),                                     # nothing survives
        ([], [("a2", a * 2 + 1), ("s", s)]),                           # no selection, a computed column
    ]


def fresh_view(p, stages, proj):
    ov, dv = apply_stages(p, stages, proj=proj)
    return ov, dv


def same_cols(want, got):
    assert len(want) == len(got)
    for i, (w, g) in enumerate(zip(want, got)):
        if isinstance(w, tuple):
            assert np.array_equal(w[0], g[0]) and np.array_equal(w[1], g[1]), f"column {i}: strings differ"
        elif isinstance(w, np.ma.MaskedArray):
            assert np.array_equal(np.ma.getmaskarray(w), np.ma.getmaskarray(g)) and np.array_equal(w.compressed(), g.compressed()), f"column {i}"
        else:
            assert w.dtype == g.dtype and np.array_equal(w.view(np.uint8), g.view(np.uint8)), f"column {i}: values differ"


@pytest.mark.parametrize("chunk", CHUNKINGS)
def test_count_indices_materialize_through_the_abi(pair, dfdb_mod, chunk):
    from dfdb import ir
    p = pair
    p.d.ctx.set_option("ooc_chunk_blocks", chunk)
    for stages, proj in views(ir):
        ov, dv = fresh_view(p, stages, proj)
        q = dv._query()
        assert not any(p.d.resident(i) for i in range(p.d.ncols)), "nothing may have become resident"
        assert q.count() == ov.nrow(), (stages, chunk)
        assert np.array_equal(q.indices(), ov.select_indices()), (stages, chunk)
        same_cols(ov.materialize(), q.materialize())
        # a second materialize on the same handle (cached count, another pass) and one on a handle that never counted
        same_cols(ov.materialize(), dv._query().materialize())


@pytest.mark.parametrize("chunk", CHUNKINGS)
def test_aggregates_through_the_abi(pair, dfdb_mod, chunk):
    from dfdb import ir
    import dfdb._native as N
    p = pair
    p.d.ctx.set_option("ooc_chunk_blocks", chunk)
    a, x = p.cols["a"], p.cols["x"]
    sel = a > 640_000
    v = p.d[("a", lambda c: c > 640_000), ["a", "x", "k"]]
    q = v._query()
    assert q.aggregate(N.AGG_SUM, 0) == int(a[sel].sum())
    assert q.aggregate(N.AGG_MIN, 0) == int(a[sel].min()) and q.aggregate(N.AGG_MAX, 0) == int(a[sel].max())
    assert q.aggregate(N.AGG_COUNT, 0) == int(sel.sum())
    want = float(np.sum(x[sel]))
    got = q.aggregate(N.AGG_SUM, 1)
    assert abs(got - want) <= 64 * np.finfo(np.float64).eps * float(np.abs(x[sel]).sum())      # DESIGN.md section 5's bound
    assert q.aggregate(N.AGG_MIN, 1) == float(x[sel].min()) and q.aggregate(N.AGG_MAX, 1) == float(x[sel].max())
    # Float64 minimum / maximum with NaN and signed zeros in the data: Julia's rules (NaN wins; -0.0 < 0.0)
    qf = p.d[dfdb_mod.ALL, ["f"]]._query()
    assert np.isnan(qf.aggregate(N.AGG_MIN, 0)) and np.isnan(qf.aggregate(N.AGG_MAX, 0))
    f = p.cols["f"]
    keep = ~np.isnan(f)
    qz = p.d[("f", lambda c: c == c), ["f"]]._query()          # NaN != NaN drops them
    mn = qz.aggregate(N.AGG_MIN, 0)
    assert mn == 0.0 and np.signbit(mn) == bool(np.any(np.signbit(f[keep]) & (f[keep] == 0)))
    # an empty selection: sum is 0, minimum raises ArgumentError
    qe = p.d[("a", lambda c: c < 0), ["a"]]._query()
    assert qe.aggregate(N.AGG_SUM, 0) == 0
    with pytest.raises(ValueError, match="empty collection"):
        qe.aggregate(N.AGG_MIN, 0)


def first_appearance(vals, keyfn=lambda v: v):
    seen, out, rows = set(), [], []
    for i, v in enumerate(vals):
        kk = keyfn(v)
        if kk not in seen:
            seen.add(kk); out.append(v); rows.append(i + 1)
    return out, rows


@pytest.mark.parametrize("chunk", CHUNKINGS)
def test_unique_through_the_abi(pair, dfdb_mod, chunk):
    import dfdb._native as N
    p = pair
    p.d.ctx.set_option("ooc_chunk_blocks", chunk)
    L = N.load()
    # Int64 keys under a selection
    a, k = p.cols["a"], p.cols["k"]
    sel = a > 250_000
    v = p.d[("a", lambda c: c > 250_000), ["k"]]
    q = v._query()
    N.check(L.dfdb_query_unique(q._h, 0))
    want, rows = first_appearance(k[sel].tolist())
    assert q.count() == len(want)
    assert np.array_equal(q.materialize()[0], np.array(want, np.int64))
    assert np.array_equal(q.indices(), (np.flatnonzero(sel) + 1)[np.array(rows) - 1])
    # narrowed: another column of the same view at the first occurrences' rows
    v2 = p.d[("a", lambda c: c > 250_000), ["k", "iota", "s"]]
    q2 = v2._query()
    N.check(L.dfdb_query_unique(q2._h, 0))
    got = q2.materialize()
    fr = (np.flatnonzero(sel) + 1)[np.array(rows) - 1]
    assert np.array_equal(got[0], np.array(want, np.int64)) and np.array_equal(got[1], fr)
    ws, wb = p.O.strings_to_flat([p.cols["s"][r - 1] for r in fr])
    assert np.array_equal(got[2][0], ws) and np.array_equal(got[2][1], wb)
    q2.reset()
    assert q2.count() == int(sel.sum())                      # the full selection again
    # String keys, Float64 keys (isequal: one NaN, -0.0 apart from 0.0), nullable keys (missing is a value)
    qs = p.d[dfdb_mod.ALL, ["s"]]._query()
    N.check(L.dfdb_query_unique(qs._h, 0))
    sizes, data = qs.materialize()[0]
    assert p.O.flat_to_strings(sizes, data) == first_appearance(p.cols["s"])[0]
    qf = p.d[dfdb_mod.ALL, ["f"]]._query()
    N.check(L.dfdb_query_unique(qf._h, 0))
    wf, _ = first_appearance(p.cols["f"].tolist(), lambda v: "nan" if v != v else (v, bool(np.signbit(v))))
    gf = qf.materialize()[0]
    assert len(gf) == len(wf) and all((np.isnan(g) and np.isnan(w)) or (g == w and np.signbit(g) == np.signbit(w)) for g, w in zip(gf, wf))
    qm = p.d[dfdb_mod.ALL, ["m"]]._query()
    N.check(L.dfdb_query_unique(qm._h, 0))
    m = p.cols["m"]
    wm, _ = first_appearance([None if mk else int(vv) for vv, mk in zip(m.data.tolist(), np.ma.getmaskarray(m).tolist())])
    gm = qm.materialize()[0]
    assert [None if mk else int(vv) for vv, mk in zip(gm.data.tolist(), np.ma.getmaskarray(gm).tolist())] == wm


@pytest.mark.parametrize("chunk", CHUNKINGS)
def test_groupreduce_through_the_abi(pair, dfdb_mod, chunk):
    import dfdb.api as api
    p = pair
    p.d.ctx.set_option("ooc_chunk_blocks", chunk)
    a, x, k, s = p.cols["a"], p.cols["x"], p.cols["k"], p.cols["s"]
    sel = a > 100_000
    for by, vals in (("k", k), ("s", np.array(s, dtype=object))):
        for stat, col in (("count", None), ("sum", "a"), ("min", "x"), ("max", "a"), ("sum", "x")):
            names = [by] if col is None else [by, col]
            v = p.d[("a", lambda c: c > 100_000), names]
            keys, counts, vi, vf, vdt = api._groupreduce_raw(api._Query(v), col is not None, stat)      # dfdb_query_groupreduce + _fetch, one call each
            keys = list(keys) if by == "s" else keys.tolist()
            wk, _ = first_appearance(vals[sel].tolist())
            assert keys == wk, (by, stat)
            kk = vals[sel]
            for jj in sorted(set(list(range(min(50, len(wk)))) + list(range(max(0, len(wk) - 5), len(wk))))):
                key = wk[jj]
                rows = kk == key
                assert counts[jj] == int(rows.sum())
                if stat == "sum" and col == "a":
                    assert vi[jj] == int(a[sel][rows].sum())
                elif stat == "max":
                    assert vi[jj] == int(a[sel][rows].max())
                elif stat == "min":
                    assert vf[jj] == float(x[sel][rows].min())
                elif stat == "sum":
                    assert abs(vf[jj] - float(x[sel][rows].sum())) <= 64 * np.finfo(np.float64).eps * float(np.abs(x[sel][rows]).sum())


def test_count_reads_the_selection_columns_only(pair, dfdb_mod):
    """nrow(v) never reads a projection-only column (BlockRowsIterator, blocksiterator.jl:46-66): dfdb_query_read_stats shows one column's rows."""
    import dfdb._native as N
    p = pair
    n = p.nrows
    q = p.d[("a", lambda c: c > 899_999), dfdb_mod.ALL]._query()
    st = N.SizeStats()
    assert q.count() == int((p.cols["a"] > 899_999).sum())
    N.check(N.load().dfdb_query_read_stats(q._h, C.byref(st)))
    assert st.rows == n and st.uncompressed == n * 8, (st.rows, st.uncompressed)      # column a, once
    # no predicate: the first projection column
    q2 = p.d[dfdb_mod.jr(10, 20_000), ["x", "s"]]._query()
    assert q2.count() == 19_991
    N.check(N.load().dfdb_query_read_stats(q2._h, C.byref(st)))
    assert st.uncompressed <= 20 * 1000 * 8, st.uncompressed                         # 20 blocks of x; the range skipped the rest
    # the select_bitmap of such a view is refused, not faked
    buf = np.zeros(1024, np.uint64)
    with pytest.raises(NotImplementedError):
        N.check(N.load().dfdb_select_bitmap(q._h, buf.ctypes.data, N.MEM_HOST))


def test_materialize_with_the_hint_is_two_passes(pair, dfdb_mod):
    """count (with the materialize hint: + the projected String column's sizes, only the blocks that kept a row) then ONE materialize pass."""
    import dfdb._native as N
    p = pair
    L = N.load()
    sel = p.cols["a"] > 990_000
    v = p.d[("a", lambda c: c > 990_000), ["s", "x"]]
    q = v._query()
    got = q.materialize()                    # hint + count + string bytes + materialize
    from dfdb import ir
    ov, _ = apply_stages(p, [("pred", ir.col(0) > 990_000)], proj=[("s", ir.col(2)), ("x", ir.col(1))])
    same_cols(ov.materialize(), got)
    st = N.SizeStats()
    N.check(L.dfdb_query_read_stats(q._h, C.byref(st)))
    blocks_kept = len({i // 1000 for i in np.flatnonzero(sel)})
    nblocks = (p.nrows + 999) // 1000
    # pass 1: a whole + s in the kept blocks; pass 2: a whole + s, x in the kept blocks  (rows are summed per column read)
    assert st.rows <= 2 * p.nrows + 3 * blocks_kept * 1000, (st.rows, blocks_kept, nblocks)


def test_prepare_loads_only_the_required_columns(pair, dfdb_mod):
    """dfdb_query_prepare: `only the required columns are opened` (view.jl:183-190).  A 7-column table answers a 1-column count with ONE column resident;
    a budget below the decoded size picks the compressed-only form; a budget below that leaves the columns on disk and the same calls stream."""
    import dfdb._native as N
    L = N.load()
    p = pair
    t = dfdb_mod.open_table(p.path, load=False)
    try:
        how = C.c_int32(-1)
        q = t[("a", lambda c: c > 899_999), ["a"]]._query()
        N.check(L.dfdb_query_prepare(q._h, C.byref(how)))
        assert how.value == 1
        assert [t.resident(i) for i in range(t.ncols)] == [True] + [False] * (t.ncols - 1)
        assert t.resident_bytes()["decoded"] >= p.nrows * 8 and t.resident_bytes("x")["decoded"] == 0
        assert q.count() == int((p.cols["a"] > 899_999).sum())
        N.check(L.dfdb_query_prepare(q._h, C.byref(how)))
        assert how.value == 0
        # a second view needs x too: only x is added
        q2 = t[("a", lambda c: c > 899_999), ["x"]]._query()
        N.check(L.dfdb_query_prepare(q2._h, C.byref(how)))
        assert how.value == 1 and t.resident(1) and not t.resident(2)
        assert np.array_equal(q2.materialize()[0], p.cols["x"][p.cols["a"] > 899_999])
        # unload everything again
        N.check(L.dfdb_table_unload(t._h, None, 0))
        assert not any(t.resident(i) for i in range(t.ncols))
        assert q2.count() == int((p.cols["a"] > 899_999).sum())          # (streams now)
    finally:
        t.close()


def test_budget_routes_decoded_compressed_streamed(oracle, dfdb_mod, tmp_path):
    import dfdb._native as N
    L = N.load()
    n = 3_000_000                                    # 24 MB decoded per column, ~9 MB compressed (values below 1e6)
    a = oracle.gen_i64(col_seed(0), 0, n)
    b = np.arange(n, dtype=np.int64)
    path = str(tmp_path / "big")
    src = dfdb_mod.DFTable.from_columns({"a": a, "b": b}, block_size=65536)
    src.save(path)
    src.close()
    want = int((a > 899_999).sum())
    seen = {}
    for budget_mb, expect in ((0, 1), (40, 2), (1, 3)):
        ctx = dfdb_mod.Context()
        ctx.set_option("hbm_budget_mb", budget_mb)
        t = dfdb_mod.open_table(path, load=False, ctx=ctx)
        try:
            q = t[("a", lambda c: c > 899_999), ["a", "b"]]._query()
            how = C.c_int32(-1)
            N.check(L.dfdb_query_prepare(q._h, C.byref(how)))
            seen[budget_mb] = how.value
            rb = t.resident_bytes()
            if how.value == 2:
                assert rb["decoded"] == 0 and rb["compressed"] > 0
            if how.value == 3:
                assert rb["decoded"] == 0 and rb["compressed"] == 0
            assert q.count() == want
            got = q.materialize()
            assert np.array_equal(got[0], a[a > 899_999]) and np.array_equal(got[1], b[a > 899_999])
            assert q.aggregate(N.AGG_SUM, 1) == int(b[a > 899_999].sum())
        finally:
            t.close()
            ctx.close()
    assert seen == {0: 1, 40: 2, 1: 3}, seen


def test_the_python_mirror_has_no_chunk_loops_left(dfdb_mod):
    """api.py's streamed helpers are one-call wrappers now: the only `for part in` loops left belong to the explicit Stream iterator's users."""
    import inspect
    import dfdb.api as api
    for fn in (api.nrow, api.nrow_streamed, api.materialize_streamed, api.materialize, api.groupreduce, api.DFColumn.unique, api.DFColumn._aggregate):
        src = inspect.getsource(fn)
        assert "Stream(" not in src and "for part in" not in src, fn.__name__
    assert not hasattr(api, "_groupreduce_streamed")


def test_minimum_and_maximum_of_signed_zeros_do_not_depend_on_the_order(dfdb_mod):
    """Base.min(-0.0, 0.0) == -0.0 and Base.max == 0.0 whichever comes first.  The device reductions kept the first zero they met (found in round 6 by the
    block-streamed aggregates, whose chunking changed the grid): both the plain reduction and the one the scan carries along (dfdb_query_hint_aggregate)."""
    import dfdb._native as N
    for first, second in ((0.0, -0.0), (-0.0, 0.0)):
        n = 70_000
        f = np.full(n, 5.0); f[::2] = first; f[1::2] = second
        g = -f                                                   # zeros again, -5.0 below them: the maximum is a zero
        t = dfdb_mod.DFTable.from_columns({"f": f, "g": g}, block_size=4096)
        for hinted in (False, True):
            q = t[("f", lambda c: c < 1.0), ["f"]]._query()      # keeps the zeros only
            if not hinted:
                q.execute()                                      # the scan runs before the aggregate is known: the plain reduction
            mn = q.aggregate(N.AGG_MIN, 0)
            q2 = t[("g", lambda c: c > -1.0), ["g"]]._query()
            if not hinted:
                q2.execute()
            mx = q2.aggregate(N.AGG_MAX, 0)
            assert mn == 0.0 and np.signbit(mn), (first, hinted, mn)
            assert mx == 0.0 and not np.signbit(mx), (first, hinted, mx)
        t.close()


def test_julia_shim_routing_transcription(pair, dfdb_mod):
    """The call sequence of julia/DataFrameDBsAMD.jl's with_query (round 6), transcribed: dfdb_table_open (NO load) -> dfdb_query_new + stages + projection ->
    dfdb_query_prepare -> the consumer's calls; on OutOfMemoryError: dfdb_table_unload(all) + dfdb_query_reset + the consumer's calls again.  Both legs give
    the oracle's answer, the first with exactly the required columns resident, the second with none."""
    import dfdb._native as N
    from dfdb import ir
    L = N.load()
    p = pair
    ov, _ = apply_stages(p, [("pred", (ir.col(0) > 640_000) & (ir.col(2) == "sony"))], proj=[("x", ir.col(1)), ("s", ir.col(2))])
    t = dfdb_mod.open_table(p.path, load=False)                                    # device_table(t): open only
    try:
        v = t[(("a", "s"), lambda a, s: (a > 640_000) & (s == "sony")), ["x", "s"]]
        q = v._query()                                                             # dfdb_query_new, dfdb_query_add_predicate, dfdb_query_set_projection
        how = C.c_int32(-1)
        N.check(L.dfdb_query_prepare(q._h, C.byref(how)))                         # PREPARE
        assert how.value == 1 and [t.resident(i) for i in range(3)] == [True, True, True] and not any(t.resident(i) for i in range(3, t.ncols))
        def consumer():                                                            # gpu_materialize_columns: hint, count, coltype, string bytes, materialize
            return q.materialize()
        first = consumer()
        same_cols(ov.materialize(), first)
        N.check(L.dfdb_table_unload(t._h, None, 0))                                # the catch branch: UNLOAD, RESET, f again
        q.reset()
        assert not any(t.resident(i) for i in range(t.ncols))
        same_cols(ov.materialize(), consumer())
        assert not any(t.resident(i) for i in range(t.ncols)), "the retry streamed: nothing became resident"
    finally:
        t.close()


def test_errors_surface_through_the_ordinary_entry_points_and_leave_the_handles_usable(oracle, dfdb_mod, tmp_path):
    """The reference's errors met while iterating blocks — DivideError inside a predicate, "decompression error" (BlockStreams.jl:112) — come out of dfdb_count /
    dfdb_materialize / dfdb_aggregate / dfdb_query_unique over a table that is not resident exactly as they do out of the explicit stream; the internal stream is
    closed (parked) behind the failure, and the same context answers the next query, streamed again."""
    import struct
    import dfdb._native as N
    n, bs = 40_000, 4096
    t = oracle.Table(block_size=bs)
    t.add_column("a", np.arange(1, n + 1, dtype=np.int64))
    t.add_column("z", (np.arange(n) < 30_000).astype(np.int64))       # zeros from row 30 001 on
    path = str(tmp_path / "tb")
    t.save(path)
    tb = dfdb_mod.open_table(path, load=False)
    tb.ctx.set_option("ooc_chunk_blocks", 2)
    try:
        bad = tb[(tb.a % tb.z) == 0, ["a"]]
        for call in (lambda q: q.count(), lambda q: q.indices(), lambda q: q.materialize(), lambda q: q.aggregate(N.AGG_SUM, 0),
                     lambda q: N.check(N.load().dfdb_query_unique(q._h, 0))):
            with pytest.raises(ZeroDivisionError):
                call(dfdb_mod.api._Query(bad))
        # rows before the first zero divisor only: no error, and the answer is the oracle's arithmetic
        ok = tb[dfdb_mod.jr(1, 30_000), ["a", "z"]]
        ok2 = dfdb_mod.selection(ok, (ok.a % ok.z) == 0)
        assert dfdb_mod.api._Query(ok2).count() == 30_000             # a % 1 == 0 everywhere there
        # a corrupt LZ4 payload in block 5 of column a
        f = tmp_path / "tb" / "1.bin"
        raw = bytearray(f.read_bytes())
        pos = 8 + 4 + len("Int64")
        for _ in range(5):
            rows, origin, comp = struct.unpack_from("<iqq", raw, pos)
            pos += 20 + comp
        raw[pos + 20: pos + 24] = b"\xff\xff\xff\xff"
        f.write_bytes(bytes(raw))
        tb2 = dfdb_mod.open_table(path, load=False)
        try:
            for call in (lambda q: q.count(), lambda q: q.materialize(), lambda q: q.aggregate(N.AGG_MAX, 0)):
                with pytest.raises(dfdb_mod.DfdbError, match="decompression error"):
                    call(dfdb_mod.api._Query(tb2[dfdb_mod.ALL, ["a"]]))
            # column z is intact: a view that needs only z is answered (only the required columns are ever read)
            assert dfdb_mod.api._Query(tb2[("z", lambda z: z == 0), ["z"]]).count() == 10_000
            # a leading range that ends before the damaged block never reads it (skip_if_can / is_finished, selection.jl:177-196)
            early = tb2[dfdb_mod.jr(1, 4 * bs), ["a"]]
            assert np.array_equal(dfdb_mod.api._Query(early).materialize()[0], np.arange(1, 4 * bs + 1))
        finally:
            tb2.close()
    finally:
        tb.close()


def test_empty_and_tiny_tables_out_of_core(oracle, dfdb_mod, tmp_path):
    """zero rows, one row, one short block: the streamed entry points agree with the resident ones"""
    import dfdb._native as N
    for n in (0, 1, 999):
        t = oracle.Table(block_size=1000)
        t.add_column("a", np.arange(n, dtype=np.int64) * 3)
        t.add_column("s", ["s%d" % (i % 5) for i in range(n)])
        path = str(tmp_path / ("t%d" % n))
        t.save(path)
        lazy, res = dfdb_mod.open_table(path, load=False), dfdb_mod.open_table(path)
        try:
            for tb in (lazy, res):
                v = tb[("a", lambda c: c % 2 == 0), dfdb_mod.ALL]
                q = dfdb_mod.api._Query(v)
                want = (np.arange(n) * 3) % 2 == 0
                assert q.count() == int(want.sum()) and np.array_equal(q.indices(), np.flatnonzero(want) + 1)
                got = q.materialize()
                assert np.array_equal(got[0], (np.arange(n) * 3)[want])
                assert dfdb_mod.api._Query(v).aggregate(N.AGG_SUM, 0) == int((np.arange(n) * 3)[want].sum())
                assert list(tb.s.unique()) == ["s%d" % i for i in range(min(n, 5))]
            assert not lazy.resident(0)
        finally:
            lazy.close(); res.close()


def test_a_new_table_made_from_a_view_that_is_not_resident(pair, oracle, dfdb_mod, tmp_path):
    """create_table(path; from = view) / add_column!(t, name, lazy_col) (creators.jl:18-60, table.jl:96-124) over a view whose columns are NOT resident:
    dfdb_table_add_from_query fills the new resident columns from the block stream (plain, computed, String and nullable columns), the saved table reads back
    through the oracle's reader as what numpy says."""
    p = pair
    p.d.ctx.set_option("ooc_chunk_blocks", 5)
    a, x, s, m = p.cols["a"], p.cols["x"], p.cols["s"], p.cols["m"]
    sel = a > 600_000
    v = p.d[("a", lambda c: c > 600_000), dfdb_mod.ALL]
    dst = dfdb_mod.DFTable.new(block_size=1000, ctx=p.d.ctx)
    try:
        dst.add_column_from("a", v.a)
        dst.add_column_from("x2", v.x * 2.0)
        dst.add_column_from("s", v.s)
        dst.add_column_from("m", v.m)
        assert not any(p.d.resident(i) for i in range(p.d.ncols))
        out = str(tmp_path / "copy")
        dst.save(out)
        back = oracle.Table.open(out)
        ba, bx2, bs, bm = back.view().materialize()
        assert np.array_equal(ba, a[sel]) and np.array_equal(bx2, x[sel] * 2.0)
        assert oracle.flat_to_strings(*bs) == [s[i] for i in np.flatnonzero(sel)]
        assert np.array_equal(np.ma.getmaskarray(bm), np.ma.getmaskarray(m)[sel]) and np.array_equal(bm.compressed(), m[sel].compressed())
    finally:
        dst.close()
