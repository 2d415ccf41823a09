"""unique / groupreduce (k_unique.hip, k_dict.hip): every form against first appearance, the optimistic inserts and their redo paths, the LDS group tables, the fetch's guards.
(re-filed by component in round 6 from the round-named files; no test body changed)"""


import ctypes as C

import numpy as np
import pytest
import pandas as pd


pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ unique / groupreduce: the table sized by the distinct values, and the dense form
def _julia_unique(values, missing=None):
    from test_gpu_parity import julia_unique
    return julia_unique(values, missing)


def test_groupreduce_fetch_refuses_a_selection_that_changed(dfdb_mod, ctx):
    """between dfdb_query_groupreduce and its fetch the query holds the narrowed selection (first occurrences) and the full one aside; a
    reset / execute / new stage in between makes the pending result stale — the fetch must refuse, not restore the old selection over the new one"""
    from dfdb import _native as N
    L = N.load()
    k = (np.arange(10_000) % 7).astype(np.int64)
    t = dfdb_mod.DFTable.from_columns({"k": k, "v": np.arange(10_000, dtype=np.int64)})
    for spoil in ("reset", "execute", "add"):
        q = dfdb_mod.DFView(t)._query() if spoil != "add" else dfdb_mod.DFView(t)[dfdb_mod.jr(1, 9000), dfdb_mod.ALL]._query()
        ng, kb = C.c_int64(), C.c_int64()
        N.check(L.dfdb_query_groupreduce(q._h, 0, 1, N.AGG_SUM, C.byref(ng), C.byref(kb)))
        assert ng.value == 7
        if spoil == "reset":
            q.reset()
        elif spoil == "execute":
            q.execute()
        else:
            N.check(L.dfdb_query_add_range(q._h, 1, 1, 100))
        out = N.OutCol(); keys = np.zeros(7, np.int64); out.data, out.memkind = keys.ctypes.data, N.MEM_HOST
        cnt = np.zeros(7, np.int64)
        with pytest.raises(ValueError, match="dfdb_query_groupreduce has not been called"):
            N.check(L.dfdb_query_groupreduce_fetch(q._h, C.byref(out), cnt.ctypes.data, None, None))
        assert q.count() == (10_000 if spoil != "add" else 100)          # and the query answers for its own selection
    t.close()


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


def test_groupreduce_by_a_string_key_skips_the_inserts_it_does_not_need(oracle, dfdb_mod, ctx):
    """groupreduce by a String key (aggregate.jl:1-36): when the second chunk of rows brings no string the first had not, the remaining rows are not inserted into
    the hash table — the accumulate pass meets every row anyway and says so if a string is missing, in which case everything runs again the slow way.  Same
    groups, in order of first appearance, same counts and sums: with every key early (optimistic path taken), with a key that first turns up in the last rows
    (found missing, redone), with the redo forced, with the option off."""
    dfdb = dfdb_mod
    n = 700_000
    rng = np.random.default_rng(31)
    brands = ["apple", "samsung", "huawei", "microsoft", "dell", "xbox", "sony", "intel", "lenovo", "asus", "a-rather-long-brand-name-over-16-bytes"]
    k = rng.integers(0, len(brands), n)
    early = [brands[i] for i in k]
    late = list(early); late[-3] = "late-comer"; late[-1] = "zz"
    a = rng.integers(-1000, 1000, n).astype(np.int64)
    ctx.set_option("unique_chunk_tiles", 8)                 # chunks of 8 K, 128 K, the rest: the rest is what the optimistic path skips
    ctx.profile(True)
    try:
        for name, keys in (("early", early), ("late", late)):
            t = dfdb.DFTable.from_columns({"s": keys, "a": a}, block_size=65536)
            arr = np.array(keys, dtype=object)
            first = {}
            for i, v in enumerate(keys):
                if v not in first:
                    first[v] = i
            order = sorted(first, key=first.get)
            for opt in (1, 2, 0):
                ctx.set_option("groupreduce_optimistic", opt)
                before, _ = ctx.profile_get("unique_insert")
                g = dfdb.groupreduce(t, "s", "a", "sum")
                after, _ = ctx.profile_get("unique_insert")
                assert list(g["s"]) == order, (name, opt)
                for v, c_, s_ in zip(g["s"], g["count"].to_numpy(), g["sum"].to_numpy()):
                    m = arr == v
                    assert c_ == int(m.sum()) and s_ == int(a[m].sum()), (name, opt, v)
                launches = after - before
                # a Float64 key (floats always take the hash table) strikes it too: the accumulate pass probes every row's key anyway and reports one without a slot
                fkeys = np.array([float(len(v)) * 0.25 if v != "late-comer" else -7.5 for v in keys])
                if "f" not in t.names():
                    t.add_column("f", fkeys)
                b3, _ = ctx.profile_get("unique_insert")
                gf = dfdb.groupreduce(t, "f", "a", "sum")
                a3, _ = ctx.profile_get("unique_insert")
                forder = list(dict.fromkeys(fkeys.tolist()))
                assert gf["f"].tolist() == forder, (name, opt)
                for v, c_, s_ in zip(gf["f"], gf["count"].to_numpy(), gf["sum"].to_numpy()):
                    m = fkeys == v
                    assert c_ == int(m.sum()) and s_ == int(a[m].sum()), (name, opt, v)
                assert a3 - b3 == launches, (name, opt, a3 - b3, launches)
                # plain unique over the same column strikes the same bargain (its compare pass meets every row)
                b2, _ = ctx.profile_get("unique_insert")
                u = list(t.s.unique())
                a2, _ = ctx.profile_get("unique_insert")
                assert u == order, (name, opt)
                assert a2 - b2 == launches, (name, opt, a2 - b2, launches)
                if opt == 1 and name == "early":
                    assert launches == 2, launches                # two prefix chunks, the rest skipped
                elif opt == 0:
                    assert launches == 3, launches
                else:
                    assert launches == 2 + 3, launches            # the optimistic attempt, then everything again
            t.close()
    finally:
        ctx.profile(False)
        ctx.set_option("groupreduce_optimistic", 1)
        ctx.set_option("unique_chunk_tiles", 0)


def test_groupreduce_by_an_integer_key_with_the_group_table_in_lds(oracle, dfdb_mod, ctx):
    """groupreduce by an Int64 key of a few thousand values (aggregate.jl:1-36): the dense form's group-number table — the occupied span of it — is copied into LDS
    beside the accumulators (k_group_acc_dense_lds).  Negative keys, a nullable key (missing is a group, the table's last entry), a filtered view, every
    statistic over Int64 and Float64 values == a numpy restatement of first-appearance numbering; a narrow value column takes the general kernel."""
    from test_gpu_parity import _np_group_ids, _np_groupreduce
    dfdb = dfdb_mod
    rng = np.random.default_rng(77)
    n = 200_000
    k = rng.integers(-1500, 1500, n).astype(np.int64) * 3 + 7            # ~3000 values spread over a span of 9000
    km = np.ma.masked_array(k.copy(), mask=rng.random(n) < 0.2)
    c = rng.integers(-1000, 1000, n).astype(np.int64)
    x = rng.normal(size=n) * 100
    u8 = rng.integers(0, 255, n).astype(np.uint8)
    fk = k * 0.5; fk[rng.random(n) < 0.01] = np.nan                   # Float64 keys take the hash table: the groups' keys go into an LDS table (k_group_acc_hash_lds)
    fkm = np.ma.masked_array(k * 0.25, mask=rng.random(n) < 0.2)
    t = dfdb.DFTable.from_columns({"k": k, "km": km, "fk": fk, "fkm": fkm, "c": c, "x": x, "u8": u8}, block_size=4096)
    ctx.profile(True)
    try:
        for view, sel in ((t[dfdb.ALL, dfdb.ALL], np.ones(n, bool)), (t[t.c > 0, dfdb.ALL], c > 0)):
            for by, keys in (("fk", fk), ("fkm", fkm)):
                ids = _np_group_ids([keys[i] for i in np.nonzero(sel)[0]])
                for col, vals, stats in (("c", c, ("count", "sum", "min", "max")), ("x", x, ("sum", "max")), ("u8", u8, ("sum",))):
                    for stat in stats:
                        before, _ = ctx.profile_get("group_accumulate.hash_lds")
                        got = dfdb.groupreduce(view, by, col, stat)
                        after, _ = ctx.profile_get("group_accumulate.hash_lds")
                        assert after - before == (0 if col == "u8" else 1), (by, col, stat)
                        order, cnt, want = _np_groupreduce(ids, vals[sel], stat)
                        gk = [None if kk is None or kk is pd.NA else ("nan" if kk != kk else float(kk)) for kk in [None if (isinstance(z, float) and False) else z for z in got[by].tolist()]]
                        wk = [None if (kk is np.ma.masked or kk is None) else ("nan" if kk != kk else float(kk)) for kk in order]
                        if by == "fkm":                                # (a masked value comes back as NaN in a float frame column: told apart by the masked order entry)
                            gk = [None if (w_ is None) else g_ for g_, w_ in zip(gk, wk)]
                        assert gk == wk, (by, col, stat)
                        assert got["count"].tolist() == cnt.tolist(), (by, col, stat)
                        if stat != "count":
                            g = got[stat].to_numpy()
                            if col == "x" and stat == "sum":
                                assert np.allclose(g, want, rtol=1e-9, atol=1e-6), (by, col, stat)
                            else:
                                assert np.array_equal(g.astype(np.float64), want.astype(np.float64)), (by, col, stat)
            for by, keys in (("k", k), ("km", km)):
                ids = _np_group_ids([keys[i] for i in np.nonzero(sel)[0]])
                for col, vals, stats in (("c", c, ("count", "sum", "min", "max")), ("x", x, ("sum", "min", "max")), ("u8", u8, ("sum",))):
                    for stat in stats:
                        before, _ = ctx.profile_get("group_accumulate.dense_lds")
                        got = dfdb.groupreduce(view, by, col, stat)
                        after, _ = ctx.profile_get("group_accumulate.dense_lds")
                        assert after - before == (0 if col == "u8" else 1), (by, col, stat)      # (a narrow value column: the general kernel)
                        order, cnt, want = _np_groupreduce(ids, vals[sel], stat)
                        assert 1024 < len(order) <= 9216
                        gk = [None if pd.isna(kk) else int(kk) for kk in got[by].tolist()]
                        wk = [None if (kk is np.ma.masked or kk is None) else int(kk) for kk in order]
                        assert gk == wk, (by, col, stat)
                        assert got["count"].tolist() == cnt.tolist(), (by, col, stat)
                        if stat != "count":
                            g = got[stat].to_numpy()
                            if col == "x" and stat == "sum":
                                assert np.allclose(g, want, rtol=1e-9, atol=1e-6), (by, col, stat)
                            else:
                                assert np.array_equal(g.astype(np.float64), want.astype(np.float64)), (by, col, stat)
    finally:
        ctx.profile(False)
    t.close()


def test_groupreduce_by_an_integer_key_makes_its_groups_from_the_head_of_the_column(oracle, dfdb_mod, ctx):
    """groupreduce by an Int64 key (aggregate.jl:1-36), dense form: the first rows / group numbers come from the head of the column, the accumulate pass — the one
    with the table in LDS — meets every row and raises a flag for a key without a group, after which everything runs again over every row.  Same groups in order
    of first appearance, same counts and sums: every key early (the head's table is used), a key / a missing value / a key outside the sampled span that first
    turn up behind the head (found, redone), the redo forced, the option off; a narrow value column is not tried (the LDS form would not take it)."""
    from test_gpu_parity import _np_group_ids, _np_groupreduce
    dfdb = dfdb_mod
    rng = np.random.default_rng(91)
    n = 80_000
    k_early = rng.integers(0, 40, n).astype(np.int64) * 5 - 60
    k_late = k_early.copy(); k_late[-7] = 33                          # inside the span, never seen in the head
    k_out = k_early.copy(); k_out[77 * 1024 + 5] = 10_000_019         # far outside what the sample saw (tile 77: the forced sample takes the even tiles)
    km = np.ma.masked_array(k_early.copy(), mask=np.zeros(n, bool)); km.mask[-5] = True          # the only missing value sits behind the head
    c = rng.integers(-1000, 1000, n).astype(np.int64)
    u8 = rng.integers(0, 255, n).astype(np.uint8)
    t = dfdb.DFTable.from_columns({"early": k_early, "late": k_late, "out": k_out, "km": km, "c": c, "u8": u8}, block_size=4096)
    ctx.set_option("dense_head_tiles", 4)                            # a head of 4096 rows; the column has 79 tiles
    ctx.set_option("unique_dense_sample", 2)                         # (a table this small is not sampled otherwise, and only a sampled span is trusted beyond the head)
    ctx.profile(True)
    try:
        for by, keys, found_late in (("early", k_early, False), ("late", k_late, True), ("out", k_out, True), ("km", km, True)):
            ids = _np_group_ids(list(keys))
            for opt in (1, 2, 0):
                ctx.set_option("groupreduce_optimistic", opt)
                for col, vals, stat in (("c", c, "sum"), ("c", c, "min"), ("c", c, "count"), ("u8", u8, "sum")):
                    h0, _ = ctx.profile_get("group_accumulate.head_table"); r0, _ = ctx.profile_get("group_accumulate.head_redo")
                    got = dfdb.groupreduce(t, by, col, stat)
                    h1, _ = ctx.profile_get("group_accumulate.head_table"); r1, _ = ctx.profile_get("group_accumulate.head_redo")
                    order, cnt, want = _np_groupreduce(ids, vals, stat)
                    gk = [None if pd.isna(kk) else int(kk) for kk in got[by].tolist()]
                    wk = [None if (kk is np.ma.masked or kk is None) else int(kk) for kk in order]
                    assert gk == wk and got["count"].tolist() == cnt.tolist(), (by, opt, col, stat)
                    if stat != "count":
                        assert np.array_equal(got[stat].to_numpy().astype(np.int64), want.astype(np.int64)), (by, opt, col, stat)
                    if opt == 0 or col == "u8":                              # (a narrow value column: the LDS form would not take it, no head table is tried)
                        assert (h1 - h0, r1 - r0) == (0, 0), (by, opt, col)
                    elif opt == 2 or found_late:
                        assert (h1 - h0, r1 - r0) == (0, 1), (by, opt, col, h1 - h0, r1 - r0)      # tried, redone over every row
                    else:
                        assert (h1 - h0, r1 - r0) == (1, 0), (by, opt, col, h1 - h0, r1 - r0)
    finally:
        ctx.profile(False)
        ctx.set_option("groupreduce_optimistic", 1)
        ctx.set_option("dense_head_tiles", 4096)
        ctx.set_option("unique_dense_sample", 1)
    t.close()
