"""unique(col) by radix partition (csrc/k_radix.hip, round 6): the many-distinct-values form of Base.unique over Base.iterate(::DFColumn)
(src/tables/column.jl:102-126; docs/src/index.md:479-487).  Same answers as the hash-table form and as first appearance computed on the host: isequal keys
(one NaN, -0.0 apart from 0.0, missing is a value), the key whose image cannot be stored (all ones), one-tile partitions, a partition that outgrows its LDS
table (the engine goes back to the hash table over the untouched selection)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def first_rows(img: np.ndarray, sel: np.ndarray) -> np.ndarray:
    """0-based rows of the first occurrence of every distinct image among the selected rows, ascending"""
    rows = np.flatnonzero(sel)
    _, idx = np.unique(img[rows], return_index=True)
    return np.sort(rows[idx])


def image(vals: np.ndarray, missing=None) -> np.ndarray:
    if vals.dtype.kind == "f":
        u = vals.view(np.uint64).copy()
        u[np.isnan(vals)] = 0x7ff8000000000000
    else:
        u = vals.astype(np.int64).view(np.uint64).copy()
    u = u.astype(object) if False else u
    if missing is not None:
        # missing is one more value: give it an image no value has (the test data avoids it)
        u = u.copy(); u[missing] = np.uint64(0x123456789ABCDEF1)
    return u


def run_unique(dfdb, t, view, radix):
    import ctypes as C
    import dfdb._native as N
    t.ctx.set_option("unique_dense", 0)
    t.ctx.set_option("unique_radix", radix)
    t.ctx.profile(True)
    try:
        q = view._query()
        N.check(N.load().dfdb_query_unique(q._h, 0))
        rows = q.indices() - 1
        taken = t.ctx.profile_get("unique_radix.taken")[0]
        fell = t.ctx.profile_get("unique_radix.fell_back")[0]
    finally:
        t.ctx.profile(False)
        t.ctx.set_option("unique_dense", 1)
        t.ctx.set_option("unique_radix", 1)
    return rows, taken, fell


@pytest.mark.parametrize("kind", ["int", "float", "nullable"])
def test_radix_unique_is_first_appearance(dfdb_mod, ctx, kind):
    rng = np.random.default_rng(7)
    n = 2_500_123
    a = rng.integers(0, 1_000_000, n).astype(np.int64)
    if kind == "int":
        k = (rng.integers(0, 300_000, n) * 40_503 - 5_000_000_000).astype(np.int64)
        k[rng.random(n) < 0.001] = -1                                   # image 0xFFFF…F: the key the table cannot store (aux[0])
        col, img = k, image(k)
    elif kind == "float":
        k = rng.integers(0, 200_000, n).astype(np.float64) / 8.0
        k[rng.random(n) < 0.01] = np.nan
        k[rng.random(n) < 0.01] = -0.0
        k[rng.random(n) < 0.001] = -np.nan
        col, img = k, image(k)
    else:
        base = (rng.integers(0, 150_000, n) * 3).astype(np.int64)
        miss = rng.random(n) < 0.05
        col, img = np.ma.masked_array(base, mask=miss), image(base, miss)
    t = dfdb_mod.DFTable.from_columns({"a": a, "k": col}, block_size=65536, ctx=ctx)
    try:
        for label, view, sel in (("all", t[dfdb_mod.ALL, ["k"]], np.ones(n, bool)), ("pred", t[("a", lambda c: c > 400_000), ["k"]], a > 400_000)):
            want = first_rows(img, sel)
            got, taken, fell = run_unique(dfdb_mod, t, view, 2)
            assert taken == 1 and fell == 0, (label, taken, fell)
            assert np.array_equal(got, want), (kind, label, len(got), len(want))
            got_h, taken_h, _ = run_unique(dfdb_mod, t, view, 0)        # the hash table over the same view
            assert taken_h == 0 and np.array_equal(got_h, want), (kind, label)
            got_f, taken_f, fell_f = run_unique(dfdb_mod, t, view, 3)   # as if a partition had overflowed: back to the hash table, selection intact
            assert taken_f == 0 and fell_f == 1 and np.array_equal(got_f, want), (kind, label)
        # the values come out in order of first appearance through the ordinary materialize
        ctx.set_option("unique_dense", 0); ctx.set_option("unique_radix", 2)
        try:
            vals = t.k.unique()
        finally:
            ctx.set_option("unique_dense", 1); ctx.set_option("unique_radix", 1)
        want_rows = first_rows(img, np.ones(n, bool))
        if kind == "nullable":
            gm = np.ma.getmaskarray(vals)
            assert np.array_equal(gm, np.ma.getmaskarray(col)[want_rows]) and np.array_equal(vals.data[~gm], col.data[want_rows][~gm])
        elif kind == "float":
            w = col[want_rows]
            assert np.array_equal(np.isnan(vals), np.isnan(w)) and np.array_equal(np.where(np.isnan(w), 0, w).view(np.uint64), np.where(np.isnan(vals), 0, vals).view(np.uint64))
        else:
            assert np.array_equal(vals, col[want_rows])
    finally:
        t.close()


@pytest.mark.parametrize("dt", ["int8", "int16", "int32", "uint8", "uint16", "uint32", "float32", "uint64"])
def test_radix_unique_narrow_and_unsigned_keys(dfdb_mod, ctx, dt):
    """the keys that are not eight raw bytes (k_radix.hip kKindAny: rkey_fixed makes the image) and UInt64, over a table whose last tile is partial"""
    rng = np.random.default_rng(11)
    n = 1_300_007
    if dt == "float32":
        k = (rng.integers(0, 90_000, n).astype(np.float32) / np.float32(4.0))
        k[rng.random(n) < 0.01] = np.nan
        k[rng.random(n) < 0.01] = -0.0
        u = k.view(np.uint32).astype(np.uint64); u[np.isnan(k)] = 0x7fc00000
        img = u
    elif dt == "uint64":
        k = (rng.integers(0, 250_000, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        k[rng.random(n) < 0.001] = np.uint64(0xFFFFFFFFFFFFFFFF)          # the image the table cannot store
        img = k.copy()
    else:
        info = np.iinfo(dt)
        k = rng.integers(max(info.min, -60_000), min(info.max, 60_000) + 1, n).astype(dt)   # (-1 of a signed type is the unstorable image too)
        img = k.astype(np.int64).view(np.uint64).copy()
    a = rng.integers(0, 100, n).astype(np.int64)
    t = dfdb_mod.DFTable.from_columns({"a": a, "k": k}, block_size=65536, ctx=ctx)
    try:
        for view, sel in ((t[dfdb_mod.ALL, ["k"]], np.ones(n, bool)), (t[("a", lambda c: c < 37), ["k"]], a < 37)):
            got, taken, fell = run_unique(dfdb_mod, t, view, 2)
            assert taken == 1 and fell == 0, (dt, taken, fell)
            assert np.array_equal(got, first_rows(img, sel)), dt
    finally:
        t.close()


def test_radix_unique_small_tables_and_one_tile_partitions(dfdb_mod, ctx):
    for n in (1, 63, 1024, 8191, 8193, 70_001):
        k = (np.arange(n, dtype=np.int64) * 2_654_435_761) % max(1, n // 3 + 1)
        t = dfdb_mod.DFTable.from_columns({"k": k}, block_size=4096, ctx=ctx)
        try:
            got, taken, fell = run_unique(dfdb_mod, t, t[dfdb_mod.ALL, ["k"]], 2)
            assert taken == 1 and fell == 0
            assert np.array_equal(got, first_rows(image(k), np.ones(n, bool))), n
        finally:
            t.close()


@pytest.mark.parametrize("share", [0.4, 0.05, 0.004])
def test_radix_unique_with_a_hot_value(dfdb_mod, ctx, share):
    """a value that a large part of the rows hold would make one partition — one workgroup's work — of all those rows: the partition pass keeps such a value in one of
    a workgroup's LDS slots (its rows never become records) and the table pass gets one list entry per workgroup (k_radix.hip, hot keys); the first occurrences are
    everybody else's, under a predicate too"""
    rng = np.random.default_rng(5)
    n = 3_000_011
    k = (rng.integers(0, 400_000, n) * 7 + 11).astype(np.int64)
    k[rng.random(n) < share] = 123_456_789
    a = rng.integers(0, 10, n).astype(np.int64)
    t = dfdb_mod.DFTable.from_columns({"a": a, "k": k}, block_size=65536, ctx=ctx)
    try:
        for view, sel in ((t[dfdb_mod.ALL, ["k"]], np.ones(n, bool)), (t[("a", lambda c: c > 3), ["k"]], a > 3)):
            got, taken, fell = run_unique(dfdb_mod, t, view, 2)
            assert (taken, fell) == (1, 0)                           # (0.4 and 0.05: the sample calls the column skewed and the kernels with the hot keys' slots run; 0.004: the plain ones)
            assert np.array_equal(got, first_rows(image(k), sel)), share
    finally:
        t.close()


def test_radix_unique_partition_overflow_falls_back(dfdb_mod, ctx):
    """20 M distinct keys over 2048 partitions = ~9 800 per 8192-slot table: the unique pass raises its flag, the selection is put back and the hash table answers."""
    n = 20_000_000
    k = (np.arange(n, dtype=np.int64) * 7919) ^ 0x5DEECE66D
    a = np.arange(n, dtype=np.int64) % 10
    t = dfdb_mod.DFTable.from_columns({"a": a, "k": k}, block_size=65536, ctx=ctx)
    try:
        got, taken, fell = run_unique(dfdb_mod, t, t[("a", lambda c: c < 9), ["k"]], 2)
        assert (taken, fell) == (0, 1)
        assert np.array_equal(got, np.flatnonzero(a < 9))                 # every key is distinct: every selected row is a first occurrence
    finally:
        t.close()
