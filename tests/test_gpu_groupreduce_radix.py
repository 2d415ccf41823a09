"""groupreduce over MORE groups than a workgroup's LDS accumulators hold (9216), by radix (csrc/k_radix.hip: 20-byte {key, row, value} records, a table with
accumulators per partition in LDS, results placed by the rank of a group's first row; csrc/query.cpp group_radix).  The reference's groupreduce numbers the groups
in order of first appearance and stops (src/tables/aggregate.jl:1-36: it prints the map); counts and one statistic per group are this repository's completion.
Checked against a numpy restatement: groups in order of first appearance, exact counts, integer sums / extrema exact, Float64 sums within n * eps * sum|x| (the
order of the additions is not fixed), isequal keys (one NaN, -0.0 apart from 0.0), the key whose image cannot be stored (-1), a predicate, and the same answers
from the form it replaces (global atomics: ctx option unique_radix = 0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def expect(keys_img, sel, vals, stat):
    rows = np.flatnonzero(sel)
    uniq, first, inv = np.unique(keys_img[rows], return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                    # groups in order of first appearance
    rank = np.empty(len(uniq), np.int64); rank[order] = np.arange(len(uniq))
    gid = rank[inv]
    ng = len(uniq)
    cnt = np.bincount(gid, minlength=ng)
    first_rows = rows[first[order]]
    if stat == "count":
        return first_rows, cnt, None
    v = vals[rows]
    if stat == "sum":
        acc = np.zeros(ng, np.float64 if v.dtype.kind == "f" else np.int64)
        np.add.at(acc, gid, v)
        return first_rows, cnt, acc
    acc = np.full(ng, v.max() if stat == "min" else v.min(), v.dtype)
    (np.minimum if stat == "min" else np.maximum).at(acc, gid, v)
    return first_rows, cnt, acc


def image(k):
    if k.dtype == np.float32:
        k = k.astype(np.float64)
    if k.dtype.kind == "f":
        u = k.astype(np.float64).view(np.uint64).copy(); u[np.isnan(k)] = 0x7ff8000000000000
        return u
    return k.astype(np.int64).view(np.uint64)


def run(dfdb, ctx, view, by, col, stat, radix):
    ctx.set_option("unique_radix", radix); ctx.profile(True)
    try:
        df = dfdb.groupreduce(view, by, col, stat)
        taken = ctx.profile_get("group_radix.taken")[0]
    finally:
        ctx.profile(False); ctx.set_option("unique_radix", 1)
    return df, taken


@pytest.mark.parametrize("kind", ["int64", "float64", "int32", "dense", "nullable", "int16", "float32"])
@pytest.mark.parametrize("stat", ["count", "sum", "min", "max"])
def test_groupreduce_by_radix(dfdb_mod, ctx, kind, stat):
    rng = np.random.default_rng(hash((kind, stat)) % 1000)
    n = 1_200_007
    if kind == "int64":
        k = (rng.integers(0, 150_000, n) * 40_503 - 3_000_000_000).astype(np.int64)
        k[rng.random(n) < 0.002] = -1                                    # the image the tables cannot store: counted and reduced by the partition pass itself
    elif kind == "float64":
        k = rng.integers(0, 120_000, n).astype(np.float64) / 8.0
        k[rng.random(n) < 0.005] = np.nan
        k[rng.random(n) < 0.005] = -0.0
    elif kind == "int32":
        k = rng.integers(-60_000, 60_000, n).astype(np.int32)
    elif kind == "int16":
        k = rng.integers(-16_000, 16_000, n).astype(np.int16)            # (-1 is the unstorable image of a narrow signed key too)
    elif kind == "float32":
        k = (rng.integers(0, 40_000, n).astype(np.float32) / np.float32(4.0))
        k[rng.random(n) < 0.004] = np.nan
    elif kind == "dense":
        k = rng.integers(0, 300_000, n).astype(np.int64)                 # a small span: unique takes its dense form, from the head of the column first
    else:
        k = (rng.integers(0, 90_000, n) * 7_919 + (1 << 41)).astype(np.int64)      # Union{Int64, Missing}: missing is a group of its own
    miss = rng.random(n) < 0.03 if kind == "nullable" else None
    kcol = np.ma.masked_array(k, mask=miss) if miss is not None else k
    kimg = image(k)
    if miss is not None:
        kimg = kimg.copy(); kimg[miss] = np.uint64(0x0123456789ABCDEF)   # (an image no key has)
    vi = rng.integers(-10**12, 10**12, n).astype(np.int64)
    vf = rng.normal(size=n) * 1e3
    v32 = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)            # a value that is not eight bytes wide: widened where the record is written
    vf32 = (rng.normal(size=n) * 100).astype(np.float32)                 # ... Float32: reduced as the Float64 it converts to
    vu8 = rng.integers(0, 256, n).astype(np.uint8)
    a = rng.integers(0, 100, n).astype(np.int64)
    t = dfdb_mod.DFTable.from_columns({"a": a, "k": kcol, "vi": vi, "vf": vf, "v32": v32, "vf32": vf32, "vu8": vu8}, block_size=65536, ctx=ctx)
    try:
        narrow = (("vf32", vf32.astype(np.float64)), ("vu8", vu8.astype(np.int64))) if kind in ("int16", "float32") else ()
        for col, vals in (("vi", vi), ("vf", vf), ("v32", v32.astype(np.int64))) + narrow:
            if stat == "count" and col != "vi":
                continue
            for view, sel in ((t, np.ones(n, bool)), (t[("a", lambda c: c < 61), dfdb_mod.ALL], a < 61)):
                first_rows, cnt, acc = expect(kimg, sel, vals, stat)
                df, taken = run(dfdb_mod, ctx, view, "k", None if stat == "count" else col, stat, 1)
                assert taken == 1, (kind, stat, col)
                want_keys = k[first_rows]
                assert len(df) == len(cnt)

                def same_keys(frame):
                    got_k = frame["k"]
                    if miss is None:
                        return np.array_equal(image(np.asarray(got_k.to_numpy(), dtype=k.dtype)), image(want_keys))
                    gm = got_k.isna().to_numpy() if hasattr(got_k, "isna") else np.ma.getmaskarray(got_k)
                    wm = miss[first_rows]
                    return np.array_equal(gm, wm) and np.array_equal(np.asarray(got_k.to_numpy()[~gm], dtype=np.int64), want_keys[~wm])
                assert same_keys(df), (kind, stat, col)
                assert np.array_equal(df["count"].to_numpy(), cnt), (kind, stat, col)
                if stat != "count":
                    got = df[stat].to_numpy()
                    if vals.dtype.kind == "f" and stat == "sum":
                        absum = expect(kimg, sel, np.abs(vals), "sum")[2]
                        assert np.all(np.abs(got - acc) <= cnt * np.finfo(np.float64).eps * absum + 1e-300), (kind, col)
                    else:
                        assert np.array_equal(got, acc), (kind, stat, col)
                # the form it replaces gives the same table
                df0, taken0 = run(dfdb_mod, ctx, view, "k", None if stat == "count" else col, stat, 0)
                assert taken0 == 0
                assert same_keys(df0) and np.array_equal(df0["count"].to_numpy(), cnt)
    finally:
        t.close()


@pytest.mark.parametrize("stat", ["count", "sum", "min", "max"])
def test_groupreduce_by_radix_with_hot_keys(dfdb_mod, ctx, stat):
    """a key that 45 % of the rows hold, another with 6 %, a third with 0.5 %, among 1e5 others: the partition pass reduces a hot key's rows in its own LDS slots and
    hands the table pass one entry per workgroup and key (k_radix.hip, hot keys) — the answers are the same as everybody else's, first rows included"""
    rng = np.random.default_rng(3)
    n = 2_000_003
    k = (rng.integers(0, 100_000, n) * 3).astype(np.int64) + (1 << 40)
    u = rng.random(n)
    k[u < 0.45] = 7
    k[(u >= 0.45) & (u < 0.51)] = -12345
    k[(u >= 0.51) & (u < 0.515)] = 1 << 50
    v = rng.integers(-10**9, 10**9, n).astype(np.int64)
    x = rng.normal(size=n)
    a = rng.integers(0, 10, n).astype(np.int64)
    t = dfdb_mod.DFTable.from_columns({"a": a, "k": k, "v": v, "x": x}, block_size=65536, ctx=ctx)
    try:
        for col, vals in (("v", v), ("x", x)):
            if stat == "count" and col == "x":
                continue
            for view, sel in ((t, np.ones(n, bool)), (t[("a", lambda c: c > 2), dfdb_mod.ALL], a > 2)):
                df, taken = run(dfdb_mod, ctx, view, "k", None if stat == "count" else col, stat, 1)
                assert taken == 1
                first_rows, cnt, acc = expect(image(k), sel, vals, stat)
                assert np.array_equal(df["k"].to_numpy(), k[first_rows]) and np.array_equal(df["count"].to_numpy(), cnt), (stat, col)
                if stat != "count":
                    got = df[stat].to_numpy()
                    if vals.dtype.kind == "f" and stat == "sum":
                        absum = expect(image(k), sel, np.abs(vals), "sum")[2]
                        assert np.all(np.abs(got - acc) <= cnt * np.finfo(np.float64).eps * absum + 1e-300)
                    else:
                        assert np.array_equal(got, acc), (stat, col)
                # the form it replaces — every row's value through a global atomic, the hot groups' through LDS slots of their own (k_unique.hip k_group_acc<0>)
                df0, taken0 = run(dfdb_mod, ctx, view, "k", None if stat == "count" else col, stat, 0)
                assert taken0 == 0 and np.array_equal(df0["k"].to_numpy(), k[first_rows]) and np.array_equal(df0["count"].to_numpy(), cnt), (stat, col)
                if stat != "count" and not (vals.dtype.kind == "f" and stat == "sum"):
                    assert np.array_equal(df0[stat].to_numpy(), acc), (stat, col)
    finally:
        t.close()


@pytest.mark.parametrize("stat", ["count", "sum", "min"])
def test_groupreduce_by_a_string_key_with_a_hot_group(dfdb_mod, ctx, stat):
    """String keys take neither the radix form nor (above 1024 groups) LDS accumulators: every row's value goes through a global atomic — and a string that 40 % of
    the rows hold made that one address.  k_str_pass<2, 0> keeps LDS slots for hot groups like k_group_acc<0>; the answers must be everybody else's."""
    rng = np.random.default_rng(9)
    n = 600_011
    ids = rng.integers(0, 3000, n)
    ids[rng.random(n) < 0.4] = 4242
    words = np.array([f"k{v:05d}" for v in range(5000)], dtype=object)
    s = words[ids]
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    t = dfdb_mod.DFTable.from_columns({"s": list(s), "v": v}, block_size=65536, ctx=ctx)
    try:
        df = dfdb_mod.groupreduce(t, "s", None if stat == "count" else "v", stat)
        first_rows, cnt, acc = expect(ids.astype(np.int64).view(np.uint64), np.ones(n, bool), v, stat)
        assert list(df["s"]) == list(s[first_rows])
        assert np.array_equal(df["count"].to_numpy(), cnt)
        if stat != "count":
            assert np.array_equal(df[stat].to_numpy(), acc)
    finally:
        t.close()


@pytest.mark.parametrize("stat", ["count", "sum", "max"])
def test_groupreduce_by_a_dictionary_of_many_strings(dfdb_mod, ctx, stat):
    """a String column with a dictionary of more than 9216 entries: its 16-bit codes are the keys of the radix form (the first occurrences come from the
    dictionary's own pass); one string holds 35 % of the rows"""
    rng = np.random.default_rng(13)
    n = 900_017
    ids = rng.integers(0, 20_000, n)
    ids[rng.random(n) < 0.35] = 777
    words = np.array([f"w{v:06d}" for v in range(20_000)], dtype=object)
    s = words[ids]
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    t = dfdb_mod.DFTable.from_columns({"s": list(s), "v": v}, block_size=65536, ctx=ctx)
    try:
        assert t.build_dictionary("s", max_entries=65535) == len(np.unique(ids))
        df, taken = run(dfdb_mod, ctx, t, "s", None if stat == "count" else "v", stat, 1)
        assert taken == 1
        first_rows, cnt, acc = expect(ids.astype(np.int64).view(np.uint64), np.ones(n, bool), v, stat)
        assert list(df["s"]) == list(s[first_rows])
        assert np.array_equal(df["count"].to_numpy(), cnt)
        if stat != "count":
            assert np.array_equal(df[stat].to_numpy(), acc)
    finally:
        t.close()


def test_groupreduce_by_radix_gives_up_when_the_estimate_was_far_too_low(dfdb_mod, ctx):
    """the first chunk of rows promises 20 000 groups, the rest of the column brings three million more: the partitions' LDS tables fill up, the table pass raises
    the abort word, the selection is put back and the form the radix form replaces answers — the same table as always"""
    rng = np.random.default_rng(21)
    n = 4_200_000
    k = np.empty(n, np.int64)
    head = 1_100_000
    k[:head] = rng.integers(0, 20_000, head) * 5 + 3
    k[head:] = np.arange(n - head, dtype=np.int64) * 7 + (1 << 45)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    t = dfdb_mod.DFTable.from_columns({"k": k, "v": v}, block_size=65536, ctx=ctx)
    try:
        ctx.profile(True)
        try:
            df = dfdb_mod.groupreduce(t, "k", "v", "sum")
            taken, fell = ctx.profile_get("group_radix.taken")[0], ctx.profile_get("group_radix.fell_back")[0]
        finally:
            ctx.profile(False)
        assert taken == 0 and fell >= 1
        first_rows, cnt, acc = expect(image(k), np.ones(n, bool), v, "sum")
        assert np.array_equal(df["k"].to_numpy(), k[first_rows]) and np.array_equal(df["count"].to_numpy(), cnt) and np.array_equal(df["sum"].to_numpy(), acc)
    finally:
        t.close()
