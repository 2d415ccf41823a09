"""Round 5 GPU tests.

* COMPRESSED-ONLY columns (ctx option keep_compressed = 2; SURVEY.md section 8f-2 "without writing decoded blocks to HBM"): a table whose fixed-width
  columns hold their LZ4 blocks and nothing decoded answers every kind of query exactly like the oracle's block iterator — simple terms through K7's
  history-ring scan (fresh and AND-ed masks, intervals, blocks without survivors skipped), projections out of the survivors' arena, everything else through
  the one-call whole-column decode — on liblz4-written and engine-written files, with corrupt blocks, and without leaving a decoded array behind.
"""
import os

import numpy as np
import pandas as pd
import pytest

from helpers import Pair, assert_same

pytestmark = pytest.mark.gpu


def _open_compressed_only(dfdb, ctx, path, **opts):
    ctx.set_option("keep_compressed", 2)
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        return dfdb.open_table(path)
    finally:
        ctx.set_option("keep_compressed", 0)


def _cols(oracle, n, seed=23):
    rng = np.random.default_rng(seed)
    far = np.concatenate([rng.integers(-2**62, 2**62, 3000), np.zeros(10, np.int64)] * (n // 3010 + 1))[:n].astype(np.int64)
    far[6000:9000] = far[0:3000]                                   # a 24-KB repeat at distance 48 KB: far sources out of the history ring
    if n > 9000:
        far[8100:8190] = far[0:90]                                 # ... and one whose source straddles the ring's 64-KB lap (mirror bytes)
    f = oracle.gen_f64(0x1234, 0, n)
    f[::977] = np.nan
    return {"a": oracle.gen_i64(0x9E37, 0, n), "u": rng.integers(0, 2**64 - 1, n, dtype=np.uint64), "f": f, "far": far,
            "z": np.zeros(n, np.int64), "i32": rng.integers(-5, 5, n).astype(np.int32), "iota": np.arange(n, dtype=np.int64)}


@pytest.mark.parametrize("bs,writer", [(65536, "liblz4"), (4096, "liblz4"), (1000, "liblz4"), (8192, "engine"), (65536, "engine")])
def test_compressed_only_table_answers_like_the_block_iterator(oracle, dfdb_mod, ctx, tmp_path, bs, writer):
    from dfdb import ir
    dfdb = dfdb_mod
    n = 200_003
    cols = _cols(oracle, n)
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "t")
    if writer == "liblz4":
        ot.save(path)
    else:                                                          # written by the device packer + LZ4 encoder
        dt = dfdb.DFTable.new(block_size=bs, ctx=ctx)
        for k, v in cols.items():
            dt.add_column(k, v)
        dt.save(path)
        dt.close()
    tb = _open_compressed_only(dfdb, ctx, path)
    rb = tb.resident_bytes()
    plain = sum(v.nbytes for k, v in cols.items())
    assert rb["decoded"] < 4096 * len(cols), rb                   # nothing decoded is resident
    assert 0 < rb["compressed"] < 1.3 * plain
    for name in cols:
        assert tb.resident_bytes(name)["decoded"] < 4096
    c = {k: i for i, k in enumerate(cols)}
    A, U, F, FAR, Z, I32, IOTA = (ir.col(c[k]) for k in ("a", "u", "f", "far", "z", "i32", "iota"))
    preds = [A > 899_999, A <= 5, A == 77, U >= 2**63, U != 12345, F < 632.456, F != 1.0, F >= 1999.0, FAR == 0, FAR < 0, Z == 0, Z > 0, I32 > 2,
             (A > 100_000) & (A < 300_000),                        # an interval: one decode
             (A > 683_771) & (F < 632.456),                        # two columns: the second launch ANDs and skips nothing (random data)
             (IOTA > int(0.9 * n)) & (A > 500_000),                # clustered first term: the second launch skips the blocks without survivors
             (IOTA > int(0.9 * n)) & (FAR < 0) & (I32 > -3),       # ... then a term no decoder takes (Int32): whole-column decode for the call
             (A + FAR > 5) & (U > 77),                             # an interpreter program over compressed-only columns + a scan term
             (A % 7 == 0)]                                         # a rem term (pre != 0): the ordinary kernel over a transient decode
    ctx.profile(True)
    try:
        for e in preds:
            ov = ot.view().add_predicate(e.to_ir())
            dv = dfdb.selection(tb.view(), e)
            q = dv._query()
            assert q.count() == ov.nrow(), (bs, e)
            assert np.array_equal(q.indices(), ov.select_indices()), (bs, e)
            assert np.array_equal(q.bitmap(), ov.select_bitmap(n)), (bs, e)
            got, want = q.materialize(), ov.materialize()
            for g, w in zip(got, want):
                assert np.array_equal(g.view(np.uint8), w.view(np.uint8)), (bs, e)
            # stages after and before the decoder's scan
            ov2 = ot.view().add_predicate(e.to_ir()).add_range(3, 2, 5000)
            dv2 = dfdb.selection(dfdb.selection(tb.view(), e), dfdb.jr(3, 2, 5000))
            assert np.array_equal(dv2._query().indices(), ov2.select_indices()), (bs, e)
            ov3 = ot.view().add_range(1000, 3, 150_000).add_predicate(e.to_ir())
            dv3 = dfdb.selection(dfdb.selection(tb.view(), dfdb.jr(1000, 3, 150_000)), e)
            q3 = dv3._query()
            assert np.array_equal(q3.indices(), ov3.select_indices()), (bs, e)
            got, want = q3.materialize(), ov3.materialize()
            for g, w in zip(got, want):
                assert np.array_equal(g.view(np.uint8), w.view(np.uint8)), (bs, e)
            assert tb.resident_bytes()["decoded"] < 4096 * len(cols), "a whole-column decode outlived its call"
        hist, _ = ctx.profile_get("lz4_decode_scan_hist")
        assert (hist > 0) == (bs % 1024 == 0), (bs, hist)          # block size 1000: no tile-aligned blocks, every term takes the transient decode
    finally:
        ctx.profile(False)
    # aggregates, unique, a computed projection, a save: all through the one-call decode, all equal to the resident table's
    full = dfdb.open_table(path)
    try:
        v_c, v_f = tb[tb.a > 500_000, dfdb.ALL], full[full.a > 500_000, dfdb.ALL]
        assert v_c.far.sum() == v_f.far.sum() and v_c.u.max() == v_f.u.max()
        assert np.array_equal(np.asarray(tb.i32.unique()), np.asarray(full.i32.unique()))
        m_c = dfdb.materialize(tb[tb.iota < 5000, {"k": tb.a * 2 + 1, "a": tb.a}])
        m_f = dfdb.materialize(full[full.iota < 5000, {"k": full.a * 2 + 1, "a": full.a}])
        assert np.array_equal(m_c["k"].to_numpy(), m_f["k"].to_numpy()) and np.array_equal(m_c["a"].to_numpy(), m_f["a"].to_numpy())
        p2 = str(tmp_path / "t_again")
        tb.save(p2)
        t3 = dfdb.open_table(p2)
        for k, v in cols.items():
            assert np.array_equal(dfdb.materialize(t3[dfdb.ALL, [k]])[k].to_numpy().view(np.uint8), v.view(np.uint8)), k
        t3.close()
        assert tb.resident_bytes()["decoded"] < 4096 * len(cols)
    finally:
        full.close()
    tb.close()


def test_compressed_only_projection_decodes_only_blocks_with_survivors(oracle, dfdb_mod, ctx, tmp_path):
    """blocksiterator.jl:111-113 for a compressed-only table: the gather of a projection column decodes the blocks that kept a row and no others
    (the arena spans first .. last such block; blocks in between without survivors are skipped)."""
    from dfdb import ir
    dfdb = dfdb_mod
    n, bs = 40 * 4096 + 17, 4096
    rng = np.random.default_rng(5)
    cols = {"i": np.arange(n, dtype=np.int64), "b": rng.integers(0, 1000, n).astype(np.int64), "x": rng.random(n)}
    p = Pair(oracle, dfdb, cols, block_size=bs, via_files=str(tmp_path / "t")); p.d.close()
    tb = _open_compressed_only(dfdb, ctx, str(tmp_path / "t"))
    ctx.profile(True)
    try:
        keep = (cols["i"] >= 5 * bs + 7) & (cols["i"] < 7 * bs) | (cols["i"] == 30 * bs + 1)          # blocks 5, 6 and 30 of 41
        e = ((ir.col(0) >= 5 * bs + 7) & (ir.col(0) < 7 * bs)) | (ir.col(0) == 30 * bs + 1)
        q = dfdb.selection(tb.view(), e)[dfdb.ALL, ["b", "x"]]._query()
        got = q.materialize()
        assert np.array_equal(got[0], cols["b"][keep]) and np.array_equal(got[1], cols["x"][keep])
        n_surv, _ = ctx.profile_get("lz4_decode.survivors")
        assert n_surv == 2                                         # one subset decode per projected column
        # nothing selected: nothing decoded, empty outputs
        q0 = dfdb.selection(tb.view(), ir.col(0) < 0)[dfdb.ALL, ["b"]]._query()
        assert q0.count() == 0 and len(q0.materialize()[0]) == 0
        # a new selection on the same query object decodes its own blocks
        q.reset()
        assert np.array_equal(q.materialize()[0], cols["b"][keep])
    finally:
        ctx.profile(False)
    assert tb.resident_bytes()["decoded"] < 4096 * 3
    tb.close()


def test_compressed_only_corrupt_resident_blocks_are_reported(oracle, dfdb_mod, ctx, tmp_path):
    """A damaged file is refused at load by the validating decode (no decoded array is ever made); and the history-ring decode does not write outside its rings
    whatever the blocks say (every byte of a block flipped in turn at a few positions: either the load fails with a format error or the table loads and every
    answer still equals the oracle's view of the SAME damaged file)."""
    from dfdb import ir
    dfdb = dfdb_mod
    n, bs = 30_000, 4096
    a = oracle.gen_i64(0x77, 0, n)
    ot = oracle.Table(block_size=bs)
    ot.add_column("a", a)
    path = str(tmp_path / "t")
    ot.save(path)
    f = os.path.join(path, [x for x in os.listdir(path) if x.endswith(".bin") and x != "meta.bin"][0])
    raw = bytearray(open(f, "rb").read())
    rng = np.random.default_rng(3)
    refused = loaded = 0
    for trial in range(40):
        b = bytearray(raw)
        pos = int(rng.integers(64, len(b)))
        b[pos] ^= int(rng.integers(1, 256))
        open(f, "wb").write(bytes(b))
        try:
            tb = _open_compressed_only(dfdb, ctx, path)
        except Exception as ex:                                    # format / decompression errors, as the reference raises them
            assert isinstance(ex, (dfdb.DfdbError, ValueError, IndexError, KeyError)), ex
            refused += 1
            continue
        loaded += 1
        try:
            try:
                oo = oracle.Table.open(path).view().add_predicate((ir.col(0) > 500_000).to_ir())
                want = oo.select_indices()
            except Exception:
                want = None                                         # liblz4 refuses what K7 let through?  then K7 must have refused too
            q = dfdb.selection(tb.view(), ir.col(0) > 500_000)._query()
            got = q.indices()
            assert want is not None and np.array_equal(got, want)
        finally:
            tb.close()
    open(f, "wb").write(bytes(raw))
    assert refused > 0 and refused + loaded == 40


def test_compress_column_in_hbm_without_a_file(oracle, dfdb_mod, ctx):
    """dfdb_table_compress_column: a resident column becomes compressed-resident (mode 1) or compressed-only (mode 2) on the device, with the bytes a saved file
    would hold; every answer stays the oracle's, dfdb_table_decode_resident decodes them back (mode 1), nullable / String columns are refused."""
    from dfdb import ir
    dfdb = dfdb_mod
    n = 150_001
    cols = _cols(oracle, n, seed=9)
    for bs in (4096, 65536):
        ot = oracle.Table(block_size=bs)
        for k, v in cols.items():
            ot.add_column(k, v)
        dt = dfdb.DFTable.new(block_size=bs, ctx=ctx)
        for k, v in cols.items():
            dt.add_column(k, v)
        dt.add_column("s", ["a", "bb", None] * (n // 3) + ["z"] * (n - 3 * (n // 3)))
        before = dt.resident_bytes()
        st = dt.compress_column("a", 1)
        assert st["rows"] == n and st["uncompressed"] == n * 8 and 0 < st["compressed"] < 1.1 * n * 8
        assert dt.resident_bytes("a")["decoded"] >= n * 8 and dt.resident_bytes("a")["compressed"] >= st["compressed"]
        dt.decode_resident("a")
        assert dt.decode_status("a") == 0
        assert np.array_equal(dfdb.materialize(dt[dfdb.ALL, ["a"]])["a"].to_numpy(), cols["a"])
        for k in cols:
            dt.compress_column(k, 2)
        with pytest.raises(NotImplementedError):
            dt.compress_column("s", 2)
        after = dt.resident_bytes()
        assert after["decoded"] < before["decoded"] - sum(v.nbytes for v in cols.values()) + 4096 * len(cols)
        c = {k: i for i, k in enumerate(cols)}
        for e in (ir.col(c["a"]) > 899_999, (ir.col(c["iota"]) > n // 2) & (ir.col(c["f"]) < 500.0), (ir.col(c["far"]) < 0) & (ir.col(c["i32"]) > 0),
                  (ir.col(c["u"]) >= 2**63) & (ir.col(c["a"]) % 3 == 0)):
            ov = ot.view().add_predicate(e.to_ir())
            q = dfdb.selection(dt.view()[dfdb.ALL, list(cols)], e)._query()
            assert q.count() == ov.nrow()
            assert np.array_equal(q.indices(), ov.select_indices())
            for g, w in zip(q.materialize(), ov.materialize()):
                assert np.array_equal(g.view(np.uint8), w.view(np.uint8))
        dt.close()


def test_progressive_load_decodes_batches_behind_their_copies(oracle, dfdb_mod, ctx, tmp_path):
    """dfdb_table_load of a plain fixed-width column decodes batch by batch on a side stream while the rest of the file is read (ctx option load_progressive):
    same columns as one launch at the end — many small pieces and batches, a corrupt block in the middle, keep_compressed = 1 beside it, a second column
    whose row count disagrees."""
    dfdb = dfdb_mod
    n, bs = 300_007, 4096
    rng = np.random.default_rng(11)
    cols = {"a": oracle.gen_i64(0x51, 0, n), "x": rng.random(n), "i32": rng.integers(-9, 9, n).astype(np.int32), "s": ["v%d" % (i % 13) for i in range(n)]}
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "t")
    ot.save(path)
    ctx.set_option("load_piece_kb", 64)                      # 64-KB pieces: the 1.2-MB files are ~20 pieces, batches of 5 blocks
    ctx.set_option("load_progressive_blocks", 5)
    ctx.profile(True)
    try:
        for keep in (0, 1):
            ctx.set_option("keep_compressed", keep)
            tb = dfdb.open_table(path)
            for k in ("a", "x", "i32"):
                assert np.array_equal(dfdb.materialize(tb[dfdb.ALL, [k]])[k].to_numpy().view(np.uint8), cols[k].view(np.uint8)), (keep, k)
            assert list(dfdb.materialize(tb[dfdb.ALL, ["s"]])["s"]) == cols["s"]
            if keep:
                tb.decode_resident("a"); assert tb.decode_status("a") == 0
                assert np.array_equal(dfdb.materialize(tb[dfdb.ALL, ["a"]])["a"].to_numpy(), cols["a"])
            tb.close()
        nprog, _ = ctx.profile_get("lz4_decode.progressive")
        assert nprog >= 2 * 3 * 5, nprog                      # several batches per plain column and load
        # a flipped byte inside a block body in the middle of the file: the load must fail like the one-launch form, not hand out a half-decoded column
        f = os.path.join(path, [x for x in sorted(os.listdir(path)) if x.endswith(".bin") and x != "meta.bin"][0])
        raw = open(f, "rb").read()
        bad = bytearray(raw); pos = len(raw) // 2
        outcomes = []
        for prog in (1, 0):
            ctx.set_option("load_progressive", prog)
            res = []
            for delta in range(0, 400, 37):
                b2 = bytearray(raw); b2[pos + delta] ^= 0x5A
                open(f, "wb").write(bytes(b2))
                try:
                    tb = dfdb.open_table(path)
                    got = dfdb.materialize(tb[dfdb.ALL, ["a"]])["a"].to_numpy()
                    res.append(("ok", bool(np.array_equal(got, cols["a"]))))
                    tb.close()
                except Exception as ex:
                    res.append(("err", type(ex).__name__))
            outcomes.append(res)
        open(f, "wb").write(raw)
        assert outcomes[0] == outcomes[1] and any(r[0] == "err" for r in outcomes[0]), outcomes
        del bad
    finally:
        ctx.profile(False)
        for k, v in (("load_piece_kb", 64 << 10), ("load_progressive_blocks", 768), ("load_progressive", 1), ("keep_compressed", 0)):
            ctx.set_option(k, v)


def test_compressed_only_shards_of_a_group(oracle, dfdb_mod, ctx, tmp_path):
    """block-range shards that are compressed-only (group option keep_compressed = 2 before the load): every answer of the sharded table — counts, indices,
    materialised columns, sums, a range after a predicate (stage bases from the exchange) — equals the oracle's single table; nothing decoded stays resident"""
    from dfdb import group as G, _native as N, ir
    n, bs = 150_003, 4096
    rng = np.random.default_rng(17)
    cols = {"a": oracle.gen_i64(0xA1, 0, n), "x": rng.random(n) * 100.0, "i": np.arange(n, dtype=np.int64)}
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "t")
    ot.save(path)
    g = G.Group.create([0, 0, 0], N.EXCHANGE_HOST)
    try:
        g.set_option("keep_compressed", 2)
        gt = G.GroupTable.open(g, path)
        for l in range(3):
            assert gt.shard(l).resident_bytes()["decoded"] < 4096 * 3
        A, X, I = ir.col(0), ir.col(1), ir.col(2)
        for stages in ([("pred", A > 700_000)], [("pred", (I > n // 3) & (X < 50.0))], [("pred", (A > 300_000) & (A < 600_000)), ("range", 5, 3, 20_000)],
                       [("range", 100, 1, 140_000), ("pred", (X < 10.0) & (A % 2 == 0))]):
            ov, gv = ot.view(), gt.view()
            for st in stages:
                if st[0] == "pred":
                    ov.add_predicate(st[1].to_ir()); gv = dfdb_mod.selection(gv, st[1])
                else:
                    ov.add_range(st[1], st[2], st[3]); gv = dfdb_mod.selection(gv, dfdb_mod.jr(st[1], st[2], st[3]))
            want = ov.select_indices()
            assert G.gnrow(gv) == len(want) and np.array_equal(G.gindices(gv), want), stages
            got, wm = G._gq(gv).materialize(), ov.materialize()
            for a_, b_ in zip(got, wm):
                assert np.array_equal(np.asarray(a_).view(np.uint8), np.asarray(b_).view(np.uint8)), stages
            assert G.gaggregate(gv[dfdb_mod.ALL, "a"], N.AGG_SUM) == int(cols["a"][want - 1].sum())
        for l in range(3):
            assert gt.shard(l).resident_bytes()["decoded"] < 4096 * 3
        gt.close()
    finally:
        g.close()


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


def test_a_streamed_count_never_reads_a_projection_only_column(oracle, dfdb_mod, ctx, tmp_path):
    """nrow over a table that is not resident (BlockRowsIterator, blocksiterator.jl:46-66): only the selection's columns are read — the first projection column
    when the queue holds no predicate.  The other columns' files are CUT SHORT after the table was opened: the counts are still the oracle's, while a streamed
    materialize of the same views meets the damage."""
    import os
    dfdb = dfdb_mod
    n = 50_000
    rng = np.random.default_rng(5)
    cols = {"a": rng.integers(0, 1000, n).astype(np.int64), "b": rng.integers(0, 1000, n).astype(np.int64), "s": [str(i % 97) for i in range(n)]}
    ot = oracle.Table(block_size=1024)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "tb")
    ot.save(path)
    tb = dfdb.open_table(path, load=False)
    for victim in ("2.bin", "3.bin"):                      # b and s: headers stay, the blocks go
        with open(os.path.join(path, victim), "r+b") as f:
            f.truncate(64)
    v = tb[("a", lambda a: a > 899), dfdb.ALL]
    assert dfdb.nrow_streamed(v, 8) == int((cols["a"] > 899).sum())
    assert dfdb.nrow_streamed(tb[dfdb.jr(10, 40_000), dfdb.ALL], 8) == 39_991          # no predicate: the first projection column (a) — or no column at all
    assert dfdb.nrow_streamed(tb[dfdb.ALL, ["a", "b"]], 8) == n
    with pytest.raises(Exception):
        dfdb.materialize_streamed(v, 8)
    tb.close()


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
