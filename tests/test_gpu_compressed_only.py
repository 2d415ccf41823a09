"""COMPRESSED-ONLY columns (keep_compressed = 2, K7's history-ring forms): answers equal to the block iterator's, survivors-only decodes, corrupt blocks, compression in HBM without a file.
(re-filed by component in round 6 from the round-named files; no test body changed)"""


import os

import numpy as np
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
