"""Write side of the block format: columns resident in HBM -> table directory in the reference's
on-disk format, bodies packed and LZ4-compressed ON THE DEVICE.  The parity statement is the one SURVEY.md §8c allows:
the compressed BYTES are not a target (liblz4 versions differ), the frozen LZ4 block format is — so every file the
engine writes is read back (a) by the CPU oracle, whose decoder is the system liblz4 and whose reader follows
BlockStreams.jl / blocks.jl, and (b) by the engine's own loaders, and must give the original columns bit for bit."""
import os
import struct

import numpy as np
import pytest

from helpers import Pair, apply_stages, assert_same

pytestmark = pytest.mark.gpu

SEED = 0x9E3779B97F4A7C15


def col_seed(k):
    return (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF


class Reopened:
    """What assert_same needs: the oracle's view of the files the ENGINE wrote, and the engine's own re-read."""

    def __init__(self, O, dfdb, path, names, nrows):
        self.O, self.dfdb, self.names, self.nrows = O, dfdb, names, nrows
        self.o = O.Table.open(path)
        self.d = dfdb.open_table(path)

    def ord(self, name):
        return self.names.index(name)


def sample_columns(oracle, n, rng):
    sizes, data = oracle.gen_str(col_seed(3), 0, n)
    strs = oracle.flat_to_strings(sizes, data)
    return {"a": oracle.gen_i64(col_seed(0), 0, n), "x": oracle.gen_f64(col_seed(1), 0, n), "s": strs,
            "sm": [None if i % 13 == 0 else s + "é" * (i % 3) for i, s in enumerate(strs)],
            "iota": np.arange(1, n + 1, dtype=np.int64), "i16": rng.integers(-300, 300, n).astype(np.int16),
            "m": np.ma.masked_array(rng.integers(0, 100, n).astype(np.int64), mask=rng.random(n) < 0.2),
            "mf": np.ma.masked_array(rng.random(n).astype(np.float32), mask=rng.random(n) < 0.5),
            "b": rng.integers(0, 2, n).astype(bool), "rnd": rng.integers(-2**62, 2**62, n).astype(np.int64),
            "zeros": np.zeros(n, np.int64), "u8": np.repeat(rng.integers(0, 7, n // 40 + 1).astype(np.uint8), 40)[:n],
            "period3": np.tile(np.array([7, -1, 2**40], np.int64), n // 3 + 1)[:n]}


@pytest.mark.parametrize("enc", [1, 0])
@pytest.mark.parametrize("n,bs", [(200_003, 65536), (200_003, 1000), (70_001, 999), (5, 65536), (12, 3), (0, 65536)])
def test_saved_table_reads_back_everywhere(oracle, dfdb_mod, ctx, tmp_path, n, bs, enc):
    from dfdb import ir
    rng = np.random.default_rng(3)
    cols = sample_columns(oracle, n, rng)
    t = dfdb_mod.DFTable.from_columns(cols, block_size=bs)
    path = str(tmp_path / "tb")
    ctx.set_option("lz4_enc_variant", enc)          # 1: window-parallel compressor (default), 0: one sequence per step
    try:
        st = t.save(path)
    finally:
        ctx.set_option("lz4_enc_variant", 1)
    assert st["rows"] == n
    assert sorted(os.listdir(path)) == sorted(["meta.bin"] + [f"{i + 1}.bin" for i in range(len(cols))])   # test/tables.jl:36-44
    # the oracle (liblz4 + the reference's reader logic) and the engine's decoders agree with the source on every observable
    if True:
        p = Reopened(oracle, dfdb_mod, path, list(cols), n)
        assert p.d.names() == list(cols) and p.d.blocksize == bs
        ov, dv = apply_stages(p, [])
        assert_same(p, ov, dv)
        if n > 100:
            ov, dv = apply_stages(p, [("pred", (ir.col(0) > 500_000) & (ir.col(2) == "sony"))])
            assert_same(p, ov, dv)
            ov, dv = apply_stages(p, [("pred", ir.ismissing(ir.col(6)) | ir.ismissing(ir.col(3)))])
            assert_same(p, ov, dv)
    # and with the data the files were made from (not only with each other)
    if n == 0:
        return                      # (an empty Python list carries no element type: nothing to compare)
    src = Pair(oracle, dfdb_mod, cols, block_size=bs)
    want = src.o.view().materialize()
    got = p.o.view().materialize()
    for w, g in zip(want, got):
        if isinstance(w, tuple):
            assert np.array_equal(w[0], g[0]) and np.array_equal(w[1], g[1])
        elif isinstance(w, np.ma.MaskedArray):
            assert np.array_equal(np.ma.getmaskarray(w), np.ma.getmaskarray(g)) and np.array_equal(w.compressed(), g.compressed())
        else:
            assert np.array_equal(w.view(np.uint8), g.view(np.uint8))


def read_blocks(fn):
    """(block_size, type string, [(rows, origin, compressed)]) of one column file: Appendix A."""
    raw = open(fn, "rb").read()
    bs, = struct.unpack_from("<q", raw, 0)
    tl, = struct.unpack_from("<i", raw, 8)
    ty = raw[12:12 + tl].decode()
    pos, out = 12 + tl, []
    while pos < len(raw):
        rows, origin, comp = struct.unpack_from("<iqq", raw, pos)
        out.append((rows, origin, comp))
        pos += 20 + comp
    assert pos == len(raw)
    return bs, ty, out


def test_file_layout_and_compression(oracle, dfdb_mod, ctx, tmp_path):
    n, bs = 300_000, 65536
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "iota": np.arange(1, n + 1, dtype=np.int64), "zeros": np.zeros(n, np.int64),
            "m": np.ma.masked_array(np.arange(n, dtype=np.int32), mask=np.arange(n) % 7 == 0), "s": ["ab", None, "xyz"] * (n // 3)}
    t = dfdb_mod.DFTable.from_columns(cols, block_size=bs)
    st = t.save(str(tmp_path / "tb"))
    meta = open(tmp_path / "tb" / "meta.bin", "rb").read()
    assert struct.unpack_from("<qqq", meta, 0) == (1, bs, len(cols))               # table_io.jl:9-19
    tot_c = tot_u = 0
    for i, (name, ty, width) in enumerate([("a", "Int64", 8), ("iota", "Int64", 8), ("zeros", "Int64", 8), ("m", "Missing(Int32)", 4), ("s", "Missing(String)", 0)]):
        fbs, fty, blocks = read_blocks(tmp_path / "tb" / f"{i + 1}.bin")
        assert fbs == bs and fty == ty
        assert [b[0] for b in blocks] == [bs] * (n // bs) + [n % bs]                # columns.jl:16-26: full blocks, then the rest
        for bi, (rows, origin, comp) in enumerate(blocks):
            if name == "m":
                assert origin == 8 * ((rows + 63) // 64) + 4 * rows                 # blocks.jl:9-18
            elif name == "s":
                datasize = sum(len(x) for x in cols["s"][bi * bs: bi * bs + rows] if x is not None)
                assert origin == 4 + 4 * rows + datasize                            # blocks.jl:21-33: datasize, sizes, bytes
            else:
                assert origin == width * rows
            assert 0 < comp <= origin + origin // 255 + 16
        tot_c += sum(b[2] + 24 for b in blocks); tot_u += sum(b[1] for b in blocks)
        ratio = sum(b[1] for b in blocks) / sum(b[2] for b in blocks)
        if name == "zeros":
            assert ratio > 200
        if name == "iota":
            assert ratio > 1.9                                                     # docs/src/index.md:53 quotes 2.0 for 1:3e6
        if name == "a":
            assert ratio > 1.25                                                    # 20 random bits per 8 bytes
    assert (st["compressed"], st["uncompressed"]) == (tot_c, tot_u)                 # SizeStats with the 24-byte quirk (Q10)
    ts = dfdb_mod.table_stats(dfdb_mod.open_table(str(tmp_path / "tb"), load=False))       # table_stats: headers only
    assert ts["column"].tolist() == list(cols) + ["Table total"] and ts["rows"].tolist() == [n] * 6
    assert ts["compressed size"].iloc[-1] == tot_c and ts["uncompressed size"].iloc[-1] == tot_u and ts["type"].iloc[3] == "Missing(Int32)"


def test_add_column_from_lazy_column_and_create_table(oracle, dfdb_mod, ctx, tmp_path):
    """add_column!(t, :k, lazy) and create_table(path; from=view): computed and filtered columns go device -> device -> disk."""
    from dfdb import ir
    n = 150_000
    rng = np.random.default_rng(9)
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "c": rng.integers(1, 50, n).astype(np.int64),
            "s": oracle.flat_to_strings(*oracle.gen_str(col_seed(3), 0, n)),
            "m": np.ma.masked_array(rng.integers(0, 100, n).astype(np.int64), mask=rng.random(n) < 0.3)}
    p = Pair(oracle, dfdb_mod, cols, block_size=4096)
    t = p.d
    t.add_column_from("k", t.a * 2 + t.c)                 # same table, every row
    t.add_column_from("r", t.a / t.c)
    got = dfdb_mod.materialize(t[dfdb_mod.ALL, ["k", "r"]])
    assert np.array_equal(np.asarray(got["k"]), cols["a"] * 2 + cols["c"])
    assert np.array_equal(np.asarray(got["r"]), cols["a"] / cols["c"])
    with pytest.raises(ValueError):                       # wrong length: add_column! rejects it
        t.add_column_from("bad", t[("a", lambda a: a > 500_000), dfdb_mod.ALL].a)
    with pytest.raises(ValueError):                       # duplicate name
        t.add_column_from("k", t.a)
    # a filtered, projected view -> new table on disk; the oracle evaluates the same view on the source data
    pred = (ir.col(0) > 600_000) & (ir.col(2) != "sony")
    v = t[pred, ["a", "s", "m", "k"]]
    path = str(tmp_path / "sub")
    sub = dfdb_mod.create_table(path, from_=v, block_size=1000)
    ov, _ = apply_stages(p, [("pred", pred)], proj=[("a", ir.col(0)), ("s", ir.col(2)), ("m", ir.col(3)), ("k", ir.col(0) * 2 + ir.col(1))])
    want = ov.materialize()
    ot = oracle.Table.open(path)
    got = ot.view().materialize()
    assert dfdb_mod.nrow(sub) == ov.nrow() == len(got[0])
    for w, g in zip(want, got):
        if isinstance(w, tuple):
            assert np.array_equal(w[0], g[0]) and np.array_equal(w[1], g[1])
        elif isinstance(w, np.ma.MaskedArray):
            assert np.array_equal(np.ma.getmaskarray(w), np.ma.getmaskarray(g)) and np.array_equal(w.compressed(), g.compressed())
        else:
            assert np.array_equal(w, g)
    with pytest.raises(dfdb_mod.DfdbError):               # create_table refuses an existing table
        sub.save(path)
    # table_exists(path) = isdir(path) (filesystem.jl:31,38): ANY existing directory is refused, nothing in it is touched;
    # an existing column file is refused too (make_column_file, filesystem.jl:16)
    empty = tmp_path / "empty_dir"
    empty.mkdir()
    (empty / "1.bin").write_bytes(b"precious")
    with pytest.raises(dfdb_mod.DfdbError, match="already exists"):
        sub.save(str(empty))
    assert (empty / "1.bin").read_bytes() == b"precious"
    with pytest.raises(dfdb_mod.DfdbError, match="already exists"):
        sub.save_column(sub.names()[0], str(empty / "1.bin"))
    # a rejected add leaves the table as it was (no phantom column, no row count set on an empty table)
    t0 = dfdb_mod.DFTable.new()
    from dfdb import _native as N
    five = np.arange(5, dtype=np.int64)
    assert N.load().dfdb_table_add_column(t0._h, b"bad", 63, 5, five.ctypes.data, None, 0, None) == N.ERR_UNSUPPORTED
    assert t0.ncols == 0
    t0.add_column("ok", np.arange(7, dtype=np.int64))                # 7 rows: the rejected 5-row column did not pin the row count
    assert dfdb_mod.nrow(t0) == 7
    with pytest.raises(ValueError, match="Duplicated"):
        t0.add_column("ok", np.arange(7, dtype=np.int64))
    assert t0.ncols == 1


def test_incompressible_and_degenerate_blocks(oracle, dfdb_mod, ctx, tmp_path):
    """Bodies of 0..40 bytes (below the 13-byte minimum the format needs for a match), pure noise, a 1-row table."""
    rng = np.random.default_rng(17)
    for k, (n, bs, dt) in enumerate([(1, 65536, np.int8), (5, 2, np.int8), (40, 13, np.uint8), (13, 65536, np.int8), (100_000, 65536, np.int64)]):
        cols = {"v": rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, n).astype(dt), "z": np.zeros(n, dt)}
        t = dfdb_mod.DFTable.from_columns(cols, block_size=bs)
        path = str(tmp_path / f"t{k}")
        t.save(path)
        ot = oracle.Table.open(path)
        got = ot.view().materialize()
        assert np.array_equal(got[0], cols["v"]) and np.array_equal(got[1], cols["z"])
        back = dfdb_mod.materialize(dfdb_mod.open_table(path))
        assert np.array_equal(np.asarray(back["v"]), cols["v"])


def test_date_datetime_time_char_columns(oracle, dfdb_mod, ctx, tmp_path):
    """Julia bits types the block format stores as plain integers (read_block_body! is a memcpy: blocks.jl:37-44): the type
    strings "Date", "DateTime", "Time", "Char", "Missing(DateTime)" must open, filter, materialise and round-trip."""
    from dfdb import ir
    rng = np.random.default_rng(2)
    n = 70_003
    days = (np.datetime64("2019-10-01") + rng.integers(0, 61, n).astype("timedelta64[D]")).astype("datetime64[D]")
    stamps = (np.datetime64("2019-10-01T00:00:00", "ms") + rng.integers(0, 61 * 86_400_000, n).astype("timedelta64[ms]"))
    tod = rng.integers(0, 86_400 * 10**9, n).astype("timedelta64[ns]")
    chars = rng.choice(list("abcé€z"), n)
    mask = rng.random(n) < 0.15
    # written by the oracle under the reference's type strings
    ot = oracle.Table(block_size=4096)
    ot.add_column("d", days.astype(np.int64) + ir.RATA_DIE_DAYS, logical="Date")
    ot.add_column("ts", stamps.astype(np.int64) + ir.RATA_DIE_MS, logical="DateTime")
    ot.add_column("tod", tod.astype(np.int64), logical="Time")
    ot.add_column("c", np.array([ir.julia_char(x) for x in chars], np.uint32), logical="Char")
    ot.add_column("tsm", stamps.astype(np.int64) + ir.RATA_DIE_MS, missing=mask, logical="DateTime")
    ot.add_column("a", np.arange(n, dtype=np.int64))
    path = str(tmp_path / "tb")
    ot.save(path)
    raw = open(os.path.join(path, "meta.bin"), "rb").read()
    for ty in (b"Date", b"DateTime", b"Time", b"Char", b"Missing(DateTime)"):
        assert ty in raw
    t = dfdb_mod.open_table(path)
    assert [m.type for m in t.columns_meta()] == ["Date", "DateTime", "Time", "Char", "Missing(DateTime)", "Int64"]
    df = dfdb_mod.materialize(t)
    assert np.array_equal(df["d"].to_numpy().astype("datetime64[D]"), days) and np.array_equal(df["ts"].to_numpy().astype("datetime64[ms]"), stamps)
    assert np.array_equal(df["tod"].to_numpy().astype("timedelta64[ns]"), tod) and df["c"].tolist() == chars.tolist()
    # predicates: constants are lowered to the Int64 instants Julia stores
    cut = np.datetime64("2019-11-01T00:00:00")
    v = t[(t.ts >= cut) & (t.d < np.datetime64("2019-11-15")) & (t.c == ir.julia_char("é")), ["a", "ts"]]
    want = np.nonzero((stamps >= cut) & (days < np.datetime64("2019-11-15")) & (chars == "é"))[0]
    got = dfdb_mod.materialize(v)
    assert got["a"].tolist() == want.tolist() and np.array_equal(got["ts"].to_numpy().astype("datetime64[ms]"), stamps[want])
    assert dfdb_mod.nrow(t[ir.ismissing(ir.col(4)), dfdb_mod.ALL]) == int(mask.sum())
    assert list(t.d.unique()) == list(dict.fromkeys(days.tolist()))
    # write side: the engine keeps the type strings; the oracle reads the same values back
    out = str(tmp_path / "out")
    t.save(out)
    ot2 = oracle.Table.open(out)
    assert [ot2.logical(i) for i in range(6)] == ["Date", "DateTime", "Time", "Char", "DateTime", ""]
    back = ot2.view().materialize()
    assert np.array_equal(back[0], days.astype(np.int64) + ir.RATA_DIE_DAYS) and np.array_equal(back[3], np.array([ir.julia_char(x) for x in chars], np.uint32))
    assert np.array_equal(np.ma.getmaskarray(back[4]), mask)
    # caller-supplied numpy datetimes become Date / DateTime / Time columns
    t2 = dfdb_mod.DFTable.from_columns({"d": days, "ts": stamps, "tod": tod})
    assert [m.type for m in t2.columns_meta()] == ["Date", "DateTime", "Time"]
    assert dfdb_mod.nrow(t2[t2.ts >= cut, dfdb_mod.ALL]) == int((stamps >= cut).sum())
    # a type outside the set is still refused, by name
    with pytest.raises(NotImplementedError, match="UndefinedType"):
        meta = bytearray(raw); i = meta.index(b"Int64"); meta[i:i + 5] = b"Int99"
        open(os.path.join(path, "meta.bin"), "wb").write(bytes(meta))
        dfdb_mod.open_table(path)


@pytest.mark.parametrize("enc", [1, 0])
def test_encoder_corner_cases(oracle, dfdb_mod, ctx, tmp_path, enc):
    """Byte columns that push the device LZ4 compressors through every emission path: periodic data of every period 1..130
    (matches that overlap their source, long extensions), literal runs of 0..400 bytes between matches (length-byte chains on the
    FIRST sequence of a window), dense short matches at mixed distances, incompressible noise, runs; block sizes that end a block
    inside a match / a literal run.  Every file must be decoded to the source by liblz4 (the oracle) and by K7."""
    rng = np.random.default_rng(11)
    n = 200_000
    periodic = np.concatenate([np.tile(rng.integers(0, 256, per).astype(np.uint8), 1600 // per + 1)[:1600] for per in range(1, 131)])
    base = rng.integers(0, 256, 20_000).astype(np.uint8)
    pieces, tot = [base], len(base)
    while tot < n:
        lit = rng.integers(0, 256, int(rng.integers(0, 401))).astype(np.uint8)
        ln = int(rng.choice([4, 5, 11, 12, 13, 18, 19, 20, 64, 65, 270, 271, 300, 1000, 9000]))
        start = int(rng.integers(0, len(base) - ln))
        pieces += [lit, base[start:start + ln]]; tot += len(lit) + ln
    mixed = np.concatenate(pieces)[:n]
    short = bytearray(rng.integers(0, 256, 4096).astype(np.uint8).tobytes())
    while len(short) < n:
        short += rng.integers(0, 256, int(rng.integers(0, 15))).astype(np.uint8).tobytes()
        ml = int(rng.integers(4, 19)); hi = (8, 64, 2000, 60_000)[int(rng.integers(0, 4))]
        d = int(rng.integers(1, min(hi, len(short)) + 1))
        for k in range(ml):
            short.append(short[len(short) - d])
    cols = {"periodic": np.resize(periodic, n), "mixed": mixed, "shortseq": np.frombuffer(bytes(short[:n]), np.uint8),
            "noise": rng.integers(0, 256, n).astype(np.uint8), "zeros": np.zeros(n, np.uint8),
            "runs": np.resize(np.repeat(rng.integers(0, 4, n // 50 + 1).astype(np.uint8), rng.integers(1, 100, n // 50 + 1)), n),
            "i64": oracle.gen_i64(col_seed(0), 0, n), "iota": np.arange(1, n + 1, dtype=np.int64)}
    ctx.set_option("lz4_enc_variant", enc)
    try:
        for bs in (65536, 4099, 13):
            m = n if bs > 100 else 2000
            sub = {k: v[:m] for k, v in cols.items()}
            t = dfdb_mod.DFTable.from_columns(sub, block_size=bs)
            path = str(tmp_path / f"e{bs}")
            st = t.save(path)
            assert st["rows"] == m
            p = Reopened(oracle, dfdb_mod, path, list(sub), m)
            ov, dv = apply_stages(p, [])
            assert_same(p, ov, dv)
            got = p.o.view().materialize()                              # and against the source itself, not only each other
            for (k, v), g in zip(sub.items(), got):
                assert np.array_equal(np.asarray(g).view(np.uint8), v.view(np.uint8)), k
    finally:
        ctx.set_option("lz4_enc_variant", 1)


@pytest.mark.parametrize("seed", range(int(os.environ.get("DFDB_FUZZ_SEED0", "0")), int(os.environ.get("DFDB_FUZZ_SEED0", "0")) + 24 * int(os.environ.get("DFDB_FUZZ_SCALE", "1"))))
def test_random_tables_written_by_the_device_read_back(oracle, dfdb_mod, ctx, tmp_path, seed):
    """Seeded fuzz of the write side: a table of random shape (row count, block size, 1-6 columns of random type — every integer width, floats, Bool,
    Strings, nullable forms — with contents from incompressible to constant, with LZ77-shaped byte structure in between), saved by the device packer +
    LZ4 encoder (both encoder variants), must come back bit for bit through the ORACLE's reader (liblz4) and through the engine's own loader, the
    resident one and the block-streamed one."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 63, 1000, 4097, 65_536, 70_001, 150_000]))
    bs = int(rng.choice([1, 7, 1000, 4096, 65536])) if n <= 4097 else int(rng.choice([1000, 4096, 65536, 50_000]))

    def ints(dt):
        info = np.iinfo(dt); kind = rng.integers(0, 5)
        if kind == 0: return rng.integers(info.min, info.max, n, dtype=np.int64 if dt != np.uint64 else np.uint64, endpoint=True).astype(dt)   # incompressible
        if kind == 1: return np.full(n, rng.integers(max(info.min, -5), min(info.max, 5)), dt)                                                   # constant
        if kind == 2: return (np.arange(n) % max(1, int(rng.integers(1, 300)))).astype(dt)                                                        # periodic
        if kind == 3: return np.repeat(rng.integers(0, 50, n // 37 + 1), 37)[:n].astype(dt)                                                      # runs
        return (rng.integers(0, 1000, n) * (1 if info.max < 2**31 else 2**33)).astype(dt)                                                       # sparse high bits
    def strs():
        pool = ["", "a", "sony", "é", "x" * int(rng.integers(1, 300))] + ["w%d" % k for k in range(int(rng.integers(1, 2000)))]
        return [pool[int(k)] for k in rng.integers(0, len(pool), n)]
    makers = [lambda: ints(np.int8), lambda: ints(np.int16), lambda: ints(np.int32), lambda: ints(np.int64), lambda: ints(np.uint8), lambda: ints(np.uint16),
              lambda: ints(np.uint32), lambda: ints(np.uint64), lambda: rng.normal(0, 1, n), lambda: rng.integers(0, 4, n).astype(np.float32),
              lambda: rng.integers(0, 2, n).astype(bool), strs,
              lambda: np.ma.masked_array(ints(np.int64), mask=rng.random(n) < rng.random()), lambda: np.ma.masked_array(rng.normal(0, 1, n), mask=rng.random(n) < 0.3),
              lambda: [None if rng.random() < 0.2 else w for w in strs()]]
    cols = {"c%d" % k: makers[int(rng.integers(0, len(makers)))]() for k in range(int(rng.integers(1, 7)))}
    t = dfdb_mod.DFTable.from_columns(cols, block_size=bs)
    path = str(tmp_path / "tb")
    ctx.set_option("lz4_enc_variant", seed % 2)
    try:
        st = t.save(path)
    finally:
        ctx.set_option("lz4_enc_variant", 1)
    t.close()
    assert st["rows"] == n
    p = Reopened(oracle, dfdb_mod, path, list(cols), n)
    ov, dv = apply_stages(p, [])
    assert_same(p, ov, dv)                                                     # oracle reader == engine loader, every column, every byte
    want = Pair(oracle, dfdb_mod, cols, block_size=bs)                         # ... and == the data the files were made from
    ov2, dv2 = apply_stages(want, [])
    a, b = ov.materialize(), ov2.materialize()
    for x, y in zip(a, b):
        if isinstance(x, tuple): assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
        elif isinstance(x, np.ma.MaskedArray): assert np.array_equal(np.ma.getmaskarray(x), np.ma.getmaskarray(y)) and np.array_equal(x.compressed().view(np.uint8), y.compressed().view(np.uint8))
        else: assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    lazy = dfdb_mod.open_table(path, load=False)
    assert dfdb_mod.nrow_streamed(lazy, 1 + seed % 4) == n
    lazy.close(); p.d.close(); want.d.close()
    # the other direction: the ORACLE writes the same table (liblz4's bytes), the device decoder + unpackers read it
    back = Pair(oracle, dfdb_mod, cols, block_size=bs, via_files=str(tmp_path / "by_oracle"))
    ov3, dv3 = apply_stages(back, [])
    assert_same(back, ov3, dv3)
    back.d.close()
    # both files again with their LZ4 blocks kept in HBM: the first resident decode of every plain fixed-width column records its sequence-start index, the
    # second decodes with it (k_decode.hip INDEX) — same bytes as the loader's decode
    ctx.set_option("keep_compressed", 1); ctx.set_option("lz4_pipeline", (seed >> 1) % 2)      # one wave per block / the two-wave pipeline (both read the index)
    try:
        for pth in (path, str(tmp_path / "by_oracle")):
            kt = dfdb_mod.open_table(pth)
            for name, v in cols.items():
                if isinstance(v, (list, np.ma.MaskedArray)) or n == 0:
                    continue
                for k in range(2):
                    kt.decode_resident(name)
                    assert kt.decode_status(name) == 0, (pth, name, k)
                    got = dfdb_mod.materialize(kt[dfdb_mod.ALL, [name]])[name].to_numpy()
                    assert np.array_equal(got.view(np.uint8), np.asarray(v).view(np.uint8)), (pth, name, k)
            kt.close()
    finally:
        ctx.set_option("keep_compressed", 0); ctx.set_option("lz4_pipeline", -1)
