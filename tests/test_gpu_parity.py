"""GPU parity: the HIP engine (through the C ABI / Python mirror) against the CPU oracle on the same
seeded inputs.  Integer, byte and index results must be bit-exact."""
import os
import shutil

import numpy as np
import pytest

from helpers import Pair, apply_stages, assert_same

pytestmark = pytest.mark.gpu

SEED = 0x9E3779B97F4A7C15


def col_seed(k):  # column k uses seed * (k+1)
    return (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF


# ------------------------------------------------------------------ generators
def test_generators_match_oracle(oracle, dfdb_mod, ctx):
    n = 200_001
    t = dfdb_mod.DFTable.new()
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, col_seed(0), n, row_first=12345)
    t.add_generated("x", dfdb_mod.GEN_F64_U2000, col_seed(1), n, row_first=12345)
    t.add_generated("s", dfdb_mod.GEN_STR_BRANDS10, col_seed(2), n, row_first=12345)
    t.add_generated("i", dfdb_mod.GEN_I64_IOTA, 0, n, row_first=12345)
    a, x, s, i = t.view()._query().materialize()
    assert np.array_equal(a, oracle.gen_i64(col_seed(0), 12345, n))
    assert np.array_equal(x.view(np.uint64), oracle.gen_f64(col_seed(1), 12345, n).view(np.uint64))
    sz, by = oracle.gen_str(col_seed(2), 12345, n)
    assert np.array_equal(s[0], sz) and np.array_equal(s[1], by)
    assert np.array_equal(i, np.arange(12346, 12346 + n, dtype=np.int64))


# ------------------------------------------------------------------ config 2 shape: x > c
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 65536, 65537, 300_001])
def test_int64_gt_sizes(oracle, dfdb_mod, ctx, n):
    from dfdb import ir
    x = oracle.gen_i64(col_seed(0), 0, n)
    p = Pair(oracle, dfdb_mod, {"x": x}, block_size=65536)
    ov, dv = apply_stages(p, [("pred", ir.col(0) > 899_999)])
    assert_same(p, ov, dv)


@pytest.mark.parametrize("op", ["==", "!=", "<", "<=", ">", ">="])
@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64])
def test_cmp_ops_all_dtypes(oracle, dfdb_mod, ctx, op, dtype):
    from dfdb import ir
    import operator
    rng = np.random.default_rng(7)
    n = 10_007
    if np.dtype(dtype).kind == "f":
        x = rng.integers(-50, 50, n).astype(dtype) / dtype(4)
        x[::97] = np.nan
        c = dtype(3.25)
    elif np.dtype(dtype).kind == "u":
        x = rng.integers(0, 100, n).astype(dtype); c = 40
    else:
        x = rng.integers(-60, 60, n).astype(dtype); c = -7
    f = {"==": operator.eq, "!=": operator.ne, "<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}[op]
    p = Pair(oracle, dfdb_mod, {"x": x}, block_size=4096)
    ov, dv = apply_stages(p, [("pred", f(ir.col(0), ir.const(c)))])
    assert_same(p, ov, dv)
    # third opinion: numpy
    want = np.nonzero(f(x, c))[0] + 1
    assert np.array_equal(dv._query().indices(), want)


def test_int_column_vs_float_constant_is_exact(oracle, dfdb_mod, ctx):
    from dfdb import ir
    x = np.array([-3, -2, 2, 3, 4, 2**53, 2**53 + 1, -(2**63), 2**63 - 1], np.int64)
    p = Pair(oracle, dfdb_mod, {"x": x}, block_size=4)
    for c in [3.5, -2.5, 3.0, float(2**53), 9.3e18, -9.3e18, float("nan"), float("inf")]:
        for mk in (lambda e, c: e > c, lambda e, c: e <= c, lambda e, c: e == c, lambda e, c: e != c, lambda e, c: c > e):
            ov, dv = apply_stages(p, [("pred", mk(ir.col(0), c))])
            assert_same(p, ov, dv)
            want = [i + 1 for i, v in enumerate(x.tolist()) if bool(mk(v, c))]   # Python int-vs-float comparison is exact too
            assert dv._query().indices().tolist() == want, (c, want)


# ------------------------------------------------------------------ selection executor known answers (test/selection.jl)
def test_selection_known_answers(oracle, dfdb_mod, ctx):
    from dfdb import ir
    a = np.arange(1, 101, dtype=np.int64)
    for bs in (100, 50, 7):
        p = Pair(oracle, dfdb_mod, {"a": a, "b": a * 5}, block_size=bs)
        # (5:20)∘(3:4) -> 7:8   (test/selection.jl:40-49)
        ov, dv = apply_stages(p, [("range", 5, 1, 20), ("range", 3, 1, 4)])
        assert_same(p, ov, dv); assert dv._query().indices().tolist() == [7, 8]
        # (10:60, 65>a>34, 15:18) -> 49:52   (:51-72)
        ov, dv = apply_stages(p, [("range", 10, 1, 60), ("pred", (65 > ir.col(0)) & (ir.col(0) > 34)), ("range", 15, 1, 18)])
        assert_same(p, ov, dv); assert dv._query().indices().tolist() == [49, 50, 51, 52]
        # (65>a>34) & (b%10==0), fused into one stage (:74-106)
        ov, dv = apply_stages(p, [("pred", (65 > ir.col(0)) & (ir.col(0) > 34)), ("pred", ir.col(1) % 10 == 0)])
        assert len(dv.selection) == 1
        assert_same(p, ov, dv)
        assert dv._query().indices().tolist() == [i for i in range(35, 65) if (i * 5) % 10 == 0]


def test_range_stage_forms(oracle, dfdb_mod, ctx):
    from dfdb import ir
    n = 10_000
    a = np.arange(1, n + 1, dtype=np.int64)
    p = Pair(oracle, dfdb_mod, {"a": a}, block_size=1000)
    cases = [
        [("range", 5, 300, 10_000)], [("range", 9990, 1, 10_000)], [("idx", [1, 200, 20])], [("idx", [7, 7, 3, 9999])], [("int", 4242)],
        [("range", 1, 10, n), ("pred", ir.col(0) % 3 == 1)], [("pred", ir.col(0) % 3 == 1), ("range", 1, 7, 3000)],
        [("pred", ir.col(0) > 5000), ("idx", [5, 1, 4000])], [("range", 100, -3, 10)], [("range", 20, 1, 10)],
        [("pred", ir.col(0) % 2 == 0), ("range", 10, 1, 4000), ("pred", ir.col(0) % 3 == 0), ("range", 2, 2, 600)],
    ]
    for st in cases:
        ov, dv = apply_stages(p, st)
        assert_same(p, ov, dv)


# ------------------------------------------------------------------ config 3 shape
def test_conjunction_and_projection(oracle, dfdb_mod, ctx):
    from dfdb import ir
    n = 150_003
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "b": oracle.gen_i64(col_seed(1), 0, n), "x": oracle.gen_f64(col_seed(2), 0, n)}
    p = Pair(oracle, dfdb_mod, cols, block_size=65536)
    pred = (ir.col(0) > 683_771) & (ir.col(2) < 632.456)
    ov, dv = apply_stages(p, [("pred", pred)], proj=[("b", ir.col(1)), ("x", ir.col(2))])
    assert_same(p, ov, dv)
    ov, dv = apply_stages(p, [("pred", pred)], proj=[("b", ir.col(1)), ("x2", ir.col(2) * 2)])
    assert_same(p, ov, dv)


# ------------------------------------------------------------------ config 4 shape
def test_string_equality_and_gather(oracle, dfdb_mod, ctx):
    from dfdb import ir
    n = 120_001
    sizes, data = oracle.gen_str(col_seed(0), 0, n)
    strs = oracle.flat_to_strings(sizes, data)
    cols = {"s": strs, "a": oracle.gen_i64(col_seed(1), 0, n)}
    p = Pair(oracle, dfdb_mod, cols, block_size=65536)
    for pred in (ir.col(0) == "sony", ir.col(0) != "sony", ir.startswith(ir.col(0), "s"), ir.endswith(ir.col(0), "l"),
                 (ir.col(0) == "dell") & (ir.col(1) > 500_000), ir.col(0) == "", ir.col(0) < "dell", ir.sizeof(ir.col(0)) == 4):
        ov, dv = apply_stages(p, [("pred", pred)])
        assert_same(p, ov, dv)


# ------------------------------------------------------------------ generic expressions (interpreter)
def test_generic_expressions(oracle, dfdb_mod, ctx):
    from dfdb import ir
    n = 20_011
    rng = np.random.default_rng(3)
    cols = {"a": rng.integers(-1000, 1000, n).astype(np.int64), "c": (rng.integers(1, 50, n)).astype(np.int64),
            "x": rng.normal(0, 100, n), "i32": rng.integers(-2**31, 2**31 - 1, n).astype(np.int32), "f": rng.normal(0, 3, n).astype(np.float32),
            "u8": rng.integers(0, 255, n).astype(np.uint8), "flag": rng.integers(0, 2, n).astype(bool)}
    p = Pair(oracle, dfdb_mod, cols, block_size=4096)
    a, c, x, i32, f, u8, flag = (ir.col(k) for k in range(7))
    preds = [a % c == 0, (a * 2 + c) > x, a / c > 1.5, ir.mod(a, c) == 3, ir.div(a, c) == -2, (a + i32) < 0, (i32 + i32) > 0, f * 2 > x,
             u8 + u8 > 300, flag & (a > 0), ~flag | (x < 0), ir.isin(a, [1, 11, 21, -5]), abs(a) < 10, -a > 990, ir.maximum(a, c) == c,
             (a ^ c) & 1 == 1, ir.float64(a) * 0.5 == x, a == x, (u8 - u8) == 0, ir.minimum(x, f) < -3, (f + 1) > 1]
    for pred in preds:
        ov, dv = apply_stages(p, [("pred", pred)])
        assert_same(p, ov, dv)
    projs = [[("k", a / 50)], [("k", a * 2 + c), ("m", x - f)], [("k", i32 * i32)], [("k", u8 + u8), ("z", a % c)], [("k", f * f)],
             [("k", flag), ("n", ~flag)], [("k", a > c)]]
    for pr in projs:
        ov, dv = apply_stages(p, [("pred", a % 7 == 0)], proj=pr)
        assert_same(p, ov, dv)


def test_divide_error_and_bad_predicates(oracle, dfdb_mod, ctx):
    from dfdb import ir
    a = np.arange(-5, 6, dtype=np.int64)
    p = Pair(oracle, dfdb_mod, {"a": a, "z": np.zeros(11, np.int64)}, block_size=4)
    ov, dv = apply_stages(p, [("pred", ir.col(0) % ir.col(1) == 0)])
    with pytest.raises(ZeroDivisionError):
        ov.nrow()
    with pytest.raises(ZeroDivisionError):
        dfdb_mod.nrow(dv)
    # two generic conjuncts, the zero divisor only on rows the FIRST one rejects: Julia's fused `&` is not short-circuit
    # (BlockBroadcasting(&, (old, elem)), selection.jl:44-47), so the second is evaluated on every row that reached the stage and raises
    z1 = np.full(11, 3, np.int64)
    z2 = np.where(a % 3 == 0, 2, 0).astype(np.int64)          # zero exactly where a % z1 != 0
    b = np.arange(11, dtype=np.int64) * 2
    p2 = Pair(oracle, dfdb_mod, {"a": a, "b": b, "z1": z1, "z2": z2}, block_size=4)
    A, B, Z1, Z2 = ir.col(0), ir.col(1), ir.col(2), ir.col(3)
    for stages in ([("pred", (A % Z1 == 0) & (B % Z2 == 0))],                       # one fused predicate
                   [("pred", A % Z1 == 0), ("pred", B % Z2 == 0)],                  # two selections: fused by the queue
                   [("pred", (A > 100) & (B % Z2 == 0))]):                          # a simple term rejects every row: still raises
        ov, dv = apply_stages(p2, stages)
        with pytest.raises(ZeroDivisionError):
            ov.nrow()
        with pytest.raises(ZeroDivisionError):
            dfdb_mod.nrow(dv)
    # ... but a range stage in between really removes the rows (a new stage, not a fused `&`): rows 3, 6, 9 (a = -3, 0, 3) have z2 != 0
    ov, dv = apply_stages(p2, [("idx", [3, 6, 9]), ("pred", B % Z2 == 0)])
    assert_same(p2, ov, dv)
    with pytest.raises(ValueError):      # non-Bool predicate: selection.jl:52-55
        dfdb_mod.selection(dfdb_mod.DFView(p.d), ir.col(0) * 3)
    with pytest.raises(ValueError):
        p.o.view().add_predicate((ir.col(0) * 3).to_ir())


# ------------------------------------------------------------------ files: oracle writer (liblz4) -> device LZ4 decode
def test_table_files_roundtrip(oracle, dfdb_mod, ctx, tmp_path):
    from dfdb import ir
    n = 200_003
    rng = np.random.default_rng(11)
    sizes, data = oracle.gen_str(col_seed(3), 0, n)
    strs = oracle.flat_to_strings(sizes, data)
    strs_m = [None if i % 13 == 0 else s for i, s in enumerate(strs)]
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "x": oracle.gen_f64(col_seed(1), 0, n), "s": strs, "sm": strs_m,
            "iota": np.arange(1, n + 1, dtype=np.int64), "i16": rng.integers(-300, 300, n).astype(np.int16),
            "m": np.ma.masked_array(rng.integers(0, 100, n).astype(np.int64), mask=rng.random(n) < 0.2),
            "b": rng.integers(0, 2, n).astype(bool), "rnd": rng.integers(-2**62, 2**62, n).astype(np.int64),
            # LZ4 corner cases for the device decoder: one endless RLE match (far beyond the 8-KB LDS ring), a period-24 pattern,
            # long literal runs followed by far matches (offset > 8 KB)
            "zeros": np.zeros(n, np.int64), "period3": np.tile(np.array([7, -1, 2**40], np.int64), n // 3 + 1)[:n],
            "far": np.concatenate([rng.integers(-2**62, 2**62, 3000), np.zeros(10, np.int64)] * (n // 3010 + 1))[:n].astype(np.int64)}
    cols["far"][6000:9000] = cols["far"][0:3000]          # a 24-KB repeat at distance 48 KB
    for bs in (65536, 1000):
        p = Pair(oracle, dfdb_mod, cols, block_size=bs, via_files=str(tmp_path / f"tb{bs}"))
        ov, dv = apply_stages(p, [])
        assert_same(p, ov, dv)
        ov, dv = apply_stages(p, [("pred", (ir.col(0) > 500_000) & (ir.col(2) == "sony"))])
        assert_same(p, ov, dv)
        ov, dv = apply_stages(p, [("pred", ir.ismissing(ir.col(6)) | ir.ismissing(ir.col(3)))])
        assert_same(p, ov, dv)


def lz4_corner_columns(variant, n=300_000):
    """the byte columns of test_lz4_decode_corner_cases (its docstring says what each is for)"""
    rng = np.random.default_rng(5 + variant)
    periodic = np.concatenate([np.tile(rng.integers(0, 256, per).astype(np.uint8), 2300 // per + 1)[:2300] for per in range(1, 131)])
    pieces = []
    base = rng.integers(0, 256, 20_000).astype(np.uint8)
    pos = 0
    while sum(len(x) for x in pieces) < n:
        lit = int(rng.integers(0, 401))
        pieces.append(rng.integers(0, 256, lit).astype(np.uint8))                    # literal run
        ln = int(rng.choice([4, 5, 18, 19, 20, 64, 65, 270, 271, 300, 1000, 9000]))
        start = int(rng.integers(0, len(base) - ln))
        pieces.append(base[start:start + ln])                                        # match somewhere in the first 20 KB of output
        if not pos:
            pieces.insert(0, base); pos = 1
    mixed = np.concatenate(pieces)[:n]
    # short sequences back to back (<= 14 literals, 4..18 match bytes) with match distances from 1 byte to 60 KB: the batch
    # decoders' candidate windows, in-chunk pointer chains, far-source prefetch slots and output-budget cuts
    short = bytearray(rng.integers(0, 256, 4096).astype(np.uint8).tobytes())
    while len(short) < n:
        short += rng.integers(0, 256, int(rng.integers(0, 15))).astype(np.uint8).tobytes()
        ml = int(rng.integers(4, 19))
        hi = (8, 64, 2000, 60_000)[int(rng.integers(0, 4))]
        d = int(rng.integers(1, min(hi, len(short)) + 1))
        for k in range(ml):
            short.append(short[len(short) - d])
    shortseq = np.frombuffer(bytes(short[:n]), np.uint8)
    cols = {"periodic": np.resize(periodic, n), "mixed": mixed, "noise": rng.integers(0, 256, n).astype(np.uint8), "shortseq": shortseq,
            "runs": np.repeat(rng.integers(0, 4, n // 50 + 1).astype(np.uint8), rng.integers(1, 100, n // 50 + 1))[:n]}
    cols["runs"] = np.resize(cols["runs"], n)
    return cols


@pytest.mark.parametrize("pipe", [0, 1, 10, 15])     # K7 one wave per block (4-window superbatch), the two-wave pipeline, round 2's 8-window shape, the 7-waves-per-SIMD shape (by default the block count chooses between the first two: these files are small)
@pytest.mark.parametrize("variant", [0, 1, 2])       # (three differently seeded data sets)
def test_lz4_decode_corner_cases(oracle, dfdb_mod, ctx, tmp_path, variant, pipe):
    """Byte columns built to hit every branch of the device LZ4 decoders: periodic data of every period 1..130 (overlapping
    matches with offset < 64, = 64, > 64), literal runs of 0..400 bytes between matches (length-byte chains), matches at
    distances beyond the 8-KB LDS ring, incompressible blocks (one 64-KB literal run), sequences that straddle the 2-KB
    staging chunks, and blocks that end right after a match / with a 5-byte literal tail."""
    n = 300_000
    cols = lz4_corner_columns(variant, n)
    ctx.set_option("lz4_pipeline", pipe)
    try:
        for bs in (65536, 50_000, 4099):
            p = Pair(oracle, dfdb_mod, cols, block_size=bs, via_files=str(tmp_path / f"c{bs}"))
            ov, dv = apply_stages(p, [])
            assert_same(p, ov, dv)
        # the same bytes as an Int64 column, decoded FUSED with a predicate (K7 SCAN: the bitmap leaves the decoder) and re-decoded in place
        ctx.set_option("keep_compressed", 1)
        for name in ("mixed", "shortseq", "periodic"):
            v8 = np.ascontiguousarray(cols[name][: n // 8 * 8]).view(np.int64)
            p8 = Pair(oracle, dfdb_mod, {"v": v8}, block_size=8192, via_files=str(tmp_path / f"v8{name}"))
            from dfdb import ir
            c = int(np.median(v8))
            ctx.set_option("decode_on_scan", 1)
            ov, dv = apply_stages(p8, [("pred", ir.col(0) > c)])
            assert_same(p8, ov, dv)
            ctx.set_option("decode_on_scan", 0)
            p8.d.decode_resident("v")
            ov, dv = apply_stages(p8, [("pred", ir.col(0) <= c)])
            assert_same(p8, ov, dv)
    finally:
        ctx.set_option("lz4_pipeline", -1); ctx.set_option("keep_compressed", 0); ctx.set_option("decode_on_scan", 0)


def test_open_table_errors(oracle, dfdb_mod, ctx, tmp_path):
    with pytest.raises(dfdb_mod.DfdbError):
        dfdb_mod.open_table(str(tmp_path / "nope"))
    t = oracle.Table(block_size=10)
    t.add_column("a", np.arange(25, dtype=np.int64))
    t.save(str(tmp_path / "tb"))
    # corrupt header: wrong block size (test/tables.jl:52-58)
    import struct
    f = tmp_path / "tb" / "1.bin"
    raw = f.read_bytes()
    f.write_bytes(struct.pack("<q", 11) + raw[8:])
    with pytest.raises(dfdb_mod.DfdbError):
        dfdb_mod.open_table(str(tmp_path / "tb"))
    # corrupt LZ4 payload (first token claims an endless literal run) -> "decompression error" (BlockStreams.jl:112)
    hdr = 8 + 4 + len("Int64")
    bad = bytearray(raw)
    bad[hdr + 20: hdr + 24] = b"\xff\xff\xff\xff"
    f.write_bytes(bytes(bad))
    with pytest.raises(dfdb_mod.DfdbError, match="decompression error"):
        dfdb_mod.open_table(str(tmp_path / "tb"))
    ot = oracle.Table.open(str(tmp_path / "tb"))
    assert ot.view().nrow() == 25            # range-only count never decompresses (isonly_range: blocksiterator.jl:135)
    with pytest.raises(OSError):
        ot.view().materialize()


@pytest.mark.parametrize("pipe", [0, 1, 10, 15])     # every compiled form of K7 (see test_lz4_decode_corner_cases)
@pytest.mark.parametrize("variant", [0, 1, 2])       # (three differently seeded damage sets)
def test_lz4_decoders_survive_corrupt_blocks(oracle, dfdb_mod, ctx, tmp_path, variant, pipe):
    """LZ4_decompress_safe semantics (BlockStreams.jl:110-112): a damaged block either still decodes to `origin` bytes or raises
    "decompression error" — it never writes outside the block, hangs or takes the device down.  Random byte flips, truncated
    sequences, zero / huge offsets and endless length chains in the payloads of valid files; after every attempt the intact file
    must still load and compare equal."""
    import struct
    rng = np.random.default_rng(17 + variant)
    n = 150_000
    short = bytearray(rng.integers(0, 256, 4096).astype(np.uint8).tobytes())
    while len(short) < n:
        short += rng.integers(0, 256, int(rng.integers(0, 15))).astype(np.uint8).tobytes()
        ml = int(rng.integers(4, 40)); d = int(rng.integers(1, min(60_000, len(short)) + 1))
        for k in range(ml):
            short.append(short[len(short) - d])
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "s": np.frombuffer(bytes(short[:n]), np.uint8)}
    good = tmp_path / "good"
    ot = oracle.Table(block_size=65536)
    for k, v in cols.items():
        ot.add_column(k, v)
    ot.save(str(good))
    ctx.set_option("lz4_pipeline", pipe)
    try:
        outcomes = {"error": 0, "decoded": 0}
        ntrials = int(os.environ.get("DFDB_FUZZ_TRIALS", "24"))          # (a soak run: DFDB_FUZZ_TRIALS=400)
        for trial in range(ntrials):
            bad = tmp_path / f"bad{trial}"
            shutil.copytree(good, bad)
            fn = bad / ("1.bin" if trial % 2 == 0 else "2.bin")
            raw = bytearray(fn.read_bytes())
            tl, = struct.unpack_from("<i", raw, 8)
            pos, blocks = 12 + tl, []
            while pos < len(raw):
                rows, origin, comp = struct.unpack_from("<iqq", raw, pos)
                blocks.append((pos + 20, comp)); pos += 20 + comp
            b0, blen = blocks[int(rng.integers(0, len(blocks)))]
            kind = trial % 6
            if kind == 0:                                   # scattered byte flips
                for _ in range(int(rng.integers(1, 20))):
                    raw[b0 + int(rng.integers(0, blen))] ^= int(rng.integers(1, 256))
            elif kind == 1:                                 # a run of 0xff: endless length chains
                at = b0 + int(rng.integers(0, max(1, blen - 600)))
                raw[at:at + 600] = b"\xff" * min(600, b0 + blen - at)
            elif kind == 2:                                 # a run of zeros: zero offsets, zero-length tokens
                at = b0 + int(rng.integers(0, max(1, blen - 300)))
                raw[at:at + 300] = b"\x00" * min(300, b0 + blen - at)
            elif kind == 3:                                 # the first sequence points before the start of the output
                raw[b0:b0 + 4] = bytes([0x10, 0x41, 0xff, 0xff])
            elif kind == 4:                                 # random garbage over the tail of the block
                k = int(rng.integers(1, min(blen, 5000)))
                raw[b0 + blen - k:b0 + blen] = rng.integers(0, 256, k).astype(np.uint8).tobytes()
            else:                                           # random garbage over the head of the block
                k = int(rng.integers(1, min(blen, 5000)))
                raw[b0:b0 + k] = rng.integers(0, 256, k).astype(np.uint8).tobytes()
            fn.write_bytes(bytes(raw))
            try:
                t = dfdb_mod.open_table(str(bad))
                assert dfdb_mod.nrow(t) == n                # decoded to origin bytes (LZ4 carries no checksum: the bytes may differ)
                t.close()
                outcomes["decoded"] += 1
            except dfdb_mod.DfdbError as e:
                assert "decompression error" in str(e)
                outcomes["error"] += 1
            shutil.rmtree(bad)
            t = dfdb_mod.open_table(str(good))              # the device is still healthy and exact
            got = dfdb_mod.materialize(t)
            assert np.array_equal(np.asarray(got["a"]), cols["a"]) and np.array_equal(np.asarray(got["s"]), cols["s"])
            t.close()
        assert outcomes["error"] >= ntrials // 3            # most of these damages cannot decode
    finally:
        ctx.set_option("lz4_pipeline", -1)


@pytest.mark.parametrize("n", [1, 1023, 1025, 70_001, 300_000])
def test_string_capture_in_match_pass(oracle, dfdb_mod, ctx, n):
    """materialize(t[s OP "const", :]) with s projected: the match kernel keeps the selected rows' sizes and bytes per tile (K5 CAP)
    and the projection of s is a contiguous copy per tile.  Every short-pattern operator, empty strings, missing values, multi-byte
    characters; the result must equal the oracle's (getindex(a, r): FlatStringsVectors.jl:136-157) and the captured path must
    actually have run."""
    from dfdb import ir
    rng = np.random.default_rng(n)
    words = ["sony", "so", "", "sonya", "apple", "x", "samsungs", "né", "sonysony", "asony", "microsoft", "microsoftware", "microsofa",
             "a-rather-long-category-name-that-needs-several-probes-to-compare", "a-rather-long-category-name-that-needs-several-probes-to-compara"]
    strs = [words[int(k)] for k in rng.integers(0, len(words), n)]
    strs_m = [None if rng.random() < 0.1 else w for w in strs]
    cols = {"s": strs, "sm": strs_m, "a": oracle.gen_i64(col_seed(0), 0, n)}
    p = Pair(oracle, dfdb_mod, cols, block_size=65536)
    ctx.profile(True)
    try:
        # (a comparison with a Union{String,Missing} column is not a Bool selection — ArgumentError on both sides — so the captured
        # column is the plain one; the nullable one rides along through the ordinary gather)
        for pred in (ir.col(0) == "sony", ir.col(0) != "sony", ir.startswith(ir.col(0), "so"), ir.endswith(ir.col(0), "ny"),
                     ir.col(0) == "", ir.col(0) == "samsungs", ir.startswith(ir.col(0), ""), ir.endswith(ir.col(0), "é"),
                     # patterns longer than one 8-byte probe (the rest is compared only where the first 8 bytes match)
                     ir.col(0) == "microsoft", ir.col(0) != "microsoftware", ir.startswith(ir.col(0), "microsoft"), ir.endswith(ir.col(0), "osoftware"),
                     ir.endswith(ir.col(0), "icrosoft"), ir.col(0) == "a-rather-long-category-name-that-needs-several-probes-to-compare",
                     ir.startswith(ir.col(0), "a-rather-long-category-name-that-needs-several-probes-to-compar")):
            n0, _ = ctx.profile_get("str_compact_captured"); f0, _ = ctx.profile_get("fill_const_strings")
            ov, dv = apply_stages(p, [("pred", pred)])
            assert_same(p, ov, dv)
            n1, _ = ctx.profile_get("str_compact_captured"); f1, _ = ctx.profile_get("fill_const_strings")
            if ov.nrow() > 0:
                # `col == "const"` pins the projected column to one value: it is written out as a constant, nothing is captured or gathered
                if pred.op == ir.EQ:
                    assert f1 > f0 and n1 == n0, "the constant-column path did not run"
                else:
                    assert n1 > n0, "the capture path did not run"
    finally:
        ctx.profile(False)


def test_disjunctions_of_simple_terms_take_the_scan_kernel(oracle, dfdb_mod, ctx):
    """(a > c1) | (x < c2) | ... is one k_scan_terms pass (combine_or), alone, beside AND terms / a string term, and after a range
    stage (ANDed with the mask so far); the interpreter is not launched for it."""
    from dfdb import ir
    n = 200_003
    sizes, data = oracle.gen_str(col_seed(3), 0, n)
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "x": oracle.gen_f64(col_seed(1), 0, n), "s": oracle.flat_to_strings(sizes, data),
            "i16": np.random.default_rng(9).integers(-300, 300, n).astype(np.int16), "b": np.random.default_rng(10).integers(0, 2, n).astype(bool)}
    p = Pair(oracle, dfdb_mod, cols, block_size=65536)
    a, x, s, i16, b = ir.col(0), ir.col(1), ir.col(2), ir.col(3), ir.col(4)
    ctx.profile(True)
    try:
        cases = [
            [("pred", (a > 900_000) | (x < 100.0))],
            [("pred", (a > 990_000) | (x < 10.0) | (i16 == 7) | (a <= 5))],
            [("pred", ((a > 900_000) | (x < 100.0)) & (i16 > 0) & (s == "sony"))],
            [("range", 1000, 1, 150_000), ("pred", (a < 100_000) | (i16 >= 299.5))],
            [("pred", ((a > 500_000) | (x < 1000.0)) & ((i16 < 0) | (a == 123_456)))],
            [("pred", ir.isin(i16, [1, 11, 21]))],                                       # in.(a, Ref([1,11,21])): test/broadcast.jl:63-71
            [("pred", ir.isin(a, [5.0, 7, 123_456.5]) | (x < 1.0))],                      # Int column, Float members: exact ==
            [("pred", ir.isin(i16, [1, 2, 3]) & (a > 100_000))],
            [("pred", b)],                                                               # a Bool column is a selection by itself
            [("pred", b & (a > 500_000))],
            [("pred", b | (x < 50.0))],
        ]
        for stages in cases:
            n0, _ = ctx.profile_get("interp_predicate")
            ov, dv = apply_stages(p, stages)
            assert_same(p, ov, dv)
            n1, _ = ctx.profile_get("interp_predicate")
            assert n1 == n0, "a disjunction of simple terms went to the interpreter"
        ov, dv = apply_stages(p, [("pred", ir.isin(i16, list(range(0, 40, 3))))])   # 14 members: the interpreter's set probe
        assert_same(p, ov, dv)
    finally:
        ctx.profile(False)


def test_ismissing_of_a_column_is_its_bitmap(oracle, dfdb_mod, ctx):
    """ismissing(col) / !ismissing(col) over a Union{T,Missing} fixed-width column never runs the interpreter: the column's missing
    bitmap is the mask (alone, negated, beside other terms, after a range stage, in blocks that do not start on a word boundary)."""
    from dfdb import ir
    n = 150_001
    rng = np.random.default_rng(21)
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "m": np.ma.masked_array(rng.integers(0, 100, n).astype(np.int64), mask=rng.random(n) < 0.3),
            "mf": np.ma.masked_array(rng.random(n).astype(np.float32), mask=rng.random(n) < 0.6)}
    for bs in (65536, 1000):
        p = Pair(oracle, dfdb_mod, cols, block_size=bs)
        a, m, mf = ir.col(0), ir.col(1), ir.col(2)
        ctx.profile(True)
        try:
            for stages in ([("pred", ir.ismissing(m))], [("pred", ~ir.ismissing(mf))], [("pred", ir.ismissing(m) & (a > 500_000) & ~ir.ismissing(mf))],
                           [("range", 100, 3, 140_000), ("pred", ~ir.ismissing(m))], [("pred", a < 300_000), ("pred", ir.ismissing(mf))]):
                n0, _ = ctx.profile_get("interp_predicate"); k0, _ = ctx.profile_get("missing_mask")
                ov, dv = apply_stages(p, stages)
                assert_same(p, ov, dv)
                n1, _ = ctx.profile_get("interp_predicate"); k1, _ = ctx.profile_get("missing_mask")
                assert n1 == n0 and k1 > k0
            # sum(ismissing.(t.m)) (docs/src/index.md:326-328) and friends: a computed Bool column is summed as one more predicate
            mm, am = np.ma.getmaskarray(cols["m"]), cols["a"]
            assert dfdb_mod.ismissing(p.d.m).sum() == int(mm.sum())
            assert (p.d.a > 500_000).sum() == int((am > 500_000).sum())
            sel = am < 300_000
            v = p.d[p.d.a < 300_000, dfdb_mod.ALL]
            assert dfdb_mod.ismissing(v.m).sum() == int(mm[sel].sum())
            assert abs(dfdb_mod.ismissing(v.m).mean() - mm[sel].mean()) < 1e-12
        finally:
            ctx.profile(False)


def test_arrow_string_output(oracle, dfdb_mod, ctx):
    """materialize() with set_string_output("arrow"): String columns come back as pyarrow-backed pandas arrays over the engine's own
    (sizes, arena) buffers; same strings, missing rows are nulls."""
    n = 30_000
    sizes, data = oracle.gen_str(col_seed(3), 0, n)
    strs = oracle.flat_to_strings(sizes, data)
    cols = {"s": strs, "sm": [None if i % 7 == 0 else w for i, w in enumerate(strs)], "a": oracle.gen_i64(col_seed(0), 0, n)}
    t = dfdb_mod.DFTable.from_columns(cols)
    v = t[t.a > 500_000, dfdb_mod.ALL]
    want = dfdb_mod.materialize(v)
    dfdb_mod.set_string_output("arrow")
    try:
        got = dfdb_mod.materialize(v)
    finally:
        dfdb_mod.set_string_output("object")
    assert len(got) == len(want) and got["a"].tolist() == want["a"].tolist()
    assert got["s"].tolist() == want["s"].tolist()
    import pandas as pd
    assert [None if pd.isna(x) else x for x in got["sm"].tolist()] == want["sm"].tolist()


# ------------------------------------------------------------------ aggregates
def test_aggregates(oracle, dfdb_mod, ctx):
    from dfdb import ir
    n = 250_000
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "x": oracle.gen_f64(col_seed(1), 0, n)}
    p = Pair(oracle, dfdb_mod, cols, block_size=65536)
    ov, dv = apply_stages(p, [("pred", ir.col(0) > 899_999)])
    sel = cols["a"] > 899_999
    ca, cx = dv[dfdb_mod.ALL, "a"], dv[dfdb_mod.ALL, "x"]
    assert ca.sum() == int(cols["a"][sel].sum()) == ov.sum_i64(0)
    assert ca.min() == int(cols["a"][sel].min()) and ca.max() == int(cols["a"][sel].max())
    # Float64 sum: the reference adds left to right (column.jl:102-126); the device sums pairwise.
    # tolerance: |err| <= n * eps * sum|x|  (stated in DESIGN.md)
    want = ov.sum_f64(1)
    tol = len(cols["x"][sel]) * np.finfo(np.float64).eps * float(np.abs(cols["x"][sel]).sum())
    assert abs(cx.sum() - want) <= tol
    assert cx.min() == float(cols["x"][sel].min()) and cx.max() == float(cols["x"][sel].max())
    assert abs(cx.mean() - want / sel.sum()) <= tol


@pytest.mark.parametrize("n", [1000, 250_001])
def test_sum_fused_into_the_scan(oracle, dfdb_mod, ctx, n):
    """sum(col) / mean(col) over a filtered view (docs/src/index.md:503-509): when the column is a simple term of the launch that
    produces the final mask, the scan adds the selected values up per tile (k_scan_terms EXTRA = 2) and dfdb_aggregate only reduces
    the partials.  Int64 sums are exact (wrapping), Float64 within n * eps * sum|x|; shapes that cannot fuse take the ordinary
    reduce and agree too."""
    from dfdb import ir
    sizes, data = oracle.gen_str(col_seed(3), 0, n)
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "x": oracle.gen_f64(col_seed(1), 0, n), "s": oracle.flat_to_strings(sizes, data),
            "big": (oracle.gen_i64(col_seed(5), 0, n).astype(np.int64) << 44) - 7}          # sums wrap around Int64
    t = dfdb_mod.DFTable.from_columns(cols, block_size=65536)
    a, x, big = cols["a"], cols["x"], cols["big"]
    strs = np.array(cols["s"], dtype=object)
    ctx.profile(True)
    try:
        def fused(fn):
            n0, _ = ctx.profile_get("reduce_partials")
            r = fn()
            n1, _ = ctx.profile_get("reduce_partials")
            return r, n1 > n0
        def ftol(sel): return max(1, int(sel.sum())) * np.finfo(np.float64).eps * float(np.abs(x[sel]).sum()) + 1e-300
        # one term, Int64
        sel = a > 500_000
        r, f = fused(lambda: t[t.a > 500_000, dfdb_mod.ALL][dfdb_mod.ALL, "a"].sum())
        assert f and r == int(a[sel].sum())
        # wrapping Int64 sum, two terms (the summed column is not the first term)
        sel = (a > 100_000) & (big > -(1 << 62))
        r, f = fused(lambda: t[(t.a > 100_000) & (t.big > -(1 << 62)), dfdb_mod.ALL][dfdb_mod.ALL, "big"].sum())
        want = int(np.sum(big[sel].astype(np.uint64), dtype=np.uint64).astype(np.int64)) if sel.any() else 0
        assert f and r == want
        # Float64 under a conjunction with a string term (the string kernel runs first, the terms last, against its mask)
        sel = (a > 300_000) & (x < 1500.0) & (strs != "sony")
        v = t[(t.a > 300_000) & (t.x < 1500.0) & (t.s != "sony"), dfdb_mod.ALL]
        r, f = fused(lambda: v[dfdb_mod.ALL, "x"].sum())
        assert f and abs(r - float(x[sel].sum())) <= ftol(sel)
        r, f = fused(lambda: v[dfdb_mod.ALL, "x"].mean())
        assert f and abs(r - float(x[sel].mean())) <= ftol(sel) / max(1, sel.sum()) * 4
        # min / max are folded the same way (EXTRA = 3 / 4); an empty selection still raises
        r, f = fused(lambda: v[dfdb_mod.ALL, "x"].min())
        assert f and r == float(x[sel].min())
        r, f = fused(lambda: v[dfdb_mod.ALL, "a"].max())
        assert f and r == int(a[sel].max())
        r, f = fused(lambda: t[(t.a > 100_000) & (t.big > -(1 << 62)), dfdb_mod.ALL][dfdb_mod.ALL, "big"].min())
        assert f and r == int(big[(a > 100_000) & (big > -(1 << 62))].min())
        with pytest.raises(ValueError):
            t[(t.a > 5_000_000) & (t.x < 1.0), dfdb_mod.ALL][dfdb_mod.ALL, "x"].max()
        assert dfdb_mod.nrow(v) == int(sel.sum())
        # a range stage first, the predicate last: still the launch that makes the final mask
        m = min(n, 100_000)
        sel = np.zeros(n, bool); sel[:m] = x[:m] < 700.0
        r, f = fused(lambda: t[dfdb_mod.jr(1, m), dfdb_mod.ALL][("x", lambda c: c < 700.0), dfdb_mod.ALL][dfdb_mod.ALL, "x"].sum())
        assert f and abs(r - float(x[sel].sum())) <= ftol(sel)
        # shapes that cannot fuse: the predicate is not the last stage / the column is not a term / an OR
        r, f = fused(lambda: t[t.x < 700.0, dfdb_mod.ALL][dfdb_mod.jr(1, 50), dfdb_mod.ALL][dfdb_mod.ALL, "x"].sum())
        idx = np.flatnonzero(x < 700.0)[:50]
        assert not f and abs(r - float(x[idx].sum())) <= 1e-9
        r, f = fused(lambda: t[t.a > 500_000, dfdb_mod.ALL][dfdb_mod.ALL, "x"].sum())
        assert not f and abs(r - float(x[a > 500_000].sum())) <= ftol(a > 500_000)
        r, f = fused(lambda: t[(t.a > 900_000) | (t.x < 100.0), dfdb_mod.ALL][dfdb_mod.ALL, "x"].sum())
        sel = (a > 900_000) | (x < 100.0)
        assert not f and abs(r - float(x[sel].sum())) <= ftol(sel)
    finally:
        ctx.profile(False)


def test_query_outlives_closed_table(oracle, dfdb_mod, ctx):
    """Handles may be released in any order: a query whose table was closed is orphaned (errors, never dangles)."""
    from dfdb import ir
    t = dfdb_mod.DFTable.from_columns({"a": np.arange(100, dtype=np.int64)})
    v = t[("a", lambda a: a > 50), dfdb_mod.ALL]
    q = v._query()
    assert q.count() == 49
    t.close()
    with pytest.raises(ValueError, match="closed"):
        q.reset(); q.count()
    del q, v


# ------------------------------------------------------------------ block-range shards on one GPU
@pytest.mark.parametrize("world", [2, 3])
def test_block_range_shards_concatenate_to_table_order(oracle, dfdb_mod, ctx, tmp_path, world):
    """Each 'rank' loads only its contiguous block range of every column (dfdb_table_load), leading range stages and row
    numbers stay global through row_base, a range stage after a predicate gets the survivors of lower ranks through
    dfdb_query_count_prefix / dfdb_query_set_stage_base; rank-order concatenation must equal the oracle's single table."""
    import ctypes as C
    from dfdb import ir, _native as N
    from dfdb.sharding import block_range
    n, bs = 300_007, 4096
    sizes, data = oracle.gen_str(col_seed(2), 0, n)
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "x": oracle.gen_f64(col_seed(1), 0, n), "s": oracle.flat_to_strings(sizes, data)}
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "tb")
    ot.save(path)
    nblocks = -(-n // bs)
    shards = []
    for r in range(world):
        b0, b1 = block_range(nblocks, r, world)
        t = dfdb_mod.open_table(path, load=False)
        t.load(None, b0, b1)
        shards.append(t)
    a, x, s = ir.col(0), ir.col(1), ir.col(2)
    queries = {
        "pred": [("pred", (a > 700_000) & (s != "sony"))],
        "lead_range": [("range", 5, 7, 250_000), ("pred", x < 1000.0)],
        "range_after_pred": [("pred", a > 500_000), ("range", 11, 3, 90_000)],
        "two_exchanges": [("pred", a % 2 == 0), ("range", 10, 1, 100_000), ("pred", s == "dell"), ("idx", [1, 5, 400, 4999, 10**9])],
        "count_all": [],
    }
    for name, stages in queries.items():
        ov = ot.view()
        views = [dfdb_mod.DFView(t) for t in shards]
        for st in stages:
            if st[0] == "pred":
                ov.add_predicate(st[1].to_ir()); views = [dfdb_mod.selection(v, st[1]) for v in views]
            elif st[0] == "range":
                ov.add_range(st[1], st[2], st[3]); views = [dfdb_mod.selection(v, dfdb_mod.jr(st[1], st[2], st[3])) for v in views]
            else:
                ov.add_indices(st[1]); views = [dfdb_mod.selection(v, list(st[1])) for v in views]
        qs = [v._query() for v in views]
        for k, st in enumerate(stages):                      # the all-gather + exclusive scan, done by hand
            if k == 0 or st[0] == "pred":
                continue
            counts = []
            for q in qs:
                c = C.c_int64()
                N.check(N.load().dfdb_query_count_prefix(q._h, k, C.byref(c)))
                counts.append(c.value)
            for r, q in enumerate(qs):
                N.check(N.load().dfdb_query_set_stage_base(q._h, k, sum(counts[:r])))
        want_idx = ov.select_indices()
        got_idx = np.concatenate([q.indices() for q in qs])
        assert np.array_equal(got_idx, want_idx), name
        assert sum(q.count() for q in qs) == ov.nrow()
        want = ov.materialize()
        got = [q.materialize() for q in qs]
        assert np.array_equal(np.concatenate([g[0] for g in got]), want[0]), name
        assert np.array_equal(np.concatenate([g[1] for g in got]).view(np.uint64), want[1].view(np.uint64)), name
        assert np.array_equal(np.concatenate([g[2][0] for g in got]), want[2][0]) and np.array_equal(np.concatenate([g[2][1] for g in got]), want[2][1]), name
    # sum(x) over shards: one all-reduce of a scalar; tolerance n*eps*sum|x|
    tot = sum(v[dfdb_mod.ALL, "x"].sum() for v in [dfdb_mod.DFView(t) for t in shards])
    assert abs(tot - float(np.sum(cols["x"]))) <= n * np.finfo(float).eps * float(np.abs(cols["x"]).sum())


# ------------------------------------------------------------------ Union{T,Missing} inside expressions
def test_missing_propagation_and_three_valued_logic(oracle, dfdb_mod, ctx):
    """The known-answer table of tests/test_oracle_cpu.py (Julia's missing propagation, three-valued & and |, coalesce,
    ismissing of computed values) through the device interpreter, then random nullable columns against the oracle."""
    from dfdb import ir
    from test_oracle_cpu import missing_cases
    cols = {"m": np.ma.masked_array(np.array([1, 99, 3, 77, 0], np.int64), mask=[0, 1, 0, 1, 0]), "c": np.array([0, 0, 5, 5, 0], np.int64),
            "sm": ["a", None, "b", "ab", None]}
    p = Pair(oracle, dfdb_mod, cols, block_size=2)
    for name, e, want, miss in missing_cases():
        if want is None:
            with pytest.raises(ZeroDivisionError):
                dfdb_mod.materialize(dfdb_mod.DFView(p.d, dfdb_mod.Projection({"k": e})))
            continue
        ov, dv = apply_stages(p, [], proj=[("k", e)])
        assert_same(p, ov, dv)
        got = dv._query().materialize()[0]
        if miss is None:
            assert not isinstance(got, np.ma.MaskedArray) and np.array_equal(got, np.array(want, got.dtype)), name
        else:
            assert np.array_equal(np.ma.getmaskarray(got), np.array(miss, bool)), name
    with pytest.raises(ValueError):          # a Union{Missing,Bool} selection function is refused (selection.jl:52-55)
        dfdb_mod.selection(dfdb_mod.DFView(p.d), ir.col(0) > 2)
    ov, dv = apply_stages(p, [("pred", ir.coalesce((ir.col(0) > 2) | (ir.col(1) > 1), False))])
    assert_same(p, ov, dv)
    assert dv._query().indices().tolist() == [3, 4]

    rng = np.random.default_rng(23)
    n = 150_011
    strs = oracle.flat_to_strings(*oracle.gen_str(col_seed(3), 0, n))
    cols = {"m": np.ma.masked_array(rng.integers(-50, 50, n).astype(np.int64), mask=rng.random(n) < 0.3),
            "f": np.ma.masked_array(rng.normal(0, 10, n), mask=rng.random(n) < 0.1),
            "b": np.ma.masked_array(rng.integers(0, 2, n).astype(bool), mask=rng.random(n) < 0.5),
            "c": rng.integers(-5, 6, n).astype(np.int64), "i8": np.ma.masked_array(rng.integers(-128, 128, n).astype(np.int8), mask=rng.random(n) < 0.2),
            "sm": [None if i % 7 == 0 else s for i, s in enumerate(strs)]}
    p = Pair(oracle, dfdb_mod, cols, block_size=4096)
    m, f, b, c, i8, sm = (ir.col(k) for k in range(6))
    preds = [ir.coalesce(m > 10, False), ir.coalesce((m > 10) & b, True), ir.coalesce(b | (f < 0.0), False) & (c != 0),
             ir.ismissing(m + f) | ir.coalesce(sm == "sony", False), ~ir.ismissing(m * i8) & ir.coalesce(ir.rem(m, ir.coalesce(i8, ir.const(1, ir.I8)) * 0 + 7) == 1, False),
             ir.coalesce(ir.startswith(sm, "s") & (m < 0), False), ir.coalesce(ir.coalesce(m, 0) + ir.coalesce(i8, ir.const(1, ir.I8)) > f, False),
             ir.coalesce(~b, False) ^ (c > 0), ir.ismissing(sm) & ir.coalesce(b, False)]
    for pred in preds:
        ov, dv = apply_stages(p, [("pred", pred)])
        assert_same(p, ov, dv)
    projs = [[("k", m + c), ("j", m * f)], [("k", (m > 0) & b), ("j", (m > 0) | b)], [("k", ir.coalesce(m, -1) + i8)], [("k", -f), ("j", abs(i8))],
             [("k", ir.sizeof(sm) + m)], [("k", ir.float64(m) / 4.0)], [("k", ir.div(m, ir.coalesce(c, 1) * 0 + 3))]]
    for pr in projs:
        ov, dv = apply_stages(p, [("pred", c > -3)], proj=pr)
        assert_same(p, ov, dv)


# ------------------------------------------------------------------ projected predicate columns captured by the scan
@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 4097, 65_536 + 7, 300_017])
def test_materialize_captures_projected_predicate_columns(oracle, dfdb_mod, ctx, n):
    """materialize() hints the scan (dfdb_query_hint_materialize): a single-stage conjunction of simple terms keeps the selected
    values of a projected 8-byte predicate column; the result must be what the gather gives (= the oracle), for one term
    (k_scan_cmp) and several (k_scan_terms), Int64 / Float64 / UInt64, full and ragged tiles, and the capture kernel must run."""
    from dfdb import ir
    rng = np.random.default_rng(n)
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n), "b": oracle.gen_i64(col_seed(1), 0, n), "x": oracle.gen_f64(col_seed(2), 0, n),
            "u": rng.integers(0, 2**63, n).astype(np.uint64) * np.uint64(2), "i32": rng.integers(-100, 100, n).astype(np.int32)}
    p = Pair(oracle, dfdb_mod, cols, block_size=65536)
    a, b, x, u, i32 = (ir.col(k) for k in range(5))
    cases = [(a > 683_771, None), ((a > 683_771) & (x < 632.456), [("b", b), ("x", x)]), (x <= 1000.0, [("x", x), ("a", a)]),
             ((u >= 2**63) & (a < 500_000) & (i32 > 0), [("u", u), ("i32", i32)]), (a != 5, [("a2", a), ("a", a)]),
             ((i32 > 0) & (a >= 0), [("i32", i32)])]
    for pred, proj in cases:
        ov, dv = apply_stages(p, [("pred", pred)], proj=proj)
        ctx.profile(True)
        assert_same(p, ov, dv)
        ncap, _ = ctx.profile_get("compact_captured")
        ctx.profile(False)
        assert ncap >= 1 or proj == [("i32", i32)] or ov.nrow() == 0        # (a 4-byte column is not captured: plain gather)
    # not the only stage / an OR / a generic conjunct: no capture, same answers
    for stages, proj in [([("pred", a > 500_000), ("range", 1, 2, n)], [("a", a)]), ([("pred", (a > 900_000) | (x < 100.0))], [("x", x)]),
                         ([("pred", (a > 500_000) & (a % 7 == 0))], [("a", a)])]:
        ov, dv = apply_stages(p, stages, proj=proj)
        assert_same(p, ov, dv)



# ------------------------------------------------------------------ unique(col): first occurrences as a selection
def julia_unique(values, missing=None):
    """Base.unique: first occurrence order, isequal (NaN == NaN, 0.0 != -0.0, missing == missing)."""
    seen, out = set(), []
    for i, v in enumerate(values):
        if missing is not None and missing[i]:
            k = ("missing",)
        elif isinstance(v, (float, np.floating)):
            k = ("nan",) if v != v else ("f", float(v), bool(np.signbit(v)))
        else:
            k = v
        if k not in seen:
            seen.add(k); out.append(None if k == ("missing",) else v)
    return out


def test_unique_columns(oracle, dfdb_mod, ctx, tmp_path):
    rng = np.random.default_rng(31)
    n = 250_007
    strs = oracle.flat_to_strings(*oracle.gen_str(col_seed(3), 0, n))
    long_strs = [f"user-{int(k):07d}-{'x' * int(k % 23)}" for k in rng.integers(0, 5000, n)]
    f = rng.integers(-3, 4, n).astype(np.float64); f[::97] = np.nan; f[5::101] = -0.0
    cols = {"a": oracle.gen_i64(col_seed(0), 0, n) % 1000, "b": rng.integers(-2**62, 2**62, n).astype(np.int64), "f": f,
            "i8": rng.integers(-128, 128, n).astype(np.int8), "u": np.where(rng.random(n) < 0.01, np.uint64(2**64 - 1), rng.integers(0, 50, n).astype(np.uint64)),
            "s": strs, "ls": long_strs, "sm": [None if i % 11 == 3 else s for i, s in enumerate(strs)],
            "m": np.ma.masked_array(rng.integers(0, 30, n).astype(np.int64), mask=rng.random(n) < 0.2), "flag": rng.integers(0, 2, n).astype(bool)}
    t = dfdb_mod.DFTable.from_columns(cols, block_size=65536)
    for name in cols:
        got = getattr(t, name).unique()
        src = cols[name]
        if isinstance(src, np.ma.MaskedArray):
            want = julia_unique(src.data.tolist(), np.ma.getmaskarray(src).tolist())
            g = [None if mm else x for x, mm in zip(np.asarray(got.data).tolist(), np.ma.getmaskarray(got).tolist())]
            assert g == want, name
        elif isinstance(src, list):
            assert list(got) == julia_unique(src, [x is None for x in src]), name
        elif src.dtype.kind == "f":
            want = julia_unique(src.tolist())
            assert len(got) == len(want) and all((x != x and y != y) or (x == y and np.signbit(x) == np.signbit(y)) for x, y in zip(got.tolist(), want)), name
        else:
            assert got.tolist() == julia_unique(src.tolist()), name
    # unique(t.brand[t.brand .!= ""]) of the docs: over a filtered view, and count == number of distinct values
    v = t[(t.a > 500) & (t.s != "sony"), ["ls"]]
    want = julia_unique([x for x, a, s in zip(long_strs, cols["a"], strs) if a > 500 and s != "sony"])
    assert list(v.ls.unique()) == want
    with pytest.raises(NotImplementedError):
        (t.a * 2).unique()
    # the same column through a table that is never resident: per-chunk unique on the device, merged in order
    path = str(tmp_path / "tb")
    t.save(path)
    tb = dfdb_mod.open_table(path, load=False)
    assert list(tb.s.unique()) == julia_unique(strs) and tb.a.unique().tolist() == julia_unique(cols["a"].tolist())
    # (numeric chunks merge as arrays on the host, by isequal images: one NaN, -0.0 apart from 0.0, the column's own dtype)
    for name in ("i8", "u", "flag", "b"):
        got = getattr(tb, name).unique()
        assert got.dtype == cols[name].dtype and got.tolist() == julia_unique(cols[name].tolist()), name
    got, want = tb.f.unique(), julia_unique(cols["f"].tolist())
    assert got.dtype == np.float64 and len(got) == len(want) and all((x != x and y != y) or (x == y and np.signbit(x) == np.signbit(y)) for x, y in zip(got.tolist(), want))
    sm = tb.sm.unique()
    assert list(sm) == julia_unique(cols["sm"], [x is None for x in cols["sm"]])
    # groupreduce over the table that is not resident: per-chunk groups merged on the host in chunk order == the resident table's answer
    from dfdb import api as _api
    keep = _api.DEFAULT_CHUNK_BLOCKS
    _api.DEFAULT_CHUNK_BLOCKS = 1                     # four chunks
    try:
        vb, vt = tb[tb.a > 100, dfdb_mod.ALL], t[t.a > 100, dfdb_mod.ALL]
        for by in ("a", "s", "sm", "m", "f", "i8"):
            for col, stats in (("b", ("count", "sum", "min", "max")), ("u", ("sum", "max", "min")), ("f", ("sum", "min", "max", "mean"))):
                if col == by:
                    continue
                for stat in stats:
                    got, want = dfdb_mod.groupreduce(vb, by, col, stat), dfdb_mod.groupreduce(vt, by, col, stat)
                    gk, wk = got[by].tolist(), want[by].tolist()
                    assert len(gk) == len(wk) and all((x == y) or (x != x and y != y) or (x is None and y is None) for x, y in zip(gk, wk)), (by, col, stat)
                    assert got["count"].tolist() == want["count"].tolist(), (by, col, stat)
                    if stat != "count":
                        g, w = got[stat].to_numpy(), want[stat].to_numpy()
                        assert g.dtype == w.dtype, (by, col, stat, g.dtype, w.dtype)
                        if g.dtype.kind == "f":
                            assert np.allclose(g, w, rtol=1e-9, atol=1e-9, equal_nan=True), (by, col, stat)
                        else:
                            assert np.array_equal(g, w), (by, col, stat)
    finally:
        _api.DEFAULT_CHUNK_BLOCKS = keep


# ------------------------------------------------------------------ groupreduce (aggregate.jl:1-36, completed to its intent)
def _np_group_ids(keys):
    """numpy restatement of the reference's numbering: group_map[elem] = length(group_map) + 1 on first appearance (aggregate.jl:21-28)"""
    order, gid, seen = [], np.empty(len(keys), np.int64), {}
    for i, k in enumerate(keys):
        kk = "missing" if k is None or k is np.ma.masked else (("nan",) if isinstance(k, float) and k != k else k)
        if kk not in seen:
            seen[kk] = len(seen); order.append(k)
        gid[i] = seen[kk]
    return order, gid


def _np_groupreduce(ids, vals, stat):
    order, gid = ids
    ng = len(order)
    cnt = np.bincount(gid, minlength=ng)
    if stat == "count":
        return order, cnt, None
    if ng == 0:
        return order, cnt, np.zeros(0)
    if stat in ("sum", "mean"):
        acc = np.zeros(ng, np.float64 if vals.dtype.kind == "f" else np.int64)
        np.add.at(acc, gid, vals.astype(acc.dtype))
        return order, cnt, acc if stat == "sum" else acc.astype(np.float64) / np.maximum(cnt, 1)
    acc = np.full(ng, vals.max() if stat == "min" else vals.min(), vals.dtype)
    (np.minimum if stat == "min" else np.maximum).at(acc, gid, vals)
    return order, cnt, acc


@pytest.mark.parametrize("n", [0, 1, 5000, 300_000])
def test_groupreduce_matches_first_appearance_numbering(oracle, dfdb_mod, ctx, n):
    """test/aggregate.jl:20 calls groupreduce(tb[:, :], (:a,), c = :c => Mean()); the reference's function stops after numbering the groups.
    Here: groups in order of first appearance, count + sum / min / max / mean per group, Int64 and String keys (few groups: LDS accumulators;
    many groups: global atomics), a nullable key (missing is a group), over a filtered view; the selection is intact afterwards."""
    rng = np.random.default_rng(31 + n)
    a = rng.integers(0, 7, n).astype(np.int64) * 1000 - 3000
    many = rng.integers(0, max(n // 3, 1), n).astype(np.int64)
    c = rng.integers(-1000, 1000, n).astype(np.int64)
    x = rng.normal(size=n) * 100
    u8 = rng.integers(0, 255, n).astype(np.uint8)
    words = ["apple", "sony", "dell", "", "microsoft", "né"]
    s = [words[i] for i in rng.integers(0, len(words), n)]
    m = np.ma.masked_array(rng.integers(0, 4, n).astype(np.int64), mask=rng.random(n) < 0.3)
    t = dfdb_mod.DFTable.from_columns({"a": a, "many": many, "c": c, "x": x, "u8": u8, "s": s, "m": m}, block_size=4096)
    sel = c > -500
    v = t[t.c > -500, dfdb_mod.ALL]
    for by, keys in (("a", a), ("many", many), ("s", np.array(s, dtype=object)), ("m", m)):
        ids = _np_group_ids([keys[i] for i in np.nonzero(sel)[0]])
        for col, vals, stats in (("c", c, ("count", "sum", "min", "max", "mean")), ("x", x, ("sum", "min", "max", "mean")), ("u8", u8, ("sum", "max"))):
            for stat in stats:
                got = dfdb_mod.groupreduce(v, by, col, stat)
                order, cnt, want = _np_groupreduce(ids, vals[sel], stat)
                assert len(got) == len(order), (by, col, stat)
                import pandas as pd
                gk = [None if (not isinstance(k, str) and pd.isna(k)) else (k if isinstance(k, str) else int(k)) for k in got[by].tolist()]
                wk = [None if (k is np.ma.masked or k is None) else (k if isinstance(k, str) else int(k)) for k in order]
                assert gk == wk, (by, col, stat)
                assert got["count"].tolist() == cnt.tolist(), (by, col, stat)
                if stat != "count" and len(order):
                    g = got[stat].to_numpy()
                    if col == "x" and stat in ("sum", "mean"):
                        assert np.allclose(g, want, rtol=1e-9, atol=1e-6), (by, col, stat)      # atomic double adds: no fixed order
                    elif stat == "mean":
                        assert np.allclose(g, want, rtol=1e-12), (by, col, stat)
                    else:
                        assert np.array_equal(g.astype(np.float64) if col == "x" else g.astype(np.int64), want.astype(np.float64) if col == "x" else want.astype(np.int64)), (by, col, stat)
    assert dfdb_mod.nrow(v) == int(sel.sum())            # the view's own query is untouched
    with pytest.raises(NotImplementedError):
        dfdb_mod.groupreduce(t[dfdb_mod.ALL, dfdb_mod.ALL], "a", "s", "sum")     # a String is no value column


@pytest.mark.parametrize("dtype", [np.int8, np.int32, np.int64, np.uint16, np.uint64, np.float32, np.float64])
def test_two_comparisons_of_one_column_fold_into_an_interval_term(oracle, dfdb_mod, ctx, dtype):
    """`65 > x > 34` (test/selection.jl:53) lowers to (65 > x) & (x > 34): two simple terms over ONE column.  The engine folds them into one
    interval term of the scan kernel (the column is read once); every pair of operators must give what the oracle and numpy give, alone,
    beside terms of other columns, as the captured / summed last term, and after a range stage (AND_EXISTING form)."""
    from dfdb import ir
    import operator
    rng = np.random.default_rng(11)
    n = 70_003
    kind = np.dtype(dtype).kind
    if kind == "f":
        x = (rng.integers(-400, 400, n) / 4).astype(dtype); x[::89] = np.nan
        lo, hi = dtype(-20.25), dtype(33.5)
    elif kind == "u":
        x = rng.integers(0, 200, n).astype(dtype); lo, hi = 34, 65
    else:
        x = rng.integers(-100, 100, n).astype(dtype); lo, hi = -34, 65
    y = rng.integers(0, 1000, n).astype(np.int64)
    z = (x.astype(np.float64) * 3).astype(np.float64) if kind != "f" else rng.random(n)
    p = Pair(oracle, dfdb_mod, {"x": x, "y": y, "z": z}, block_size=8192)
    cx, cy = ir.col(0), ir.col(1)
    ops = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge, "==": operator.eq, "!=": operator.ne}
    ctx.profile(True)
    try:
        for o1, f1 in ops.items():
            for o2, f2 in ops.items():
                n0, _ = ctx.profile_get("scan_terms")
                ov, dv = apply_stages(p, [("pred", f1(ir.const(hi), cx) & f2(cx, ir.const(lo)))])
                assert_same(p, ov, dv)
                with np.errstate(invalid="ignore"):
                    want = np.nonzero(f1(hi, x) & f2(x, lo))[0] + 1
                assert np.array_equal(dv._query().indices(), want), (o1, o2)
                n1, _ = ctx.profile_get("scan_terms")
                assert n1 > n0                                  # the folded term runs in the multi-term scan kernel, not the interpreter
        n0i, _ = ctx.profile_get("interp_predicate")
        # beside another column's term; three comparisons of x (the third starts a second term); after a range stage
        for stages in ([("pred", (cx > lo) & (cy < 700) & (cx <= hi))], [("pred", (cx > lo) & (cx <= hi) & (cx != lo + 1) & (cy >= 10))],
                       [("range", 100, 3, n - 5), ("pred", (cx >= lo) & (cx < hi))]):
            ov, dv = apply_stages(p, stages)
            assert_same(p, ov, dv)
        n1i, _ = ctx.profile_get("interp_predicate")
        assert n1i == n0i
    finally:
        ctx.profile(False)
    if dtype in (np.int64, np.uint64, np.float64):              # the interval as the capturing / summing last term
        ov, dv = apply_stages(p, [("pred", (cy < 900) & (cx > lo) & (cx < hi))], proj=[("x", cx), ("y", cy)])
        assert_same(p, ov, dv)
        t = p.d
        with np.errstate(invalid="ignore"):
            sel = (x > lo) & (x < hi)
        got = t[(t.x > lo) & (t.x < hi), dfdb_mod.ALL][dfdb_mod.ALL, "x"].sum()
        if kind == "f":
            assert abs(got - float(x[sel].astype(np.float64).sum())) <= 1e-9 * max(1.0, float(np.abs(x[sel]).sum()))
        else:
            assert got == int(x[sel].astype(np.int64).sum())


@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.int32, np.int64])
def test_rem_by_a_constant_is_a_scan_term(oracle, dfdb_mod, ctx, dtype):
    """`a % 50 == 0` (test/selection.jl:21, test/view.jl): rem(col, m) OP const over a signed integer column runs in the scan kernel (division by
    the invariant |m| as multiply + shift), not in the interpreter.  Julia's rem takes the sign of the dividend; the result must equal the
    oracle's generic evaluation and numpy's fmod for every operator, small / large / negative / power-of-two divisors and the extreme values."""
    from dfdb import ir
    import operator
    rng = np.random.default_rng(5)
    n = 50_021
    info = np.iinfo(dtype)
    x = rng.integers(info.min, info.max, n, dtype=np.int64, endpoint=True).astype(dtype)
    x[:6] = [info.min, info.max, 0, -1, 1, info.min + 1]
    y = rng.integers(0, 100, n).astype(np.int64)
    p = Pair(oracle, dfdb_mod, {"x": x, "y": y}, block_size=4096)
    cx, cy = ir.col(0), ir.col(1)
    ops = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge, "==": operator.eq, "!=": operator.ne}
    divisors = [2, 3, 50, -7, 64, -128, 1000003, 2**40 + 1, -(2**62) - 5, -(2**63), 2**63 - 1]
    ctx.profile(True)
    try:
        n0i, _ = ctx.profile_get("interp_predicate")
        for m in divisors:
            r = np.fmod(x.astype(np.int64), np.int64(m)) if m != -(2**63) else np.where(x.astype(np.int64) == -(2**63), 0, x.astype(np.int64))
            for name, f in ops.items():
                c = 0 if name in ("==", "!=") else (1 if m > 0 else -1)
                ov, dv = apply_stages(p, [("pred", f(cx % ir.const(m), ir.const(c)))])
                assert_same(p, ov, dv)
                assert np.array_equal(dv._query().indices(), np.nonzero(f(r, c))[0] + 1), (m, name)
        # constant on the left, an interval of remainders, beside another column's term, after a range stage, in a disjunction
        for stages in ([("pred", ir.const(0) == cx % ir.const(50))], [("pred", (cx % ir.const(50) > -10) & (cx % ir.const(50) <= 3))],
                       [("pred", (cx % ir.const(7) == 0) & (cy < 93))], [("range", 5, 2, n - 3), ("pred", cx % ir.const(50) == 0)],
                       [("pred", (cx % ir.const(9) == 4) | (cy > 97))]):
            ov, dv = apply_stages(p, stages)
            assert_same(p, ov, dv)
        n1i, _ = ctx.profile_get("interp_predicate")
        assert n1i == n0i                                        # none of the above went through the interpreter
        # m = 1 / -1 / 0 stay with the interpreter (0 raises DivideError on both sides)
        ov, dv = apply_stages(p, [("pred", cx % ir.const(1) == 0)])
        assert_same(p, ov, dv)
        ov, dv = apply_stages(p, [("pred", cx % ir.const(0) == 0)])
        with pytest.raises(Exception, match="DivideError"):
            dv._query().count()
    finally:
        ctx.profile(False)


def test_string_dictionary_gives_the_same_answers(oracle, dfdb_mod, ctx, tmp_path):
    """K9: a dictionary (16-bit codes + the distinct strings) beside a low-cardinality String column turns == / != / startswith / endswith into a
    bit-table lookup of the codes and the column's projection into copies out of the dictionary.  Every answer must be the oracle's, with the dictionary
    built explicitly, automatically (ctx option string_dictionary) for host arrays / generated columns / files, alone and inside longer queues; columns
    it cannot take (too many distinct values, nullable, a 5-KB string) keep working through the flat kernels."""
    from dfdb import ir
    rng = np.random.default_rng(9)
    n = 70_001
    words = ["", "a", "sony", "sonya", "huawei", "apple", "apple pie", "xiaomi-redmi-note-12", "é", "日本語", "0123456789abcdef0123456789abcdef0123456789", "so"]
    s = [words[int(k)] for k in rng.integers(0, len(words), n)]
    a = rng.integers(0, 1000, n).astype(np.int64)
    many = ["v%d" % int(k) for k in rng.integers(0, 9000, n)]                 # 9000 distinct values
    p = Pair(oracle, dfdb_mod, {"s": s, "a": a, "many": many}, block_size=4096)
    t = p.d
    assert t.build_dictionary("s") == len(set(s))
    assert t.build_dictionary("many", 4096) == 0 and t.build_dictionary("many", 20000) == len(set(many))
    with pytest.raises(Exception, match="not a String column"):
        t.build_dictionary("a")
    S, A, M = ir.col(0), ir.col(1), ir.col(2)
    ctx.profile(True)
    try:
        n0, _ = ctx.profile_get("str_match")
        d0, _ = ctx.profile_get("dict_scan")
        cases = [[("pred", S == "sony")], [("pred", S != "sony")], [("pred", S == "nokia")], [("pred", S != "nokia")], [("pred", S == "")],
                 [("pred", ir.startswith(S, "so"))], [("pred", ir.endswith(S, "a"))], [("pred", ir.startswith(S, ""))], [("pred", ir.endswith(S, "日本語"))],
                 [("pred", S == "0123456789abcdef0123456789abcdef0123456789")], [("pred", (S == "apple") & (A > 500))], [("pred", (A > 500) & (S != "apple") & ir.startswith(S, "a"))],
                 [("range", 100, 3, 60_000), ("pred", S == "huawei")], [("pred", A < 300), ("range", 5, 1, 2000), ("pred", ir.endswith(S, "i"))],
                 [("pred", M == "v17")], [("pred", ir.startswith(M, "v89") & (S == "é"))], [("pred", (S == "sony") | (A == 7))], [("idx", [5, 77, 4096, 70_001])]]
        for stages in cases:
            ov, dv = apply_stages(p, stages)
            assert_same(p, ov, dv)
            ov, dv = apply_stages(p, stages, proj=[("s", S), ("k", A * 2)])
            assert_same(p, ov, dv)
        # Boolean combinations of string terms over the one dictionary column: a function of the entry, one lookup (never the interpreter)
        i0, _ = ctx.profile_get("interp_predicate")
        for pred in ((S == "sony") | (S == "apple"), ~(S == "sony"), ~ir.startswith(S, "a") | (S == ""), (S == "sony") ^ ir.endswith(S, "y"),
                     ((S == "sony") | (S == "so") | ir.startswith(S, "hua")) & (A > 100), ~((S != "é") & ~ir.endswith(S, "語")),
                     ir.isin(S, ["sony", "dell", "no such brand", ""]), ~ir.isin(S, ["asus"]) & (A < 500_000)):       # in.(s, Ref([...])) over strings
            ov, dv = apply_stages(p, [("pred", pred)], proj=[("s", S), ("a", A)])
            assert_same(p, ov, dv)
        i1, _ = ctx.profile_get("interp_predicate")
        assert i1 == i0
        ov, dv = apply_stages(p, [("pred", (S == "sony") | (M == "v17"))])          # two different columns: the interpreter, same answer
        assert_same(p, ov, dv)
        n1, _ = ctx.profile_get("str_match")
        d1, _ = ctx.profile_get("dict_scan")
        g1, _ = ctx.profile_get("dict_expand_bytes")
        assert n1 == n0 and d1 > d0 and g1 > 0                       # K5 never ran: the dictionary kernels did
    finally:
        ctx.profile(False)
    # unique / groupreduce keyed by the dictionary column: the codes are the group labels (no hash table); same frames as over the flat column
    import pandas as pd
    flat = dfdb_mod.DFTable.from_columns({"s": s, "a": a}, block_size=4096)
    for view_of in (lambda tb: tb, lambda tb: tb[(tb.a > 300) & (tb.s != "sony"), dfdb_mod.ALL], lambda tb: tb[tb.a > 10_000, dfdb_mod.ALL]):
        for stat in ("count", "sum", "min", "max", "mean"):
            got, want = dfdb_mod.groupreduce(view_of(t), "s", "a", stat), dfdb_mod.groupreduce(view_of(flat), "s", "a", stat)
            pd.testing.assert_frame_equal(got, want)
    assert list(t.s.unique()) == list(flat.s.unique()) == list(dict.fromkeys(s))
    sel = a > 300
    assert list(t[t.a > 300, dfdb_mod.ALL][dfdb_mod.ALL, "s"].unique()) == list(dict.fromkeys(np.array(s, dtype=object)[sel]))
    flat.close()
    # automatic, for every way a String column becomes resident
    ctx.set_option("string_dictionary", 64)
    try:
        path = str(tmp_path / "tb")
        p2 = Pair(oracle, dfdb_mod, {"s": s, "a": a, "many": many}, block_size=5000, via_files=path)      # files (device LZ4 decode + unpack)
        g = dfdb_mod.DFTable.new(); g.add_generated("b", dfdb_mod.GEN_STR_BRANDS10, 3, 50_000)             # generated
        nullable = [None if k % 7 == 0 else w for k, w in enumerate(s)]
        p3 = Pair(oracle, dfdb_mod, {"s": nullable, "big": ["x" * 5000 if k == 9 else "y" for k in range(n)]}, block_size=4096)
        ctx.profile(True)
        d0, _ = ctx.profile_get("dict_scan"); k0, _ = ctx.profile_get("str_match")
        for stages in ([("pred", S == "sony")], [("pred", ir.startswith(S, "app") & (A < 900))]):
            ov, dv = apply_stages(p2, stages, proj=[("s", S), ("a", A)])
            assert_same(p2, ov, dv)
        assert dfdb_mod.nrow(g[g.b == "sony", dfdb_mod.ALL]) == int((np.array(oracle.flat_to_strings(*oracle.gen_str(3, 0, 50_000)), dtype=object) == "sony").sum())
        d1, _ = ctx.profile_get("dict_scan"); k1, _ = ctx.profile_get("str_match")
        assert d1 >= d0 + 3 and k1 == k0
        ov, dv = apply_stages(p2, [("pred", M == "v17")], proj=[("many", M)])                              # 9000 distinct > 64: the flat kernels
        assert_same(p2, ov, dv)
        for stages in ([("pred", ir.col(1) == "y")], [("pred", ir.startswith(ir.col(1), "xx"))], [("range", 3, 2, 5000)]):   # nullable / a 5-KB string: no dictionary
            ov, dv = apply_stages(p3, stages, proj=[("s", ir.col(0)), ("big", ir.col(1))])
            assert_same(p3, ov, dv)
        k2, _ = ctx.profile_get("str_match")
        assert k2 > k1
        ctx.profile(False)
        g.close()
    finally:
        ctx.set_option("string_dictionary", 0)


def test_affine_forms_of_one_column_are_scan_terms(oracle, dfdb_mod, ctx):
    """(col * k + d) OP const — `a * 2 + 1 > c`, `x - 5.0 > 0`, `1.2 * price > 100` — runs in the scan kernel: wrapping Int64 arithmetic for signed integer
    columns with integer constants, Float64 with a rounding after the multiplication and another after the addition (never a fused multiply-add)
    otherwise.  Same answers as the oracle's generic evaluation, including overflow wrap-around, NaN / Inf / -0.0, Int64 values beyond 2^53 converted to
    Float64, mixed forms the term cannot take (they stay with the interpreter), and the interpreter is not used for the forms it can."""
    from dfdb import ir
    rng = np.random.default_rng(21)
    n = 60_007
    x = rng.normal(0, 1000, n); x[::97] = np.nan; x[3::501] = np.inf; x[5::499] = -0.0; x[7::503] = 1e308
    cols = {"a": rng.integers(-1000, 1000, n).astype(np.int64), "b": rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64), "i8": rng.integers(-128, 127, n).astype(np.int8),
            "x": x, "f": rng.normal(0, 10, n).astype(np.float32), "u16": rng.integers(0, 65535, n).astype(np.uint16), "i32": rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)}
    p = Pair(oracle, dfdb_mod, cols, block_size=8192)
    a, b, i8, X, f, u16, i32 = (ir.col(k) for k in range(7))
    term_forms = [a * 2 + 1 > 100, 3 * a - 7 <= -50, a + 5 == 10, 100 - a > 1050, b * 3 > 0, b * 2**62 + 2**62 < 0, b - 1 >= 2**63 - 2, i8 * 100 + 28 > 12_000,
                  i32 * 4_000_000_000 < 0, X * 1.5 + 0.25 > 100.0, X - 5.0 > 0, 1.2 * X > 100, 2.0 - X >= 1e308, X * 3 < 7, X + 1e308 > 1.5e308, a * 0.1 > 12.3,
                  b * 1.0 > 9.2e18, b + 0.5 < -9.2e18, f * 2.5 + 1 > 3.5, u16 * 0.5 + 0.25 >= 1000.25, i32 * 1e10 > 1e19, a / 50 > 10.5, b / 3 < -1e18, X / 0.1 >= 33.0, X / 0 > 0, 7 <= i8 / -3, u16 / 7.5 == 8.0,
                  (a * 2 + 1 > 100) & (a * 2 + 1 < 900), (X * 1.5 > 10.0) & (a - 3 != 0) & (b * 3 > 0)]
    ctx.profile(True)
    try:
        i0, _ = ctx.profile_get("interp_predicate")
        for k, pred in enumerate(term_forms):
            j0, _ = ctx.profile_get("interp_predicate")
            ov, dv = apply_stages(p, [("pred", pred)])
            assert_same(p, ov, dv)
            j1, _ = ctx.profile_get("interp_predicate")
            assert j1 == j0, f"form {k} ({pred!r}) went through the interpreter"
        ov, dv = apply_stages(p, [("range", 10, 3, n - 11), ("pred", (a * 7 - 1 > 0) & (X * 0.5 < 3.0))], proj=[("a", a), ("x", X)])
        assert_same(p, ov, dv)
        # the same transforms as PROJECTIONS ride on the gather (k_gather_transform).  Two roundings, never an fma: with |b| ~ 1e18 the addend is
        # below one ulp of the product and a fused multiply-add lands one ulp away in ~2 % of the rows (found by the fuzz soak)
        ov, dv = apply_stages(p, [("pred", a > -2000)], proj=[("k", b * -1.91 - (-36)), ("j", X * 3 + 1), ("r", i8 % 7), ("d", b / 3), ("w", b * 3 + 1), ("h", 0.5 - u16 * 0.1),
                                                             ("z", a * -2.5), ("z2", -3.0 * X), ("z3", X * -1.0 + 0.0)])       # (a product of zero keeps its sign: -2.5 * 0 is -0.0)
        assert_same(p, ov, dv)
        with np.errstate(invalid="ignore", over="ignore"):
            got = dv._query().materialize()
            assert np.array_equal(got[0], cols["b"].astype(np.float64) * -1.91 + 36.0) and np.array_equal(got[4], cols["b"] * np.int64(3) + np.int64(1))
        i1, _ = ctx.profile_get("interp_predicate")
        j1, _ = ctx.profile_get("interp_project")
        assert i1 == i0 and j1 == 0, "an affine form went through the interpreter"
        # not one form: a Float32 product (Float32 * Int stays Float32), two multiplications, an Int64 product under a Float64 sum, an Int8 product (stays Int8 and wraps there), unsigned wrap-around
        for pred in (f * 2 > 3.5, (a * 2) * 3 > 5, a * 2 + 0.5 > 10, (a + 1) * 2 > 10, i8 * ir.const(100, ir.I8) > 10, u16 * 70000 > 100, X * X > 4.0, a * b > 0):
            ov, dv = apply_stages(p, [("pred", pred)])
            assert_same(p, ov, dv)
    finally:
        ctx.profile(False)
    # third opinion on a few (numpy evaluates in the same types)
    with np.errstate(invalid="ignore", over="ignore"):
        av, bv, xv = cols["a"], cols["b"], cols["x"]
        for pred, want in ((a * 2 + 1 > 100, av * 2 + 1 > 100), (b * 3 > 0, bv * np.int64(3) > 0), (X * 1.5 + 0.25 > 100.0, xv * 1.5 + 0.25 > 100.0),
                           (2.0 - X >= 1e308, 2.0 - xv >= 1e308), (b * 1.0 > 9.2e18, bv * 1.0 > 9.2e18)):
            ov, dv = apply_stages(p, [("pred", pred)])
            assert np.array_equal(dv._query().indices(), np.nonzero(want)[0] + 1)


def test_errors_surface_only_where_the_block_iteration_gets(oracle, dfdb_mod, ctx):
    """DivideError / InexactError of a predicate are raised by the reference only if its block-by-block iteration EVALUATES the block the offending
    row lives in: blocks wholly before the first element of a leading range are skipped unread (skip_if_can, selection.jl:177-190) and once any range
    stage has seen its last element nothing more is read (is_finished :192-196).  The engine evaluates whole columns, records the smallest erroring row
    and decides the same way (query.cpp error_is_reached); oracle and engine must agree case by case, raising and not raising."""
    from dfdb import ir
    n, bs = 10_000, 1000
    a = np.arange(1, n + 1, dtype=np.int64)
    z = np.ones(n, np.int64); z[7_500] = 0                       # one zero divisor, in block 7 (rows 7001..8000)
    f = np.full(n, 2.0); f[3_200] = 2.5                          # one inexact conversion, in block 3
    p = Pair(oracle, dfdb_mod, {"a": a, "z": z, "f": f}, block_size=bs)
    A, Z, F = ir.col(0), ir.col(1), ir.col(2)
    risky_div, risky_cast = (A % Z == 0), (ir.cast(F, ir.I64) == 2)
    cases = [
        ([("pred", risky_div)], "ZeroDivisionError"),                                        # no range stage: every block is evaluated
        ([("pred", risky_div), ("range", 1, 1, 50)], None),                                  # the range is satisfied inside block 0
        ([("pred", risky_div), ("range", 1, 1, 7_000)], None),                               # ... exactly at the end of block 6: block 7 is never read
        ([("pred", risky_div), ("range", 1, 1, 7_001)], "ZeroDivisionError"),                # one more survivor needed: block 7 is read
        ([("range", 1, 1, 7_500), ("pred", risky_div)], None),                               # a predicate after a range sees the range's rows only: row 7501 is not one
        ([("range", 1, 1, 7_501), ("pred", risky_div)], "ZeroDivisionError"),
        ([("range", 7_502, 1, 9_000), ("pred", risky_div)], None),
        ([("range", 7_501, 7, 9_000), ("pred", risky_div)], "ZeroDivisionError"),
        ([("pred", risky_div), ("range", 6_900, 1, 7_000), ("range", 1, 1, 2)], None),       # two range stages: the LAST one is done in block 6
        ([("pred", A > 6_990), ("idx", [3, 5, 9]), ("pred", risky_div)], None),              # survivors 3, 5, 9 of stage 1 all live in block 6
        ([("pred", A > 6_990), ("idx", [3, 5, 511]), ("pred", risky_div)], "ZeroDivisionError"),   # survivor 511 is row 7501: the zero divisor itself
        ([("pred", risky_div), ("pred", A > 6_990), ("idx", [3, 5, 11])], "ZeroDivisionError"),    # survivor 11 is row 7001: block 7 is read, all of it evaluated
        ([("pred", risky_div), ("pred", A > 6_990), ("idx", [3, 5, 10])], None),                   # ... row 7000: done in block 6
        ([("pred", risky_cast), ("range", 1, 2, 3_000)], None),                              # InexactError lurking in block 3, range done in block 2
        ([("pred", risky_cast), ("range", 1, 2, 3_001)], "ValueError"),
        ([("pred", risky_cast & risky_div), ("range", 1, 1, 9_999)], "ValueError"),          # both lurk: the iteration meets block 3 first
        ([("pred", risky_div), ("pred", A < 100)], "ZeroDivisionError"),                     # predicates only: fused, every row evaluated
    ]
    for stages, want_err in cases:
        ov, dv = apply_stages(p, stages)
        o = None
        try: o_n = ov.nrow()
        except Exception as e: o = type(e).__name__      # noqa: E722,BLE001
        d = None
        try: d_n = dfdb_mod.nrow(dv)
        except Exception as e: d = type(e).__name__      # noqa: BLE001
        assert o == want_err, (stages, "oracle", o)
        assert d == want_err, (stages, "engine", d)
        if want_err is None:
            assert_same(p, ov, dv)
    # computed projection columns that raise: the reference evaluates block by block and, inside a block, the projection's columns in order
    # (projection.jl:149-154) — first erroring BLOCK, then first such COLUMN, then first such row
    div, cast = A % Z, ir.cast(F, ir.I64)
    f2 = np.full(n, 2.0); f2[7_100] = 2.5                        # a second inexact conversion in block 7, BEFORE the zero divisor's row 7500
    p2 = Pair(oracle, dfdb_mod, {"a": a, "z": z, "f": f2}, block_size=bs)
    proj_cases = [
        (p, [("d", div), ("c", cast)], "ValueError"),              # block 3 (the conversion) comes before block 7 (the division), column order regardless
        (p, [("c", cast), ("d", div)], "ValueError"),
        (p2, [("d", div), ("c", cast)], "ZeroDivisionError"),      # both in block 7: the first COLUMN raises, though the other column's row is earlier
        (p2, [("c", cast), ("d", div)], "ValueError"),
        (p2, [("k", A + 1), ("c", cast), ("d", div)], "ValueError"),
    ]
    for pr, proj, want_err in proj_cases:
        ov, dv = apply_stages(pr, [("pred", A > 5)], proj=proj)
        assert ov.nrow() == dfdb_mod.nrow(dv) == n - 5
        o = d = None
        try: ov.materialize()
        except Exception as e: o = type(e).__name__      # noqa: BLE001
        try: dv._query().materialize()
        except Exception as e: d = type(e).__name__      # noqa: BLE001
        assert (o, d) == (want_err, want_err), ([k for k, _ in proj], "oracle", o, "engine", d)
