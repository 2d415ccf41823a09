"""Seeded differential fuzz of the whole path: random typed expression trees (every arithmetic / comparison / logical operator of the IR over
every column type, missing values, strings) inside random selection queues (ranges, index vectors, predicates) with random projections —
the engine's count, indices, bitmap and materialised columns must equal the oracle's, byte for byte, and where Julia would raise
(DivideError, InexactError) both must raise the same error.  The fixed shapes of the other suites pick the kernels; this one picks nothing."""
import os

import numpy as np
import pytest

from helpers import Pair, apply_stages, apply_stages_both, assert_same

pytestmark = pytest.mark.gpu
N = int(os.environ.get("DFDB_FUZZ_ROWS", "12345"))          # (DFDB_FUZZ_ROWS=300007 DFDB_FUZZ_BLOCK=65536: the same queues over a bigger table)
BLOCK = int(os.environ.get("DFDB_FUZZ_BLOCK", "1000"))
SCALE = int(os.environ.get("DFDB_FUZZ_SCALE", "1"))      # DFDB_FUZZ_SCALE=20 python -m pytest tests/test_gpu_fuzz.py: 20 x the seeds
SEED0 = int(os.environ.get("DFDB_FUZZ_SEED0", "0"))      # DFDB_FUZZ_SEED0=10000000: the same number of cases from fresh seeds (soaks)


@pytest.fixture(scope="module", params=["flat strings", "string dictionary", "compressed-only"])
def pair(oracle, dfdb_mod, request):
    rng = np.random.default_rng(2024)
    f64 = rng.normal(0, 50, N); f64[::101] = np.nan; f64[5::997] = np.inf; f64[7::991] = -0.0
    f32 = rng.normal(0, 8, N).astype(np.float32); f32[::113] = np.nan
    cols = {
        "a": rng.integers(-60, 60, N).astype(np.int64),                       # 0
        "b": rng.integers(-2**62, 2**62, N).astype(np.int64),                 # 1  (overflow territory)
        "c": rng.integers(1, 40, N).astype(np.int64),                         # 2  (never zero: a safe divisor)
        "i32": rng.integers(-2**31, 2**31 - 1, N).astype(np.int32),           # 3
        "i8": rng.integers(-128, 127, N).astype(np.int8),                     # 4
        "u16": rng.integers(0, 2**16 - 1, N).astype(np.uint16),               # 5
        "u64": rng.integers(0, 2**63, N).astype(np.uint64) * np.uint64(2),    # 6
        "x": f64,                                                             # 7
        "f": f32,                                                             # 8
        "flag": rng.integers(0, 2, N).astype(bool),                           # 9
        "m": np.ma.masked_array(rng.integers(-9, 9, N).astype(np.int64), mask=rng.random(N) < 0.25),   # 10
        "s": ["%s%d" % ("ab"[i % 2] * (i % 3), i % 23) for i in range(N)],   # 11
        "z": rng.integers(-2, 3, N).astype(np.int64),                         # 12 (zeros: a divisor that raises)
        "sm": [None if i % 11 == 3 else "%s%d" % ("xy"[i % 2] * (i % 4), i % 7) for i in range(N)],                   # 13 Union{String,Missing}
        "mf": np.ma.masked_array(rng.normal(0, 5, N), mask=rng.random(N) < 0.4),                                         # 14 Union{Float64,Missing}
        # 15: a divisor whose zeros all sit in the LAST third of the table: whether `a % zl` raises depends on whether the reference's block-by-block
        # iteration gets that far (a later range stage that has seen its last element ends it: is_finished, selection.jl:192-196)
        "zl": np.where(np.arange(N) >= 2 * N // 3, rng.integers(0, 2, N), rng.integers(1, 5, N)).astype(np.int64),
    }
    # (the compressed-only variant takes blocks of whole 1024-row tiles, so that its simple 8-byte terms run inside the decoder: K7's history-ring scan)
    p = Pair(oracle, dfdb_mod, cols, block_size=BLOCK if request.param != "compressed-only" or BLOCK % 1024 == 0 else 1024)
    if request.param == "string dictionary":             # K9: every string predicate and projection of `s` goes through the codes
        assert p.d.build_dictionary("s") == len(set(cols["s"]))
    if request.param == "compressed-only":               # round 5: every plain fixed-width column holds its LZ4 blocks and nothing decoded (keep_compressed = 2's form,
        for name, v in cols.items():                     # made in HBM by dfdb_table_compress_column): simple 8-byte terms run inside the decoder, gathers out of the
            if isinstance(v, np.ndarray) and not isinstance(v, np.ma.MaskedArray):      # survivors' arena, everything else over a one-call decode
                p.d.compress_column(name, 2)
        assert p.d.resident_bytes("b")["decoded"] < 4096
    return p


NUM_COLS = [0, 1, 2, 3, 4, 5, 6, 7, 8]
INT_COLS = [0, 1, 2, 3, 4, 5, 6]


class Gen:
    ZCOL, ZLCOL = 12, 15            # ordinals of the zero-holding divisor columns

    def __init__(self, ir, seed, risky):
        self.ir, self.rng, self.risky = ir, np.random.default_rng(seed), risky

    def pick(self, xs):
        return xs[int(self.rng.integers(0, len(xs)))]

    def const(self):
        r = self.rng.random()
        if r < 0.5:
            return self.ir.const(int(self.rng.integers(-70, 70)))
        if r < 0.8:
            return self.ir.const(float(np.round(self.rng.normal(0, 30), 2)))
        return self.ir.const(self.pick([0, 1, -1, 2**31, -2**31 - 1, 2**53 + 1, 0.5, -0.0, float("inf"), float("nan"), 255, 65536]))

    def num(self, depth):
        ir = self.ir
        if depth <= 0 or self.rng.random() < 0.3:
            return ir.col(self.pick(NUM_COLS)) if self.rng.random() < 0.75 else self.const()
        k = int(self.rng.integers(0, 12))
        a = self.num(depth - 1)
        if k == 0: return -a
        if k == 1: return abs(a)
        if k == 2:                                                               # T(x): InexactError where the value does not fit (risky seeds only)
            return ir.cast(a, self.pick([ir.I8, ir.I32, ir.I64, ir.U16, ir.U64, ir.F32, ir.F64])) if self.risky else ir.float64(a)
        b = self.num(depth - 1)
        if k == 3: return a + b
        if k == 4: return a - b
        if k == 5: return a * b
        if k == 6: return ir.minimum(a, b)
        if k == 7: return ir.maximum(a, b)
        if k == 8: return a / (b if self.risky else ir.col(2))
        # integer-only operators: integer operands (÷, rem, mod are defined for floats too, but keep the divisor's zero under control)
        ia = ir.col(self.pick(INT_COLS)) if self.rng.random() < 0.8 else ir.col(self.pick([7, 8]))     # (rem / mod / div of floats too)
        ib = (ir.col(self.ZLCOL) if self.risky and self.rng.random() < 0.35 else ir.col(self.ZCOL) if self.risky and self.rng.random() < 0.5 else self.pick([ir.col(2), ir.const(7), ir.const(-3), ir.col(2) * 2 + 1]))
        if k == 9: return ia % ib
        if k == 10: return ir.mod(ia, ib)
        return ir.div(ia, ib)

    def boolean(self, depth):
        ir = self.ir
        r = self.rng.random()
        if depth <= 0 or r < 0.45:
            k = int(self.rng.integers(0, 9))
            if k <= 3:
                f = self.pick([lambda p, q: p == q, lambda p, q: p != q, lambda p, q: p < q, lambda p, q: p <= q, lambda p, q: p > q, lambda p, q: p >= q])
                return f(self.num(min(depth, 2)), self.num(min(depth, 2)) if self.rng.random() < 0.6 else self.const())
            if k == 4: return ir.col(9)
            if k == 5: return ir.isin(ir.col(self.pick([0, 3, 4, 5])), [int(v) for v in self.rng.integers(-60, 60, int(self.rng.integers(1, 9)))])
            if k == 6 and self.rng.random() < 0.4:
                return self.pick([ir.ismissing(ir.col(13)), ~ir.ismissing(ir.col(13)), ir.ismissing(ir.col(14)), ir.coalesce(ir.col(14), 0.5) * 2 > ir.col(8),
                                  ir.coalesce(ir.col(14), ir.col(7)) <= self.const(), ir.ismissing(ir.col(14)) | (ir.col(0) > 3), ir.sizeof(ir.col(13)) > 2])
            if k == 6: return ir.ismissing(ir.col(10)) if self.rng.random() < 0.5 else (ir.coalesce(ir.col(10), ir.const(int(self.rng.integers(-3, 3)))) > self.const())
            if k == 7: return self.pick([ir.col(11) == "a7", ir.col(11) != "bb11", ir.startswith(ir.col(11), "aa"), ir.endswith(ir.col(11), "2"), ir.sizeof(ir.col(11)) > 2,
                                         ir.isin(ir.col(11), ["a7", "bb11", "nope", "0"])])
            return ir.coalesce(ir.col(10), ir.col(0)) * 2 >= ir.col(4)           # a nullable column made whole by another column
        a, b = self.boolean(depth - 1), self.boolean(depth - 1)
        k = int(self.rng.integers(0, 4))
        return (a & b) if k == 0 else (a | b) if k == 1 else (a ^ b) if k == 2 else ~a

    def stages(self):
        # `bound`: the statically known size of the queue so far (Julia checks a range / index stage against it when the stage before is one)
        out, bound = [], N
        for _ in range(int(self.rng.integers(1, 4))):
            k = self.rng.random()
            if k < 0.55 or bound < 2:
                out.append(("pred", self.boolean(int(self.rng.integers(0, 3)))))
            elif k < 0.8:
                lo = int(self.rng.integers(1, bound // 2 + 1)); step = int(self.pick([1, 1, 2, 3, 7, 64, 1000]))
                hi = int(self.rng.integers(lo, bound + 1))
                out.append(("range", lo, step, hi)); bound = len(range(lo, hi + 1, step))
            elif k < 0.9:
                idx = [int(v) for v in self.rng.integers(1, bound + 1, int(self.rng.integers(0, 40)))]
                out.append(("idx", idx)); bound = len(set(idx))
            else:
                out.append(("int", int(self.rng.integers(1, bound + 1))))
                break                                                            # nothing indexes a scalar selection
        return out

    def proj(self):
        if self.rng.random() < 0.3:
            return None
        ir, out = self.ir, []
        for k in range(int(self.rng.integers(1, 4))):
            r = self.rng.random()
            e = ir.col(int(self.rng.integers(0, 15))) if r < 0.5 else (self.num(2) if r < 0.85 else self.boolean(1))
            out.append(("p%d" % k, e))
        return out


def outcome(fn):
    try:
        return ("ok", fn())
    except Exception as e:          # noqa: BLE001 — the class is what is compared
        return ("err", type(e).__name__)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + 400 * SCALE))
def test_random_queue_equals_the_oracle(pair, dfdb_mod, seed):
    from dfdb import ir
    g = Gen(ir, seed, risky=seed % 4 == 3)
    stages, proj = g.stages(), g.proj()
    ov, dv = apply_stages_both(pair, stages, proj=proj)      # skips only when oracle AND engine refuse the queue with the same exception class; a one-sided refusal fails
    want = outcome(lambda: ov.nrow())
    got = outcome(lambda: dfdb_mod.nrow(dv))
    assert want[0] == got[0], f"oracle {want}, engine {got} for {stages} / {proj}"
    # which error: Julia raises the one of the first row that fails; oracle and engine both report the kind of the EARLIEST erroring row
    if want[0] == "err":
        assert want[1] == got[1], f"oracle raises {want[1]}, engine {got[1]} for {stages}"
        return
    w2 = outcome(lambda: ov.materialize())
    if w2[0] == "err":              # the projection raises (DivideError / InexactError on a selected row)
        g2 = outcome(lambda: dv._query().materialize())
        assert g2 == w2, f"oracle {w2}, engine {g2} for {proj}"
        return
    assert_same(pair, ov, dv)


# ---------------------------------------------------------------- the same random queues, block-streamed and block-range sharded
@pytest.fixture(scope="module")
def filed(oracle, dfdb_mod, tmp_path_factory):
    """The fuzz table without its nullable column (group tables split host columns by block range: plain arrays), written by the oracle (liblz4),
    opened three ways: resident, not loaded (for streaming), and as a 3-shard group on device 0 with the host exchange."""
    from dfdb import group as G, _native as NAT
    rng = np.random.default_rng(77)
    x = rng.normal(0, 50, N); x[::101] = np.nan
    cols = {"a": rng.integers(-60, 60, N).astype(np.int64), "b": rng.integers(-2**62, 2**62, N).astype(np.int64), "c": rng.integers(1, 40, N).astype(np.int64),
            "i32": rng.integers(-2**31, 2**31 - 1, N).astype(np.int32), "i8": rng.integers(-128, 127, N).astype(np.int8),
            "u16": rng.integers(0, 2**16 - 1, N).astype(np.uint16), "u64": rng.integers(0, 2**63, N).astype(np.uint64) * np.uint64(2), "x": x,
            "f": rng.normal(0, 8, N).astype(np.float32), "flag": rng.integers(0, 2, N).astype(bool),
            "s": ["%s%d" % ("ab"[i % 2] * (i % 3), i % 23) for i in range(N)]}
    cols["z"] = rng.integers(-2, 3, N).astype(np.int64)                                                        # 11: zeros everywhere
    cols["zl"] = np.where(np.arange(N) >= 2 * N // 3, rng.integers(0, 2, N), rng.integers(1, 5, N)).astype(np.int64)   # 12: zeros in the last third only
    path = str(tmp_path_factory.mktemp("fuzz") / "tb")
    # DFDB_FUZZ_KEEP=1 (with DFDB_FUZZ_BLOCK a multiple of 1024): the resident table keeps its LZ4 blocks and every fresh single-term scan of an 8-byte column
    # decodes them again on the way (decode_on_scan: K7 fused with the predicate, or the two-wave pipeline + scan, with the sequence-start index after the first)
    keep = os.environ.get("DFDB_FUZZ_KEEP", "0") == "1"
    ctx0 = dfdb_mod.default_context(0)
    if keep:
        ctx0.set_option("keep_compressed", 1)
    try:
        p = Pair(oracle, dfdb_mod, cols, block_size=BLOCK, via_files=path)
    finally:
        ctx0.set_option("keep_compressed", 0)
    if keep:
        ctx0.set_option("decode_on_scan", 1)
    lazy = dfdb_mod.open_table(path, load=False)
    g = G.Group.create([0, 0, 0], NAT.EXCHANGE_HOST)
    gt = G.GroupTable.open(g, path)
    yield p, lazy, gt
    ctx0.set_option("decode_on_scan", 0)
    gt.close(); g.close(); lazy.close()


class GenNoMissing(Gen):
    """column ordinals of `filed`: a b c i32 i8 u16 u64 x f flag s z zl  (no nullable column)"""
    ZCOL, ZLCOL = 11, 12

    def boolean(self, depth):
        ir = self.ir
        if depth <= 0 or self.rng.random() < 0.45:
            k = int(self.rng.integers(0, 7))
            if k <= 3:
                f = self.pick([lambda p, q: p == q, lambda p, q: p != q, lambda p, q: p < q, lambda p, q: p <= q, lambda p, q: p > q, lambda p, q: p >= q])
                return f(self.num(min(depth, 2)), self.num(min(depth, 2)) if self.rng.random() < 0.6 else self.const())
            if k == 4: return ir.col(9)
            if k == 5: return ir.isin(ir.col(self.pick([0, 3, 4, 5])), [int(v) for v in self.rng.integers(-60, 60, int(self.rng.integers(1, 9)))])
            return self.pick([ir.col(10) == "a7", ir.col(10) != "bb11", ir.startswith(ir.col(10), "aa"), ir.endswith(ir.col(10), "2"), ir.sizeof(ir.col(10)) > 2])
        a, b = self.boolean(depth - 1), self.boolean(depth - 1)
        k = int(self.rng.integers(0, 4))
        return (a & b) if k == 0 else (a | b) if k == 1 else (a ^ b) if k == 2 else ~a


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + 120 * SCALE))
def test_random_queue_streamed_and_sharded(filed, dfdb_mod, seed):
    """Range and index stages after predicates number the SURVIVORS: across chunk boundaries (streaming) and across shards (the stage-base exchange)
    their running offsets must continue exactly where the previous chunk / the lower ranks stopped."""
    from dfdb import ir, group as G
    pair, lazy, gt = filed
    g = GenNoMissing(ir, 10_000 + seed, risky=False)
    stages = g.stages()
    proj = [("a", ir.col(0)), ("x", ir.col(7)), ("k", ir.col(3) * 2 - ir.col(4))]
    ov, dv = apply_stages_both(pair, stages, proj=proj)      # skips only when oracle AND engine refuse the queue with the same exception class; a one-sided refusal fails
    want_idx = ov.select_indices()
    want = ov.materialize()
    assert np.array_equal(dv._query().indices(), want_idx)
    # block-streamed over the table that is not resident, chunk size 1..5 blocks
    sv = dfdb_mod.DFView(lazy, dv.projection, dv.selection)
    assert dfdb_mod.nrow_streamed(sv, 1 + seed % 5) == len(want_idx)
    got = dfdb_mod.materialize_streamed(sv, 1 + (seed // 5) % 5)
    assert np.array_equal(np.asarray(got["a"], dtype=np.int64), want[0]) and np.array_equal(np.asarray(got["k"], dtype=np.int64), want[2])
    gx, wx = np.asarray(got["x"], dtype=np.float64), want[1]
    assert np.array_equal(np.isnan(gx), np.isnan(wx)) and np.array_equal(gx[~np.isnan(gx)], wx[~np.isnan(wx)])
    # round 6: aggregates, unique and groupreduce over the table that is NOT resident — the ordinary calls, streamed inside the library (csrc/ooc.cpp): per-chunk
    # device results merged in chunk order — against the same calls over the resident table (which the other fuzz tests hold to the oracle)
    lazy.ctx.set_option("ooc_chunk_blocks", 1 + (seed // 7) % 5)
    rsel = dfdb_mod.DFView(pair.d, None, dv.selection)
    lsel = dfdb_mod.DFView(lazy, None, dv.selection)
    for name in ("b", "u64", "i8"):
        assert lsel[dfdb_mod.ALL, name].sum() == rsel[dfdb_mod.ALL, name].sum(), (name, stages)
    if len(want_idx):
        for name in ("a", "x", "f"):
            lo, ro = lsel[dfdb_mod.ALL, name].min(), rsel[dfdb_mod.ALL, name].min()
            assert (lo == ro and np.signbit(lo) == np.signbit(ro)) or (lo != lo and ro != ro), (name, lo, ro, stages)
            lo, ro = lsel[dfdb_mod.ALL, name].max(), rsel[dfdb_mod.ALL, name].max()
            assert (lo == ro and np.signbit(lo) == np.signbit(ro)) or (lo != lo and ro != ro), (name, lo, ro, stages)
    key = ("c", "s", "i8", "a")[seed % 4]
    lu, ru = lsel[dfdb_mod.ALL, key].unique(), rsel[dfdb_mod.ALL, key].unique()
    assert list(lu) == list(ru), (key, stages)
    stat, val = (("count", None), ("sum", "b"), ("min", "i32"), ("max", "u16"))[(seed // 4) % 4]
    lg, rg = dfdb_mod.groupreduce(lsel, key, val, stat), dfdb_mod.groupreduce(rsel, key, val, stat)
    assert list(lg[key]) == list(rg[key]) and np.array_equal(lg["count"].to_numpy(), rg["count"].to_numpy()), (key, stat, stages)
    if stat != "count":
        assert np.array_equal(lg[stat].to_numpy(), rg[stat].to_numpy()), (key, stat, stages)
    # three block-range shards
    gv = dfdb_mod.DFView(gt.view().table, dv.projection, dv.selection)
    assert G.gnrow(gv) == len(want_idx)
    assert np.array_equal(G.gindices(gv), want_idx)
    gm = G._gq(gv).materialize()
    assert np.array_equal(gm[0], want[0]) and np.array_equal(gm[2], want[2])


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + 120 * SCALE))
def test_random_risky_queue_streamed_and_sharded(filed, dfdb_mod, seed):
    """Queues whose predicates may raise (zero divisors, inexact conversions), block-streamed and over three shards: WHETHER and WHICH error surfaces
    must not depend on where the chunk / shard boundaries fall (the survivors fed to a later range stage continue across them)."""
    from dfdb import ir, group as G
    pair, lazy, gt = filed
    g = GenNoMissing(ir, 20_000 + seed, risky=True)
    stages = g.stages()
    ov, dv = apply_stages_both(pair, stages, proj=[("a", ir.col(0))])      # skips only when oracle AND engine refuse the queue with the same exception class; a one-sided refusal fails
    want = outcome(lambda: ov.nrow())
    assert outcome(lambda: dfdb_mod.nrow(dv)) == want, stages
    sv = dfdb_mod.DFView(lazy, dv.projection, dv.selection)
    assert outcome(lambda: dfdb_mod.nrow_streamed(sv, 1 + seed % 5)) == want, ("streamed", stages)
    gv = dfdb_mod.DFView(gt.view().table, dv.projection, dv.selection)
    assert outcome(lambda: G.gnrow(gv)) == want, ("sharded", stages)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + 200 * SCALE))
def test_random_aggregates(pair, dfdb_mod, seed):
    """sum / min / max / count of a random numeric column over a random queue: the fused forms (the scan adds up or keeps the extremum of a column
    that is itself a term of the last launch) and the plain reduce must both equal numpy over the rows the oracle selects — Int sums exactly
    (wrapping, small integer types widened as Julia's `sum` does), Float sums within n * eps * sum|x|, min / max exactly (NaN wins both)."""
    from dfdb import ir
    g = Gen(ir, 50_000 + seed, risky=False)
    stages = g.stages()
    ci = int(g.pick([0, 1, 2, 3, 4, 5, 6, 7, 8]))
    name = pair.names[ci]
    ov, dv = apply_stages_both(pair, stages)      # skips only when oracle AND engine refuse the queue with the same exception class; a one-sided refusal fails
    idx = ov.select_indices() - 1
    host = np.asarray(pair.d.ctx and pair_columns(pair)[name])[idx]
    col = dv[dfdb_mod.ALL, name]
    assert dfdb_mod.nrow(dv) == len(idx)
    if len(idx) == 0:
        assert col.sum() == 0
        with pytest.raises(ValueError, match="empty collection"):
            col.min()
        return
    if host.dtype.kind == "f":
        want = float(host.astype(np.float64).sum())
        with np.errstate(invalid="ignore"):
            tol = len(idx) * np.finfo(np.float64).eps * float(np.nansum(np.abs(host.astype(np.float64)))) + 1e-300
        got = col.sum()
        assert (np.isnan(want) and np.isnan(got)) or (np.isinf(want) and got == want) or abs(got - want) <= tol, (got, want)
        wmin, wmax = (float("nan"), float("nan")) if np.isnan(host).any() else (float(host.min()), float(host.max()))
        gmin, gmax = col.min(), col.max()
        assert (np.isnan(wmin) and np.isnan(gmin)) or gmin == wmin
        assert (np.isnan(wmax) and np.isnan(gmax)) or gmax == wmax
    else:
        wide = np.uint64 if host.dtype.kind == "u" else np.int64
        with np.errstate(over="ignore"):
            want = int(host.astype(wide).sum(dtype=wide))
        assert col.sum() == want
        assert col.min() == int(host.min()) and col.max() == int(host.max())


_PAIR_COLUMNS = {}


def pair_columns(pair):
    """the host copy of the fuzz table's columns (materialised once per fixture through the oracle: what the engine must agree with)"""
    key = id(pair)
    if key not in _PAIR_COLUMNS:
        out = pair.o.view().materialize()
        _PAIR_COLUMNS[key] = {n: (c if not isinstance(c, tuple) else None) for n, c in zip(pair.names, out)}
    return _PAIR_COLUMNS[key]


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + 100 * SCALE))
def test_random_groupreduce_and_unique(pair, dfdb_mod, seed):
    """groupreduce / unique over a random queue, keyed by an integer column, the nullable column (missing is a group) or the String column (through the
    hash table when it is flat, through the 16-bit codes when it has a dictionary): groups in order of first appearance among the rows the oracle selects,
    exact counts, integer sums / extrema exact, Float sums within n * eps * sum|x|."""
    from dfdb import ir
    from test_gpu_parity import _np_group_ids, _np_groupreduce
    g = Gen(ir, 90_000 + seed, risky=False)
    stages = g.stages()
    key = g.pick(["a", "i8", "s", "m", "c", "i32", "u64", "x"])
    val = g.pick([v for v in ["a", "b", "i32", "u16", "x", "c"] if v != key])      # (a projection cannot name a column twice)
    stat = g.pick(["count", "sum", "min", "max", "mean"])
    # every form of unique's machinery in turn (round 4): the defaults; the hash table only; a 1024-slot table fed one tile at a time (aborted chunks, migrations);
    # the dense form with a span of 100 values (keys outside the first placement, then the hash table for the wider columns), one-tile launches, the exact range
    # (round 6) the radix-partitioned form at any size (k_radix.hip: fixed-width keys; the others go on to the hash table)
    forms = [{}, {"unique_dense": 0}, {"unique_dense": 0, "unique_cap0_log2": 10, "unique_chunk_tiles": 1}, {"unique_dense_range": 100, "unique_chunk_tiles": 1, "unique_dense_sample": seed % 8 < 4},
             {"unique_dense": 0, "unique_radix": 2}]
    defaults = {"unique_dense": 1, "unique_cap0_log2": 21, "unique_chunk_tiles": 0, "unique_dense_range": 1 << 40, "unique_dense_sample": 1, "unique_radix": 1}
    ctx0 = dfdb_mod.default_context(0)
    for k, v in {**defaults, **forms[seed % 5]}.items():
        ctx0.set_option(k, int(v))
    try:
        _groupreduce_and_unique_case(pair, dfdb_mod, g, stages, key, val, stat)
    finally:
        for k, v in defaults.items():
            ctx0.set_option(k, v)


def _groupreduce_and_unique_case(pair, dfdb_mod, g, stages, key, val, stat):
    from test_gpu_parity import _np_group_ids, _np_groupreduce
    ov, dv = apply_stages_both(pair, stages)      # skips only when oracle AND engine refuse the queue with the same exception class; a one-sided refusal fails
    idx = ov.select_indices() - 1
    cols = full_columns(pair)
    keys = cols[key][idx] if not isinstance(cols[key], list) else [cols[key][i] for i in idx]
    if isinstance(keys, np.ma.MaskedArray):
        keys = [None if m else int(v) for v, m in zip(keys.data, np.ma.getmaskarray(keys))]
    elif not isinstance(keys, list):
        keys = keys.tolist()
    if key == "x":                                          # isequal on Float64: one NaN, -0.0 apart from 0.0 (the keys come back as floats: compare images)
        keys = [("nan",) if v != v else (v, bool(np.signbit(v))) for v in keys]
    vals = cols[val][idx]
    ids = _np_group_ids(keys)
    order, cnt, acc = _np_groupreduce(ids, vals, "sum" if stat in ("min", "max") else stat)
    if stat in ("min", "max") and len(order):       # per group, the way Julia's minimum / maximum do it: a NaN in the group wins
        gid = ids[1]
        acc = np.array([(np.nan if (vals.dtype.kind == "f" and np.isnan(vals[gid == k]).any()) else (vals[gid == k].min() if stat == "min" else vals[gid == k].max()))
                        for k in range(len(order))], dtype=np.float64 if vals.dtype.kind == "f" else vals.dtype)
    got = dfdb_mod.groupreduce(dv, key, val, stat)
    if key == "x":
        gk = [("nan",) if v != v else (v, bool(np.signbit(v))) for v in got[key].tolist()]
    else:
        gk = [None if (k is None or k is np.ma.masked or (isinstance(k, float) and k != k)) else k for k in got[key].tolist()]
    assert gk == [None if k is None else k for k in order], (key, stages)
    assert got["count"].tolist() == cnt.tolist()
    if stat != "count" and len(order):
        gv = np.asarray(got[stat])
        if vals.dtype.kind == "f" or stat == "mean":
            with np.errstate(invalid="ignore"):
                tol = max(1, len(idx)) * np.finfo(np.float64).eps * float(np.nansum(np.abs(vals.astype(np.float64)))) + 1e-300
                a, b = gv.astype(np.float64), np.asarray(acc, np.float64)
                ok = (np.abs(a - b) <= tol) | np.isnan(a) | (a == b)
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.all(ok), (stat, a[:4], b[:4])
        else:
            assert gv.astype(np.int64).tolist() == np.asarray(acc).astype(np.int64).tolist(), (stat, val)
    # unique of the key column over the same queue
    uk = dv[dfdb_mod.ALL, key].unique()
    if key == "x":
        uk = [("nan",) if v != v else (v, bool(np.signbit(v))) for v in uk.tolist()]
    else:
        uk = [None if (k is None or k is np.ma.masked) else k for k in (uk.tolist() if hasattr(uk, "tolist") else list(uk))]
    assert uk == [None if k is None else k for k in order]


_FULL_COLUMNS = {}


def full_columns(pair):
    key = id(pair)
    if key not in _FULL_COLUMNS:
        out = pair.o.view().materialize()
        d = {}
        for n, c in zip(pair.names, out):
            d[n] = pair.O.flat_to_strings(*c) if isinstance(c, tuple) else c
        _FULL_COLUMNS[key] = d
    return _FULL_COLUMNS[key]
