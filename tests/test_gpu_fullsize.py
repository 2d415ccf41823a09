"""GPU: BASELINE.json's full sizes, checked through size-independent properties (the oracle cannot finish 1e9 rows in
seconds): sortedness, count == popcount(bitmap) == torch's own count, every gathered value satisfies the predicate,
idempotence, and bit-exact agreement with the oracle on SAMPLED 65 536-row blocks of the same seeded column."""
import ctypes as C
import json
import os
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
SEED = 0x9E3779B97F4A7C15
THR = 899_999


def _dev_outcol(N, tensor, bytes_t=None):
    o = N.OutCol()
    o.data, o.memkind = tensor.data_ptr(), N.MEM_DEVICE
    if bytes_t is not None:
        o.bytes, o.bytes_cap = bytes_t.data_ptr(), bytes_t.numel()
    return o


@pytest.mark.parametrize("n", [1_000_000_000])
def test_config2_full_size_properties(oracle, dfdb_mod, ctx, n):
    import torch
    from dfdb import _native as N
    free, _ = torch.cuda.mem_get_info()
    if free < 30e9:
        pytest.skip("needs ~20 GB of HBM")
    dev = torch.device("cuda", 0)
    t = dfdb_mod.DFTable.new()
    t.add_generated("x", dfdb_mod.GEN_I64_MOD1M, SEED, n)
    v = t[("x", lambda x: x > THR), dfdb_mod.ALL]
    q = v._query()
    nsel = q.count()
    assert abs(nsel / n - 0.1) < 1e-3
    idx = torch.empty(nsel, dtype=torch.int64, device=dev)
    assert q.indices_device(idx.data_ptr(), nsel, want_count=True) == nsel
    torch.cuda.synchronize()
    # sortedness / range
    assert bool((idx[1:] > idx[:-1]).all()) and int(idx[0]) >= 1 and int(idx[-1]) <= n
    # the whole column, materialized to the device through the ABI, is torch's third opinion
    full = torch.empty(n, dtype=torch.int64, device=dev)
    qa = dfdb_mod.DFView(t)._query()
    outs = (N.OutCol * 1)(_dev_outcol(N, full))
    N.check(N.load().dfdb_materialize(qa._h, outs, 1))
    torch.cuda.synchronize()
    mask = full > THR
    assert int(mask.sum()) == nsel
    assert bool(mask[idx - 1].all())                                   # every selected row satisfies the predicate
    assert int(idx.sum()) == int((torch.nonzero(mask).flatten() + 1).sum())   # checksum of the index list
    del mask
    # gathered values == column[idx]
    xs = torch.empty(nsel, dtype=torch.int64, device=dev)
    outs = (N.OutCol * 1)(_dev_outcol(N, xs))
    N.check(N.load().dfdb_materialize(q._h, outs, 1))
    torch.cuda.synchronize()
    assert torch.equal(xs, full[idx - 1])
    # bitmap popcount and sampled blocks against the oracle's generator + predicate
    bm = q.bitmap()
    assert int(np.unpackbits(bm.view(np.uint8)).sum()) == nsel
    rng = np.random.default_rng(1)
    nblocks = -(-n // 65536)
    for b in [0, 1, nblocks - 1, nblocks - 2] + rng.integers(0, nblocks, 12).tolist():
        r0 = b * 65536
        rows = min(65536, n - r0)
        want = oracle.gen_i64(SEED, r0, rows) > THR
        got = np.unpackbits(bm[r0 // 64:(r0 + rows + 63) // 64].view(np.uint8), bitorder="little")[:rows].astype(bool)
        assert np.array_equal(got, want), f"block {b}"
        lo, hi = int(torch.searchsorted(idx, r0 + 1)), int(torch.searchsorted(idx, r0 + rows + 1))
        assert np.array_equal(idx[lo:hi].cpu().numpy(), np.nonzero(want)[0] + r0 + 1), f"block {b}"
    # idempotence: a fresh evaluation gives the same bytes
    idx2 = torch.empty_like(idx)
    q.reset()
    q.indices_device(idx2.data_ptr(), nsel)
    torch.cuda.synchronize()
    assert torch.equal(idx, idx2)
    # sum over the filtered column == torch's exact integer sum
    assert v[dfdb_mod.ALL, "x"].sum() == int(xs.sum())
    t.close()


def _need_hbm(gb):
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < gb * 1e9:
        pytest.skip(f"needs ~{gb} GB of free HBM")


def _seed(k):
    return (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF


def test_config1_file_table_materialize(oracle, dfdb_mod, ctx, tmp_path):
    """BASELINE config 1 at its stated size: a single Int64 column of 1e6 rows in a reference-format file table (16 blocks of 65 536
    rows, the last one 16 960 rows), written by the oracle's liblz4 writer, opened and fully materialised by the engine == the generator;
    and the same table written by the DEVICE encoder, read back by the oracle (liblz4) == the generator."""
    n = 1_000_000
    x = oracle.gen_i64(SEED, 0, n)
    ot = oracle.Table(block_size=65536)
    ot.add_column("x", x)
    ot.save(str(tmp_path / "c1"))
    t = dfdb_mod.open_table(str(tmp_path / "c1"))
    st = dfdb_mod.table_stats(t)
    assert int(st["rows"].iloc[0]) == n and -(-n // 65536) == 16 and n - 15 * 65536 == 16_960
    got = dfdb_mod.materialize(t)
    assert list(got.columns) == ["x"] and got["x"].dtype == np.int64 and np.array_equal(got["x"].to_numpy(), x)
    assert dfdb_mod.nrow(t) == n
    assert np.array_equal(dfdb_mod.materialize(t[dfdb_mod.jr(n - 16_959, n), dfdb_mod.ALL])["x"].to_numpy(), x[-16_960:])      # the short last block
    t.save(str(tmp_path / "c1_dev"))
    back = oracle.Table.open(str(tmp_path / "c1_dev")).view().materialize()[0]
    assert np.array_equal(back, x)
    t.close()


@pytest.mark.parametrize("n", [200_000_000, 1_000_000_000])
def test_config3_conjunction_projection(oracle, dfdb_mod, ctx, n):
    """BASELINE config 3 (at 2e8 and at its stated 1e9 rows): 3-column Int64 + Float64 table, conjunctive predicate + projection.
    Properties at full size (torch as the third opinion: the count, every projected value, the index checksum) + sampled 65 536-row
    blocks against the oracle's generators bit for bit."""
    import torch
    from dfdb import _native as N
    _need_hbm(n * 24 / 1e9 + n * 18 / 1e9 + 6)
    dev = torch.device("cuda", 0)
    t = dfdb_mod.DFTable.new()
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, _seed(0), n)
    t.add_generated("b", dfdb_mod.GEN_I64_MOD1M, _seed(1), n)
    t.add_generated("x", dfdb_mod.GEN_F64_U2000, _seed(2), n)
    v = t[(t.a > 683_771) & (t.x < 632.456), ["b", "x"]]
    q = v._query()
    q.hint_materialize(True)
    nsel = q.count()
    assert abs(nsel / n - 0.1) < 2e-3
    ob = torch.empty(nsel, dtype=torch.int64, device=dev)
    ox = torch.empty(nsel, dtype=torch.float64, device=dev)
    outs = (N.OutCol * 2)(_dev_outcol(N, ob), _dev_outcol(N, ox))
    N.check(N.load().dfdb_materialize(q._h, outs, 2))
    torch.cuda.synchronize()
    assert bool((ox < 632.456).all())
    # third opinion: the whole columns through the ABI into torch, torch's own mask
    full = {}
    for name, dt in (("a", torch.int64), ("b", torch.int64), ("x", torch.float64)):
        full[name] = torch.empty(n, dtype=dt, device=dev)
        qa = dfdb_mod.DFView(t)[dfdb_mod.ALL, [name]]._query()
        N.check(N.load().dfdb_materialize(qa._h, (N.OutCol * 1)(_dev_outcol(N, full[name])), 1))
    torch.cuda.synchronize()
    mask = (full["a"] > 683_771) & (full["x"] < 632.456)
    assert int(mask.sum()) == nsel
    assert torch.equal(ob, full["b"][mask]) and torch.equal(ox.view(torch.int64), full["x"][mask].view(torch.int64))
    didx = torch.empty(nsel, dtype=torch.int64, device=dev)
    q.indices_device(didx.data_ptr(), nsel)
    torch.cuda.synchronize()
    assert torch.equal(didx, torch.nonzero(mask).flatten() + 1)
    del mask, full
    idx = didx.cpu().numpy()
    for r0 in (0, 65536 * 1000, (n // 65536 - 1) * 65536, n - 65536):      # sampled blocks vs the oracle's generators
        a = oracle.gen_i64(_seed(0), r0, 65536); b = oracle.gen_i64(_seed(1), r0, 65536); x = oracle.gen_f64(_seed(2), r0, 65536)
        m = (a > 683_771) & (x < 632.456)
        lo, hi = np.searchsorted(idx, r0 + 1), np.searchsorted(idx, r0 + 65536 + 1)
        assert np.array_equal(idx[lo:hi], np.nonzero(m)[0] + r0 + 1)
        assert np.array_equal(ob[lo:hi].cpu().numpy(), b[m]) and np.array_equal(ox[lo:hi].cpu().numpy().view(np.uint64), x[m].view(np.uint64))
    t.close()


@pytest.mark.parametrize("n", [100_000_000, 500_000_000])
def test_config4_string_equality_materialize(oracle, dfdb_mod, ctx, n):
    """BASELINE config 4 (at 1e8 and at its stated 5e8 rows): FlatStringsVector + Int64, string-equality filter + materialize."""
    import torch
    from dfdb import _native as N
    _need_hbm(n * 30 / 1e9 + 6)
    dev = torch.device("cuda", 0)
    t = dfdb_mod.DFTable.new()
    t.add_generated("s", dfdb_mod.GEN_STR_BRANDS10, _seed(0), n)
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, _seed(1), n)
    v = t[t.s == "sony", dfdb_mod.ALL]
    q = v._query()
    q.hint_materialize(True)
    nsel = q.count()
    assert abs(nsel / n - 0.1) < 2e-3
    nb = C.c_int64()
    N.check(N.load().dfdb_result_string_bytes(q._h, 0, C.byref(nb)))
    assert nb.value == 4 * nsel                                          # every selected string is "sony"
    osz = torch.empty(nsel, dtype=torch.int32, device=dev)
    oby = torch.empty(nb.value + 64, dtype=torch.uint8, device=dev)
    oa = torch.empty(nsel, dtype=torch.int64, device=dev)
    outs = (N.OutCol * 2)(_dev_outcol(N, osz, oby), _dev_outcol(N, oa))
    N.check(N.load().dfdb_materialize(q._h, outs, 2))
    torch.cuda.synchronize()
    assert bool((osz == 4).all())
    assert bool((oby[:nb.value].view(nsel, 4) == torch.tensor(list(b"sony"), dtype=torch.uint8, device=dev)).all())
    # third opinion for the Int64 projection: the whole column through the ABI, gathered by torch at the engine's row numbers
    didx = torch.empty(nsel, dtype=torch.int64, device=dev)
    q.indices_device(didx.data_ptr(), nsel)
    full = torch.empty(n, dtype=torch.int64, device=dev)
    qa = dfdb_mod.DFView(t)[dfdb_mod.ALL, ["a"]]._query()
    N.check(N.load().dfdb_materialize(qa._h, (N.OutCol * 1)(_dev_outcol(N, full)), 1))
    torch.cuda.synchronize()
    assert bool((didx[1:] > didx[:-1]).all()) and torch.equal(oa, full[didx - 1])
    # ... and for the selection itself: the sizes column says which rows CAN be "sony" (4 bytes: sony, dell, xbox, asus): the plain
    # != selection over the same column must select the complement
    q2 = t[t.s != "sony", ["a"]]._query()
    assert q2.count() == n - nsel
    del full
    idx = didx.cpu().numpy()
    for r0 in (0, 65536 * 700, n - 65536):
        sz, by = oracle.gen_str(_seed(0), r0, 65536)
        strs = oracle.flat_to_strings(sz, by)
        m = np.array([s == "sony" for s in strs])
        lo, hi = np.searchsorted(idx, r0 + 1), np.searchsorted(idx, r0 + 65536 + 1)
        assert np.array_equal(idx[lo:hi], np.nonzero(m)[0] + r0 + 1)
        assert np.array_equal(oa[lo:hi].cpu().numpy(), oracle.gen_i64(_seed(1), r0, 65536)[m])
    t.close()


def _exact_sum(torch, x):
    """the exact real value of sum(x) for a Float64 device tensor, as a Fraction: every value is m * 2^(e - 53) with an integer mantissa m < 2^53;
    per exponent the mantissas are added as integers (split into 27 low and 26 high bits so that 2^28 terms cannot overflow Int64)"""
    mant, exp = torch.frexp(x)
    m = (mant * float(2 ** 53)).to(torch.int64)
    total = Fraction(0)
    for e in torch.unique(exp).tolist():
        me = m[exp == e]
        lo, hi = int((me & ((1 << 27) - 1)).sum().item()), int((me >> 27).sum().item())
        total += Fraction((hi << 27) + lo) * Fraction(2) ** (int(e) - 53)
    return total


def test_config5_one_shard_count_and_sum(oracle, dfdb_mod, ctx):
    """BASELINE config 5's per-GPU share at its stated size: 1e10 rows over 8 GPUs = 1.25e9 rows x (Int64, Float64, String) on one device
    (32 GB), conjunctive predicate over all three columns, count() + sum(x) — through a ONE-rank group (dfdb_group_*: the code path the
    8-GPU run takes, with its RCCL all-reduce issued for real) and through the plain query; torch is the third opinion for the numeric
    conjuncts and for the sum, sampled blocks go against the oracle's generators."""
    import torch
    from dfdb import _native as N, group as G
    n = 1_250_000_000
    _need_hbm(n * 26 / 1e9 + n * 17 / 1e9 + 10)
    dev = torch.device("cuda", 0)
    g = G.Group.create([0], N.EXCHANGE_RCCL)
    try:
        gt = G.GroupTable.new(g)
        gt.add_generated("a", dfdb_mod.GEN_I64_MOD1M, _seed(0), n)
        gt.add_generated("x", dfdb_mod.GEN_F64_U2000, _seed(1), n)
        gt.add_generated("s", dfdb_mod.GEN_STR_BRANDS10, _seed(2), n)
        t = gt.shard(0)
        base = gt.view()
        v = base[(base.a > 500_000) & (base.x < 800.0) & (base.s != "sony"), dfdb_mod.ALL]
        cnt = G.gnrow(v)
        sx = G.gaggregate(v[dfdb_mod.ALL, "x"], N.AGG_SUM)
        # the plain single-GPU query over the same shard agrees exactly on the count and within tolerance on the sum
        pv = t[(t.a > 500_000) & (t.x < 800.0) & (t.s != "sony"), dfdb_mod.ALL]
        assert dfdb_mod.nrow(pv) == cnt
        assert abs(cnt / n - 0.5 * 0.4 * 0.9) < 1e-3
        # third opinion: numeric conjuncts by torch, the string conjunct through the engine's own != selection as a bitmap
        full = {}
        for name, dt in (("a", torch.int64), ("x", torch.float64)):
            full[name] = torch.empty(n, dtype=dt, device=dev)
            qa = dfdb_mod.DFView(t)[dfdb_mod.ALL, [name]]._query()
            N.check(N.load().dfdb_materialize(qa._h, (N.OutCol * 1)(_dev_outcol(N, full[name])), 1))
        torch.cuda.synchronize()
        num = (full["a"] > 500_000) & (full["x"] < 800.0)
        nnum = int(num.sum())
        assert dfdb_mod.nrow(t[(t.a > 500_000) & (t.x < 800.0), dfdb_mod.ALL]) == nnum
        assert dfdb_mod.nrow(t[(t.a > 500_000) & (t.x < 800.0) & (t.s == "sony"), dfdb_mod.ALL]) == nnum - cnt      # the complement inside A & B
        bm = torch.from_numpy(pv._query().bitmap().view(np.int64)).to(dev)                                    # the final mask, 1 bit per row
        bits = ((bm.view(-1, 1) >> torch.arange(64, device=dev, dtype=torch.int64)) & 1).to(torch.bool).view(-1)[:n]
        assert int(bits.sum()) == cnt and bool((bits <= num).all())                                              # selected rows are a subset of A & B
        # Float64 tolerance, measured: the EXACT real sum of the selected values (mantissas added as integers, exponent by
        # exponent) is the yardstick — Julia's own left-to-right sum is only within n * eps * sum|x| of it, the engine's fixed-shape tree
        # (lane -> wave -> block -> final) must be within 64 * eps * sum|x| (log2(2.25e8) = 27.7 levels, twice over), and torch's
        # reduction is a third opinion held to the same bound.  The achieved error goes on record (gpurun_out/r3_float_sum_error.json).
        sel_x = full["x"][bits]
        exact = _exact_sum(torch, sel_x)
        eps_abs = np.finfo(np.float64).eps * float(sel_x.abs().sum().item())
        err_engine, err_torch = abs(Fraction(sx) - exact), abs(Fraction(float(sel_x.sum().item())) - exact)
        rec = {"terms": cnt, "eps_times_sum_abs": eps_abs, "engine_error_in_eps_sum_abs": float(err_engine / Fraction(eps_abs)),
               "torch_error_in_eps_sum_abs": float(err_torch / Fraction(eps_abs)), "bound_in_eps_sum_abs": 64.0,
               "left_to_right_worst_case_in_eps_sum_abs": float(cnt)}
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "r3_float_sum_error.json"), "w") as f:
                json.dump(rec, f)
        except OSError:
            pass
        assert err_engine <= 64 * Fraction(eps_abs), rec
        assert err_torch <= 64 * Fraction(eps_abs), rec
        del sel_x
        del full, num, bits, bm
        idx = pv._query().indices()
        for r0 in (0, 65536 * 9000, (n // 65536) * 65536 - 65536):
            a = oracle.gen_i64(_seed(0), r0, 65536); x = oracle.gen_f64(_seed(1), r0, 65536)
            sz, by = oracle.gen_str(_seed(2), r0, 65536)
            strs = oracle.flat_to_strings(sz, by)
            m = (a > 500_000) & (x < 800.0) & np.array([s != "sony" for s in strs])
            lo, hi = np.searchsorted(idx, r0 + 1), np.searchsorted(idx, r0 + 65536 + 1)
            assert np.array_equal(idx[lo:hi], np.nonzero(m)[0] + r0 + 1)
        gt.close()
    finally:
        g.close()


def test_hinted_paths_equal_the_plain_ones_at_scale(oracle, dfdb_mod, ctx):
    """The two execution hints never change a result: at 1e8 / 2e8 rows the captured projections (numeric terms: k_scan_terms
    EXTRA = 1, String: K5 CAP) are byte-identical to the gathered ones, the sum folded into the scan (EXTRA = 2) equals the
    separate reduce pass exactly for Int64 and within n * eps * sum|x| for Float64."""
    import torch
    from dfdb import _native as N, ir
    dev = torch.device("cuda", 0)

    seed = _seed
    lib = N.load()
    # ---- numeric capture + fused sums
    n = 200_000_000
    t = dfdb_mod.DFTable.new()
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, seed(0), n)
    t.add_generated("x", dfdb_mod.GEN_F64_U2000, seed(2), n)
    v = t[(t.a > 683_771) & (t.x < 632.456), ["a", "x"]]
    q = v._query()
    outs_by_hint = {}
    for hint in (False, True):
        q.hint_materialize(hint)
        q.reset()
        nsel = q.count()
        oa = torch.empty(nsel, dtype=torch.int64, device=dev); ox = torch.empty(nsel, dtype=torch.float64, device=dev)
        outs = (N.OutCol * 2)(_dev_outcol(N, oa), _dev_outcol(N, ox))
        N.check(lib.dfdb_materialize(q._h, outs, 2))
        torch.cuda.synchronize()
        outs_by_hint[hint] = (oa, ox)
    assert torch.equal(outs_by_hint[False][0], outs_by_hint[True][0])
    assert torch.equal(outs_by_hint[False][1].view(torch.int64), outs_by_hint[True][1].view(torch.int64))
    q.hint_materialize(False)
    sums = {}
    for hint in (0, N.AGG_SUM):
        for col in (0, 1):
            q.reset()
            N.check(lib.dfdb_query_hint_aggregate(q._h, hint, col))
            oi, of = C.c_int64(), C.c_double()
            N.check(lib.dfdb_aggregate(q._h, N.AGG_SUM, col, C.byref(oi), C.byref(of)))
            sums[(hint, col)] = (oi.value, of.value)
    assert sums[(0, 0)][0] == sums[(N.AGG_SUM, 0)][0] == int(outs_by_hint[True][0].sum().item())
    xs = outs_by_hint[True][1]
    tol = xs.numel() * np.finfo(np.float64).eps * float(xs.abs().sum().item())
    assert abs(sums[(0, 1)][1] - sums[(N.AGG_SUM, 1)][1]) <= tol and abs(sums[(N.AGG_SUM, 1)][1] - float(xs.sum().item())) <= tol
    del outs_by_hint, xs
    t.close()
    # ---- String capture: variable-length survivors (startswith "s": sony, samsung) and a 90 % selection (!= "sony")
    n = 100_000_000
    t = dfdb_mod.DFTable.new()
    t.add_generated("s", dfdb_mod.GEN_STR_BRANDS10, seed(0), n)
    for v in (t[dfdb_mod.startswith(t.s, "s"), dfdb_mod.ALL], t[t.s != "sony", dfdb_mod.ALL]):
        q = v._query()
        got = {}
        for hint in (False, True):
            q.hint_materialize(hint)
            q.reset()
            nsel = q.count()
            nb = C.c_int64()
            N.check(lib.dfdb_result_string_bytes(q._h, 0, C.byref(nb)))
            osz = torch.empty(nsel, dtype=torch.int32, device=dev)
            oby = torch.zeros(nb.value + 64, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()          # (torch's fill runs on torch's stream, the engine writes on its own)
            outs = (N.OutCol * 1)(_dev_outcol(N, osz, oby))
            N.check(lib.dfdb_materialize(q._h, outs, 1))
            torch.cuda.synchronize()
            got[hint] = (nsel, nb.value, osz, oby)
        assert got[False][0] == got[True][0] and got[False][1] == got[True][1] and got[True][1] == int(got[True][2].sum().item())
        assert torch.equal(got[False][2], got[True][2]) and torch.equal(got[False][3][:got[True][1]], got[True][3][:got[True][1]])
        del got
    t.close()


def test_interpreter_large_properties(oracle, dfdb_mod, ctx):
    """A predicate and a computed column that only the device interpreter can run, at 2e8 rows: torch evaluates the same
    Int64 / Float64 arithmetic on the device as a third opinion, the oracle checks sampled blocks bit for bit."""
    import torch
    from dfdb import _native as N, ir
    dev = torch.device("cuda", 0)

    def seed(k):
        return (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF
    n = 200_000_000
    t = dfdb_mod.DFTable.new()
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, seed(0), n)
    t.add_generated("b", dfdb_mod.GEN_I64_MOD1M, seed(1), n)
    t.add_generated("x", dfdb_mod.GEN_F64_U2000, seed(2), n)
    pred = ((t.a * 2 + t.b) % 7 == 0) & (t.x * 0.5 < 400.0)
    v = t[pred, {"k": t.a * 3 - t.b, "r": t.x / 3.0 + t.a}]
    q = v._query()
    nsel = q.count()
    cols = []
    for name, dt in (("a", torch.int64), ("b", torch.int64), ("x", torch.float64)):
        full = torch.empty(n, dtype=dt, device=dev)
        qa = dfdb_mod.DFView(t)[dfdb_mod.ALL, [name]]._query()
        N.check(N.load().dfdb_materialize(qa._h, (N.OutCol * 1)(_dev_outcol(N, full)), 1))
        cols.append(full)
    torch.cuda.synchronize()
    a, b, x = cols
    mask = (torch.remainder(a * 2 + b, 7) == 0) & (x * 0.5 < 400.0)     # operands are non-negative: rem == mod
    assert int(mask.sum()) == nsel
    idx = torch.empty(nsel, dtype=torch.int64, device=dev)
    q.indices_device(idx.data_ptr(), nsel)
    ok = torch.empty(nsel, dtype=torch.int64, device=dev); orr = torch.empty(nsel, dtype=torch.float64, device=dev)
    N.check(N.load().dfdb_materialize(q._h, (N.OutCol * 2)(_dev_outcol(N, ok), _dev_outcol(N, orr)), 2))
    torch.cuda.synchronize()
    assert torch.equal(idx, torch.nonzero(mask).flatten() + 1)
    assert torch.equal(ok, (a * 3 - b)[mask])
    # (torch divides by a Python scalar as x * (1/3.0): not IEEE division.  A tensor divisor takes the true-division kernel.)
    three = torch.full((1,), 3.0, dtype=torch.float64, device=dev)
    assert torch.equal(orr.view(torch.int64), (torch.div(x, three) + a.to(torch.float64))[mask].view(torch.int64))   # IEEE: same two roundings
    hidx = idx.cpu().numpy(); hk = ok.cpu().numpy(); hr = orr.cpu().numpy()
    for r0 in (0, 65536 * 777, n - 65536):
        oa = oracle.gen_i64(seed(0), r0, 65536); ob_ = oracle.gen_i64(seed(1), r0, 65536); ox_ = oracle.gen_f64(seed(2), r0, 65536)
        m = ((oa * 2 + ob_) % 7 == 0) & (ox_ * 0.5 < 400.0)
        lo, hi = np.searchsorted(hidx, r0 + 1), np.searchsorted(hidx, r0 + 65536 + 1)
        assert np.array_equal(hidx[lo:hi], np.nonzero(m)[0] + r0 + 1)
        assert np.array_equal(hk[lo:hi], (oa * 3 - ob_)[m])
        assert np.array_equal(hr[lo:hi].view(np.uint64), (ox_ / 3.0 + oa)[m].view(np.uint64))     # numpy: IEEE division and addition
    t.close()


def test_writer_large_roundtrip(oracle, dfdb_mod, ctx, tmp_path):
    """5e7 rows x (Int64, Float64, String) written by the device encoder, read back by the device decoder: equal on the device."""
    import torch
    from dfdb import _native as N
    dev = torch.device("cuda", 0)
    n = 50_000_000
    t = dfdb_mod.DFTable.new()
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, SEED, n)
    t.add_generated("x", dfdb_mod.GEN_F64_U2000, SEED + 1, n)
    t.add_generated("s", dfdb_mod.GEN_STR_BRANDS10, SEED + 2, n)
    st = t.save(str(tmp_path / "big"))
    assert st["rows"] == n and st["compressed"] < st["uncompressed"]
    t2 = dfdb_mod.open_table(str(tmp_path / "big"))

    def column(tb, name, dt, count):
        out = torch.empty(count, dtype=dt, device=dev)
        qa = dfdb_mod.DFView(tb)[dfdb_mod.ALL, [name]]._query()
        N.check(N.load().dfdb_materialize(qa._h, (N.OutCol * 1)(_dev_outcol(N, out)), 1))
        torch.cuda.synchronize()          # device outputs are ordered on the ENGINE's stream, not on torch's
        return out
    for name, dt in (("a", torch.int64), ("x", torch.int64)):          # Float64 compared as bit patterns
        assert torch.equal(column(t, name, dt, n), column(t2, name, dt, n))
    v1, v2 = t[t.s == "sony", ["a"]], t2[t2.s == "sony", ["a"]]
    assert dfdb_mod.nrow(v1) == dfdb_mod.nrow(v2) > 0
    assert v1.a.sum() == v2.a.sum()
    # and liblz4 (the oracle) agrees on the first and the last block of the Int64 column
    ot = oracle.Table.open(str(tmp_path / "big"))
    ov = ot.view(); ov.set_projection([("a", __import__("dfdb").ir.col(0).to_ir())]); ov.add_range(n - 70_000, 1, n)
    got = ov.materialize()[0]
    assert np.array_equal(got, oracle.gen_i64(SEED, n - 70_001, 70_001))
    t.close(); t2.close()


def test_compressed_only_table_3e9_rows_answers_config3(oracle, dfdb_mod, ctx):
    """SURVEY.md section 8f-2 at scale: a 3e9-row x 3-column table (72 GB decoded) held COMPRESSED-ONLY on one GPU — every column encoded on
    the device into its LZ4 blocks, the decoded arrays released — answers config 3's query: the count, the projection [b, x] and the row indices equal the
    oracle's on sampled 65 536-row blocks, every projected x satisfies the predicate, and no decoded column is resident before or after."""
    import torch
    from dfdb import _native as N
    n = 3_000_000_000
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs ~150 GB of free HBM (one decoded column at a time beside 41 GB of blocks and 24 GB of results)")
    dev = torch.device("cuda", 0)
    t = dfdb_mod.DFTable.new()
    comp = 0
    for k, (name, gen) in enumerate((("a", dfdb_mod.GEN_I64_MOD1M), ("b", dfdb_mod.GEN_I64_MOD1M), ("x", dfdb_mod.GEN_F64_U2000))):
        t.add_generated(name, gen, _seed(k), n)
        st = t.compress_column(name, 2)                   # one decoded column at a time
        assert st["rows"] == n
        comp += st["compressed"]
    rb = t.resident_bytes()
    assert rb["decoded"] < 1 << 20 and comp <= rb["compressed"] < 0.8 * n * 24, rb
    v = t[(t.a > 683_771) & (t.x < 632.456), ["b", "x"]]
    q = v._query()
    nsel = q.count()
    assert abs(nsel / n - 0.1) < 2e-3
    assert t.resident_bytes()["decoded"] < 1 << 20          # the two scans decoded into the history rings only
    ob = torch.empty(nsel, dtype=torch.int64, device=dev)
    ox = torch.empty(nsel, dtype=torch.float64, device=dev)
    outs = (N.OutCol * 2)(_dev_outcol(N, ob), _dev_outcol(N, ox))
    N.check(N.load().dfdb_materialize(q._h, outs, 2))
    torch.cuda.synchronize()
    assert bool((ox < 632.456).all())
    didx = torch.empty(nsel, dtype=torch.int64, device=dev)
    q.indices_device(didx.data_ptr(), nsel)
    torch.cuda.synchronize()
    assert bool((didx[1:] > didx[:-1]).all())
    nb = -(-n // 65536)
    for blk in (0, 1000, 17_123, nb // 2, nb - 2, nb - 1):   # sampled blocks vs the oracle's generators
        r0 = blk * 65536
        m_rows = min(65536, n - r0)
        a = oracle.gen_i64(_seed(0), r0, m_rows); b = oracle.gen_i64(_seed(1), r0, m_rows); x = oracle.gen_f64(_seed(2), r0, m_rows)
        m = (a > 683_771) & (x < 632.456)
        lo = int(torch.searchsorted(didx, torch.tensor([r0 + 1], device=dev))[0]); hi = int(torch.searchsorted(didx, torch.tensor([r0 + m_rows + 1], device=dev))[0])
        assert np.array_equal(didx[lo:hi].cpu().numpy(), np.nonzero(m)[0] + r0 + 1), blk
        assert np.array_equal(ob[lo:hi].cpu().numpy(), b[m]) and np.array_equal(ox[lo:hi].cpu().numpy().view(np.uint64), x[m].view(np.uint64)), blk
    del q
    t.close()


def test_unique_by_radix_at_full_size(dfdb_mod, ctx):
    """unique(col) over 1e9 rows of 1e6 distinct values (round 6: csrc/k_radix.hip — the partition pass's page pool, the LDS tables): the first occurrences it
    marks are the hash-table form's, row for row, for the Int64 column, for a Float64 image of it and under a predicate; every value of the range is there once;
    a second call gives the same rows (the order of a partition's records depends on timing, the answer must not)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("needs ~45 GB of HBM")
    import dfdb._native as N
    n = 1_000_000_000
    t = dfdb_mod.DFTable.new(ctx=ctx)
    t.add_generated("x", dfdb_mod.GEN_I64_MOD1M, SEED, n)
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, SEED * 2, n)
    t.add_column_from("f", t.x * 0.5)

    def first_rows(view, radix):
        ctx.set_option("unique_dense", 0); ctx.set_option("unique_radix", radix); ctx.profile(True)
        try:
            q = view._query()
            N.check(N.load().dfdb_query_unique(q._h, 0))
            rows = q.indices()
            taken = ctx.profile_get("unique_radix.taken")[0]
        finally:
            ctx.profile(False); ctx.set_option("unique_dense", 1); ctx.set_option("unique_radix", 1)
        return rows, taken

    try:
        for label, view in (("x", t[dfdb_mod.ALL, ["x"]]), ("f", t[dfdb_mod.ALL, ["f"]]), ("x where a > 499999", t[("a", lambda c: c > 499_999), ["x"]])):
            r1, taken = first_rows(view, 1)
            assert taken == 1, label
            r2, _ = first_rows(view, 1)
            h, taken_h = first_rows(view, 0)
            assert taken_h == 0
            assert np.array_equal(r1, h) and np.array_equal(r1, r2), label
            assert len(r1) == 1_000_000 and bool(np.all(r1[1:] > r1[:-1])), label
        vals = t.x.unique()                                                  # (the dense form: a third opinion on the values)
        assert len(vals) == 1_000_000 and np.array_equal(np.sort(vals), np.arange(1_000_000))
    finally:
        t.close()
