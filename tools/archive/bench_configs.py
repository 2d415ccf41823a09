#!/usr/bin/env python3
"""Per-kernel timings of BASELINE.json configs 2-4 and of the device LZ4 decode stage on one MI355X.

Not the driver's bench (that is /bench.py, config 2 only): this prints one JSON object per config with the
HIP-event time of every kernel family, the algorithmic bytes of SURVEY.md §8d and the resulting GB/s, for
DESIGN.md's roofline table.   python tools/bench_configs.py [--scale 1.0] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402,F401  (before any HIP call of libdfdb: torch must load ITS libamdhip64 first)
import numpy as np  # noqa: E402

import dfdb  # noqa: E402
from dfdb import ir  # noqa: E402

SEED = 0x9E3779B97F4A7C15
KERNELS = ["lz4_compress", "compact_captured", "scan_cmp", "scan_terms", "str_match", "interp_predicate", "interp_project", "scan_counts", "range_stage", "fill_ones",
           "compact_indices", "gather", "str_gather_sizes", "str_gather_bytes", "str_compact_captured", "reduce", "reduce_partials", "lz4_decode", "dict_scan", "dict_expand_sizes", "dict_expand_bytes", "dict_encode"]


def seed(k):
    return (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF


def timed(ctx, fn, reps):
    fn()
    ctx.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ks = {}
    for k in KERNELS:
        n, ms = ctx.profile_get(k)
        if n:
            ks[k] = round(ms / n, 4)
    ctx.profile(False)
    # un-profiled wall time (profiling serialises each launch on an event)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return ks, (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--lz4-rows", type=int, default=50_000_000)
    ap.add_argument("--compact-store", type=int, default=None, help="A/B: K2 index stores: 0 plain, 1 nontemporal (default), 2 write-through")
    args = ap.parse_args()
    ctx = dfdb.default_context(0)
    if args.compact_store is not None:
        ctx.set_option("compact_store", args.compact_store)
    info = ctx.device_info()
    print(json.dumps({"device": info}))
    import torch
    dev = torch.device("cuda", 0)

    # ---- config 2
    n = int(1e9 * args.scale)
    t = dfdb.DFTable.new()
    t.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, n)
    v = t[("x", lambda x: x > 899_999), dfdb.ALL]
    q = v._query()
    nsel = q.count()
    out = torch.empty(nsel, dtype=torch.int64, device=dev)

    def step2():
        q.execute(); q.indices_device(out.data_ptr(), nsel)
    ks, wall = timed(ctx, step2, args.reps)
    byts = n * 8 + nsel * 8
    print(json.dumps({"config": 2, "rows": n, "selected": nsel, "kernels_ms": ks, "wall_ms": wall * 1e3, "algorithmic_GB": byts / 1e9,
                      "job_GBps": byts / wall / 1e9, "rows_per_s": n / wall}))
    # A/B in ONE process on ONE device (boxes differ by several %): default-policy vs nontemporal column loads
    ab = {0: [], 1: []}
    ctx.profile(True)
    for r in range(6):
        for nt in (0, 1):
            ctx.set_option("scan_nt", nt)
            n0, ms0 = ctx.profile_get("scan_cmp")
            q.execute()
            n1, ms1 = ctx.profile_get("scan_cmp")
            if r:
                ab[nt].append(ms1 - ms0)
    ctx.profile(False)
    ctx.set_option("scan_nt", 1)
    print(json.dumps({"config": "2-nt-ab", "scan_cmp_ms_default": sorted(ab[0]), "scan_cmp_ms_nt": sorted(ab[1])}))
    ab = {0: [], 1: []}
    ctx.profile(True)
    for r in range(6):
        for wt in (0, 1):
            ctx.set_option("scan_wt_store", wt)
            n0, ms0 = ctx.profile_get("scan_cmp")
            q.execute()
            n1, ms1 = ctx.profile_get("scan_cmp")
            if r:
                ab[wt].append(ms1 - ms0)
    ctx.profile(False)
    ctx.set_option("scan_wt_store", 1)
    print(json.dumps({"config": "2-wt-store-ab", "scan_cmp_ms_plain_store": sorted(ab[0]), "scan_cmp_ms_wt_store": sorted(ab[1])}))
    # materialize(t[x > c, :]) with the values on the device: gather vs capture-in-scan (dfdb_query_hint_materialize)
    import ctypes as C
    from dfdb import _native as N
    xs = torch.empty(nsel, dtype=torch.int64, device=dev)
    o1 = (N.OutCol * 1)(); o1[0].data, o1[0].memkind = xs.data_ptr(), N.MEM_DEVICE
    for hint in (False, True):
        q.hint_materialize(hint)
        ks, wall = timed(ctx, lambda: (q.execute(), N.check(N.load().dfdb_materialize(q._h, o1, 1))), args.reps)
        print(json.dumps({"config": "2-materialize", "hint": hint, "kernels_ms": ks, "wall_ms": wall * 1e3}))
    q.hint_materialize(False)
    del xs
    cx = v[dfdb.ALL, "x"]
    cq = cx.view._query()
    for hint in (False, True):          # True is what sum() does: the scan adds the selected values up per tile (k_scan_terms EXTRA = 2)
        def step_sum():
            cq.reset()
            N.check(N.load().dfdb_query_hint_aggregate(cq._h, N.AGG_SUM if hint else 0, 0))
            oi, of = C.c_int64(), C.c_double()
            N.check(N.load().dfdb_aggregate(cq._h, N.AGG_SUM, 0, C.byref(oi), C.byref(of)))
            return oi.value
        ks, wall = timed(ctx, step_sum, args.reps)
        print(json.dumps({"config": "2-sum", "hint": hint, "sum": step_sum(), "kernels_ms": ks, "wall_ms": wall * 1e3}))
    # leading range stage then predicate: t[1:100000, :][x -> x > c, :] -- later stages skip tiles without survivors
    vh = t[dfdb.jr(1, 100_000), dfdb.ALL][("x", lambda x: x > 899_999), dfdb.ALL]
    qh = vh._query()
    ks, wall = timed(ctx, lambda: (qh.reset(), qh.execute()), args.reps)
    print(json.dumps({"config": "2-head", "rows": n, "selected": qh.count(), "kernels_ms": ks, "wall_ms": wall * 1e3}))
    del out, q, v, cx, qh, vh
    t.close()

    # ---- config 3: (a > 683771) & (x < 632.456), project [b, x]
    t = dfdb.DFTable.new()
    t.add_generated("a", dfdb.GEN_I64_MOD1M, seed(0), n)
    t.add_generated("b", dfdb.GEN_I64_MOD1M, seed(1), n)
    t.add_generated("x", dfdb.GEN_F64_U2000, seed(2), n)
    v = t[(t.a > 683_771) & (t.x < 632.456), ["b", "x"]]
    q = v._query()
    nsel = q.count()
    ob = torch.empty(nsel, dtype=torch.int64, device=dev)
    ox = torch.empty(nsel, dtype=torch.float64, device=dev)
    import ctypes as C
    from dfdb import _native as N
    outs = (N.OutCol * 2)()
    outs[0].data, outs[0].memkind = ob.data_ptr(), N.MEM_DEVICE
    outs[1].data, outs[1].memkind = ox.data_ptr(), N.MEM_DEVICE

    def step3():
        q.execute(); N.check(N.load().dfdb_materialize(q._h, outs, 2))
    ks, wall = timed(ctx, step3, args.reps)
    byts = n * 16 + nsel * 8 + nsel * 16
    print(json.dumps({"config": "3-gather-only", "rows": n, "kernels_ms": ks, "wall_ms": wall * 1e3, "job_GBps": byts / wall / 1e9}))
    q.hint_materialize(True)        # what materialize() does: the scan keeps the selected x values, b is still gathered
    ks, wall = timed(ctx, step3, args.reps)
    print(json.dumps({"config": 3, "rows": n, "selected": nsel, "kernels_ms": ks, "wall_ms": wall * 1e3, "algorithmic_GB": byts / 1e9,
                      "job_GBps": byts / wall / 1e9, "rows_per_s": n / wall}))
    # same predicate through the generic interpreter (x*1 defeats the term matcher)
    v2 = t[(t.a > 683_771) & (t.x * 1.0 < 632.456), ["b", "x"]]
    q2 = v2._query()
    ks, wall = timed(ctx, lambda: q2.execute(), max(2, args.reps // 2))
    print(json.dumps({"config": "3-interp", "kernels_ms": ks, "wall_ms": wall * 1e3, "count_equal": q2.count() == nsel}))
    del ob, ox, q, q2, v, v2
    t.close()

    # ---- config 4: s == "sony", materialize s and a
    n4 = int(5e8 * args.scale)
    t = dfdb.DFTable.new()
    t.add_generated("s", dfdb.GEN_STR_BRANDS10, seed(0), n4)
    t.add_generated("a", dfdb.GEN_I64_MOD1M, seed(1), n4)
    v = t[t.s == "sony", dfdb.ALL]
    q = v._query()
    nsel = q.count()
    nb = C.c_int64()
    N.check(N.load().dfdb_result_string_bytes(q._h, 0, C.byref(nb)))
    osz = torch.empty(nsel, dtype=torch.int32, device=dev)
    oby = torch.empty(nb.value + 64, dtype=torch.uint8, device=dev)
    oa = torch.empty(nsel, dtype=torch.int64, device=dev)
    outs = (N.OutCol * 2)()
    outs[0].data, outs[0].bytes, outs[0].bytes_cap, outs[0].memkind = osz.data_ptr(), oby.data_ptr(), nb.value, N.MEM_DEVICE
    outs[1].data, outs[1].memkind = oa.data_ptr(), N.MEM_DEVICE

    def step4():
        q.execute(); N.check(N.load().dfdb_materialize(q._h, outs, 2))
    lbar = 5.4
    byts = n4 * (4 + lbar) + nsel * (8 + 8 + 4 + 4)
    for hint in (False, True):      # True is what materialize() does: the match pass keeps the selected rows of s (K5 CAP)
        q.hint_materialize(hint)
        ks, wall = timed(ctx, step4, args.reps)
        print(json.dumps({"config": 4 if hint else "4-gather-only", "rows": n4, "selected": nsel, "string_bytes_out": nb.value, "kernels_ms": ks, "wall_ms": wall * 1e3,
                          "algorithmic_GB": byts / 1e9, "job_GBps": byts / wall / 1e9, "rows_per_s": n4 / wall}))
    # the same query with a dictionary beside s (K9: 16-bit codes + the 10 distinct strings): the predicate is a bit-table lookup of the codes,
    # the projection of s copies out of the dictionary
    t0 = time.perf_counter()
    nd = t.build_dictionary("s")
    ctx.synchronize()
    build_s = time.perf_counter() - t0
    q.hint_materialize(True)
    ks, wall = timed(ctx, step4, args.reps)
    print(json.dumps({"config": "4-dictionary", "rows": n4, "selected": nsel, "dictionary_entries": nd, "dictionary_build_s": build_s, "kernels_ms": ks, "wall_ms": wall * 1e3,
                      "algorithmic_GB": byts / 1e9, "job_GBps": byts / wall / 1e9, "rows_per_s": n4 / wall}))
    del osz, oby, oa, q, v
    t.close()

    # ---- LZ4 stages: a reference-format table written by the device compressor, decoded by each decoder variant
    # (blocks compressed by the system liblz4 instead: tools/bench_lz4, which also checks every decoded block)
    import shutil
    import tempfile
    m = args.lz4_rows
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        src = dfdb.DFTable.new()
        src.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, m)
        ctx.profile(True)
        t0 = time.perf_counter()
        st = src.save(os.path.join(d, "tb"))
        wall = time.perf_counter() - t0
        nl, ms = ctx.profile_get("lz4_compress")
        ctx.profile(False)
        print(json.dumps({"config": "lz4-write", "rows": m, "lz4_compress_ms": ms, "compress_GBps_in": m * 8 / (ms * 1e-3) / 1e9 if ms else None,
                          "ratio": st["uncompressed"] / max(1, st["compressed"]), "save_wall_s": wall}))
        want = dfdb.nrow(src[("x", lambda x: x > 899_999), dfdb.ALL])
        src.close()
        for variant in (4, 4):
            ctx.profile(True)
            t0 = time.perf_counter()
            tb = dfdb.open_table(os.path.join(d, "tb"))
            wall = time.perf_counter() - t0
            nl, ms = ctx.profile_get("lz4_decode")
            ctx.profile(False)
            assert dfdb.nrow(tb[("x", lambda x: x > 899_999), dfdb.ALL]) == want
            tb.close()
            print(json.dumps({"config": "lz4", "variant": variant, "rows": m, "blocks": -(-m // 65536), "compressed_MB": st["compressed"] / 1e6,
                              "lz4_decode_ms": ms, "decode_GBps_out": m * 8 / (ms * 1e-3) / 1e9 if ms else None, "open_table_wall_s": wall}))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
