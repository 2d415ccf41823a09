#!/usr/bin/env python3
"""open_table (file -> HBM, decoded) per column type: Int64, Float64, String, Union{Int64,Missing}.  python tools/bench_open.py [rows]"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch  # noqa
import dfdb

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ctx = dfdb.default_context(0)
S = 0x9E3779B97F4A7C15
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    kinds = {"int64": lambda t: t.add_generated("v", dfdb.GEN_I64_MOD1M, S, n), "float64": lambda t: t.add_generated("v", dfdb.GEN_F64_U2000, S, n),
             "string": lambda t: t.add_generated("v", dfdb.GEN_STR_BRANDS10, S, n)}
    for kind, make in kinds.items():
        t = dfdb.DFTable.new(); make(t)
        path = os.path.join(d, kind)
        t0 = time.perf_counter(); st = t.save(path); ws = time.perf_counter() - t0
        t.close()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); tb = dfdb.open_table(path); dt = time.perf_counter() - t0
            best = min(best, dt); tb.close()
        print(json.dumps({"column": kind, "rows": n, "file_MB": st["compressed"] / 1e6, "body_MB": st["uncompressed"] / 1e6, "save_s": ws,
                          "open_s": best, "open_rows_per_s": n / best, "open_decoded_GBps": st["uncompressed"] / best / 1e9}))
    m = min(n, 20_000_000)
    rng = np.random.default_rng(1)
    t = dfdb.DFTable.from_columns({"v": np.ma.masked_array(rng.integers(0, 1000, m).astype(np.int64), mask=rng.random(m) < 0.2)})
    path = os.path.join(d, "nullable")
    st = t.save(path); t.close()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); tb = dfdb.open_table(path); dt = time.perf_counter() - t0
        best = min(best, dt); tb.close()
    print(json.dumps({"column": "Union{Int64,Missing}", "rows": m, "file_MB": st["compressed"] / 1e6, "open_s": best, "open_rows_per_s": m / best}))
finally:
    shutil.rmtree(d, ignore_errors=True)
