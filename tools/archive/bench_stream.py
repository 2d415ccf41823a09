#!/usr/bin/env python3
"""Block-streamed scan of a table that is never resident: files -> pread -> PCIe -> K7 -> K1, chunk by chunk,
the loader thread one chunk ahead on its own HIP stream.  Prints one JSON line per chunk size (the CPU baseline of the same scan is bench.py's cpu_baseline leg).   python tools/bench_stream.py [--rows 2.5e8]"""
import argparse, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=float, default=2.5e8)
args = ap.parse_args()
n = int(args.rows)
SEED = 0x9E3779B97F4A7C15
ctx = dfdb.default_context(0)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    t = dfdb.DFTable.new()
    t.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, n)
    t0 = time.perf_counter()
    st = t.save(os.path.join(d, "tb"))
    print(json.dumps({"write_s": time.perf_counter() - t0, "rows": n, "file_MB": st["compressed"] / 1e6, "ratio": st["uncompressed"] / st["compressed"]}))
    want = t[("x", lambda x: x > 899_999), dfdb.ALL]._query().count()
    t.close()
    tb = dfdb.open_table(os.path.join(d, "tb"), load=False)
    v = tb[("x", lambda x: x > 899_999), dfdb.ALL]
    for chunk in (256, 512, 1024):
        for rep in range(2):
            t0 = time.perf_counter()
            got = dfdb.nrow_streamed(v, chunk)
            dt = time.perf_counter() - t0
        assert got == want
        print(json.dumps({"config": "stream-count", "chunk_blocks": chunk, "rows": n, "seconds": dt, "rows_per_s": n / dt,
                          "decoded_GBps": n * 8 / dt / 1e9, "compressed_GBps": st["compressed"] / dt / 1e9}))
    t0 = time.perf_counter()
    tr = dfdb.open_table(os.path.join(d, "tb"))
    got = dfdb.nrow(tr[("x", lambda x: x > 899_999), dfdb.ALL])
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "load-all-then-count", "rows": n, "seconds": dt, "rows_per_s": n / dt}))
    tr.close()
finally:
    shutil.rmtree(d, ignore_errors=True)
