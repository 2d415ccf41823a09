#!/usr/bin/env python3
"""round 4: block-streamed count over one Int64 column that is never resident, swept over ctx options stream_slots / io_threads and chunk sizes.
python tools/r4_stream_sweep.py [--rows 2e9]"""
import argparse, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=float, default=2e9)
args = ap.parse_args()
n = int(args.rows)
ctx = dfdb.default_context(0)
d = tempfile.mkdtemp(dir="/dev/shm")
try:
    t = dfdb.DFTable.new()
    t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
    st = t.save(os.path.join(d, "tb"))
    want = t[("x", lambda x: x > 899_999), dfdb.ALL]._query().count()
    t.close()
    tb = dfdb.open_table(os.path.join(d, "tb"), load=False)
    v = tb[("x", lambda x: x > 899_999), dfdb.ALL]
    cfgs = ((4, 3, 8, 1024), (6, 5, 8, 1024), (6, 3, 8, 1024), (6, 2, 8, 1024), (8, 3, 8, 1024), (8, 2, 8, 1024), (8, 4, 8, 1024), (8, 3, 16, 1024), (8, 2, 16, 1024), (8, 3, 8, 512), (8, 2, 8, 512), (8, 3, 8, 256))
    if os.environ.get("SWEEP_SHORT"):
        cfgs = ((6, 5, 8, 1024), (8, 3, 8, 1024), (8, 2, 8, 512))
    if os.environ.get("SWEEP_READERS"):
        cfgs = ((8, 2, 8, 1024), (8, 3, 8, 1024), (8, 4, 8, 1024), (8, 5, 8, 1024), (8, 7, 8, 1024), (8, 3, 12, 1024), (8, 4, 12, 1024), (8, 4, 6, 1024), (8, 5, 6, 1024), (8, 3, 8, 512), (8, 4, 8, 512), (8, 4, 8, 2048))
    for slots, readers, io, chunk in cfgs:
        ctx.set_option("stream_slots", slots); ctx.set_option("io_threads", io); ctx.set_option("stream_readers", readers)
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            got = dfdb.nrow_streamed(v, chunk)
            ts.append(time.perf_counter() - t0)
        best = min(ts); med = sorted(ts)[len(ts) // 2]
        assert got == want
        print(json.dumps({"median_file_GBps": round(st["compressed"] / med / 1e9, 2), "stream_slots": slots, "stream_readers": readers, "io_threads": io, "chunk_blocks": chunk, "rows": n, "seconds": best, "rows_per_s": n / best, "file_GBps": st["compressed"] / best / 1e9}), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
