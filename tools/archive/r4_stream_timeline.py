#!/usr/bin/env python3
"""round 4: one block-streamed count with DFDB_STREAM_DEBUG=1 timelines (run with the env var set).  python tools/r4_stream_timeline.py [--rows 2e9] [--slots 6]"""
import argparse, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=float, default=2e9)
ap.add_argument("--slots", type=int, default=6)
ap.add_argument("--chunk", type=int, default=1024)
ap.add_argument("--readers", type=int, default=3)
args = ap.parse_args()
n = int(args.rows)
ctx = dfdb.default_context(0)
d = tempfile.mkdtemp(dir="/dev/shm")
try:
    t = dfdb.DFTable.new()
    t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
    st = t.save(os.path.join(d, "tb"))
    t.close()
    tb = dfdb.open_table(os.path.join(d, "tb"), load=False)
    v = tb[("x", lambda x: x > 899_999), dfdb.ALL]
    ctx.set_option("stream_slots", args.slots); ctx.set_option("stream_readers", args.readers)
    for rep in range(3):
        print(f"---- rep {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        got = dfdb.nrow_streamed(v, args.chunk)
        dt = time.perf_counter() - t0
        print(json.dumps({"rep": rep, "seconds": dt, "file_GBps": st["compressed"] / dt / 1e9}), file=sys.stderr, flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
