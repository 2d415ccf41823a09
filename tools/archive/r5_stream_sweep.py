#!/usr/bin/env python3
"""block-streamed count of a 1e9-row Int64 column from /dev/shm: file GB/s for several (stream_readers, io_threads) — the host's pread rate peaks near 16 threads
and FALLS beyond (tools/bench_pread.cpp), so fewer concurrent readers can mean more bytes per second"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch
torch.cuda.init()
import dfdb
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = dfdb.default_context(0)
d = tempfile.mkdtemp(dir="/dev/shm")
try:
    t = dfdb.DFTable.new()
    t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
    st = t.save(os.path.join(d, "tb")); t.close()
    tb = dfdb.open_table(os.path.join(d, "tb"), load=False)
    v = tb[("x", lambda x: x > 899_999), dfdb.ALL]
    dfdb.nrow_streamed(v, 1024)
    for readers, io, piece in ((3, 8, 32), (3, 8, 64), (3, 8, 128), (2, 8, 64), (2, 8, 128), (1, 8, 64), (1, 8, 128), (1, 16, 128), (2, 6, 96), (3, 8, 32), (3, 8, 64)):
        ctx.set_option("stream_readers", readers); ctx.set_option("io_threads", io); ctx.set_option("stream_piece_mb", piece)
        best = None
        for rep in range(3):
            t0 = time.perf_counter(); got = dfdb.nrow_streamed(v, 1024); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(json.dumps({"stream_readers": readers, "io_threads": io, "piece_mb": piece, "best_s": round(best, 4), "file_GBps": round(st["compressed"] / best / 1e9, 1)}), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
