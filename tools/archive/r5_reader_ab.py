import os
#!/usr/bin/env python3
"""the same 1e9-row Int64 column file on the same box, DFDB_STREAM_DEBUG=1: dfdb_table_load's reader against the block stream's (one reading turn, three turns,
one turn with 16 pread threads) — per-piece read times go to stderr, one JSON line per run to stdout"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
ctx = dfdb.default_context(0)
d = tempfile.mkdtemp(dir="/dev/shm")
try:
    t = dfdb.DFTable.new()
    t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, 1_000_000_000)
    if "three" in sys.argv:                              # (bench.py's cold table: three columns, 13 GB of files in the page cache)
        t.add_generated("i", dfdb.GEN_I64_IOTA, 0, 1_000_000_000)
        t.add_generated("b", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C16, 1_000_000_000)
    st = t.save(os.path.join(d, "tb")); t.close()
    xbytes = os.path.getsize(os.path.join(d, "tb", "1.bin"))
    for rep in range(int(os.environ.get('DFDB_AB_REPEATS', '5'))):
        for what, readers, io, chunk in (("load", 0, 8, 1024), ("stream", 3, 8, 1024), ("stream", 2, 8, 1024), ("stream", 1, 8, 1024)):
            ctx.set_option("io_threads", io)
            print(f"---- {what} readers={readers} io={io} rep={rep}", file=sys.stderr, flush=True)
            tb = dfdb.open_table(os.path.join(d, "tb"), load=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if what == "load":
                tb.load(["x"]); torch.cuda.synchronize()
            else:
                ctx.set_option("stream_slots", 8); ctx.set_option("stream_readers", readers)
                dfdb.nrow_streamed(tb[("x", lambda x: x > 899_999), dfdb.ALL], chunk)
            dt = time.perf_counter() - t0
            print(json.dumps({"what": what, "readers": readers, "io_threads": io, "chunk_blocks": chunk, "rep": rep, "seconds": round(dt, 4), "file_GBps": round(xbytes / dt / 1e9, 1)}), flush=True)
            tb.close()
finally:
    shutil.rmtree(d, ignore_errors=True)
