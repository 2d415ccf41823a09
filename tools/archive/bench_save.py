#!/usr/bin/env python3
"""save() of a generated Int64 column with both device LZ4 compressors: kernel time, ratio, wall.  python tools/bench_save.py [rows]"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 250_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new()
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    for enc in (0, 1, 0, 1):
        ctx.set_option("lz4_enc_variant", enc)
        ctx.profile(True)
        path = os.path.join(d, f"tb{enc}")
        shutil.rmtree(path, ignore_errors=True)
        t0 = time.perf_counter()
        st = t.save(path)
        wall = time.perf_counter() - t0
        nl, ms = ctx.profile_get("lz4_compress")
        ctx.profile(False)
        print(json.dumps({"enc_variant": enc, "rows": n, "lz4_compress_ms": ms, "compress_GBps_in": n * 8 / (ms * 1e-3) / 1e9,
                          "ratio": st["uncompressed"] / st["compressed"], "save_wall_s": wall, "save_GBps_in": n * 8 / wall / 1e9}))
finally:
    shutil.rmtree(d, ignore_errors=True)
