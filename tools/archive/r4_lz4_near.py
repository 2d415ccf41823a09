#!/usr/bin/env python3
"""round 4: what the compressor's preference for near matches (ctx option lz4_enc_near) does to the file size and to K7: for each setting the 1e9-row
benchmark column is written by the device encoder, loaded with its LZ4 blocks kept in HBM, and decoded again without and with the sequence-start index.
python tools/r4_lz4_near.py [--rows 1e9]"""
import argparse, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=float, default=1e9)
ap.add_argument("--gen", default="mod1m")
args = ap.parse_args()
n = int(args.rows)
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(ctx=ctx)
if args.gen == "mod1m":
    t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
elif args.gen == "f64":
    t.add_generated("x", dfdb.GEN_F64_U2000, 0x9E3779B97F4A7C15, n)
else:
    t.add_generated("x", dfdb.GEN_I64_IOTA, 0, n)
want_sum = t.x.sum()
for near in (0, 1984, 1024, 4032, 8000):
    ctx.set_option("lz4_enc_near", near)
    d = tempfile.mkdtemp(dir="/dev/shm")
    try:
        t0 = time.perf_counter()
        st = t.save(os.path.join(d, "tb"))
        save_s = time.perf_counter() - t0
        ctx.set_option("keep_compressed", 1)
        t2 = dfdb.open_table(os.path.join(d, "tb"), ctx=ctx, load=False)
        t2.load()
        ctx.set_option("keep_compressed", 0)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    rec = {"lz4_enc_near": near, "rows": n, "ratio": round(st["uncompressed"] / st["compressed"], 4), "compressed_GB": round(st["compressed"] / 1e9, 3), "save_s": round(save_s, 3)}
    for idx in (0, 1):
        ctx.set_option("lz4_index", idx)
        t2.decode_resident("x"); ctx.synchronize()          # (with the index: this one records it)
        ctx.profile(True)
        for _ in range(5):
            t2.decode_resident("x")
        ctx.synchronize()
        nl, ms = ctx.profile_get("lz4_decode")
        ctx.profile(False)
        rec["indexed" if idx else "plain"] = {"ms": round(ms / nl, 3), "decoded_GBps": round(n * 8 / (ms / nl * 1e-3) / 1e9, 1)}
    ctx.set_option("lz4_index", 1)
    assert t2.decode_status("x") == 0
    assert t2.x.sum() == want_sum, "the decoded column is not the column that was written"
    t2.close()
    print(json.dumps(rec), flush=True)
