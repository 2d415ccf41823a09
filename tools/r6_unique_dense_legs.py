#!/usr/bin/env python3
"""per-kernel device ms of unique(col) in its dense form at 1e9 rows / 1e6 values (bench.py's `unique` leg)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")]
import torch
torch.cuda.init()
import dfdb
n = 1_000_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
keys = ("unique_insert", "unique_mark", "unique", "unique_first", "unique_minmax", "unique_presence", "gather", "scan_counts", "fill_ones", "compact_indices")
for rep in range(3):
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u = t.x.unique()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = {k: ctx.profile_get(k) for k in keys}
    ctx.profile(False)
    print("dense unique ms %.3f" % (dt * 1e3), len(u), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
