#!/usr/bin/env python3
"""the skewed column of tools/r6_groupreduce_skew.py through the OLD form only (ctx option unique_radix = 0), for a kernel trace"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")]
import torch
torch.cuda.init()
import dfdb
n = 200_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
t.add_generated("a", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15 * 2, n)
t.add_column_from("ks", (t.x % 100000) * (t.x > 299999) + (1 << 40))
ctx.set_option("unique_radix", 0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = dfdb.groupreduce(t, "ks", "a", "sum")
    torch.cuda.synchronize(); print("old form, 2e8 rows: ms %.1f" % ((time.perf_counter() - t0) * 1e3), len(g), flush=True)
