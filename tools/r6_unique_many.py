#!/usr/bin/env python3
"""unique(col) with MANY distinct values at 1e9 rows (1e7, 1e8): beyond what the radix form's LDS tables hold (~5.6 M) — what the hash table costs there"""
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
t.add_generated("a", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15 * 2, n)
t.add_column_from("k7", t.x * 10 + t.a % 10 + (1 << 40))
t.add_column_from("k8", t.x * 100 + t.a % 100 + (1 << 40))
keys = ("unique_insert", "unique_mark", "unique_migrate", "unique", "radix_partition", "radix_unique", "unique_radix.taken", "gather")
for col in ("k7", "k8"):
    for rep in range(2):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u = getattr(t, col).unique()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        p = {k: ctx.profile_get(k) for k in keys}
        ctx.profile(False)
        print(col, "ms %.3f" % (dt * 1e3), len(u), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
