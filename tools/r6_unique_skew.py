#!/usr/bin/env python3
"""unique(col) over a SKEWED column at 1e9 rows (30 % of the rows hold one value, the rest 1e6 values evenly): the radix form leaves it to the hash table — what that costs"""
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
t.add_column_from("ks", t.a * (t.x > 299999) + (1 << 40))
ctx.set_option("unique_dense", 0)
keys = ("unique_insert", "unique_mark", "unique_migrate", "unique", "radix_sample", "radix_partition", "radix_unique", "unique_radix.taken", "unique_radix.skewed")
for rep in range(3):
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u = t.ks.unique()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = {k: ctx.profile_get(k) for k in keys}
    ctx.profile(False)
    print("skewed unique ms %.3f" % (dt * 1e3), len(u), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
ctx.set_option("unique_dense", 1)
