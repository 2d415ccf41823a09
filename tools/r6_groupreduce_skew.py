#!/usr/bin/env python3
"""groupreduce over a SKEWED key at 1e9 rows: 30 % of the rows hold key 0, the rest 1e5 keys evenly — what the forms cost there"""
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
t.add_column_from("ks", (t.x % 100000) * (t.x > 299999) + (1 << 40))
keys = ("unique_insert", "unique", "group_accumulate", "radix_sample", "radix_partition", "radix_group", "group_radix.taken", "group_radix.skewed", "group_radix.fell_back")
for radix in (1, 0, 1, 0):
    ctx.set_option("unique_radix", radix)
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = dfdb.groupreduce(t, "ks", "a", "sum")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = {k: ctx.profile_get(k) for k in keys}
    ctx.profile(False)
    print("radix" if radix else "old  ", "ms %.3f" % (dt * 1e3), len(g), int(g["count"].max()), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
ctx.set_option("unique_radix", 1)
