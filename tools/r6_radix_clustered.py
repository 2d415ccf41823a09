#!/usr/bin/env python3
"""unique / groupreduce over a CLUSTERED selection (the first 5 % of 1e9 rows: a range predicate over rows in order): most tiles of the partition pass hold no selected row"""
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
t.add_generated("i", dfdb.GEN_I64_IOTA, 0, n)
ctx.set_option("unique_dense", 0)
keys = ("unique_insert", "unique", "radix_sample", "radix_partition", "radix_unique", "radix_group", "unique_radix.taken", "group_radix.taken", "scan_cmp", "group_accumulate")
v = t[("i", lambda c: c < 50_000_000), ["x", "a"]]
for what in ("unique", "unique", "groupreduce", "groupreduce"):
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = v.x.unique() if what == "unique" else dfdb.groupreduce(v, "x", "a", "sum")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = {k: ctx.profile_get(k) for k in keys}
    ctx.profile(False)
    print(what, "over the first 5 %% of the rows: ms %.3f" % (dt * 1e3), len(r), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
ctx.set_option("unique_dense", 1)
