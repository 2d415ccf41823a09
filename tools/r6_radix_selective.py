#!/usr/bin/env python3
"""unique over a FILTERED view (1e9 rows, 1e6 distinct values, a predicate that keeps 50 / 10 / 3.3 %): the radix form against the hash table"""
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
ctx.set_option("unique_dense", 0)
for thr, label in ((499_999, "50 %"), (899_999, "10 %"), (966_666, "3.3 %"), (989_999, "1 %"), (996_999, "0.3 %")):
    v = t[("a", lambda c, thr=thr: c > thr), ["x"]]
    for radix in (1, 0, 1, 0):
        ctx.set_option("unique_radix", radix)
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u = v.x.unique()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        keys = ("radix_sample", "radix_partition", "radix_unique", "unique_insert", "unique", "unique_radix.taken", "scan_cmp")
        p = {k: ctx.profile_get(k) for k in keys}
        ctx.profile(False)
        print(label, "radix" if radix else "hash ", "ms %.3f" % (dt * 1e3), "distinct", len(u), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
ctx.set_option("unique_radix", 1); ctx.set_option("unique_dense", 1)
