#!/usr/bin/env python3
"""groupreduce with MANY groups (1e9 rows; 1e6 groups: Int64 key x, Float64 key x * 0.5; 5e4 groups) — what the existing forms cost there"""
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
t.add_column_from("f", t.x * 0.5)
t.add_column_from("k50", t.x % 50000)
t.add_column_from("f50", (t.x % 50000) * 0.5)
if os.environ.get("DFDB_2M"): t.add_column_from("k2m", t.x * 2 + t.a % 2)
keys = ("unique_insert", "unique_mark", "unique_migrate", "unique", "unique_first", "unique_minmax", "unique_presence", "group_accumulate", "radix_sample", "radix_partition", "radix_unique", "radix_group", "group_radix.taken", "group_radix.fell_back", "group_radix.skewed", "gather", "scan_counts")
LEGS = (("Int64 key, 1e6 groups", "x"), ("Float64 key, 1e6 groups", "f"), ("Int64 key, 5e4 groups", "k50"), ("Float64 key, 5e4 groups", "f50"))
if os.environ.get("DFDB_ONE_LEG"): LEGS = LEGS[2:3]
if os.environ.get("DFDB_2M"): LEGS = (("Int64 key, 2e6 groups", "k2m"),)
for label, key in LEGS:
    for rep in range(2):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = dfdb.groupreduce(t, key, "a", "sum")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        p = {k: ctx.profile_get(k) for k in keys}
        ctx.profile(False)
        print(label, "ms %.3f" % (dt * 1e3), len(g), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
