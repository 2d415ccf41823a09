#!/usr/bin/env python3
"""per-kernel device ms of groupreduce by an Int64 / Float64 key at 1e9 rows, 5000 groups (bench.py's legs)"""
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
t.add_column_from("fk", (t.x % 5000) * 0.5)
t.add_column_from("k", t.x % 5000)
keys = ("unique_insert", "unique_mark", "unique_migrate", "unique", "unique_first", "unique_minmax", "unique_presence", "group_accumulate", "reduce", "gather", "scan_counts", "scan_terms", "scan_cmp")
LEGS = (("Float64 key, 5000 groups", "fk"), ("Int64 key, 5000 groups", "k"))
if os.environ.get("DFDB_FLOAT_ONLY"): LEGS = LEGS[:1]
for label, key in LEGS:
    for rep in range(3):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = dfdb.groupreduce(t, key, "x", "sum")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        p = {k: ctx.profile_get(k) for k in keys}
        ctx.profile(False)
        print(label, "ms %.3f" % (dt * 1e3), len(g), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
if os.environ.get("DFDB_FLOAT_ONLY"): sys.exit(0)
t.close()
# by a 10-value String key, flat and with the dictionary (5e8 rows)
n = 500_000_000
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("s", dfdb.GEN_STR_BRANDS10, 0x9E3779B97F4A7C15, n)
t.add_generated("a", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15 * 2, n)
keys = keys + ("dict_scan", "str_match")
for label in ("String key (flat), 10 groups", "String key (dictionary), 10 groups"):
    if "dictionary" in label:
        t.build_dictionary("s")
    for rep in range(3):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = dfdb.groupreduce(t, "s", "a", "sum")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        p = {k: ctx.profile_get(k) for k in keys}
        ctx.profile(False)
        print(label, "ms %.3f" % (dt * 1e3), len(g), {k: (v2[0], round(v2[1], 3)) for k, v2 in p.items() if v2[0]}, flush=True)
