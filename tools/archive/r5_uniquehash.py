#!/usr/bin/env python3
"""bench.py's unique_hash_table leg alone (1e9 Int64 rows, 1e6 distinct values, the dense form switched off)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, os.environ.get("DFDB_PKG", "dataframedbs.jl_amd"))):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
n = 1_000_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
ctx.set_option("unique_dense", 0)
best = None
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); u = t.x.unique(); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(json.dumps({"leg": "unique_hash_table", "best_ms": round(best * 1e3, 3), "n": len(u)}), flush=True)
