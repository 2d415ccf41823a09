#!/usr/bin/env python3
"""unique / groupreduce over a Float64 key column with FEW distinct values (the hash-table form: floats never take the dense form), 1e9 rows"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, os.environ.get("DFDB_PKG", "dataframedbs.jl_amd"))):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
for m in (7, 5000):
    t.add_column_from("f%d" % m, (t.x % m) * 0.5)
for m in (7, 5000):
    for name, fn in (("unique", lambda: getattr(t, "f%d" % m).unique()), ("groupreduce", lambda: dfdb.groupreduce(t, "f%d" % m, "x", "sum"))):
        best = None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(json.dumps({"values": m, "call": name, "best_ms": round(best * 1e3, 3), "n": len(r)}), flush=True)
