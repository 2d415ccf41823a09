#!/usr/bin/env python3
"""groupreduce by a 10-value String key over 5e8 rows (bench.py's `groupreduce` leg alone), a few calls: what tools/r5_strpass_pmc.sh profiles.  argv: rows reps"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, os.environ.get("DFDB_PKG", "dataframedbs.jl_amd"))):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = dfdb.default_context(0)
use_dict = "dict" in sys.argv[3:]
for k in sys.argv[3:]:
    if "=" in k:
        a, b = k.split("="); ctx.set_option(a, int(b))
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("s", dfdb.GEN_STR_BRANDS10, 0x9E3779B97F4A7C15, n)
t.add_generated("a", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C16, n)
ctx.set_option("profile", 1) if "profile" in os.environ.get("DFDB_STRPASS", "") else None
if use_dict:
    t.build_dictionary("s")
best = None
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = dfdb.groupreduce(t, "s", "a", "sum")
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(json.dumps({"rows": n, "groups": len(g), "best_ms": round(best * 1e3, 3), "GBps": round(n * 17.4 / best / 1e9, 1), "counts": [int(x) for x in g["count"].to_numpy()[:3]]}), flush=True)
