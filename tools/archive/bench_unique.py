#!/usr/bin/env python3
"""unique(col) on the device: first occurrences as a selection (k_unique.hip).  Reference: 7-11 MRows/s (docs/src/index.md:479-486)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
S = 0x9E3779B97F4A7C15
ctx = dfdb.default_context(0)
for name, gen, n in (("Int64 mod 1e6 (1e6 distinct)", dfdb.GEN_I64_MOD1M, 1_000_000_000), ("String brands (10 distinct)", dfdb.GEN_STR_BRANDS10, 500_000_000),
                     ("Float64 (all distinct)", dfdb.GEN_F64_U2000, 200_000_000)):
    t = dfdb.DFTable.new()
    t.add_generated("c", gen, S, n)
    col = t.c
    best = None
    for _ in range(3):
        ctx.synchronize(); t0 = time.perf_counter(); u = col.unique(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ctx.profile(True); col.unique(); k, ms = ctx.profile_get("unique"); ctx.profile(False)
    print(json.dumps({"config": "unique", "column": name, "rows": n, "distinct": len(u), "seconds": best, "rows_per_s": n / best, "unique_kernels_ms": ms}))
    t.close()
