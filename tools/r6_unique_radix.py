#!/usr/bin/env python3
"""unique over 1e9 rows / 1e6 distinct values: the hash table against the radix-partitioned form (k_radix.hip), per-pass device times.
usage: python tools/r6_unique_radix.py [rows]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")]
import torch  # noqa: E402
torch.cuda.init()
import dfdb  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
t.add_column_from("f", t.x * 0.5)
ctx.set_option("unique_dense", 0)
out = {"rows": n}
for col in ("x", "f"):
    for radix in (0, 1):
        ctx.set_option("unique_radix", radix)
        best, prof = None, {}
        for rep in range(3):
            ctx.profile(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            u = getattr(t, col).unique()
            dt = time.perf_counter() - t0
            p = {k: ctx.profile_get(k) for k in ("radix_sample", "radix_partition", "radix_unique", "unique_insert", "unique_mark", "unique_migrate", "unique", "unique_radix.taken", "unique_radix.fell_back", "gather", "scan_counts")}
            ctx.profile(False)
            if best is None or dt < best:
                best, prof = dt, {k: [v[0], round(v[1], 3)] for k, v in p.items() if v[0]}
        out[f"{col}.radix{radix}"] = {"ms": round(best * 1e3, 3), "distinct": int(len(u)), "frac_of_8TBps": round(n * 8 / best / 8e12, 4), "kernels": prof}
        print(col, radix, out[f"{col}.radix{radix}"], flush=True)
print(json.dumps(out))
