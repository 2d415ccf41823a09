#!/usr/bin/env python3
"""unique over 1e9 Int64 rows with 1e6 distinct values through the general hash table: the XCD-partitioned insert pass on / off.
(The record of an experiment: ctx option unique_xcd_parts and the kernel forms it selected were measured — 34-68 ms against 17.9 — and removed)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch  # noqa
import dfdb
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
ref = t.x.unique()
ctx.set_option("unique_dense", 0)
for xp in (1, 0, 1, 0):
    ctx.set_option("unique_xcd_parts", xp)
    for rep in range(2):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u = t.x.unique()
        ms = (time.perf_counter() - t0) * 1e3
        ks = {k: ctx.profile_get(k) for k in ("unique_insert", "unique_migrate", "unique_mark")}
        ctx.profile(False)
    print(json.dumps({"xcd_parts": xp, "wall_ms": round(ms, 2), "same_as_dense": bool(np.array_equal(np.asarray(u), np.asarray(ref))), "kernels": {k: [v[0], round(v[1], 2)] for k, v in ks.items()}}), flush=True)
