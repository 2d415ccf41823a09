#!/usr/bin/env python3
"""timing experiments on the radix passes (DFDB_RADIX_XP set by the caller): per-pass device ms; results are NOT checked (they are wrong with a bit set)"""
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
ctx.set_option("unique_dense", 0)
ctx.set_option("unique_radix", 1)
for rep in range(2):
    ctx.profile(True)
    try:
        u = t.x.unique()
    except Exception as e:
        print("raised", type(e).__name__, str(e)[:80])
    p = {k: ctx.profile_get(k) for k in ("radix_sample", "radix_partition", "radix_unique")}
    ctx.profile(False)
print("XP", os.environ.get("DFDB_RADIX_XP"), {k: round(v[1] / max(v[0], 1), 3) for k, v in p.items()}, flush=True)
