#!/usr/bin/env python3
"""round 4: the four benchmark expressions through the device interpreter (ctx option jit = 0) and through their hipRTC-compiled kernels (jit = 2), a few launches
each: run it under `rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES --kernel-trace` for instructions per kernel (k_interp<...> vs dfdb_jit_kernel).
    python tools/r4_interp.py [--rows 1000000000] [--reps 3]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import dfdb  # noqa: E402
from dfdb import ir  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=1_000_000_000); ap.add_argument("--reps", type=int, default=3)
a_ = ap.parse_args()
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(ctx=ctx)
t.add_generated("a", dfdb.GEN_I64_MOD1M, 1, a_.rows)
t.add_generated("b", dfdb.GEN_I64_MOD1M, 2, a_.rows)
t.add_generated("x", dfdb.GEN_F64_U2000, 3, a_.rows)
a, b, x = ir.col(0), ir.col(1), ir.col(2)
cases = {"a*3 + b*2 - 7 > 4e6": a * 3 + b * 2 - 7 > 4_000_000, "(a > b) | (x*2 > a)": (a > b) | (x * 2 > a), "(a + b) * x > 3e9": (a + b) * x > 3e9, "a + b > 1.8e6": a + b > 1_800_000}
out = {}
for name, pred in cases.items():
    for jit, key, kern in ((0, "interpreter", "interp_predicate"), (2, "compiled", "jit_predicate")):
        ctx.set_option("jit", jit)
        q = t[pred, dfdb.ALL]._query(); n = q.count()
        ctx.profile(True)
        for _ in range(a_.reps):
            q.reset(); q.execute()
        ctx.synchronize()
        nl, ms = ctx.profile_get(kern)
        ctx.profile(False)
        out.setdefault(name, {"selected": n})[key] = round(ms / nl, 4) if nl else None
ctx.set_option("jit", 1)
print(json.dumps(out))
