#!/usr/bin/env python3
"""round 4 experiment: a single-column comparison through the run-time compiled expression kernel (`abs(a) > c`: one load, one abs, one compare per row) against the
hand-written K1 (`a > c`) and the two-column forms against k_scan_pair, same process, 1e9 rows.  python tools/r4_jit_vs_k1.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
from dfdb import ir
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = dfdb.default_context(0)
ctx.set_option("jit", 2)
t = dfdb.DFTable.new(ctx=ctx)
t.add_generated("a", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
t.add_generated("b", dfdb.GEN_I64_MOD1M, 2, n)
a, b = ir.col(0), ir.col(1)
cases = [("K1  a > c", a > 899_999), ("JIT abs(a) > c", abs(a) > 899_999), ("pair (a > c1) & (b < c2)", (a > 683_771) & (b < 316_228)), ("JIT (abs(a) > c1) & (abs(b) < c2)", (abs(a) > 683_771) & (abs(b) < 316_228)),
         ("K1  a > c", a > 899_999), ("JIT abs(a) > c", abs(a) > 899_999)]
for name, pred in cases:
    q = t[pred, dfdb.ALL]._query(); cnt = q.count()
    ctx.profile(True)
    for _ in range(10):
        q.reset(); q.execute()
    ctx.synchronize()
    ks = {k: ctx.profile_get(k) for k in ("scan_cmp", "scan_terms", "jit_predicate", "interp_predicate")}
    ctx.profile(False)
    print(json.dumps({"case": name, "selected": cnt, "ms": {k: round(v[1] / v[0], 4) for k, v in ks.items() if v[0]}}), flush=True)
