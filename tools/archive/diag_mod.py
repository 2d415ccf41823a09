#!/usr/bin/env python3
"""Time a few interpreter predicates on 1e9 rows (HIP-event kernel times): python tools/diag_mod.py [rows]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new()
t.add_generated("a", dfdb.GEN_I64_MOD1M, 1, n)
t.add_generated("x", dfdb.GEN_F64_U2000, 2, n)
t.add_generated("b", dfdb.GEN_I64_MOD1M, 3, n)
preds = {"a % 50 == 0": lambda: t.a % 50 == 0, "a * 2 + 1 > 1000000": lambda: t.a * 2 + 1 > 1_000_000, "x * 1.0 < 632.456": lambda: t.x * 1.0 < 632.456,
         "a / 50 > 10000.5": lambda: t.a / 50 > 10000.5, "(a > 5e5) | (x < 100)": lambda: (t.a > 500_000) | (t.x < 100.0), "a > b": lambda: t.a > t.b, "a == b": lambda: t.a == t.b,
         "(a % 50 == 0) & (b < 930000)": lambda: (t.a % 50 == 0) & (t.b < 930_000)}
for name, mk in preds.items():
    q = t[mk(), dfdb.ALL]._query()
    q.execute(); ctx.synchronize()
    ctx.profile(True)
    for _ in range(3):
        q.reset(); q.execute()
    cnt = q.count()
    ks = {k: ctx.profile_get(k) for k in ("interp_predicate", "scan_cmp", "scan_terms")}
    ctx.profile(False)
    print(json.dumps({"predicate": name, "selected": cnt, "kernels_ms": {k: round(v[1] / v[0], 4) for k, v in ks.items() if v[0]}}))
