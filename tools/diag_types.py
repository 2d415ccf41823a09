#!/usr/bin/env python3
"""K1 / interpreter kernel times by column type (1e8 rows): python tools/diag_types.py [rows]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch  # noqa
import dfdb
from dfdb import ir

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ctx = dfdb.default_context(0)
rng = np.random.default_rng(0)
base = rng.integers(0, 1_000_000, n)
cols = {"i64": base.astype(np.int64), "i32": base.astype(np.int32), "i16": (base % 30000).astype(np.int16), "i8": (base % 100).astype(np.int8),
        "f32": (base / 500.0).astype(np.float32), "f64": base / 500.0, "b": (base % 10 == 0),
        "m": np.ma.masked_array(base.astype(np.int64), mask=(base % 7 == 0))}
t = dfdb.DFTable.from_columns(cols)
preds = {"i64 > c": lambda: t.i64 > 899_999, "i32 > c": lambda: t.i32 > 899_999, "i16 > c": lambda: t.i16 > 27_000, "i8 > c": lambda: t.i8 > 89,
         "f32 < c": lambda: t.f32 < 200.0, "f64 < c": lambda: t.f64 < 200.0, "b": lambda: t.b, "i32 > 899999.5 (float const)": lambda: t.i32 > 899_999.5,
         "ismissing(m)": lambda: dfdb.ismissing(t.m), "coalesce(m, 0) > c": lambda: dfdb.coalesce(t.m, 0) > 899_999,
         "(i64 > c) & (i32 < c2) & (f32 < c3)": lambda: (t.i64 > 500_000) & (t.i32 < 800_000) & (t.f32 < 1500.0)}
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
