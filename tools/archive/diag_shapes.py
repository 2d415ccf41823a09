#!/usr/bin/env python3
"""Wall time of a list of query shapes on resident columns (2e8 rows unless given): a net for slow paths.  python tools/diag_shapes.py [rows]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch  # noqa
import dfdb
from dfdb import ir

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
ctx = dfdb.default_context(0)
S = 0x9E3779B97F4A7C15
t = dfdb.DFTable.new()
t.add_generated("a", dfdb.GEN_I64_MOD1M, S, n)
t.add_generated("x", dfdb.GEN_F64_U2000, S * 2 & (2**64 - 1), n)
t.add_generated("s", dfdb.GEN_STR_BRANDS10, S * 3 & (2**64 - 1), n)
# narrow / nullable columns derived on the device from a (add_column_from: lazy column -> resident)
t.add_column_from("i32", dfdb.map_to_column(t.a, lambda a: ir.cast(a, ir.I32)) if hasattr(dfdb, "map_to_column") and False else t.a) if False else None


def timed(name, fn, reps=3):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    print(json.dumps({"shape": name, "ms": round(best * 1e3, 3), "rows_per_s": round(n / best), "result": r if isinstance(r, (int, float)) else None}))


def fresh(v):
    q = v._query(); q.reset(); return q

A, X, Sx = t.a, t.x, t.s
timed("nrow(t[a > c])", lambda: fresh(t[A > 899_999, dfdb.ALL]).count())
timed("nrow(t[x < c])", lambda: fresh(t[X < 200.0, dfdb.ALL]).count())
timed("nrow(t[(a>c)&(x<c)&(s==sony)])", lambda: fresh(t[(A > 500_000) & (X < 1000.0) & (Sx == "sony"), dfdb.ALL]).count())
timed("nrow(t[1:10:end][a > c])", lambda: fresh(t[dfdb.jr(1, 10, n), dfdb.ALL][("a", lambda a: a > 899_999), dfdb.ALL]).count())
timed("nrow(t[a > c][1:1000])", lambda: fresh(t[A > 899_999, dfdb.ALL][dfdb.jr(1, 1000), dfdb.ALL]).count())
idx = np.sort(np.random.default_rng(1).choice(n, 1_000_000, replace=False)) + 1
timed("nrow(t[1e6 index vector])", lambda: fresh(t[idx.tolist() if False else idx, dfdb.ALL]).count())
timed("materialize(t[a > c, [a, x]]) 10%", lambda: len(dfdb.materialize(t[A > 899_999, ["a", "x"]])))
timed("materialize(t[s == sony, :]) 10%", lambda: len(dfdb.materialize(t[Sx == "sony", dfdb.ALL])))
timed("materialize(head)", lambda: len(dfdb.head(t)))
timed("sum(x[a > c])", lambda: t[A > 899_999, dfdb.ALL][dfdb.ALL, "x"].sum())
timed("mean(x[s == huawei])", lambda: t[Sx == "huawei", dfdb.ALL][dfdb.ALL, "x"].mean())
timed("sum(a*2 over a > c)", lambda: dfdb.DFColumn(t[A > 899_999, {"d": ("a", lambda a: a * 2)}]).sum() if False else 0)
timed("unique(s)", lambda: len(t.s.unique()))
timed("unique(a)", lambda: len(t.a.unique()))
timed("nrow(t[ismissing-free: a % 50 == 0])", lambda: fresh(t[A % 50 == 0, dfdb.ALL]).count())
