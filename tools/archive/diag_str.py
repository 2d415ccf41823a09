#!/usr/bin/env python3
"""K5 timings by pattern length / operator on the 10-brand column (5e8 rows): python tools/diag_str.py [rows]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new()
t.add_generated("s", dfdb.GEN_STR_BRANDS10, 0x9E3779B97F4A7C15, n)
preds = {'s == "sony"': lambda: t.s == "sony", 's == "samsung"': lambda: t.s == "samsung", 's == "microsoft"': lambda: t.s == "microsoft",
         's != "microsoft"': lambda: t.s != "microsoft", 'startswith(s, "micro")': lambda: dfdb.startswith(t.s, "micro"),
         'startswith(s, "microsoft")': lambda: dfdb.startswith(t.s, "microsoft"), 'endswith(s, "soft")': lambda: dfdb.endswith(t.s, "soft")}
for name, mk in preds.items():
    q = t[mk(), dfdb.ALL]._query()
    q.execute(); ctx.synchronize()
    ctx.profile(True)
    for _ in range(3):
        q.reset(); q.execute()
    cnt = q.count()
    nl, ms = ctx.profile_get("str_match")
    ctx.profile(False)
    print(json.dumps({"predicate": name, "selected": cnt, "str_match_ms": round(ms / nl, 4)}))
