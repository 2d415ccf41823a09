#!/usr/bin/env python3
"""K1 by column type at full size: `col OP const` over 1-, 2-, 4- and 8-byte columns of N rows made on the device (casts of the generated Int64 column,
dfdb_table_add_from_query), the kernel's HIP-event average, its algorithmic GB/s (width + 1/8 + 4/1024 bytes per row) and fraction of the HBM peak;
ctx option scan_narrow = 0 beside it (the one-element-per-lane kernel).   python tools/diag_types.py [rows=1e9]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch  # noqa
import dfdb
from dfdb import ir

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = dfdb.default_context(0)
peak = float(ctx.device_info().get("peak_hbm_gbps") or 8000.0)
t = dfdb.DFTable.new(ctx=ctx)
t.add_generated("i64", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
t.add_generated("f64", dfdb.GEN_F64_U2000, 0x1234, n)
def add(name, expr):
    v = dfdb.DFView(t, dfdb.Projection({name: expr}), dfdb.DFView(t).selection)
    t.add_column_from(name, v)
a, x = ir.col(0), ir.col(1)
add("i32", ir.cast(a, ir.I32))
add("u32", ir.cast(a, ir.U32))
add("f32", ir.cast(x, ir.F32))
add("i16", ir.cast(a % 30000, ir.I16))
add("u16", ir.cast(a % 60000, ir.U16))
add("i8", ir.cast(a % 100, ir.I8))
add("u8", ir.cast(a % 200, ir.U8))
add("b", a % 10 == 0)
width = {"i64": 8, "f64": 8, "i32": 4, "u32": 4, "f32": 4, "i16": 2, "u16": 2, "i8": 1, "u8": 1, "b": 1}
preds = [("i64 > 899999", "i64", lambda: t.i64 > 899_999), ("f64 < 200.0", "f64", lambda: t.f64 < 200.0), ("i32 > 899999", "i32", lambda: t.i32 > 899_999),
         ("u32 > 899999", "u32", lambda: t.u32 > 899_999), ("f32 < 200.0", "f32", lambda: t.f32 < 200.0), ("i16 > 27000", "i16", lambda: t.i16 > 27_000),
         ("u16 >= 54000", "u16", lambda: t.u16 >= 54_000), ("i8 > 89", "i8", lambda: t.i8 > 89), ("u8 == 7", "u8", lambda: t.u8 == 7), ("b (Bool column)", "b", lambda: t.b),
         ("i32 != 5", "i32", lambda: t.i32 != 5)]
for name, col, mk in preds:
    rec = {"predicate": name, "rows": n, "bytes_per_row": width[col] + 1 / 8 + 4 / 1024}
    for narrow in (2, 0):
        if narrow == 0 and width[col] == 8:
            continue
        ctx.set_option("scan_narrow", narrow)
        q = t[mk(), dfdb.ALL]._query()
        q.execute(); ctx.synchronize()
        ctx.profile(True)
        for _ in range(5):
            q.reset(); q.execute()
        cnt = q.count()
        ks = {k: ctx.profile_get(k) for k in ("interp_predicate", "scan_cmp", "scan_terms")}
        ctx.profile(False)
        ms = {k: v[1] / v[0] for k, v in ks.items() if v[0]}
        (kname, kms), = ms.items() if len(ms) == 1 else [max(ms.items(), key=lambda kv: kv[1])]
        gbps = n * rec["bytes_per_row"] / (kms * 1e-3) / 1e9
        key = "narrow" if (narrow and width[col] < 8) else ("one_per_lane" if width[col] < 8 else "k_scan_cmp")
        rec[key] = {"kernel": kname, "ms": round(kms, 4), "GBps": round(gbps, 1), "frac_of_peak": round(gbps / peak, 3), "selected": cnt}
    ctx.set_option("scan_narrow", 1)
    print(json.dumps(rec), flush=True)
