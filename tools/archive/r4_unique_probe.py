#!/usr/bin/env python3
"""round 4: where unique / groupreduce spend their time, per kernel family (ctx profile timers), dense form / hash table.  python tools/r4_unique_probe.py [rows]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
ctx = dfdb.default_context(0)
KS = ("unique", "unique_minmax", "unique_presence", "unique_first", "unique_insert", "unique_migrate", "unique_mark", "group_accumulate", "scan_counts", "gather")


def run(name, f, reps=3):
    for _ in range(reps):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f()
        ms = (time.perf_counter() - t0) * 1e3
        ctx.synchronize()
        ks = {k: ctx.profile_get(k) for k in KS}
        ctx.profile(False)
        print(json.dumps({"case": name, "result_len": len(r), "wall_ms": round(ms, 3), "kernels_ms": {k: [v[0], round(v[1], 3)] for k, v in ks.items() if v[0]}}), flush=True)


t = dfdb.DFTable.new(ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
t.add_generated("v", dfdb.GEN_I64_MOD1M, 7, n)
run("unique int64 mod 1e6, dense", lambda: t.x.unique())
run("groupreduce by int64 mod 1e6 (sum of v), dense", lambda: dfdb.groupreduce(t, "x", "v", "sum"), 2)
t.add_column_from("k5", t.x % 5000)
t.add_column_from("k900", t.x % 900)
run("groupreduce by x mod 5000 (sum of v): 144 KB of accumulators per CU", lambda: dfdb.groupreduce(t, "k5", "v", "sum"), 2)
run("groupreduce by x mod 900 (sum of v): 16 KB of accumulators per workgroup", lambda: dfdb.groupreduce(t, "k900", "v", "sum"), 2)
ctx.set_option("unique_dense", 0)
run("unique int64 mod 1e6, hash table", lambda: t.x.unique())
run("groupreduce by int64 mod 1e6 (sum of v), hash table", lambda: dfdb.groupreduce(t, "x", "v", "sum"), 2)
ctx.set_option("unique_dense", 1)
t.close()
n2 = n // 2
t = dfdb.DFTable.new(ctx=ctx)
t.add_generated("s", dfdb.GEN_STR_BRANDS10, 3, n2)
t.add_generated("a", dfdb.GEN_I64_MOD1M, 5, n2)
run("groupreduce by String (10 brands), sum of a", lambda: dfdb.groupreduce(t, "s", "a", "sum"))
run("unique String (10 brands)", lambda: t.s.unique(), 2)
t.close()
