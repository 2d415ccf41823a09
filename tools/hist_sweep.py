#!/usr/bin/env python3
"""K7's history-ring form (compressed-only column, 1e9 Int64 rows): ms per launch of the fused decode + predicate for several numbers of rings (= workgroups),
fresh mask and AND-ed masks with / without the block skip.  Prints one JSON line per setting."""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, os.environ.get("DFDB_PKG", "dataframedbs.jl_amd"))):
    sys.path.insert(0, p)
import torch
torch.cuda.init()
import dfdb
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
waves_list = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4096, 3072, 2048, 8192]
SEED = 0x9E3779B97F4A7C15
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, rows)
d = tempfile.mkdtemp(prefix="dfdb_sweep_", dir="/dev/shm")
try:
    st = t.save(os.path.join(d, "tb"))
    t.close()
    ctx.set_option("keep_compressed", 2)
    t0 = time.perf_counter()
    tb = dfdb.open_table(os.path.join(d, "tb"), ctx=ctx)
    torch.cuda.synchronize()
    print(json.dumps({"load_s": time.perf_counter() - t0, "resident": tb.resident_bytes(), "file": st}), flush=True)
finally:
    ctx.set_option("keep_compressed", 0)
    shutil.rmtree(d, ignore_errors=True)
q = tb[("x", lambda x: x > 899_999), dfdb.ALL]._query()
want = q.count()
for w in waves_list:
    ctx.set_option("lz4_hist_waves", w)
    q.reset(); n = q.count()
    ctx.profile(True)
    for _ in range(5):
        q.reset(); q.execute()
    torch.cuda.synchronize()
    k, ms = ctx.profile_get("lz4_decode_scan_hist")
    ctx.profile(False)
    print(json.dumps({"waves": w, "ms": ms / k, "decoded_GBps": rows * 8 / (ms / k * 1e-3) / 1e9, "count_ok": n == want}), flush=True)
