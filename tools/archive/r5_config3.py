#!/usr/bin/env python3
"""bench.py's config 3 step alone (1e9 rows, (a > c1) & (x < c2) at 10 %, materialize [b, x] into device buffers), a few steps: for rocprofv3 --kernel-trace"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, os.environ.get("DFDB_PKG", "dataframedbs.jl_amd"))):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
from dfdb import _native as N
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = dfdb.default_context(0)
for k in sys.argv[3:]:
    if "=" in k:
        a, b = k.split("="); ctx.set_option(a, int(b))
SEED = 0x9E3779B97F4A7C15
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
for k, (name, gen) in enumerate((("a", dfdb.GEN_I64_MOD1M), ("b", dfdb.GEN_I64_MOD1M), ("x", dfdb.GEN_F64_U2000))):
    t.add_generated(name, gen, (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF, rows)
q = t[(t.a > 683_771) & (t.x < 632.456), ["b", "x"]]._query()
q.hint_materialize("nohint" not in sys.argv)
nsel = q.count()
dev = torch.device("cuda:0")
ob = torch.empty(max(nsel, 1), dtype=torch.int64, device=dev); ox = torch.empty(max(nsel, 1), dtype=torch.float64, device=dev)
outs = (N.OutCol * 2)()
outs[0].data, outs[0].memkind = ob.data_ptr(), N.MEM_DEVICE
outs[1].data, outs[1].memkind = ox.data_ptr(), N.MEM_DEVICE
lib = N.load()
best = None
for _ in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q.execute()
    if "noout" not in sys.argv: N.check(lib.dfdb_materialize(q._h, outs, 2))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(json.dumps({"config": 3, "rows": rows, "selected": nsel, "best_ms": round(best * 1e3, 3)}), flush=True)
