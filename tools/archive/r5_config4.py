#!/usr/bin/env python3
"""bench.py's config 4 step alone (5e8 rows, s == "sony" at 10 %, materialize [s, a] into device buffers), flat and with the dictionary: for rocprofv3 --kernel-trace"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, os.environ.get("DFDB_PKG", "dataframedbs.jl_amd"))):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
from dfdb import _native as N
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = dfdb.default_context(0)
SEED = 0x9E3779B97F4A7C15
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("s", dfdb.GEN_STR_BRANDS10, SEED, rows)
t.add_generated("a", dfdb.GEN_I64_MOD1M, (SEED * 2) & 0xFFFFFFFFFFFFFFFF, rows)
q = t[t.s == "sony", dfdb.ALL]._query()
q.hint_materialize(True)
nsel = q.count()
lib = N.load()
nb = C.c_int64()
N.check(lib.dfdb_result_string_bytes(q._h, 0, C.byref(nb)))
dev = torch.device("cuda:0")
osz = torch.empty(max(nsel, 1), dtype=torch.int32, device=dev); oby = torch.empty(nb.value + 64, dtype=torch.uint8, device=dev); oa = torch.empty(max(nsel, 1), dtype=torch.int64, device=dev)
outs = (N.OutCol * 2)()
outs[0].data, outs[0].bytes, outs[0].bytes_cap, outs[0].memkind = osz.data_ptr(), oby.data_ptr(), nb.value, N.MEM_DEVICE
outs[1].data, outs[1].memkind = oa.data_ptr(), N.MEM_DEVICE
for leg in ("4", "4_dictionary"):
    if leg.endswith("dictionary"):
        t.build_dictionary("s")
    best = None
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q.execute(); N.check(lib.dfdb_materialize(q._h, outs, 2)); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(json.dumps({"config": leg, "rows": rows, "selected": nsel, "best_ms": round(best * 1e3, 3)}), flush=True)
    print("---- end of leg", leg, file=sys.stderr, flush=True)
