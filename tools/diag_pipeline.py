#!/usr/bin/env python3
"""Same-process A/B of the headline step: plain K1 + scan + K2 vs the pipelined pieces (ctx option "pipeline").  Wall ms per step."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new()
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
q = t[("x", lambda x: x > 899_999), dfdb.ALL]._query()
nsel = q.count()
out = torch.empty(nsel, dtype=torch.int64, device=dev)
res = {0: [], 1: []}
for rnd in range(6):
    for mode in (0, 1):
        ctx.set_option("pipeline", mode)
        for _ in range(3):
            q.reset(); q.indices_device(out.data_ptr(), nsel)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            q.reset(); q.indices_device(out.data_ptr(), nsel)
        ctx.synchronize()
        if rnd:
            res[mode].append((time.perf_counter() - t0) / 20 * 1e3)
ctx.set_option("pipeline", 0)
print(json.dumps({"rows": n, "selected": nsel, "ms_per_step_plain": [round(v, 4) for v in res[0]], "ms_per_step_pipelined": [round(v, 4) for v in res[1]]}))
