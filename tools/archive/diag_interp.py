#!/usr/bin/env python3
"""Run the generic-interpreter form of config 3 a few times (target of rocprofv3 --pmc / --kernel-trace runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new()
t.add_generated("a", dfdb.GEN_I64_MOD1M, 1, n)
t.add_generated("x", dfdb.GEN_F64_U2000, 2, n)
v = t[(t.x * 1.0 < 632.456), dfdb.ALL]
q = v._query()
for _ in range(reps):
    q.reset(); q.execute()
ctx.synchronize()
print("selected", q.count())
