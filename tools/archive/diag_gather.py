#!/usr/bin/env python3
"""config 3 at a given scale, a few materialize() calls (rocprofv3 --pmc target for the gather kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
from dfdb import _native as N
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
S = 0x9E3779B97F4A7C15
seed = lambda k: (S * (k + 1)) & 0xFFFFFFFFFFFFFFFF
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new()
t.add_generated("a", dfdb.GEN_I64_MOD1M, seed(0), n)
t.add_generated("b", dfdb.GEN_I64_MOD1M, seed(1), n)
t.add_generated("x", dfdb.GEN_F64_U2000, seed(2), n)
v = t[(t.a > 683_771) & (t.x < 632.456), ["b", "x"]]
q = v._query()
nsel = q.count()
dev = torch.device("cuda", 0)
ob = torch.empty(nsel, dtype=torch.int64, device=dev); ox = torch.empty(nsel, dtype=torch.float64, device=dev)
outs = (N.OutCol * 2)()
outs[0].data, outs[0].memkind = ob.data_ptr(), N.MEM_DEVICE
outs[1].data, outs[1].memkind = ox.data_ptr(), N.MEM_DEVICE
for _ in range(2):
    q.execute(); N.check(N.load().dfdb_materialize(q._h, outs, 2))
ctx.synchronize()
print("rows", n, "selected", nsel)
