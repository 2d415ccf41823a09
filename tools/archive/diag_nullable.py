import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/dataframedbs.jl_amd')
import torch, numpy as np, dfdb
from dfdb import ir
ctx = dfdb.default_context(0)
rng = np.random.default_rng(23)
n = 150_011
cols = {"m": np.ma.masked_array(rng.integers(-50, 50, n).astype(np.int64), mask=rng.random(n) < 0.3), "c": rng.integers(-5, 6, n).astype(np.int64),
        "sm": [None if i % 7 == 0 else "s%d" % (i % 5) for i in range(n)]}
t = dfdb.DFTable.from_columns(cols, block_size=4096)
for name, pred in [("plain", ir.col(1) * 2 > 3), ("nul", ir.coalesce(ir.col(0) > 10, False)), ("nulstr", ir.coalesce(ir.col(2) == "s1", False))]:
    v = dfdb.selection(dfdb.DFView(t), pred)
    for rep in range(2):
        ctx.profile(True)
        t0 = time.perf_counter(); q = v._query(); c = q.count(); dt = time.perf_counter() - t0
        print(name, rep, c, "wall %.4f" % dt, ctx.profile_get("interp_predicate"))
        ctx.profile(False)
