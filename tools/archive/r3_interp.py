#!/usr/bin/env python3
"""The device interpreter on predicates no specialised kernel takes: time per launch (1e9 rows), for counters run it under rocprofv3 --pmc.
    python tools/r3_interp.py [--rows 1000000000]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from dfdb import _native as _N  # noqa: E402
if os.environ.get("R3_LIB"):                 # A/B against another build of the library (this tool only)
    _N.LIB_PATH = os.path.abspath(os.environ["R3_LIB"])
import dfdb  # noqa: E402
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=1_000_000_000); ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
    ctx = dfdb.Context(0, stream=s.cuda_stream)
    t = dfdb.DFTable.new(ctx=ctx)
    t.add_generated("a", dfdb.GEN_I64_MOD1M, 1, a.rows)
    t.add_generated("b", dfdb.GEN_I64_MOD1M, 2, a.rows)
    t.add_generated("x", dfdb.GEN_F64_U2000, 3, a.rows)
    out = {}
    cases = {"a + b > 1.8e6 (2 cols, 16 B/row)": t[t.a + t.b > 1_800_000, dfdb.ALL],
             "a * 3 + b * 2 - 7 > 4e6 (2 cols)": t[t.a * 3 + t.b * 2 - 7 > 4_000_000, dfdb.ALL],
             "(a + b) * x > 3e9 (3 cols, 24 B/row)": t[(t.a + t.b) * t.x > 3e9, dfdb.ALL],
             "(a > b) | (x * 2 > a) (3 cols)": t[(t.a > t.b) | (t.x * 2 > t.a), dfdb.ALL]}
    # a nullable column (Union{Int64,Missing}): the kernels that carry missing flags
    nn = min(a.rows, 200_000_000)
    tn = dfdb.DFTable.from_columns({"m": np.ma.masked_array(np.arange(nn, dtype=np.int64) % 1000, mask=(np.arange(nn) % 7 == 0)), "a": (np.arange(nn, dtype=np.int64) * 7919) % 1000}, ctx=ctx)
    from dfdb import ir
    cases["coalesce(m, 0) + a > 900 (nullable, 2e8 rows)"] = tn[ir.coalesce(ir.col(0), 0) + ir.col(1) > 900, dfdb.ALL]
    cases["(m > 500) & (a < 900) three-valued (nullable, 2e8 rows)"] = tn[ir.coalesce((ir.col(0) > 500) & (ir.col(1) < 900), False), dfdb.ALL]
    for name, v in cases.items():
        q = v._query(); n = q.count()
        ctx.profile(True)
        for _ in range(a.reps):
            q.reset(); q.execute()
        torch.cuda.synchronize()
        r = {}
        for k in ("interp_predicate", "scan_terms", "scan_cmp"):
            nl, ms = ctx.profile_get(k)
            if nl:
                r[k] = round(ms / nl, 4)
        ctx.profile(False)
        out[name] = dict(selected=n, **r)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
