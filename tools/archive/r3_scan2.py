#!/usr/bin/env python3
"""How far is the generic multi-term scan (k_scan_terms) from twice the single-term scan (k_scan_cmp) on a two-column conjunction?
One process, 1e9 rows: K1 on a, K1 on x, then (a > c1) & (x < c2) as a count, with sum(x), and with x captured for materialize.
    python tools/r3_scan2.py [--rows 1000000000] [--option name=value ...]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from dfdb import _native as _N  # noqa: E402
if os.environ.get("R3_LIB"):                 # A/B against another build of the library (this tool only)
    _N.LIB_PATH = os.path.abspath(os.environ["R3_LIB"])
import dfdb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
    ctx = dfdb.Context(0, stream=s.cuda_stream)
    for o in a.option:
        k, v = o.split("="); ctx.set_option(k, int(v))
    n = a.rows
    t = dfdb.DFTable.new(ctx=ctx)
    t.add_generated("a", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
    t.add_generated("x", dfdb.GEN_F64_U2000, 0x9E3779B97F4A7C15 + 2, n)
    out = {}

    def timed(name, q, fn, kernels):
        fn(q)
        ctx.profile(True)
        for _ in range(a.reps):
            q.reset(); fn(q)
        torch.cuda.synchronize()
        r = {}
        for k in kernels:
            nl, ms = ctx.profile_get(k)
            if nl:
                r[k] = round(ms / nl, 4)
        ctx.profile(False)
        out[name] = r

    timed("K1 a > c", t[t.a > 683_771, dfdb.ALL]._query(), lambda q: q.execute(), ["scan_cmp"])
    timed("K1 x < c", t[t.x < 632.456, dfdb.ALL]._query(), lambda q: q.execute(), ["scan_cmp"])
    timed("two terms, count", t[(t.a > 683_771) & (t.x < 632.456), dfdb.ALL]._query(), lambda q: q.execute(), ["scan_terms", "scan_cmp"])
    timed("two terms, sum(x)", t[(t.a > 683_771) & (t.x < 632.456), ["x"]]._query(), lambda q: q.aggregate(dfdb.AGG_SUM, 0) if hasattr(q, "aggregate") else q.execute(), ["scan_terms", "reduce_partials"])
    qm = t[(t.a > 683_771) & (t.x < 632.456), ["x"]]._query(); qm.hint_materialize(True)
    timed("two terms, x captured", qm, lambda q: q.execute(), ["scan_terms"])
    qk = t[t.a > 899_999, ["a"]]._query(); qk.hint_materialize(True)
    timed("K1 a > c, a captured", qk, lambda q: q.execute(), ["scan_cmp"])
    ob = torch.empty(qm.count() + 1, dtype=torch.float64, device=dev)
    outs = (_N.OutCol * 1)(); outs[0].data, outs[0].memkind = ob.data_ptr(), _N.MEM_DEVICE
    lib = _N.load()

    def mat(q):
        q.execute(); _N.check(lib.dfdb_materialize(q._h, outs, 1))
    timed("two terms, materialize x", qm, mat, ["scan_terms", "compact_captured", "gather"])
    timed("three terms", t[(t.a > 683_771) & (t.x < 632.456) & (t.a < 990_000), dfdb.ALL]._query(), lambda q: q.execute(), ["scan_terms"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
