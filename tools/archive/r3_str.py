#!/usr/bin/env python3
"""The flat string scan alone (config 4's predicate: s == "sony" over 5e8 rows of the 10-brand vocabulary), per-kernel time; R3_LIB points at another
build of the library for A/B runs.   python tools/r3_str.py [--rows 500000000]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from dfdb import _native as _N  # noqa: E402
if os.environ.get("R3_LIB"):
    _N.LIB_PATH = os.path.abspath(os.environ["R3_LIB"])
import dfdb  # noqa: E402


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=500_000_000); ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
    ctx = dfdb.Context(0, stream=s.cuda_stream)
    ctx.set_option("string_dictionary", 0)
    t = dfdb.DFTable.new(ctx=ctx)
    t.add_generated("s", dfdb.GEN_STR_BRANDS10, 0x9E3779B97F4A7C15, a.rows)
    out = {}
    for name, v in (("s == sony", t[t.s == "sony", dfdb.ALL]), ("s != sony", t[t.s != "sony", dfdb.ALL]), ("s == microsoft", t[t.s == "microsoft", dfdb.ALL])):
        q = v._query(); n = q.count()
        ctx.profile(True)
        for _ in range(a.reps):
            q.reset(); q.execute()
        torch.cuda.synchronize()
        nl, ms = ctx.profile_get("str_match"); ctx.profile(False)
        out[name] = dict(selected=n, str_match_ms=round(ms / max(nl, 1), 4))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
