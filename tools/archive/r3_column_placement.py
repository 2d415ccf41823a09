#!/usr/bin/env python3
"""Does the COLUMN's allocation matter once the bitmap has been calibrated?  Six 8-GB columns with the same contents in one process, each scanned by its own query with
the bitmap placement calibration on: per column the best and the worst of the nine bitmap candidates, and the K1 time the query then runs at.
    python tools/r3_column_placement.py [--columns 6]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import dfdb  # noqa: E402


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--columns", type=int, default=6); a = ap.parse_args()
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
    ctx = dfdb.Context(0, stream=s.cuda_stream)
    n = 1_000_000_000
    tabs = []
    for k in range(a.columns):
        t = dfdb.DFTable.new(ctx=ctx); t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n); tabs.append(t)
    rows = []
    for cal in (0, 1):
        ctx.set_option("placement_calibrate", cal)
        for k, t in enumerate(tabs):
            q = t[("x", lambda x: x > 899_999), dfdb.ALL]._query()
            b0 = ctx.profile_get("placement_best_us"); w0 = ctx.profile_get("placement_worst_us")
            q.count()
            b1 = ctx.profile_get("placement_best_us"); w1 = ctx.profile_get("placement_worst_us")
            ctx.profile(True)
            for _ in range(10):
                q.reset(); q.execute()
            torch.cuda.synchronize()
            nl, ms = ctx.profile_get("scan_cmp"); ctx.profile(False)
            rows.append(dict(column=k, calibrated=cal, k1_ms=round(ms / nl, 4), candidates_best_ms=round((b1[1] - b0[1]) / 1e3, 4) if b1[0] > b0[0] else None,
                             candidates_worst_ms=round((w1[1] - w0[1]) / 1e3, 4) if w1[0] > w0[0] else None))
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
