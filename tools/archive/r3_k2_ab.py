#!/usr/bin/env python3
"""K2 (k_compact_indices) store forms A/B'd INSIDE one process, round robin, in the bench step (K1 -> count scan -> K2, back to back): placement of the
column / bitmap / output is the same for every form, so the differences are the kernels'.  Forms (ctx option compact_store): 0 plain 8-byte stores,
1 nontemporal 8-byte (round 2's default), 3 wide (two ctiles per trip) with nontemporal 16-byte stores, 4 wide with plain 16-byte stores.
Prints one JSON line.   python tools/r3_k2_ab.py [--rows 1e9] [--rounds 6] [--steps 20]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import dfdb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=1e9)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--sigma-threshold", type=int, default=899_999)
    a = ap.parse_args()
    n = int(a.rows)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
    ctx = dfdb.Context(0, stream=s.cuda_stream)
    t = dfdb.DFTable.new(ctx=ctx)
    t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
    q = t[("x", lambda x: x > a.sigma_threshold), dfdb.ALL]._query()
    nsel = q.count()
    out = torch.empty(max(nsel, 1), dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        q.reset(); q.indices_device(out.data_ptr(), nsel); q.count_device(cnt.data_ptr())
    forms = [("nt8", 1, -1), ("wide_nt16", 3, -1), ("wide_nt16_grid16k", 3, 16384), ("wide_nt16_grid64k", 3, 65536), ("wide_nt16_lds4k", 5, -1),
             ("wide_nt16_lds4k_grid16k", 5, 16384), ("wide_nt16_lds4k_grid64k", 5, 65536), ("wide_plain16_lds4k_grid64k", 6, 65536), ("plain8", 0, -1)]
    res = {f[0]: {"k2": [], "k1": [], "step": []} for f in forms}
    for r in range(a.rounds):
        for name, f, cap in forms:
            ctx.set_option("compact_store", f)
            ctx.set_option("compact_grid_cap", cap)       # -1: back to the shipped rule
            for _ in range(3):
                step()
            ctx.profile(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / a.steps * 1e3
            n2, ms2 = ctx.profile_get("compact_indices"); n1, ms1 = ctx.profile_get("scan_cmp")
            ctx.profile(False)
            if r:
                res[name]["k2"].append(round(ms2 / n2, 4)); res[name]["k1"].append(round(ms1 / n1, 4)); res[name]["step"].append(round(el, 4))
    print(json.dumps({"rows": n, "selected": nsel, "forms": {k: {kk: {"min": min(v), "median": sorted(v)[len(v) // 2], "all": v} for kk, v in d.items()} for k, d in res.items()}}))


if __name__ == "__main__":
    main()
