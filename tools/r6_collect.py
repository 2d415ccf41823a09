#!/usr/bin/env python3
"""Turn what tools/r6_profiles.sh left under gpurun_out/r6p/ into the tracked files under profiles/ (r6_*).  Nothing here touches the GPU or oracle/."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r6p")
DST = os.path.join(ROOT, "profiles")


def json_lines(path):
    with open(path) as f:
        return [l for l in f if l.startswith("{")]


def counters(path):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    shutil.copy(os.path.join(SRC, "stats", "b_kernel_stats.csv"), os.path.join(DST, "r6_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "stats_step", "b_kernel_stats.csv"), os.path.join(DST, "r6_kernel_stats_step_only.csv"))
    for a, b in (("bench_under_rocprof.json", "r6_bench_under_rocprof.json"), ("bench_step_under_rocprof.json", "r6_bench_step_under_rocprof.json")):
        with open(os.path.join(DST, b), "w") as f:
            f.writelines(json_lines(os.path.join(SRC, a)))
    f = counters(os.path.join(SRC, "pmc_FETCH_SIZE", "p_counter_collection.csv"))
    w = counters(os.path.join(SRC, "pmc_WRITE_SIZE", "p_counter_collection.csv"))
    rows = 1_000_000_000

    def pick(acc, counter, needle, exclude=()):
        v = []
        for (k, c), vals in acc.items():
            if c == counter and needle in k and not any(x in k for x in exclude):
                v += vals
        return v

    def per_launch(needle, exclude=()):
        fv, wv = pick(f, "FETCH_SIZE", needle, exclude), pick(w, "WRITE_SIZE", needle, exclude)
        fv = [x for x in fv if x > 0.98 * max(fv)]
        wv = [x for x in wv if x > 0.9 * max(wv)]
        return sum(fv) / len(fv), sum(wv) / len(wv), [len(fv), len(wv)]

    fk, wk, n1 = per_launch("k_scan_cmp")
    out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-cold ; the same with "
                      "--pmc WRITE_SIZE (separate passes: the TCC cannot hold both). Round 6; tools/r6_profiles.sh + tools/r6_collect.py.",
           "kernel": "dfdb::k_scan_cmp<long, GT, false, nt=true, false>", "rows": rows, "launches": n1,
           "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
           "correction": "gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while a request is a 128-byte line (MI355X_MICROARCH.md, HBM section; calibrated in round 1 on 8e9 "
                         "known bytes, in round 2 by tools/bench_gather): bytes = 2 x FETCH_SIZE + WRITE_SIZE, for streams and for sparse reads alike",
           "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024, "algorithmic_bytes_per_launch": rows * (8 + 1 / 8 + 4 / 1024)}
    fk, wk, n = per_launch("k_compact_indices")
    out["k_compact_indices_wide"] = {"FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk, "launches": n, "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024}
    # K7: the forms by their template arguments (the mangled tail: ..., SCAN, PIPE, OCC, FARMAX, INDEX, HIST>)
    forms = {}
    for (k, c), vals in list(f.items()) + list(w.items()):
        if "k_lz4_decode" in k:
            forms.setdefault(k, {})[c] = {"launches": len(vals), "KB_per_launch_min": min(vals), "KB_per_launch_max": max(vals), "KB_per_launch_mean": sum(vals) / len(vals)}
    out["k_lz4_decode_forms"] = forms
    with open(os.path.join(DST, "r6_pmc_scan_cmp.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    sys.exit(main())
