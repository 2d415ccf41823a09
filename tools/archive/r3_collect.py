#!/usr/bin/env python3
"""Turn what tools/r3_profiles.sh left under gpurun_out/r3p/ into the tracked files under profiles/ (r3_*).
Run from the repo root after `gpurun -- bash tools/r3_profiles.sh`.  Nothing here touches the GPU or oracle/."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3p")
DST = os.path.join(ROOT, "profiles")


def json_lines(path):
    with open(path) as f:
        return [l for l in f if l.startswith("{")]


def counters(path, want):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            for w in want:
                if w in r["Kernel_Name"]:
                    acc[(w, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    shutil.copy(os.path.join(SRC, "stats", "b_kernel_stats.csv"), os.path.join(DST, "r3_kernel_stats.csv"))
    if os.path.exists(os.path.join(SRC, "stats_step", "b_kernel_stats.csv")):
        shutil.copy(os.path.join(SRC, "stats_step", "b_kernel_stats.csv"), os.path.join(DST, "r3_kernel_stats_step_only.csv"))
        with open(os.path.join(DST, "r3_bench_step_under_rocprof.json"), "w") as f:
            f.writelines(json_lines(os.path.join(SRC, "bench_step_under_rocprof.json")))
    for a, b in (("bench_under_rocprof.json", "r3_bench_under_rocprof.json"), ("bench_default.json", "r3_bench_default.json"),
                 ("bench_exchange_lib.json", "r3_bench_exchange_lib.json"), ("bench_2ranks_gloo_device0.json", "r3_bench_2ranks_gloo_device0.json"),
                 ("bench_config5_host_shards4.json", "r3_bench_config5_host_shards4.json")):
        with open(os.path.join(DST, b), "w") as f:
            f.writelines(json_lines(os.path.join(SRC, a)))
    want = ("k_scan_cmp", "k_compact_indices", "k_lz4_decode")
    f = counters(os.path.join(SRC, "pmc_FETCH_SIZE", "p_counter_collection.csv"), want)
    w = counters(os.path.join(SRC, "pmc_WRITE_SIZE", "p_counter_collection.csv"), want)
    rows = 1_000_000_000

    def per_launch(k):
        fv, wv = f[(k, "FETCH_SIZE")], w[(k, "WRITE_SIZE")]
        fv = [v for v in fv if v > 0.98 * max(fv)]
        wv = [v for v in wv if v > 0.98 * max(wv)]
        return sum(fv) / len(fv), sum(wv) / len(wv), [len(fv), len(wv)]

    fk, wk, n1 = per_launch("k_scan_cmp")
    out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-configs ; the same with "
                      "--pmc WRITE_SIZE (separate passes: the TCC cannot hold both). Round 3, final code; tools/r3_profiles.sh + tools/r3_collect.py.",
           "kernel": "dfdb::k_scan_cmp<long, GT, false, nt=true, false>", "rows": rows, "launches": n1,
           "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
           "correction": "gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while a request is a 128-byte line (MI355X_MICROARCH.md, HBM section; calibrated in round 1 on 8e9 "
                         "known bytes, in round 2 by tools/bench_gather): bytes = 2 x FETCH_SIZE + WRITE_SIZE, for streams and for sparse reads alike",
           "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024, "algorithmic_bytes_per_launch": rows * (8 + 1 / 8 + 4 / 1024)}
    bd = json.loads(json_lines(os.path.join(SRC, "bench_default.json"))[0])
    nsel = bd["config"]["selected_per_gpu"]
    fk, wk, n = per_launch("k_compact_indices")
    out["k_compact_indices_wide"] = {"FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk, "launches": n, "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024,
                                     "algorithmic_bytes_per_launch": rows // 8 + rows // 1024 * 8 + nsel * 8}
    # K7: both passes see the same launches in the same order (the load's decode, the `without_index` leg, the recording launch, the indexed legs, the fused
    # legs): pair them by position and sort them into the forms by what they moved
    st = bd.get("decode_scan", {})
    fv, wv = f[("k_lz4_decode", "FETCH_SIZE")], w[("k_lz4_decode", "WRITE_SIZE")]
    forms = {"plain": [], "recording": [], "indexed": [], "indexed_fused_with_the_predicate": []}
    if len(fv) == len(wv) and fv:
        fmin = min(fv)
        for a_, b_ in zip(fv, wv):
            if b_ > 1.2 * min(wv): forms["recording"].append((a_, b_))            # the index's atomics count as writes
            elif a_ < 1.01 * fmin: forms["plain"].append((a_, b_))
            elif b_ > 1.01 * min(wv): forms["indexed_fused_with_the_predicate"].append((a_, b_))   # + the bitmap and the tile counts
            else: forms["indexed"].append((a_, b_))
    lz = {"algorithmic_bytes_per_launch": (st.get("compressed_bytes") or 0) + rows * 8,
          "note": "bench.py's decode_scan leg (15 259 blocks of the ENGINE-compressed column per launch): compressed bytes + the 128-byte lines the 24-byte far-source "
                  "reads pull in (+ the sequence-start index where it is read), decoded bytes written.  profiles/r3_pmc_lz4.txt has the request counters of the "
                  "liblz4-compressed column and the reconciliation with round 2's two records; profiles/r3_lz4_index.txt the instruction counters of the indexed form."}
    for k, v in forms.items():
        if v:
            fk, wk = sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)
            lz[k] = {"launches": len(v), "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk, "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024}
    out["k_lz4_decode"] = lz
    with open(os.path.join(DST, "r3_pmc_scan_cmp.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
