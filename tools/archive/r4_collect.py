#!/usr/bin/env python3
"""Turn what tools/r4_profiles.sh left under gpurun_out/r4p/ into the tracked files under profiles/ (r4_*).
Run from the repo root after `gpurun -- bash tools/r4_profiles.sh`.  Nothing here touches the GPU or oracle/."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r4p")
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
    shutil.copy(os.path.join(SRC, "stats", "b_kernel_stats.csv"), os.path.join(DST, "r4_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "stats_step", "b_kernel_stats.csv"), os.path.join(DST, "r4_kernel_stats_step_only.csv"))
    for a, b in (("bench_under_rocprof.json", "r4_bench_under_rocprof.json"), ("bench_step_under_rocprof.json", "r4_bench_step_under_rocprof.json"),
                 ("bench_8ranks_gloo_device0.json", "r4_bench_8ranks_gloo_device0.json"), ("bench_threads3_device0.json", "r4_bench_threads3_device0.json")):
        with open(os.path.join(DST, b), "w") as f:
            f.writelines(json_lines(os.path.join(SRC, a)))
    shutil.copy(os.path.join(SRC, "types.txt"), os.path.join(DST, "r4_types.txt"))
    # ---- K1 / K2 / K7 traffic
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
    out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-cold ; the same with "
                      "--pmc WRITE_SIZE (separate passes: the TCC cannot hold both). Round 4, final code; tools/r4_profiles.sh + tools/r4_collect.py.",
           "kernel": "dfdb::k_scan_cmp<long, GT, false, nt=true, false>", "rows": rows, "launches": n1,
           "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
           "correction": "gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while a request is a 128-byte line (MI355X_MICROARCH.md, HBM section; calibrated in round 1 on 8e9 "
                         "known bytes, in round 2 by tools/bench_gather): bytes = 2 x FETCH_SIZE + WRITE_SIZE, for streams and for sparse reads alike",
           "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024, "algorithmic_bytes_per_launch": rows * (8 + 1 / 8 + 4 / 1024)}
    fk, wk, n = per_launch("k_compact_indices")
    out["k_compact_indices_wide"] = {"FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk, "launches": n, "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024}
    fv, wv = f[("k_lz4_decode", "FETCH_SIZE")], w[("k_lz4_decode", "WRITE_SIZE")]
    if fv and wv:
        out["k_lz4_decode"] = {"launches": [len(fv), len(wv)], "FETCH_SIZE_KB_per_launch_min_max": [min(fv), max(fv)], "WRITE_SIZE_KB_per_launch_min_max": [min(wv), max(wv)],
                               "note": "bench.py's decode_scan leg over the ENGINE-compressed column (lz4_enc_near = 0, the default): see profiles/r3_pmc_scan_cmp.json for the forms; "
                                       "profiles/r4_lz4_near.txt for what the compressor's near-match option does to ratio and decode rate"}
    with open(os.path.join(DST, "r4_pmc_scan_cmp.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    # ---- interpreter vs its run-time compiled kernels
    names = ("k_interp", "dfdb_jit_kernel")
    ic = counters(os.path.join(SRC, "pmc_interp", "p_counter_collection.csv"), names)
    lines = ["# Round 4: the device interpreter against the SAME programs compiled at run time by hipRTC (csrc/jit.cpp), 1e9 rows, tools/r4_interp.py under",
             "# rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES --kernel-trace (and FETCH_SIZE / WRITE_SIZE in separate passes).  Launch order in both kernels:",
             "# a*3 + b*2 - 7 > 4e6 | (a > b) | (x*2 > a) | (a + b) * x > 3e9 | a + b > 1.8e6, each 1 warm + 2 timed launches (the compiled form first runs once more: the count).", ""]
    with open(os.path.join(SRC, "interp.json")) as fi:
        lines.append("times without counters (ms per launch): " + fi.read().strip())
    lines.append("")
    for k in names:
        sal, val, wav = ic[(k, "SQ_INSTS_SALU")], ic[(k, "SQ_INSTS_VALU")], ic[(k, "SQ_WAVES")]
        lines.append(f"{k}: {len(sal)} launches")
        for i, (s_, v_, w_) in enumerate(zip(sal, val, wav)):
            lines.append(f"  launch {i:2d}: SQ_INSTS_SALU {s_:.4g}  SQ_INSTS_VALU {v_:.4g}  SQ_WAVES {w_:.4g}  -> per 256-row group (1e9 rows = 3.906e6 groups): SALU {s_ / 3.90625e6:.1f}  VALU {v_ / 3.90625e6:.1f}")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(SRC, f"pmc_interp_{c}", "p_counter_collection.csv")
        if os.path.exists(p):
            tc = counters(p, names)
            for k in names:
                v = tc[(k, c)]
                if v:
                    lines.append(f"{k} {c} KB per launch: " + ", ".join(f"{x:.4g}" for x in v) + ("   (HBM bytes read = 2 x FETCH_SIZE on gfx950)" if c == "FETCH_SIZE" else ""))
    with open(os.path.join(DST, "r4_interp_pmc.txt"), "w") as fo:
        fo.write("\n".join(lines) + "\n")
    print(json.dumps(out, indent=1)[:1500])
    print("\n".join(lines)[:3000])


if __name__ == "__main__":
    sys.exit(main())
