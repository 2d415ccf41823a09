#!/usr/bin/env python3
"""print the interesting figures of one bench.py JSON line: python tools/r4_show_bench.py file.json"""
import json, sys
r = json.load(open(sys.argv[1]))
print("value", r["value"], "ms/step", r["ms_per_step"], "K1 frac", r["roofline"]["frac"], "|", r["config"].get("options"))
for k, v in r.get("configs", {}).items():
    if k == "interp":
        for e, x in v.items():
            if isinstance(x, dict):
                print("  interp", e, {kk: (round(vv["ms"], 3), round(vv["frac_of_peak"], 3), round(vv["first_execution_s"], 2)) for kk, vv in x.items() if isinstance(vv, dict) and "ms" in vv})
    elif isinstance(v, dict) and "ms_per_step" in v:
        print(k, round(v["ms_per_step"], 3), v["kernels_avg_ms"], round(v["roofline"]["frac"], 3))
    else:
        print(k, {kk: vv for kk, vv in v.items() if kk != "what"} if isinstance(v, dict) else v)
c = r.get("cold", {})
for k, v in c.items():
    if isinstance(v, dict):
        print("cold", k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk not in ("what", "file_bytes_read")})
    else:
        print("cold", k, v)
d = r.get("decode_scan", {})
print("decode_scan", {k: d.get(k) for k in ("ms_per_step", "rows_per_s", "decoded_GBps", "error")}, "unfused", (d.get("unfused") or {}).get("decoded_GBps"), "no index", (d.get("without_index") or {}).get("decoded_GBps"))
print("cpu", (r.get("cpu_baseline") or {}).get("value"))
