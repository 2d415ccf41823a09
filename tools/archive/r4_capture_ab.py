#!/usr/bin/env python3
"""round 4 A/B: the two-column capture of config 5's materialize leg (k_scan_terms EXTRA = 5: the first captured column parked in LDS) against the count-only scan,
1.25e9 rows, same process.  The variant that read the first captured column a SECOND time once the final mask existed instead of parking it (EXTRA = 6, ctx option
scan_capture_reload in the build that measured it) ran at 5.33-5.42 ms against 4.37-4.39 parked in LDS and 3.30 count-only: the tile is gone from L2 by then; not kept.
python tools/r4_capture_ab.py"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
torch.cuda.init()
import dfdb
from dfdb import _native as N
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_250_000_000
S = 0x9E3779B97F4A7C15
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(ctx=ctx)
t.add_generated("a", dfdb.GEN_I64_MOD1M, S, n)
t.add_generated("x", dfdb.GEN_F64_U2000, (S * 2) & (2**64 - 1), n)
t.add_generated("s", dfdb.GEN_STR_BRANDS10, (S * 3) & (2**64 - 1), n)
v = t[(t.a > 683_771) & (t.x < 632.456) & (t.s != "sony"), ["a", "x"]]
lib = N.load()
outs_by = {}
for rounds in range(2):
    for name, hint in (("count only", False), ("two captures, LDS stash", True)):
        q = v._query()
        q.hint_materialize(hint)
        nsel = q.count()
        oa = torch.empty(nsel, dtype=torch.int64, device="cuda"); ox = torch.empty(nsel, dtype=torch.float64, device="cuda")
        outs = (N.OutCol * 2)()
        outs[0].data, outs[0].memkind = oa.data_ptr(), N.MEM_DEVICE
        outs[1].data, outs[1].memkind = ox.data_ptr(), N.MEM_DEVICE
        ctx.profile(True)
        for _ in range(5):
            q.reset(); q.execute()
            if hint:
                N.check(lib.dfdb_materialize(q._h, outs, 2))
        ctx.synchronize()
        ks = {k: ctx.profile_get(k) for k in ("scan_terms", "str_match", "compact_captured", "gather")}
        ctx.profile(False)
        if hint:
            key = (oa.sum().item(), ox.sum().item())
            outs_by[name] = key
        print(json.dumps({"case": name, "selected": nsel, "ms": {k: round(v_[1] / v_[0], 4) for k, v_ in ks.items() if v_[0]}}), flush=True)
assert len(set(outs_by.values())) == 1, outs_by
