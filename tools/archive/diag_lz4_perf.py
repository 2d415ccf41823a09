#!/usr/bin/env python3
"""Open one LZ4-compressed Int64 table (written by the device compressor) with a given decoder variant (rocprofv3 target)."""
import os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
variant = int(sys.argv[1]); m = int(float(sys.argv[2]))
ctx = dfdb.default_context(0)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    t = dfdb.DFTable.new(); t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, m); t.save(os.path.join(d, "tb")); t.close()
    for _ in range(2):
        ctx.profile(True)
        tb = dfdb.open_table(os.path.join(d, "tb"))
        print(ctx.profile_get("lz4_decode"))
        ctx.profile(False)
        tb.close()
finally:
    shutil.rmtree(d, ignore_errors=True)
