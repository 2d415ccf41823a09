#!/usr/bin/env python3
"""Open one LZ4-compressed Int64 table with a given decoder variant (rocprofv3 target)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import dfdb
from oracle import oracle as O
variant = int(sys.argv[1]); m = int(float(sys.argv[2]))
ctx = dfdb.default_context(0)
ctx.set_option("lz4_variant", variant)
ot = O.Table(block_size=65536)
ot.add_column("x", O.gen_i64(0x9E3779B97F4A7C15, 0, m))
d = tempfile.mkdtemp()
ot.save(os.path.join(d, "tb"))
for _ in range(2):
    ctx.profile(True)
    tb = dfdb.open_table(os.path.join(d, "tb"))
    print(ctx.profile_get("lz4_decode"))
    ctx.profile(False)
