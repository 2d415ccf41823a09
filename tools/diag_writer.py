#!/usr/bin/env python3
"""Save one generated Int64 column with the device encoder, read it back with liblz4 (oracle) and with the device decoder."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import numpy as np
import dfdb
from oracle import oracle as O
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
SEED = 0x9E3779B97F4A7C15
t = dfdb.DFTable.new()
t.add_generated("a", dfdb.GEN_I64_MOD1M, SEED, n)
d = tempfile.mkdtemp()
t.save(os.path.join(d, "tb"))
want = O.gen_i64(SEED, 0, n)
ot = O.Table.open(os.path.join(d, "tb"))
try:
    got = ot.view().materialize()[0]
    bad = np.flatnonzero(got != want)
    print("oracle read: mismatches", len(bad), "first", bad[:3], "blocks", sorted(set((bad // 65536).tolist()))[:10])
except Exception as e:
    print("oracle read failed:", e)
back = np.asarray(dfdb.materialize(dfdb.open_table(os.path.join(d, "tb")))["a"])
bad = np.flatnonzero(back != want)
print("device read: mismatches", len(bad), "first", bad[:3], "blocks", sorted(set((bad // 65536).tolist()))[:10])
src = np.asarray(dfdb.materialize(t)["a"])
print("source column vs generator:", int((src != want).sum()))
