#!/usr/bin/env python3
"""Decode single-column tables with one LZ4 decoder variant and report the first differing byte per column."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa
import numpy as np
import dfdb
from oracle import oracle as O

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = dfdb.default_context(0)
ctx.set_option("lz4_variant", variant)
rng = np.random.default_rng(1)
n = 200_003
cols = {"a": O.gen_i64(0x9E3779B97F4A7C15, 0, n), "x": O.gen_f64(1, 0, n), "iota": np.arange(1, n + 1, dtype=np.int64), "zeros": np.zeros(n, np.int64),
        "rnd": rng.integers(-2**62, 2**62, n).astype(np.int64), "i16": rng.integers(-300, 300, n).astype(np.int16),
        "period3": np.tile(np.array([7, -1, 2**40], np.int64), n // 3 + 1)[:n]}
d = tempfile.mkdtemp()
for name, arr in cols.items():
    for bs in (65536, 1000):
        t = O.Table(block_size=bs)
        t.add_column(name, arr)
        path = os.path.join(d, f"{name}{bs}")
        t.save(path)
        try:
            tb = dfdb.open_table(path)
        except Exception as e:
            print(name, bs, "open failed:", e); continue
        got = np.asarray(dfdb.materialize(tb[dfdb.ALL, dfdb.ALL])[name])
        w, g = arr.view(np.uint8), got.view(np.uint8)
        if np.array_equal(w, g):
            print(name, bs, "ok")
        else:
            i = int(np.flatnonzero(w != g)[0])
            blk = i // (bs * arr.itemsize)
            print(name, bs, "first diff at byte", i, "block", blk, "byte in block", i - blk * bs * arr.itemsize, "ndiff", int((w != g).sum()),
                  "want", w[i:i + 12].tolist(), "got", g[i:i + 12].tolist())
