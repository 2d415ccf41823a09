#!/usr/bin/env python3
"""where the wall clock of groupreduce over 1e6 groups goes: the two C calls (dfdb_query_groupreduce, dfdb_query_groupreduce_fetch) against the Python mirror around them"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")]
import numpy as np
import torch
torch.cuda.init()
import dfdb
import dfdb._native as N
from dfdb import api
n = 1_000_000_000
ctx = dfdb.default_context(0)
t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
t.add_generated("a", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15 * 2, n)
t.add_column_from("f", t.x * 0.5)
L = N.load()
for key in ("x", "f"):
    for rep in range(3):
        sub, with_value = api._groupreduce_view(t, key, "a", "sum")
        q = api._Query(sub)
        ng, kb = C.c_int64(), C.c_int64()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N.check(L.dfdb_query_groupreduce(q._h, 0, 1, api._STATS["sum"], C.byref(ng), C.byref(kb)))
        t1 = time.perf_counter()
        m = ng.value
        karr = np.empty(m, np.int64 if key == "x" else np.float64); counts = np.empty(m, np.int64); vi = np.empty(m, np.int64); vf = np.empty(m, np.float64)
        out = N.OutCol(); out.memkind = N.MEM_HOST; out.data = karr.ctypes.data
        t2 = time.perf_counter()
        N.check(L.dfdb_query_groupreduce_fetch(q._h, C.byref(out), counts.ctypes.data, vi.ctypes.data, vf.ctypes.data))
        t3 = time.perf_counter()
        t4 = time.perf_counter(); g = dfdb.groupreduce(t, key, "a", "sum"); t5 = time.perf_counter()
        print(key, "groups", m, "dfdb_query_groupreduce %.2f ms" % ((t1 - t0) * 1e3), "numpy.empty x4 %.2f" % ((t2 - t1) * 1e3), "dfdb_query_groupreduce_fetch %.2f ms" % ((t3 - t2) * 1e3),
              "| the mirror's dfdb.groupreduce, whole: %.2f ms" % ((t5 - t4) * 1e3), flush=True)
