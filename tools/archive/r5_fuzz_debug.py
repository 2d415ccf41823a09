#!/usr/bin/env python3
"""debug aid: one seed of tests/test_gpu_fuzz.py::test_random_queue_equals_the_oracle, stage by stage and conjunct by conjunct"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()
import dfdb
from dfdb import ir
from oracle import oracle as O
O.build()
import test_gpu_fuzz as F
from helpers import apply_stages_both
seed = int(sys.argv[1]); variant = sys.argv[2] if len(sys.argv) > 2 else "flat strings"
class R: param = variant
fx = F.pair
fn = getattr(fx, "__wrapped__", None) or getattr(getattr(fx, "_fixture_function", None), "__call__", None) or fx.__pytest_wrapped__.obj
pair = fn(O, dfdb, R)
g = F.Gen(ir, seed, risky=seed % 4 == 3)
stages, proj = g.stages(), g.proj()
print("stages:", stages)
print("proj:", proj)
ov, dv = apply_stages_both(pair, stages, proj=proj)
print("oracle nrow", ov.nrow(), "engine nrow", dfdb.nrow(dv), "engine q.count", dv._query().count())
for k in range(1, len(stages) + 1):
    o2, d2 = apply_stages_both(pair, stages[:k], proj=None)
    print("prefix", k, stages[k - 1][0], "oracle", o2.nrow(), "engine", d2._query().count())
def conj(e, out):
    if e.op == ir.AND: conj(e.args[0], out); conj(e.args[1], out)
    else: out.append(e)
for st in stages:
    if st[0] == "pred":
        cs = []; conj(st[1], cs)
        for c in cs:
            o2, d2 = apply_stages_both(pair, [("pred", c)], proj=None)
            print("  conjunct", repr(c)[:150], "oracle", o2.nrow(), "engine", d2._query().count())
