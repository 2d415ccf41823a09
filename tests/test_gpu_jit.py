"""Round 4: the second tier of expression evaluation — the interpreter's own source compiled at run time by hipRTC for one program shape (csrc/jit.cpp,
k_interp_device.inc).  The reference JIT-fuses every broadcast (src/tables/broadcast.jl:60-68); here the compiled kernel must give, bit for bit, what the
interpreter gives and what the oracle gives: the same random queues as tests/test_gpu_fuzz.py with ctx option jit = 2 (wait for the compiler, every table
size), the two benchmark expressions at 2e8 rows, and the background path (jit = 1: the interpreter answers until the compiler is done)."""
import os
import time

import numpy as np
import pytest

import test_gpu_fuzz as F
from test_gpu_fuzz import pair  # noqa: F401  (the fuzz table, flat strings and with a dictionary)
from helpers import Pair, apply_stages, assert_same

pytestmark = pytest.mark.gpu
NSEEDS = int(os.environ.get("DFDB_JIT_SEEDS", "30"))
SEED0 = int(os.environ.get("DFDB_JIT_SEED0", "0"))


def _jit_launches(ctx):
    return ctx.profile_get("jit_predicate")[0] + ctx.profile_get("jit_project")[0]


@pytest.fixture()
def jit_forced(ctx):
    ctx.set_option("jit", 2)
    ctx.set_option("jit_min_rows", 0)
    yield ctx
    ctx.set_option("jit", 1)
    ctx.set_option("jit_min_rows", 1 << 22)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + NSEEDS))
def test_compiled_kernels_equal_the_oracle_on_random_queues(pair, dfdb_mod, jit_forced, seed):  # noqa: F811
    """the differential fuzz of tests/test_gpu_fuzz.py — typed expression trees over every operator and column type, missing values, strings, zero
    divisors, inexact casts, random projections — with every interpreter program replaced by its run-time compiled kernel"""
    # (seeds 3 mod 4 are the risky ones: DivideError / InexactError must surface exactly as in the interpreter)
    F.test_random_queue_equals_the_oracle(pair, dfdb_mod, 4 * seed + 3 if seed % 2 else seed)


def test_compiled_kernels_were_the_ones_that_ran(pair, dfdb_mod, jit_forced):  # noqa: F811
    """a generic predicate and a computed projection under jit = 2: the launches are the compiled kernel's, not the interpreter's"""
    from dfdb import ir
    ctx = jit_forced
    ov, dv = apply_stages(pair, [("pred", (ir.col(0) * 3 + ir.col(2) * 2 - 7 > ir.col(4)) & (ir.col(7) * 2.0 < ir.col(0) + 50))],
                          proj=[("k", ir.col(0) * ir.col(2) - ir.col(3)), ("q", ir.col(7) / ir.col(2))])
    ctx.profile(True)
    assert_same(pair, ov, dv)
    nj, ni = _jit_launches(ctx), ctx.profile_get("interp_predicate")[0] + ctx.profile_get("interp_project")[0]
    ctx.profile(False)
    assert nj >= 3 and ni == 0, (nj, ni)


def test_background_compile_takes_over(dfdb_mod, oracle, ctx):
    """jit = 1 (the default): the first execution of a new shape is the interpreter's — nothing waits for the compiler — and a later one is the compiled
    kernel's; both give the oracle's answer"""
    from dfdb import ir
    n = 50_000
    rng = np.random.default_rng(5)
    cols = {"a": rng.integers(-1000, 1000, n).astype(np.int64), "b": rng.integers(-1000, 1000, n).astype(np.int32), "x": rng.normal(0, 100, n)}
    p = Pair(oracle, dfdb_mod, cols, block_size=4096)
    pred = (ir.col(0) * 5 - ir.col(1) * 3 + 11 > ir.col(2)) | (ir.col(1) % 7 == 3)       # a shape no other test uses
    ctx.set_option("jit", 1); ctx.set_option("jit_min_rows", 0)
    try:
        ov, dv = apply_stages(p, [("pred", pred)])
        want = ov.select_indices()
        ctx.profile(True)
        q = dv._query()
        assert np.array_equal(q.indices(), want)
        deadline = time.time() + 30
        took_over = False
        while time.time() < deadline and not took_over:
            q.reset()
            assert np.array_equal(q.indices(), want)
            took_over = ctx.profile_get("jit_predicate")[0] > 0
            if not took_over:
                time.sleep(0.05)
        ni = ctx.profile_get("interp_predicate")[0]
        ctx.profile(False)
        assert took_over, "the compiled kernel never replaced the interpreter"
        assert ni >= 1                                                  # the first answer did not wait for the compiler
    finally:
        ctx.set_option("jit", 1); ctx.set_option("jit_min_rows", 1 << 22)


def test_compiled_kernels_large_properties(dfdb_mod, ctx):
    """2e8 rows: the interpreter's and the compiled kernel's bitmaps are the same words for the two expressions the bench times, a nullable program and a
    computed Float64 projection; the compiled kernel is not slower"""
    import torch
    from dfdb import ir, _native as N
    n = 200_000_000
    t = dfdb_mod.DFTable.new(ctx=ctx)
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, 1, n)
    t.add_generated("b", dfdb_mod.GEN_I64_MOD1M, 2, n)
    t.add_generated("x", dfdb_mod.GEN_F64_U2000, 3, n)
    a, b, x = ir.col(0), ir.col(1), ir.col(2)
    preds = {"a*3 + b*2 - 7 > 4e6": a * 3 + b * 2 - 7 > 4_000_000, "(a > b) | (x*2 > a)": (a > b) | (x * 2 > a), "(a + b) * x > 3e9": (a + b) * x > 3e9,
             "a ÷ (b % 5 + 1) == 7": ir.div(a, b % 5 + 1) == 7}
    for name, pred in preds.items():
        res = {}
        for jit in (0, 2):
            ctx.set_option("jit", jit)
            try:
                q = t[pred, dfdb_mod.ALL]._query()
                cnt = q.count()
                ctx.profile(True)
                for _ in range(3):
                    q.reset(); q.execute()
                ctx.synchronize()
                k = "jit_predicate" if jit else "interp_predicate"
                nl, ms = ctx.profile_get(k)
                ctx.profile(False)
                assert nl == 3, (name, jit, nl)
                bm = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
                N.check(N.load().dfdb_select_bitmap(q._h, bm.data_ptr(), N.MEM_DEVICE))
                ctx.synchronize()
                res[jit] = (cnt, bm, ms / nl)
            finally:
                ctx.set_option("jit", 1)
        assert res[0][0] == res[2][0] and torch.equal(res[0][1], res[2][1]), name
        print(f"{name}: interpreter {res[0][2]:.3f} ms, compiled {res[2][2]:.3f} ms per 2e8 rows")
        assert res[2][2] <= res[0][2] * 1.05, (name, res[0][2], res[2][2])
    # a computed projection over a filtered view: the compacted values are the same bytes
    outs = {}
    for jit in (0, 2):
        ctx.set_option("jit", jit)
        try:
            v = t[t.a > 900_000, {"k": t.a * 2 + t.b, "r": t.x * 0.5 - t.a}]
            outs[jit] = [c for c in v._query().materialize()]
        finally:
            ctx.set_option("jit", 1)
    for c0, c1 in zip(outs[0], outs[2]):
        assert np.array_equal(c0.view(np.uint8), c1.view(np.uint8))
    t.close()


def test_exit_while_the_compiler_is_busy(ctx):
    """a process that ends while the background compiler is inside hipRTC must end cleanly (exit status 0, no crash in comgr's static destructors): the
    engine's exit handler drops pending shapes and waits for the compile in flight"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'dataframedbs.jl_amd')!r}]\n"
        "import torch; torch.cuda.init()\n"
        "import dfdb\n"
        "from dfdb import ir\n"
        "ctx = dfdb.default_context(0)\n"
        "ctx.set_option('jit', 1); ctx.set_option('jit_min_rows', 0)\n"
        "t = dfdb.DFTable.new(ctx=ctx)\n"
        "t.add_generated('a', dfdb.GEN_I64_MOD1M, 1, 100000)\n"
        "n = 0\n"
        "for k in range(6):\n"                      # six new shapes queued: the first is being compiled when the process ends
        "    n += t[abs(ir.col(0) * (k + 2)) % (k + 3) > k, dfdb.ALL]._query().count()\n"
        "print('counted', n)\n")
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stderr.decode(errors="replace")[-2000:])
    assert b"counted" in p.stdout


def test_code_objects_come_back_from_the_disk_cache(tmp_path):
    """a shape compiled by one process is read from $DFDB_JIT_CACHE_DIR by the next (no hipRTC call), gives the same answer, and a torn file in the cache is
    discarded and compiled again"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'dataframedbs.jl_amd')!r}]\n"
        "import torch; torch.cuda.init()\n"
        "import dfdb\n"
        "from dfdb import ir\n"
        "ctx = dfdb.default_context(0)\n"
        "ctx.set_option('jit', 2); ctx.set_option('jit_min_rows', 0)\n"
        "t = dfdb.DFTable.new(ctx=ctx)\n"
        "t.add_generated('a', dfdb.GEN_I64_MOD1M, 1, 300000)\n"
        "t.add_generated('b', dfdb.GEN_I64_MOD1M, 2, 300000)\n"
        "ctx.profile(True)\n"
        "n = t[(abs(ir.col(0) * 3 - ir.col(1)) % 11 > 4) | (ir.col(1) * 2 < ir.col(0)), dfdb.ALL]._query().count()\n"
        "print('RESULT', n, ctx.profile_get('jit.compiled')[0], ctx.profile_get('jit.from_disk')[0], ctx.profile_get('jit_predicate')[0])\n")
    env = dict(os.environ, DFDB_JIT_CACHE="1", DFDB_JIT_CACHE_DIR=str(tmp_path))

    def run():
        p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
        assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
        line = [l for l in p.stdout.decode().splitlines() if l.startswith("RESULT")][0].split()
        return [int(x) for x in line[1:]]
    n1, compiled1, disk1, ran1 = run()
    files = [f for f in os.listdir(tmp_path) if f.endswith(".co")]
    assert compiled1 >= 1 and disk1 == 0 and ran1 >= 1 and len(files) == compiled1
    n2, compiled2, disk2, ran2 = run()
    assert n2 == n1 and compiled2 == 0 and disk2 == compiled1 and ran2 >= 1
    victim = os.path.join(tmp_path, files[0])
    with open(victim, "r+b") as f:
        f.truncate(os.path.getsize(victim) // 2)
    n3, compiled3, disk3, ran3 = run()
    assert n3 == n1 and compiled3 == 1 and disk3 == compiled1 - 1 and ran3 >= 1
