"""Union{String,Missing}: the device generator and three-valued equality.
(re-filed by component in round 6 from the round-named files; no test body changed)"""


import numpy as np
import pytest

from helpers import Pair, apply_stages, assert_same


pytestmark = pytest.mark.gpu


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


@pytest.mark.parametrize("n", [1, 1023, 70_001])
def test_nullable_string_generator_and_three_valued_equality(oracle, dfdb_mod, ctx, n):
    """DFDB_GEN_STR_BRANDS10_MISSING (the bench's Union{String,Missing} column, like the docs' real data set: docs/src/index.md:264-272): row i is missing when
    (h >> 32) mod 8 == 7, else brands10[h mod 10] — rebuilt here in numpy from the oracle's plain generator — and `s == "sony"` over it selects what the oracle
    selects over the same column (a comparison with missing is missing; the selection needs coalesce(., false): selection.jl:52-55 wants plain Bool)."""
    from dfdb import ir
    seed = 0x9E3779B97F4A7C15
    t = dfdb_mod.DFTable.new(block_size=4096, ctx=ctx)
    t.add_generated("s", dfdb_mod.GEN_STR_BRANDS10_MISSING, seed, n)
    t.add_generated("a", dfdb_mod.GEN_I64_MOD1M, seed, n)
    with np.errstate(over="ignore"):
        h = _splitmix64(np.uint64(seed) + np.arange(n, dtype=np.uint64))
    miss = ((h >> np.uint64(32)) & np.uint64(7)) == np.uint64(7)
    plain = oracle.flat_to_strings(*oracle.gen_str(seed, 0, n))
    want = [None if m else s for s, m in zip(plain, miss.tolist())]
    got = dfdb_mod.materialize(t[dfdb_mod.ALL, ["s"]])["s"].tolist()
    assert [g if isinstance(g, str) else None for g in got] == want
    p = Pair(oracle, dfdb_mod, {"s": want, "a": oracle.gen_i64(seed, 0, n)}, block_size=4096)
    pred = ir.coalesce(ir.col(0) == "sony", False)
    ov, dv = apply_stages(p, [("pred", pred)])
    assert_same(p, ov, dv)
    dq = t[pred, dfdb_mod.ALL]._query()
    assert np.array_equal(dq.indices(), ov.select_indices())
    # coalesce(<string term>, false) over a nullable String column is K5's own answer (a missing row selects nothing), not an interpreter program: every
    # term kind, the empty pattern (which every NON-missing row matches), a long pattern, and a conjunction with a numeric term
    ctx.profile(True)
    for term in (ir.col(0) != "sony", ir.startswith(ir.col(0), "s"), ir.endswith(ir.col(0), "y"), ir.col(0) == "", ir.col(0) != "", ir.startswith(ir.col(0), ""),
                 ir.col(0) == "a-pattern-that-is-longer-than-sixteen-bytes", ir.col(0) != "a-pattern-that-is-longer-than-sixteen-bytes"):
        ov, dv = apply_stages(p, [("pred", ir.coalesce(term, False))])
        assert_same(p, ov, dv)
        ov, dv = apply_stages(p, [("pred", ir.coalesce(term, False) & (ir.col(1) > 300_000))], proj=[("s", ir.col(0)), ("a", ir.col(1))])
        assert_same(p, ov, dv)
    assert ctx.profile_get("str_match")[0] >= 16 and ctx.profile_get("interp_predicate")[0] + ctx.profile_get("jit_predicate")[0] == 0
    ctx.profile(False)
    # ismissing counts (docs/src/index.md:326-328)
    assert t[ir.ismissing(ir.col(0)), dfdb_mod.ALL]._query().count() == int(miss.sum())
    t.close()
