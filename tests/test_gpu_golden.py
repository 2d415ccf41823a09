"""GPU: the HIP engine against the committed golden vectors (the reference tests' known answers), both for
in-memory columns and for tables written in the reference's on-disk format and LZ4-decoded on the device."""
import numpy as np
import pytest

import golden_cases as G
from helpers import Pair, apply_stages, assert_same

pytestmark = pytest.mark.gpu
CASES = G.load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_engine_matches_reference_known_answers(oracle, dfdb_mod, ctx, case, tmp_path):
    cols = G.build_columns(case["table"])
    names = list(cols.keys())
    for k, bs in enumerate(case["block_sizes"]):
        via = str(tmp_path / f"t{k}") if k == 0 else None          # first block size goes through files + device LZ4
        p = Pair(oracle, dfdb_mod, cols, block_size=bs, via_files=via)
        ov, dv = apply_stages(p, G.stages_for(case, names), G.proj_for(case, names))
        q = dv._query()
        assert q.indices().tolist() == case["expect_rows"], f"{case['name']} ({case['ref']}) block_size={bs}"
        assert q.count() == len(case["expect_rows"])
        G.check_columns(case, names, q.materialize(), oracle.flat_to_strings)
        assert_same(p, ov, dv)                                      # and bit-exact against the oracle
