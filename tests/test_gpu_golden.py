"""GPU: the HIP engine against the committed golden vectors (the reference tests' known answers), both for
in-memory columns and for tables written in the reference's on-disk format and LZ4-decoded on the device."""
import os

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


def test_engine_reads_and_rewrites_the_hand_assembled_format(oracle, dfdb_mod, ctx, tmp_path):
    """tests/golden/format_v1 (round 6): meta.bin + an Int64 and a Missing(String) column file assembled with struct.pack + liblz4 from the Julia lines that
    define the format, through neither writer of this repo.  The ENGINE's reader (dfdb_table_open + K7 + the block-body unpackers) gives the values the
    script started from — resident and block-streamed —, and the ENGINE's writer, given those values, produces the same meta.bin and column headers byte for
    byte, the same {rows, origin} in every block header, and LZ4 blocks (its own compressor's) that liblz4 decodes to the fixture's block bodies."""
    import json
    import struct
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = json.load(open(os.path.join(g, "format_v1.json")))
    files = {n: open(os.path.join(g, "format_v1", n), "rb").read() for n in ("meta.bin", "1.bin", "2.bin")}
    path = os.path.join(g, "format_v1")
    for load in (True, False):                                     # resident, then out of core (csrc/ooc.cpp streams the same files)
        t = dfdb_mod.open_table(path, load=load)
        try:
            assert t.blocksize == spec["block_size"] and t.names() == ["a", "s"] and [m.id for m in t.columns_meta()] == [1, 2]
            assert [m.type for m in t.columns_meta()] == ["Int64", "Missing(String)"]
            df = dfdb_mod.materialize(t)
            assert df["a"].tolist() == spec["a"] and [None if v is None else v for v in df["s"].tolist()] == spec["s"]
            assert dfdb_mod.nrow(t[("a", lambda a: a > 4), dfdb_mod.ALL]) == sum(1 for v in spec["a"] if v > 4)
        finally:
            t.close()
    w = dfdb_mod.DFTable.from_columns({"a": np.array(spec["a"], np.int64), "s": spec["s"]}, block_size=spec["block_size"])
    out = str(tmp_path / "engine_written")
    w.save(out)
    w.close()
    assert open(os.path.join(out, "meta.bin"), "rb").read() == files["meta.bin"]
    for n in ("1.bin", "2.bin"):
        got, want = open(os.path.join(out, n), "rb").read(), files[n]
        tylen = struct.unpack_from("<i", want, 8)[0]
        head = 12 + tylen
        assert got[:head] == want[:head], n                          # Int64 block size + the type string
        go, wo = head, head
        for k in range(3):
            gr, gorig, gcomp = struct.unpack_from("<iqq", got, go)
            wr, worig, wcomp = struct.unpack_from("<iqq", want, wo)
            assert (gr, gorig) == (wr, worig), (n, k)
            gbody = oracle.block_decode(got, go)[1]
            wbody = oracle.block_decode(want, wo)[1]
            assert gbody == wbody, (n, k)
            go += 20 + gcomp; wo += 20 + wcomp
        assert go == len(got) and wo == len(want)
