"""Compressed-resident columns (keep_compressed = 1): the LZ4 blocks beside the decoded array, the sequence-start index, decode statuses.
(re-filed by component in round 6 from the round-named files; no test body changed)"""


import numpy as np
import pytest


pytestmark = pytest.mark.gpu


def test_reloading_a_column_forgets_its_compressed_blocks(oracle, dfdb_mod, ctx, tmp_path):
    """a column loaded with keep_compressed = 1 and loaded AGAIN without it (dfdb_table_load_image over a resident column) must not keep
    the first load's LZ4 descriptors: dfdb_table_decode_resident would decode the old blocks into the new array"""
    from helpers import Pair
    x = oracle.gen_i64(0x5151, 0, 70_000)
    y = oracle.gen_i64(0x7777, 0, 70_000)
    Pair(oracle, dfdb_mod, {"x": x}, block_size=4096, via_files=str(tmp_path / "tx")).d.close()
    Pair(oracle, dfdb_mod, {"x": y}, block_size=4096, via_files=str(tmp_path / "ty")).d.close()
    image_y = open(str(tmp_path / "ty" / "1.bin"), "rb").read()            # `<id>.bin`: header + blocks of the one column
    t = dfdb_mod.open_table(str(tmp_path / "tx"), load=False)
    ctx.set_option("keep_compressed", 1)
    try:
        t.load(["x"])
        t.decode_resident("x")                                   # the blocks of the first load are there
        assert np.array_equal(t.view()._query().materialize()[0], x)
        ctx.set_option("keep_compressed", 0)
        t.load_image("x", image_y)                                # the same column, other bytes, nothing kept this time
        with pytest.raises(ValueError, match="holds no compressed blocks"):
            t.decode_resident("x")
        assert np.array_equal(t.view()._query().materialize()[0], y)
    finally:
        ctx.set_option("keep_compressed", 0)
        t.close()


@pytest.mark.parametrize("pipe", [0, 1])
@pytest.mark.parametrize("variant", [0, 1])
def test_lz4_sequence_index_changes_no_byte(oracle, dfdb_mod, tmp_path, variant, pipe):
    """A column that keeps its LZ4 blocks in HBM (ctx option keep_compressed; BlockStreams.jl:101-119 is what every decode restates) records where its
    sequences start during its first resident decode and decodes with that index afterwards (k_decode.hip INDEX; ctx option lz4_index, default 1).
    Every corner-case body of test_lz4_decode_corner_cases, files written by liblz4 (the oracle's writer) and by the device encoder, at block sizes that
    leave ragged last blocks: the plain decode (lz4_index = 0), the recording decode and the indexed decodes — alone and fused with a predicate — all
    leave exactly the bytes liblz4 decodes."""
    from test_gpu_parity import lz4_corner_columns
    from helpers import Pair
    n = 200_000
    cols = lz4_corner_columns(variant, n)
    c = dfdb_mod.Context(0)
    try:
        c.set_option("lz4_pipeline", pipe)              # 0: one wave per block; 1: the two-wave pipeline (what these small files get by default), whose parser reads the index and fetches the far sources (the recording launch is one wave per block either way)
        c.set_option("keep_compressed", 1)
        for writer in ("liblz4", "device"):
            for bs in (65536, 8192, 4099):
                d = str(tmp_path / f"ix{writer}{bs}")
                if writer == "liblz4":
                    ot = oracle.Table(block_size=bs)
                    for k, v in cols.items():
                        ot.add_column(k, v)
                    ot.save(d)
                else:
                    wt = dfdb_mod.DFTable.from_columns(cols, block_size=bs, ctx=c)
                    wt.save(d); wt.close()
                t = dfdb_mod.open_table(d, ctx=c)
                for name, want in cols.items():
                    c.set_option("lz4_index", 0)
                    c.profile(True)
                    t.decode_resident(name)
                    assert np.array_equal(dfdb_mod.materialize(t[dfdb_mod.ALL, [name]])[name].to_numpy(), want), (writer, bs, name, "plain")
                    c.set_option("lz4_index", 1)
                    for k in range(3):
                        t.decode_resident(name)
                        assert t.decode_status(name) == 0, (writer, bs, name, "index", k)          # every block ended on its stored size
                        assert np.array_equal(dfdb_mod.materialize(t[dfdb_mod.ALL, [name]])[name].to_numpy(), want), (writer, bs, name, "index", k)
                    got = {k: c.profile_get("lz4_decode." + k)[0] for k in ("plain", "recording", "indexed")}
                    c.profile(False)
                    assert got == {"plain": 1, "recording": 1, "indexed": 2}, got
                t.close()
        # fused with a predicate (K7 SCAN), 8-byte view of the same bytes: the first fused decode records, the later ones read the index
        c.set_option("decode_on_scan", 1)
        for name in ("mixed", "shortseq", "periodic", "runs"):
            v8 = np.ascontiguousarray(cols[name][: n // 8 * 8]).view(np.int64)
            d = str(tmp_path / f"ix8{name}")
            ot = oracle.Table(block_size=8192); ot.add_column("v", v8); ot.save(d)
            t = dfdb_mod.open_table(d, ctx=c)
            med = int(np.median(v8))
            c.profile(True)
            for k in range(3):
                q = t[t.v > med, dfdb_mod.ALL]._query()
                assert np.array_equal(q.indices(), np.flatnonzero(v8 > med).astype(np.int64) + 1), (name, k)
                assert t.decode_status("v") == 0, (name, k)
                assert np.array_equal(dfdb_mod.materialize(t)["v"].to_numpy(), v8), (name, k)
            fam = "lz4_decode." if pipe == 1 else "lz4_decode_scan."           # (the pipeline decodes, then the ordinary scan runs: few blocks)
            got = {k: c.profile_get(fam + k)[0] for k in ("plain", "recording", "indexed")}
            c.profile(False)
            assert got == {"plain": 0, "recording": 1, "indexed": 2}, got
            t.close()
    finally:
        c.close()


def test_decode_status_reports_what_the_resident_decode_said(oracle, dfdb_mod, tmp_path):
    """dfdb_table_decode_status: 0 bad blocks after a resident decode of valid blocks (BlockStreams.jl:112's assertion holds for each); a column that
    kept no blocks is an ArgumentError, an ordinal out of range a KeyError."""
    n = 150_000
    x = (np.arange(n, dtype=np.int64) * 7919) % 1000
    ot = oracle.Table(block_size=4096); ot.add_column("x", x); ot.add_column("s", ["r%d" % (i % 7) for i in range(n)]); ot.save(str(tmp_path / "t"))
    c = dfdb_mod.Context(0)
    try:
        c.set_option("keep_compressed", 1)
        t = dfdb_mod.open_table(str(tmp_path / "t"), ctx=c)
        for pipe in (0, 1):
            c.set_option("lz4_pipeline", pipe)
            for _ in range(2):
                t.decode_resident("x")
                assert t.decode_status("x") == 0
        with pytest.raises(ValueError, match="holds no compressed blocks"):      # DFDB_ERR_ARGUMENT -> ArgumentError
            t.decode_status("s")                       # String columns keep none
        t.close()
        c.set_option("keep_compressed", 0)
        t2 = dfdb_mod.open_table(str(tmp_path / "t"), ctx=c)
        with pytest.raises(ValueError, match="holds no compressed blocks"):      # DFDB_ERR_ARGUMENT -> ArgumentError
            t2.decode_status("x")
        t2.close()
    finally:
        c.close()
