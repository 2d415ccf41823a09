"""GPU: the multi-GPU group API of include/dfdb.h (block-range shards + RCCL / host exchange) against the oracle's single table.

A 1-GPU box cannot hold two RCCL ranks on distinct devices, so the sharded paths are driven three ways:
  * one-process groups whose shards all live on device 0 with DFDB_EXCHANGE_HOST (every code path of group.cpp except the
    two RCCL calls: worker threads, block-range loads, stage-base planning, rank-order concatenation, aggregates),
  * a one-rank group with DFDB_EXCHANGE_RCCL (dlopen of librccl, ncclCommInitAll / ncclCommInitRank, all-reduce and all-gather
    really issued on the engine stream),
  * two PROCESSES on device 0 over torch.distributed gloo running the real engine per rank (dfdb/sharding.py), and bench.py's
    own launcher.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import Pair  # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def col_seed(k):
    return (0x9E3779B97F4A7C15 + 0x1234567 * k) & 0xFFFFFFFFFFFFFFFF


def _columns(oracle, n):
    sizes, data = oracle.gen_str(col_seed(2), 0, n)
    return {"a": oracle.gen_i64(col_seed(0), 0, n), "x": oracle.gen_f64(col_seed(1), 0, n), "s": oracle.flat_to_strings(sizes, data)}


def _stage_sets(ir):
    a, x, s = ir.col(0), ir.col(1), ir.col(2)
    return {
        "pred": [("pred", (a > 700_000) & (s != "sony"))],
        "lead_range": [("range", 5, 7, 250_000), ("pred", x < 1000.0)],
        "range_after_pred": [("pred", a > 500_000), ("range", 11, 3, 90_000)],
        "two_exchanges": [("pred", a % 2 == 0), ("range", 10, 1, 100_000), ("pred", s == "dell"), ("idx", [1, 5, 400, 4999, 10**9])],
        "integer_after_pred": [("pred", a > 999_000), ("int", 17)],
        "nothing_selected": [("pred", a > 2_000_000)],
        "count_all": [],
    }


def _build(dfdb, ov, gv, stages):
    for st in stages:
        if st[0] == "pred":
            ov.add_predicate(st[1].to_ir()); gv = dfdb.selection(gv, st[1])
        elif st[0] == "range":
            ov.add_range(st[1], st[2], st[3]); gv = dfdb.selection(gv, dfdb.jr(st[1], st[2], st[3]))
        elif st[0] == "int":
            ov.add_integer(st[1]); gv = dfdb.selection(gv, st[1])
        else:
            ov.add_indices(st[1]); gv = dfdb.selection(gv, list(st[1]))
    return ov, gv


def _check_group_against_oracle(oracle, dfdb, ot, gt, cols, n):
    from dfdb import ir, group as G, _native as N
    for name, stages in _stage_sets(ir).items():
        ov, gv = _build(dfdb, ot.view(), gt.view(), stages)
        want_idx = ov.select_indices()
        assert G.gnrow(gv) == ov.nrow() == len(want_idx), name
        assert np.array_equal(G.gindices(gv), want_idx), name
        assert sum(G._gq(gv).shard_counts()) == len(want_idx), name
        want = ov.materialize()
        got = G._gq(gv).materialize()
        assert np.array_equal(got[0], want[0]), name
        assert np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64)), name
        assert np.array_equal(got[2][0], want[2][0]) and np.array_equal(got[2][1], want[2][1]), name
        sel = want_idx - 1
        # aggregates: Int64 exact (wrapping sums are associative), Float64 sum within n*eps*sum|x|
        av = gv[dfdb.ALL, "a"]
        assert G.gaggregate(av, N.AGG_SUM) == int(cols["a"][sel].sum()), name
        assert G.gaggregate(av, N.AGG_COUNT) == len(sel), name
        xv = gv[dfdb.ALL, "x"]
        xs = cols["x"][sel]
        assert abs(G.gaggregate(xv, N.AGG_SUM) - float(xs.sum())) <= max(1, n) * np.finfo(float).eps * float(np.abs(xs).sum()) + 0.0, name
        if len(sel):
            assert G.gaggregate(av, N.AGG_MIN) == int(cols["a"][sel].min()) and G.gaggregate(av, N.AGG_MAX) == int(cols["a"][sel].max()), name
            assert G.gaggregate(xv, N.AGG_MIN) == float(xs.min()) and G.gaggregate(xv, N.AGG_MAX) == float(xs.max()), name
        else:
            with pytest.raises(ValueError, match="empty collection"):
                G.gaggregate(av, N.AGG_MIN)


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_group_host_exchange_files_vs_oracle(oracle, dfdb_mod, ctx, tmp_path, world):
    """world shards of one file-backed table on device 0 (world = 5 over 74 blocks; the last shard is short)"""
    from dfdb import group as G, _native as N
    n, bs = 300_007, 4096
    cols = _columns(oracle, n)
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "tb")
    ot.save(path)
    g = G.Group.create([0] * world, N.EXCHANGE_HOST if world > 1 else N.EXCHANGE_AUTO)
    try:
        assert (g.world, g.nlocal, g.first_rank) == (world, world, 0)
        gt = G.GroupTable.open(g, path)
        assert gt.nrows == n
        rows = [gt.shard(l).view()._query().count() for l in range(world)]
        assert sum(rows) == n and all(r % bs == 0 for r in rows[:-1])
        _check_group_against_oracle(oracle, dfdb_mod, ot, gt, cols, n)
        gt.close()
    finally:
        g.close()


def test_group_more_shards_than_blocks_and_host_columns(oracle, dfdb_mod, ctx):
    """3 blocks over 5 shards: shards 3 and 4 hold zero rows; columns come from host arrays split by block range"""
    from dfdb import group as G, _native as N
    n, bs = 2 * 4096 + 17, 4096
    cols = _columns(oracle, n)
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    g = G.Group.create([0] * 5, N.EXCHANGE_HOST)
    try:
        gt = G.GroupTable.from_columns(g, cols, block_size=bs)
        assert [gt.shard(l).view()._query().count() for l in range(5)] == [4096, 4096, 17, 0, 0]
        _check_group_against_oracle(oracle, dfdb_mod, ot, gt, cols, n)
        gt.close()
    finally:
        g.close()


def test_group_generated_columns_match_single_table(oracle, dfdb_mod, ctx):
    """dfdb_group_table_add_generated: every shard generates the rows of its own block range (row_first = its first row)"""
    from dfdb import group as G, _native as N, ir
    n = 5 * 65536 + 123
    g = G.Group.create([0, 0, 0], N.EXCHANGE_HOST)
    try:
        gt = G.GroupTable.new(g)
        gt.add_generated("x", dfdb_mod.GEN_I64_MOD1M, col_seed(0), n)
        gt.add_generated("f", dfdb_mod.GEN_F64_U2000, col_seed(1), n)
        x = oracle.gen_i64(col_seed(0), 0, n)
        f = oracle.gen_f64(col_seed(1), 0, n)
        v = gt.view()[(ir.col(0) > 899_999) & (ir.col(1) < 1000.0), dfdb_mod.ALL]
        m = (x > 899_999) & (f < 1000.0)
        assert G.gnrow(v) == int(m.sum())
        assert np.array_equal(G.gindices(v), np.nonzero(m)[0] + 1)
        got = G._gq(v).materialize()
        assert np.array_equal(got[0], x[m]) and np.array_equal(got[1].view(np.uint64), f[m].view(np.uint64))
        # an async count (no host wait) followed by the read
        q = G._gq(v)
        q.reset(); q.count_async()
        assert q.count() == int(m.sum())
        gt.close()
    finally:
        g.close()


def test_group_rccl_single_rank(oracle, dfdb_mod, ctx):
    """DFDB_EXCHANGE_RCCL with one rank: librccl is dlopen'ed, ncclCommInitAll / ncclCommInitRank run, and the all-reduce /
    all-gather of count, aggregates and stage bases are really issued on the engine stream (world = 1 makes them copies)"""
    from dfdb import group as G, _native as N, ir
    n = 200_000
    x = oracle.gen_i64(col_seed(0), 0, n)
    for make in (lambda: G.Group.create([0], N.EXCHANGE_RCCL), lambda: G.Group.create_rank(0, None, 0, 1)):
        g = make()
        try:
            gt = G.GroupTable.from_columns(g, {"x": x}, block_size=4096)
            v = gt.view()[ir.col(0) > 500_000, dfdb_mod.ALL]
            m = x > 500_000
            assert G.gnrow(v) == int(m.sum())
            assert G.gaggregate(v, N.AGG_SUM) == int(x[m].sum()) and G.gaggregate(v, N.AGG_MAX) == int(x[m].max())
            v2 = dfdb_mod.selection(v, dfdb_mod.jr(3, 2, 1000))
            assert np.array_equal(G.gindices(v2), (np.nonzero(m)[0] + 1)[2:1000:2])
            g.barrier()
            assert g.allreduce([[1.5, -2.0]], N.AGG_MAX) == [[1.5, -2.0]]
            gt.close()
        finally:
            g.close()
    uid = G.Group.unique_id()
    assert len(uid) == N.GROUP_ID_BYTES and any(uid)


def test_group_rejects_bad_arguments(dfdb_mod, ctx):
    from dfdb import group as G, _native as N
    with pytest.raises(ValueError, match="distinct"):
        G.Group.create([0, 0], N.EXCHANGE_RCCL)
    with pytest.raises(ValueError):
        G.Group.create([], N.EXCHANGE_AUTO)
    with pytest.raises(ValueError, match="out of range"):
        G.Group.create([0, 99], N.EXCHANGE_HOST)
    g = G.Group.create([0, 0], N.EXCHANGE_HOST)
    try:
        gt = G.GroupTable.from_columns(g, {"a": np.arange(10, dtype=np.int64)}, block_size=4)
        with pytest.raises(ValueError, match="rows"):
            gt.add_column("b", np.arange(11, dtype=np.int64))
        v = gt.view()
        with pytest.raises(IndexError):                     # range[range] out of bounds is rejected on every shard alike
            G.gnrow(dfdb_mod.selection(dfdb_mod.selection(v, dfdb_mod.jr(1, 1, 5)), dfdb_mod.jr(1, 1, 9)))
        gt.close()
    finally:
        g.close()


# ---------------------------------------------------------------- two processes, one device, gloo: the torch.distributed path
_RANK_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "dataframedbs.jl_amd"))
import numpy as np, torch, torch.distributed as dist
import dfdb
from dfdb import ir, sharding
rank, world, path = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = dfdb.Context(0)
t = dfdb.open_table(path, ctx=ctx, load=False)
nrows = int(dfdb.table_stats(t)["rows"].iloc[0])
nb = -(-nrows // t.blocksize)
b0, b1 = sharding.block_range(nb, rank, world)
t.load(None, b0, b1)
a, x, s = ir.col(0), ir.col(1), ir.col(2)
views = {
  "pred": dfdb.selection(t.view(), (a > 700_000) & (s != "sony")),
  "range_after_pred": dfdb.selection(dfdb.selection(t.view(), a > 500_000), dfdb.jr(11, 3, 90_000)),
  "two_exchanges": dfdb.selection(dfdb.selection(dfdb.selection(dfdb.selection(t.view(), a % 2 == 0), dfdb.jr(10, 1, 100_000)), s == "dell"), [1, 5, 400, 4999, 10**9]),
}
out = {}
for name, v in views.items():
    total = sharding.sharded_count(v)
    out[name] = dict(total=total, idx=v._query().indices().tolist())
json.dump(out, open(sys.argv[3] + f".{rank}", "w"))
dist.destroy_process_group()
"""


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_processes_gloo_real_engine(oracle, dfdb_mod, ctx, tmp_path):
    """world_size 2, both ranks on device 0, gloo: each rank loads its block range into the real engine and
    dfdb/sharding.py's exchanges (all-gather + exclusive scan, all-reduce) give the oracle's single-table answer"""
    from dfdb import ir
    n, bs = 300_007, 4096
    cols = _columns(oracle, n)
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "tb")
    ot.save(path)
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, path, str(tmp_path / "out")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = [json.load(open(str(tmp_path / "out") + f".{r}")) for r in range(2)]
    a, x, s = ir.col(0), ir.col(1), ir.col(2)
    want = {
        "pred": ot.view().add_predicate(((a > 700_000) & (s != "sony")).to_ir()),
        "range_after_pred": ot.view().add_predicate((a > 500_000).to_ir()).add_range(11, 3, 90_000),
        "two_exchanges": ot.view().add_predicate((a % 2 == 0).to_ir()).add_range(10, 1, 100_000).add_predicate((s == "dell").to_ir()).add_indices([1, 5, 400, 4999, 10**9]),
    }
    for name, ov in want.items():
        w = ov.select_indices().tolist()
        assert got[0][name]["idx"] + got[1][name]["idx"] == w, name            # rank order = table order
        assert got[0][name]["total"] == got[1][name]["total"] == len(w), name


def test_bench_self_launches_ranks(ctx):
    """`python bench.py --gpus 2` with WORLD_SIZE unset spawns its own two ranks (here both on device 0 over gloo) and reports
    n_gpus = 2; the aggregate count is the sum of both shards"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--all-on-device0", "--backend", "gloo", "--rows", "20000000",
                        "--steps", "3", "--warmup", "1", "--no-cpu", "--config-scale", "0.004"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak"
    assert r["config"]["global_selected"] > 2 * 0.09 * 20_000_000
    assert r["value"] > 0
    # round 3: the same line carries configs 3 / 4 / 5 at every N (here 0.4 % of their sizes, both ranks on device 0, the exchange over gloo)
    cf = r["configs"]
    for k in ("3", "4", "4_dictionary", "5_shard", "5_shard_dictionary", "5_shard_materialize"):
        assert "error" not in cf[k], cf[k]
        assert cf[k]["ms_per_step"] > 0 and cf[k]["rows_per_s"] > 0 and 0 < cf[k]["roofline"]["frac"] < 2 and cf[k]["kernels_avg_ms"], (k, cf[k])
    n5 = cf["5_shard"]["rows_per_gpu"]
    assert cf["5_shard"]["total_rows"] == 2 * n5 and 0.05 * 2 * n5 < cf["5_shard"]["global_count"] < 0.12 * 2 * n5
    assert "dfdb_group_create_rank_callbacks" in cf["5_shard"]["exchange"]          # the LIBRARY's group path, its exchanges through gloo
    assert "library defaults" in r["config"]["options"] and r["config"]["placement_calibration"] == "off"      # `value` is the default configuration
    assert "calibrated_config" not in r


def _is_sony(sz, by):
    """rows of a flat string column (sizes, bytes) that hold exactly "sony" """
    off = np.concatenate([[0], np.cumsum(sz)])[:-1]
    out = np.asarray(sz) == 4
    idx = off[out]
    ok = (by[idx] == ord("s")) & (by[idx + 1] == ord("o")) & (by[idx + 2] == ord("n")) & (by[idx + 3] == ord("y"))
    out[np.nonzero(out)[0][~ok]] = False
    return out


@pytest.mark.parametrize("exchange", ["torch", "lib"])
def test_bench_eight_ranks_on_one_device(oracle, ctx, exchange):
    """what the driver's 8-GPU run will do, on the one GPU there is — `bench.py --gpus 8` spawns eight ranks (all on device 0, gloo), the
    block count of the whole table is not divisible by 8, every config leg is present at every rank count, and the global counts equal the oracle's for the
    same rows; with --exchange lib the headline's per-step all-reduce goes through the library's own group (its exchanges from the host's collectives here,
    RCCL between eight devices on a real node)."""
    rows = 20_000_000
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--all-on-device0", "--backend", "gloo", "--rows", str(rows), "--steps", "3", "--warmup", "1",
           "--no-cpu", "--config-scale", "0.002"] + (["--exchange", "lib"] if exchange == "lib" else [])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["steps"] == 3 and r["scaling"] == "weak" and r["value"] > 0
    total = 8 * rows
    assert (-(-total // 65536)) % 8 != 0                                       # the last rank's block range is shorter
    want = int((oracle.gen_i64(0x9E3779B97F4A7C15, 0, total) > 899_999).sum())
    assert r["config"]["global_selected"] == want
    cf = r["configs"]
    for k in ("3", "3_computed", "4", "4_dictionary", "5_shard", "5_shard_dictionary", "5_shard_materialize", "unique", "unique_hash_table", "groupreduce", "groupreduce_int_key", "nullable_string_eq"):
        assert k in cf and "error" not in cf[k], (k, cf.get(k), {n: v for n, v in cf.items() if isinstance(v, dict) and "error" in v})
    assert "error" not in cf["interp"], cf["interp"]
    n5 = cf["5_shard"]["rows_per_gpu"]
    a = oracle.gen_i64(0x9E3779B97F4A7C15, 0, 8 * n5); x = oracle.gen_f64((0x9E3779B97F4A7C15 * 2) & 0xFFFFFFFFFFFFFFFF, 0, 8 * n5)
    sz, by = oracle.gen_str((0x9E3779B97F4A7C15 * 3) & 0xFFFFFFFFFFFFFFFF, 0, 8 * n5)
    sony = _is_sony(sz, by)
    sel = (a > 683_771) & (x < 632.456) & ~sony
    assert cf["5_shard"]["total_rows"] == 8 * n5 and cf["5_shard"]["global_count"] == int(sel.sum())
    assert abs(cf["5_shard"]["global_sum_x"] - float(x[sel].sum())) <= 64 * np.finfo(np.float64).eps * float(np.abs(x[sel]).sum())
    if exchange == "lib":
        assert "libdfdb_hip's group" in r["config"]["sharding"]


def test_bench_threads_mode_one_process_drives_the_shards(oracle, ctx):
    """`bench.py --mode threads` — ONE process, dfdb_group_create over N devices, a host worker thread per shard: what a Julia session gets
    and what no torch.distributed launch exercises.  On a 1-GPU box the three shards share device 0 and exchange through the host; the global count and config
    5's count / sum equal the oracle's."""
    rows = 10_000_000
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "threads", "--gpus", "3", "--all-on-device0", "--rows", str(rows), "--steps", "3", "--warmup", "1",
                        "--config-scale", "0.002"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    r = json.loads(lines[0])
    assert r["n_gpus"] == 3 and r["value"] > 0 and "one process" in r["config"]["launcher"] and r["roofline"]["frac"] > 0
    assert r["config"]["global_selected"] == int((oracle.gen_i64(0x9E3779B97F4A7C15, 0, 3 * rows) > 899_999).sum())
    cf = r["configs"]
    for k in ("5_shard", "5_shard_materialize", "5_shard_dictionary"):
        assert "error" not in cf[k], cf[k]
    n5 = cf["5_shard"]["rows_per_gpu"]
    assert cf["5_shard"]["total_rows"] == 3 * n5
    a = oracle.gen_i64(0x9E3779B97F4A7C15, 0, 3 * n5); x = oracle.gen_f64((0x9E3779B97F4A7C15 * 2) & 0xFFFFFFFFFFFFFFFF, 0, 3 * n5)
    sz, by = oracle.gen_str((0x9E3779B97F4A7C15 * 3) & 0xFFFFFFFFFFFFFFFF, 0, 3 * n5)
    sony = _is_sony(sz, by)
    sel = (a > 683_771) & (x < 632.456) & ~sony
    assert cf["5_shard"]["global_count"] == int(sel.sum())


def test_bench_config_legs_through_the_library_group(oracle, ctx):
    """N = 1: the config legs at 0.4 % of BASELINE.json's sizes.  Config 5 runs twice through the library's own group path — a one-rank RCCL group
    (the RCCL all-reduce of {sum, count} issued for real) and, with --config5-host-shards 3, a one-process host-exchange group of three shards — and
    both must report the count and the sum of x the ORACLE gets for the same rows (sum within the Float64 tolerance of DESIGN.md section 5)."""
    from dfdb import ir
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    outs = []
    for extra in ([], ["--config5-host-shards", "3"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "5000000", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-decode-leg",
                            "--config-scale", "0.004"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
        outs.append(json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0]))
    n = int(1_250_000_000 * 0.004)
    S = 0x9E3779B97F4A7C15
    sd = lambda k: (S * (k + 1)) & 0xFFFFFFFFFFFFFFFF
    a, x = oracle.gen_i64(sd(0), 0, n), oracle.gen_f64(sd(1), 0, n)
    sz, by = oracle.gen_str(sd(2), 0, n)
    off = np.concatenate([[0], np.cumsum(sz)])
    sony = np.array([sz[i] == 4 and bytes(by[off[i]:off[i] + 4]) == b"sony" for i in range(n)])
    sel = (a > 683_771) & (x < 632.456) & ~sony
    want_n, want_s = int(sel.sum()), float(x[sel].sum())
    for r, label in zip(outs, ("rccl", "host")):
        cf = r["configs"]
        for k in ("5_shard", "5_shard_dictionary", "5_shard_materialize", "3", "4", "4_dictionary"):
            assert "error" not in cf[k], (label, k, cf[k])
        for k in ("5_shard", "5_shard_dictionary"):
            assert cf[k]["global_count"] == want_n, (label, k, cf[k]["global_count"], want_n)
            assert abs(cf[k]["global_sum_x"] - want_s) <= 64 * np.finfo(np.float64).eps * float(np.abs(x[sel]).sum()), (label, k)
        assert cf["5_shard_materialize"]["selected_per_gpu"] == want_n
        assert ("RCCL" in cf["5_shard"]["exchange"]) == (label == "rccl")


@pytest.mark.parametrize("world,dictionary", [(1, 4096), (3, 4096), (5, 0)])
def test_group_unique_and_groupreduce_match_the_single_table(oracle, dfdb_mod, ctx, world, dictionary):
    """unique / groupreduce over block-range shards (per-shard reduction on the device, one record per distinct key merged in rank order)
    against the same calls on ONE table holding every row: keys in order of first appearance in the whole table, counts, sums (Int64 sums
    wrap), min / max; selections with range stages that count across shards; a shard without rows; nullable and String keys, with the
    String column dictionary-coded or flat."""
    from dfdb import group as G, _native as N
    n, bs = 9 * 4096 + 123, 4096
    base = _columns(oracle, n)
    a = base["a"]
    rng = np.random.default_rng(5)
    cols = {"a": a, "x": base["x"], "s": base["s"],
            "k": (a % 37).astype(np.int16),
            "u": (a.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)),                      # UInt64 sums wrap
            "big": (a * 9_000_000_000_000 + 1),                                               # Int64 sums wrap
            "nk": np.ma.masked_array((a % 11).astype(np.int32), mask=(a % 7 == 0)),
            "ns": [None if v % 13 == 0 else "k%d" % (v % 19) for v in a.tolist()],
            "f": np.where(rng.random(n) < 0.01, np.nan, np.round(base["x"] / 100.0))}
    ctx.set_option("string_dictionary", dictionary)
    g = G.Group.create([0] * world, N.EXCHANGE_HOST if world > 1 else N.EXCHANGE_AUTO)
    g.set_option("string_dictionary", dictionary)
    try:
        t1 = dfdb_mod.DFTable.from_columns(cols, block_size=bs, ctx=ctx)
        gt = G.GroupTable.from_columns(g, cols, block_size=bs)
        for name, stages in _stage_sets(__import__("dfdb").ir).items():
            v1, gv = dfdb_mod.DFView(t1), gt.view()
            for st in stages:
                sel = st[1] if st[0] == "pred" else dfdb_mod.jr(st[1], st[2], st[3]) if st[0] == "range" else st[1] if st[0] == "int" else list(st[1])
                v1, gv = dfdb_mod.selection(v1, sel), dfdb_mod.selection(gv, sel)
            for key in ("k", "s", "nk", "ns", "f"):
                want, got = getattr(v1, key).unique(), G.gunique(getattr(gv, key))
                assert _same_keys(want, got), (name, key, want, got)
                for col, stat in ((None, "count"), ("a", "sum"), ("big", "sum"), ("u", "sum"), ("u", "max"), ("a", "min"), ("x", "max"), ("a", "mean")):
                    w = dfdb_mod.groupreduce(v1, key, col, stat)
                    r = G.ggroupreduce(gv, key, col, stat)
                    assert _same_keys(w[key].to_numpy(), r[key].to_numpy()), (name, key, stat)
                    assert (w["count"].to_numpy() == r["count"].to_numpy()).all(), (name, key, stat)
                    if stat != "count":
                        assert w[stat].dtype == r[stat].dtype and (w[stat].to_numpy() == r[stat].to_numpy()).all(), (name, key, col, stat, w, r)
                # Float64 sums: the sharded sum is a sum of per-shard sums — equal up to the reassociation
                w, r = dfdb_mod.groupreduce(v1, key, "x", "sum"), G.ggroupreduce(gv, key, "x", "sum")
                assert np.allclose(w["sum"].to_numpy(), r["sum"].to_numpy(), rtol=1e-12, atol=0.0), (name, key)
        gt.close(); t1.close()
    finally:
        g.close()
        ctx.set_option("string_dictionary", 0)


def _same_keys(want, got):
    def canon(v):
        if isinstance(v, np.ma.MaskedArray):
            return [None if m else x for x, m in zip(v.data.tolist(), np.ma.getmaskarray(v).tolist())]
        return ["nan" if isinstance(x, float) and x != x else x for x in (v.tolist() if isinstance(v, np.ndarray) else list(v))]
    return canon(want) == canon(got)


# ---------------------------------------------------------------- one process per GPU through the LIBRARY's group path, the host's own collectives
_CALLBACK_RANK_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "dataframedbs.jl_amd"))
import numpy as np, torch, torch.distributed as dist
import dfdb
from dfdb import ir, group as G, _native as N
rank, world, path = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
g = G.Group.create_rank_torch(0)                       # dfdb_group_create_rank_callbacks: allreduce / allgather over gloo
assert (g.world, g.nlocal, g.first_rank, g.exchange) == (world, 1, rank, N.EXCHANGE_CALLBACK)
gt = G.GroupTable.open(g, path)
a, x, s, z = ir.col(0), ir.col(1), ir.col(2), ir.col(3)
base = gt.view()
views = {
  "pred": dfdb.selection(base, (a > 700_000) & (s != "sony")),
  "range_after_pred": dfdb.selection(dfdb.selection(base, a > 500_000), dfdb.jr(11, 3, 90_000)),
  "two_exchanges": dfdb.selection(dfdb.selection(dfdb.selection(dfdb.selection(base, a % 2 == 0), dfdb.jr(10, 1, 100_000)), s == "dell"), [1, 5, 400, 4999, 10**9]),
}
out = {}
for name, v in views.items():
    o = dict(total=G.gnrow(v), idx=G.gindices(v).tolist())
    o["sum_a"] = G.gaggregate(v[dfdb.ALL, ["a"]], N.AGG_SUM); o["min_x"] = G.gaggregate(v[dfdb.ALL, ["x"]], N.AGG_MIN); o["max_x"] = G.gaggregate(v[dfdb.ALL, ["x"]], N.AGG_MAX)
    o["sum_x"] = G.gaggregate(v[dfdb.ALL, ["x"]], N.AGG_SUM)
    o["unique_s"] = list(G.gunique(v.s))
    gr = G.ggroupreduce(v, "s", "a", "sum")
    o["groups"] = [list(gr["s"]), [int(c) for c in gr["count"]], [int(c) for c in gr["sum"]]]
    m = G._gq(v[dfdb.ALL, ["a", "s"]]).materialize()
    o["mat_a"] = m[0].tolist(); o["mat_s_sizes"] = m[1][0].tolist()
    out[name] = o
# a DivideError that only the LAST rank's rows reach: every rank must come back from the same call with the same error, none may hang
bad = dfdb.selection(base, a % z == 0)
errs = []
for call in (lambda: G.gnrow(bad), lambda: G.gaggregate(bad[dfdb.ALL, ["a"]], N.AGG_SUM), lambda: G.gunique(bad.s), lambda: G.ggroupreduce(bad, "s", "a", "sum")):
    try:
        call(); errs.append("none")
    except Exception as e:
        errs.append(type(e).__name__)
q = G.GroupQuery(gt, bad)
try:
    q.count_async(); errs.append("none")                # enqueue only: the failing rank hears of it now ...
except Exception as e:
    errs.append(type(e).__name__)
try:
    q.count(); errs.append("none")                      # ... every rank here
except Exception as e:
    errs.append(type(e).__name__)
q.close()
out["errors"] = errs
out["after"] = G.gnrow(views["pred"])                   # the group is fine afterwards
g.barrier()
json.dump(out, open(sys.argv[3] + f".{rank}", "w"))
gt.close(); g.close()
dist.destroy_process_group()
"""


def test_three_processes_through_the_library_group_with_host_collectives(oracle, dfdb_mod, ctx, tmp_path):
    """One process per GPU, the mode the 8-GPU run uses, exercised for real on a 1-GPU box: three processes on device 0 each hold one rank of a
    dfdb_group whose exchanges go through the host's own collectives (dfdb_group_create_rank_callbacks over torch.distributed gloo; RCCL refuses ranks that
    share a device).  Everything csrc/group.cpp does between processes runs: the all-reduce of the count and of {value, count}, the all-gather + exclusive
    scan behind a range stage after a predicate, the all-gather of the packed unique / groupreduce records, rank-order results, and the fault agreement —
    a DivideError only the last rank's rows reach comes back from the same call on EVERY rank (the others used to wait in the collective)."""
    from dfdb import ir
    n, bs, world = 300_007, 4096, 3
    cols = _columns(oracle, n)
    z = np.ones(n, np.int64); z[n - 777] = 0                 # a zero divisor in the last rank's block range only
    cols = {"a": cols["a"], "x": cols["x"], "s": cols["s"], "z": z}
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "tb")
    ot.save(path)
    script = tmp_path / "rank_cb.py"
    script.write_text(_CALLBACK_RANK_SCRIPT)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, path, str(tmp_path / "out")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    got = [json.load(open(str(tmp_path / "out") + f".{r}")) for r in range(world)]
    a, x, s = ir.col(0), ir.col(1), ir.col(2)
    want = {
        "pred": ot.view().add_predicate(((a > 700_000) & (s != "sony")).to_ir()),
        "range_after_pred": ot.view().add_predicate((a > 500_000).to_ir()).add_range(11, 3, 90_000),
        "two_exchanges": ot.view().add_predicate((a % 2 == 0).to_ir()).add_range(10, 1, 100_000).add_predicate((s == "dell").to_ir()).add_indices([1, 5, 400, 4999, 10**9]),
    }
    strs = oracle.flat_to_strings(*cols["s"]) if isinstance(cols["s"], tuple) else list(cols["s"])
    for name, ov in want.items():
        w = ov.select_indices()
        rows = w - 1
        assert sum((g[name]["idx"] for g in got), []) == w.tolist(), name                       # rank order = table order
        for g in got:
            o = g[name]
            assert o["total"] == len(w), name
            assert o["sum_a"] == int(cols["a"][rows].sum()), name
            if len(rows):
                assert o["min_x"] == float(cols["x"][rows].min()) and o["max_x"] == float(cols["x"][rows].max()), name
                assert abs(o["sum_x"] - float(cols["x"][rows].sum())) <= 64 * np.finfo(np.float64).eps * float(np.abs(cols["x"][rows]).sum()), name
            sel = [strs[i] for i in rows.tolist()]
            first = list(dict.fromkeys(sel))                                                     # distinct values in order of first appearance
            assert o["unique_s"] == first, name
            assert o["groups"][0] == first and o["groups"][1] == [sel.count(k) for k in first], name
            assert o["groups"][2] == [int(sum(int(cols["a"][i]) for i in rows.tolist() if strs[i] == k)) for k in first], name
        assert sum((g[name]["mat_a"] for g in got), []) == cols["a"][rows].tolist(), name
    for g in got[:-1]:
        assert g["errors"] == ["ZeroDivisionError"] * 4 + ["none", "ZeroDivisionError"], g["errors"]        # the healthy ranks hear of it at their next read
    assert got[-1]["errors"] == ["ZeroDivisionError"] * 6, got[-1]["errors"]                              # the failing rank at once
    assert all(g["after"] == len(want["pred"].select_indices()) for g in got)


def test_bench_deadline_keeps_the_headline(ctx):
    """the config legs exchange between ranks; should one ever be left waiting in an exchange, a deadline ends every rank's process and rank 0 prints the
    line first — headline intact, the finished legs, and a note — instead of the run dying silently in the driver's timeout; the process then exits
    with status 3, so a hung exchange never looks like a clean run"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "5000000", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-decode-leg",
                        "--config-scale", "0.05", "--config-deadline", "0.05"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 3, (p.returncode, p.stderr.decode(errors="replace")[-2000:])      # the line is printed, and the exit status still says the run was cut short
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["value"] > 0 and r["roofline"]["frac"] > 0 and "did not finish within" in r["configs"]["error"]


def test_bench_exchange_lib_over_two_ranks(ctx):
    """`bench.py --gpus 2 --exchange lib` on one device over gloo: the HEADLINE step's count all-reduce goes through the library's group too (its exchanges
    from the host's collectives here, RCCL on a real node), every rank generating its own block range of the one 2-shard table"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--all-on-device0", "--backend", "gloo", "--exchange", "lib", "--rows", "10000000",
                        "--steps", "3", "--warmup", "1", "--no-cpu", "--no-configs"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and "libdfdb_hip's group" in r["config"]["sharding"]
    assert abs(r["config"]["global_selected"] / 2e7 - 0.1) < 0.002                       # both shards' survivors, reduced by the library


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_group_streams_its_block_ranges_when_the_shards_are_not_resident(oracle, dfdb_mod, ctx, tmp_path, world):
    """Round 6: a sharded table whose columns do not fit (dfdb_group_query_prepare -> 3) — or that was simply never loaded — is answered by every shard
    STREAMING its own block range of the column files (csrc/ooc.cpp behind group.cpp): count, indices, shard counts, materialize, aggregates, range
    stages that number the survivors of the lower ranks, unique and groupreduce == the oracle / the single resident table.  Chunks of 4 blocks over 74."""
    from dfdb import group as G, _native as N
    n, bs = 300_007, 4096
    cols = _columns(oracle, n)
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "tb")
    ot.save(path)
    g = G.Group.create([0] * world, N.EXCHANGE_HOST if world > 1 else N.EXCHANGE_AUTO)
    try:
        g.set_option("ooc_chunk_blocks", 4)
        g.set_option("hbm_budget_mb", 1)                    # nothing fits
        gt = G.GroupTable.open(g, path, load=False)
        assert gt.nrows == n
        gq = G._gq(gt[("a", lambda c: c > 700_000), dfdb_mod.ALL])
        assert gq.prepare() == 3
        for l in range(world):
            assert gt.shard(l).resident_bytes() == {"decoded": 0, "compressed": 0}
        _check_group_against_oracle(oracle, dfdb_mod, ot, gt, cols, n)
        for l in range(world):
            assert gt.shard(l).resident_bytes() == {"decoded": 0, "compressed": 0}, "a streamed shard keeps nothing"
        # unique / groupreduce: the shards' chunk-merged parts merged in rank order
        t1 = dfdb_mod.open_table(path, ctx=ctx)
        try:
            v1 = t1[("a", lambda c: c > 300_000), dfdb_mod.ALL]
            gv = gt[("a", lambda c: c > 300_000), dfdb_mod.ALL]
            want, got = v1.s.unique(), G.gunique(gv.s)
            assert list(want) == list(got)
            for col, stat in ((None, "count"), ("a", "sum"), ("x", "max"), ("a", "min")):
                w, r = dfdb_mod.groupreduce(v1, "s", col, stat), G.ggroupreduce(gv, "s", col, stat)
                assert list(w["s"]) == list(r["s"]) and np.array_equal(w["count"].to_numpy(), r["count"].to_numpy()), stat
                if stat != "count":
                    assert np.array_equal(w[stat].to_numpy(), r[stat].to_numpy()), stat
        finally:
            t1.close()
        # with room, prepare loads every shard's block range of exactly the columns the view needs
        g.set_option("hbm_budget_mb", 0)
        gq2 = G._gq(gt[("a", lambda c: c > 700_000), ["a"]])
        assert gq2.prepare() == 1 and gq2.prepare() == 0
        for l in range(world):
            sh = gt.shard(l)
            assert sh.resident(0) and not sh.resident(1) and not sh.resident(2)
        assert gq2.count() == int((cols["a"] > 700_000).sum())
        gt.close()
    finally:
        g.close()
