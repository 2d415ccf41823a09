"""GPU tests that need MORE THAN ONE physical device.  On a 1-GPU box every test here reports "skipped: 1 device"; on a node they are
the first place RCCL runs between distinct GPUs: one process driving every device through dfdb_group_create (a host thread per GPU, ncclCommInitAll,
grouped all-reduce / all-gather on the shards' engine streams), and bench.py's two N-rank modes with real nccl — all answers against the oracle's single table
(count() reduced as view.jl:192-206 sums it over blocks; block ranges shard, only the final count / aggregate crosses devices)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_group import _check_group_against_oracle, _columns, _is_sony

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x9E3779B97F4A7C15


def _ndev():
    import torch
    return torch.cuda.device_count()             # (counting devices does not initialise them)


def _need_devices(k=2):
    n = _ndev()
    if n < k:
        pytest.skip(f"skipped: {n} device" + ("" if n == 1 else "s") + f" (needs {k} distinct GPUs)")
    return n


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_rccl_smoke_allreduce_of_rank_ids_over_all_devices(dfdb_mod, ctx):
    """the first thing to run on a node: dfdb_group_create over every device (ncclCommInitAll) and one Float64 all-reduce per operator through the library"""
    from dfdb import group as G, _native as N
    n = _need_devices(2)
    n = min(n, 8)
    g = G.Group.create(list(range(n)), N.EXCHANGE_AUTO)
    try:
        assert (g.world, g.nlocal, g.first_rank) == (n, n, 0) and g.exchange == N.EXCHANGE_RCCL
        vals = [[float(r + 1), float(-r)] for r in range(n)]          # every shard contributes (rank id + 1, -rank id)
        assert g.allreduce(vals, N.AGG_SUM) == [[n * (n + 1) / 2.0, -n * (n - 1) / 2.0]] * n
        assert g.allreduce(vals, N.AGG_MIN) == [[1.0, float(-(n - 1))]] * n
        assert g.allreduce(vals, N.AGG_MAX) == [[float(n), 0.0]] * n
        g.barrier()
    finally:
        g.close()


@pytest.mark.parametrize("bs", [4096, 65536])
def test_group_over_distinct_devices_vs_oracle(oracle, dfdb_mod, ctx, tmp_path, bs):
    """every shard on ITS OWN GPU: count / indices / materialize / sum / min / max / range-after-predicate (all-gather + exclusive scan of the stage bases)
    equal the oracle's single table; then unique and groupreduce over the whole sharded table (records all-gathered by RCCL)"""
    from dfdb import group as G, _native as N
    n = min(_need_devices(2), 8)
    rows = 300_007 if bs == 4096 else 65536 * (2 * n + 1) + 123
    cols = _columns(oracle, rows)
    ot = oracle.Table(block_size=bs)
    for k, v in cols.items():
        ot.add_column(k, v)
    path = str(tmp_path / "tb")
    ot.save(path)
    g = G.Group.create(list(range(n)), N.EXCHANGE_AUTO)
    try:
        assert g.exchange == N.EXCHANGE_RCCL
        gt = G.GroupTable.open(g, path)
        assert gt.nrows == rows
        _check_group_against_oracle(oracle, dfdb_mod, ot, gt, cols, rows)
        # unique / groupreduce: first appearance in table order = lowest rank, then lowest row
        one = dfdb_mod.open_table(path)
        try:
            for key in ("a", "s"):
                sel_g = dfdb_mod.selection(gt.view(), dfdb_mod.jr(1, 1, 50_000)) if key == "a" else gt.view()
                sel_1 = dfdb_mod.selection(dfdb_mod.DFView(one), dfdb_mod.jr(1, 1, 50_000)) if key == "a" else dfdb_mod.DFView(one)
                assert list(G.gunique(getattr(sel_g, key))) == list(getattr(sel_1, key).unique()), key
            w = dfdb_mod.groupreduce(dfdb_mod.DFView(one), "s", "x", "sum")
            r = G.ggroupreduce(gt.view(), "s", "x", "sum")
            assert list(w["s"]) == list(r["s"]) and (w["count"].to_numpy() == r["count"].to_numpy()).all() and np.allclose(w["sum"], r["sum"], rtol=1e-12)
        finally:
            one.close()
        gt.close()
    finally:
        g.close()


def test_a_failing_shard_on_another_device_raises_the_same_error_everywhere(oracle, dfdb_mod, ctx):
    """a zero divisor that only the LAST device's rows reach: the fault key travels with the RCCL exchange and the one call raises DivideError"""
    from dfdb import group as G, _native as N, ir
    n = min(_need_devices(2), 8)
    rows = 4096 * 4 * n
    z = np.ones(rows, np.int64); z[-5] = 0
    cols = {"a": np.arange(rows, dtype=np.int64), "z": z}
    g = G.Group.create(list(range(n)), N.EXCHANGE_AUTO)
    try:
        gt = G.GroupTable.from_columns(g, cols, block_size=4096)
        v = dfdb_mod.selection(gt.view(), ir.col(0) % ir.col(1) == 0)
        with pytest.raises(ZeroDivisionError):
            G.gnrow(v)
        assert G.gnrow(dfdb_mod.selection(gt.view(), ir.col(0) > 10)) == rows - 11          # the group is still usable
        gt.close()
    finally:
        g.close()


@pytest.mark.parametrize("exchange", ["torch", "lib"])
def test_bench_over_real_devices_with_nccl(oracle, exchange):
    """`bench.py --gpus N` (one process per GPU, real nccl = RCCL between distinct devices), with the per-step all-reduce through torch.distributed and through
    the library's own communicator: one line, N GPUs, the global count the oracle's, every config leg present"""
    n = min(_need_devices(2), 8)
    rows = 20_000_000
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--exchange", exchange, "--rows", str(rows), "--steps", "3", "--warmup", "1", "--no-cpu",
                        "--config-scale", "0.004"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    r = json.loads(lines[0])
    assert r["n_gpus"] == n and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["global_selected"] == int((oracle.gen_i64(SEED, 0, n * rows) > 899_999).sum())
    cf = r["configs"]
    for k in ("3", "4", "5_shard", "5_shard_materialize", "5_shard_dictionary"):
        assert k in cf and "error" not in cf[k], (k, cf.get(k))
    assert "RCCL" in cf["5_shard"]["exchange"]
    n5 = cf["5_shard"]["rows_per_gpu"]
    a = oracle.gen_i64(SEED, 0, n * n5); x = oracle.gen_f64((SEED * 2) & 0xFFFFFFFFFFFFFFFF, 0, n * n5)
    sz, by = oracle.gen_str((SEED * 3) & 0xFFFFFFFFFFFFFFFF, 0, n * n5)
    sel = (a > 683_771) & (x < 632.456) & ~_is_sony(sz, by)
    assert cf["5_shard"]["global_count"] == int(sel.sum())
    assert "summary" in r and list(r)[-1] == "summary"


def test_bench_threads_mode_over_real_devices(oracle):
    """`bench.py --mode threads --gpus N`: ONE process, dfdb_group_create over N distinct devices (what a Julia session gets)"""
    n = min(_need_devices(2), 8)
    rows = 20_000_000
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "threads", "--gpus", str(n), "--rows", str(rows), "--steps", "3", "--warmup", "1",
                        "--config-scale", "0.004"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    r = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == n and "RCCL" in r["config"]["sharding"]
    assert r["config"]["global_selected"] == int((oracle.gen_i64(SEED, 0, n * rows) > 899_999).sum())
