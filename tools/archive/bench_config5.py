#!/usr/bin/env python3
"""BASELINE.json config 5: 1e10 rows x (a::Int64, x::Float64, s::String) sharded by contiguous block
ranges over the GPUs of one node, conjunctive predicate, count() + sum(x) through ONE all-reduce of two scalars.

    python tools/bench_config5.py                      # one GPU = one 1/8 shard of the 8-GPU job (1.25e9 rows, 32 GB)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/bench_config5.py --total-rows 1e10

Rank g generates ITS rows of the global columns on the device (row_first = first row of its block range), so the
data is the same 1e10-row table whatever the world size; no bulk byte crosses xGMI.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SEED = 0x9E3779B97F4A7C15


def seed(k):
    return (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total-rows", type=float, default=0.0, help="rows of the whole table (default: 1.25e9 per rank)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-dictionary", action="store_true", help="keep the String column flat only (no K9 codes beside it)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    import dfdb
    from dfdb import sharding
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = dfdb.Context(local, stream=stream.cuda_stream)
    if not args.no_dictionary:
        ctx.set_option("string_dictionary", 4096)       # s has 10 distinct values: 16-bit codes beside the flat column, s != "sony" scans those
    bs = 65536
    total = int(args.total_rows) if args.total_rows else 1_250_000_000 * world
    r0, r1 = sharding.row_range(total, bs, rank, world)
    n = r1 - r0
    t = dfdb.DFTable.new(block_size=bs, ctx=ctx)
    t.add_generated("a", dfdb.GEN_I64_MOD1M, seed(0), n, row_first=r0)
    t.add_generated("x", dfdb.GEN_F64_U2000, seed(1), n, row_first=r0)
    t.add_generated("s", dfdb.GEN_STR_BRANDS10, seed(2), n, row_first=r0)
    t.set_row_base(r0)
    v = t[(t.a > 683_771) & (t.x < 632.456) & (t.s != "sony"), dfdb.ALL]
    cx = v[dfdb.ALL, "x"]
    q = cx.view._query()               # ONE evaluation of the selection gives both numbers: sum() hints the scan to add x up while
    res = torch.zeros(2, dtype=torch.float64, device=dev)   # it holds it (dfdb_query_hint_aggregate), count() reads the same execution

    def step():
        q.reset()
        sx = cx.sum()
        cnt = q.count()
        res[0] = float(cnt); res[1] = sx
        if world > 1:
            dist.all_reduce(res)
        return res

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        sec = dt.item() / args.steps
        bytes_row = 8 + 8 + 4 + 5.4            # predicate columns a, x, sizes + bytes of s
        print(json.dumps({"config": 5, "n_gpus": world, "total_rows": total, "rows_per_gpu": n, "count": int(res[0].item()), "sum_x": res[1].item(),
                          "ms_per_step": sec * 1e3, "rows_per_s": total / sec, "algorithmic_GBps": total * bytes_row / sec / 1e9,
                          "string_dictionary": not args.no_dictionary,
                          "note": "algorithmic bytes count the flat String column (4 + 5.4 B/row); with the dictionary the scan of s reads 2 B/row, so the figure can exceed what HBM delivers"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
