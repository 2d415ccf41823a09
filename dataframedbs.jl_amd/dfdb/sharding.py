"""Block-range sharding of one table over the GPUs of a node.

The reference is single-process; its blocks are independent 65 536-row units (every column shares the block
boundaries: check_column_head, src/io/filesystem.jl:47-54), so rank g of G simply owns a contiguous block
range of EVERY required column and results concatenate in rank order = table order.  One process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Collectives:

  * count()/sum/min/max  -> one all-reduce of a few scalars
  * a range stage AFTER a predicate stage numbers the survivors globally (selection.jl:94-111), so each
    such stage needs the survivors that live on lower ranks: all-gather of one Int64 per rank, exclusive scan
  * a LEADING range stage numbers table rows: no exchange (dfdb_table_set_row_base)

No bulk data ever crosses xGMI.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple


def block_range(nblocks: int, rank: int, world: int) -> Tuple[int, int]:
    """Blocks [first, last) owned by `rank`: g * ceil(nb/G) .. (g+1) * ceil(nb/G), clipped."""
    per = -(-nblocks // world) if world > 0 else nblocks
    first = min(rank * per, nblocks)
    return first, min(first + per, nblocks)


def row_range(nrows: int, block_size: int, rank: int, world: int) -> Tuple[int, int]:
    nb = -(-nrows // block_size)
    b0, b1 = block_range(nb, rank, world)
    return min(b0 * block_size, nrows), min(b1 * block_size, nrows)


def _dist():
    import torch.distributed as dist
    return dist


def exclusive_base(local_count: int, group=None, device=None) -> Tuple[int, int]:
    """(survivors on lower ranks, global total) from one all-gather of an Int64 per rank."""
    import torch
    dist = _dist()
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0, int(local_count)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    counts = [int(p.item()) for p in parts]
    return sum(counts[:rank]), sum(counts)


def all_reduce_scalars(values: Sequence[float], op: str = "sum", group=None, device=None, dtype=None) -> List:
    """One all-reduce over a handful of scalars (count, sum(x), …)."""
    import torch
    dist = _dist()
    dt = dtype or (torch.int64 if all(isinstance(v, int) for v in values) else torch.float64)
    t = torch.tensor(list(values), dtype=dt, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        rop = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op]
        dist.all_reduce(t, op=rop, group=group)
    return t.tolist()


def plan_stage_bases(stage_kinds: Sequence[str], count_prefix: Callable[[int], int], set_stage_base: Callable[[int, int], None],
                     group=None, device=None) -> None:
    """For every range-like stage that follows at least one earlier stage, give the engine the number of
    survivors of the preceding stages that live on lower ranks.

    stage_kinds[i] in {"range", "integer", "indices", "predicate"}; count_prefix(k) evaluates stages [0, k)
    locally and returns the local survivor count (dfdb_query_count_prefix); set_stage_base(k, base) is
    dfdb_query_set_stage_base.  Stages are resolved left to right because a later base depends on the
    earlier ones being set.
    """
    for k, kind in enumerate(stage_kinds):
        if k == 0 or kind == "predicate":
            continue
        base, _ = exclusive_base(count_prefix(k), group=group, device=device)
        set_stage_base(k, base)


def sharded_count(view, group=None, device=None) -> int:
    """nrow(v) over all ranks: local scans + exchanges for range-after-predicate + one all-reduce."""
    from . import api, ir
    import ctypes as C
    from . import _native as N
    q = view._query()
    kinds = ["predicate" if isinstance(s, ir.Expr) else ("range" if isinstance(s, api.JRange) else ("integer" if isinstance(s, int) else "indices"))
             for s in view.selection.queue]

    def count_prefix(k: int) -> int:
        n = C.c_int64()
        N.check(N.load().dfdb_query_count_prefix(q._h, k, C.byref(n)))
        return n.value

    def set_base(k: int, base: int):
        N.check(N.load().dfdb_query_set_stage_base(q._h, k, base))

    plan_stage_bases(kinds, count_prefix, set_base, group=group, device=device)
    return int(all_reduce_scalars([q.count()], "sum", group=group, device=device)[0])
