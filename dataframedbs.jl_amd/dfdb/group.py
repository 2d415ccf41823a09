"""Multi-GPU groups: one table block-range sharded over the GPUs of a node, through the C ABI (include/dfdb.h, "multi-GPU groups").

The lazy algebra is unchanged: `GroupTable.view()` is an ordinary DFView (built over the metadata of shard 0), and
`gnrow / gindices / gmaterialize / gaggregate` replay its SelectionQueue and Projection onto a group query, the way
`nrow / materialize` (view.jl:192-206, materialization.jl:27-40) do for one GPU.  The exchanges (all-reduce of the count /
aggregate over RCCL, all-gather + exclusive scan for a range stage after a predicate) happen inside libdfdb_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import _native as N
from . import api, ir


class Group:
    """dfdb_group: `Group.create([0, 1, ...])` = one process drives the GPUs; `Group.create_rank(...)` = one process per GPU."""

    def __init__(self, handle):
        self._h = handle
        w, nl, fr, ex = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        N.check(N.load().dfdb_group_info(self._h, C.byref(w), C.byref(nl), C.byref(fr), C.byref(ex)))
        self.world, self.nlocal, self.first_rank, self.exchange = w.value, nl.value, fr.value, ex.value

    @classmethod
    def create(cls, devices: Sequence[int], exchange: int = N.EXCHANGE_AUTO) -> "Group":
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        N.check(N.load().dfdb_group_create(devs, len(devices), exchange, C.byref(h)))
        return cls(h)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(N.GROUP_ID_BYTES)
        N.check(N.load().dfdb_group_unique_id(buf))
        return buf.raw

    @classmethod
    def create_rank(cls, device: int, uid: Optional[bytes], rank: int, world: int, stream: Optional[int] = None) -> "Group":
        h = C.c_void_p()
        buf = C.create_string_buffer(uid, N.GROUP_ID_BYTES) if uid is not None else None
        N.check(N.load().dfdb_group_create_rank(device, C.c_void_p(stream) if stream else None, buf, rank, world, C.byref(h)))
        return cls(h)

    @classmethod
    def create_rank_callbacks(cls, device: int, rank: int, world: int, allreduce, allgather, stream: Optional[int] = None) -> "Group":
        """one process per GPU with the HOST's own collectives (dfdb_group_create_rank_callbacks, DFDB_EXCHANGE_CALLBACK):
        allreduce(vals: np.ndarray[uint64 view of n 8-byte values], dtype, op) reduces in place, allgather(send: bytes) -> bytes of every rank in rank order"""
        def _ar(user, vals, n, dtype, op):
            try:
                allreduce(np.ctypeslib.as_array((C.c_uint64 * n).from_address(vals)), dtype, op)
                return 0
            except Exception:      # noqa: BLE001 — an exception must not unwind through the C frames
                import traceback; traceback.print_exc()
                return 1

        def _ag(user, send, recv, nbytes):
            try:
                out = allgather(C.string_at(send, nbytes))
                C.memmove(recv, out, len(out))
                return 0
            except Exception:      # noqa: BLE001
                import traceback; traceback.print_exc()
                return 1
        fns = N.ExchangeFns(None, N.ALLREDUCE_FN(_ar), N.ALLGATHER_FN(_ag))
        h = C.c_void_p()
        N.check(N.load().dfdb_group_create_rank_callbacks(device, C.c_void_p(stream) if stream else None, rank, world, C.byref(fns), C.byref(h)))
        g = cls(h)
        g._keep = fns                      # the C side copied the table; the CFUNCTYPE thunks must stay alive
        return g

    @classmethod
    def create_rank_torch(cls, device: int, stream: Optional[int] = None) -> "Group":
        """the same over torch.distributed (any backend that moves CPU tensors: gloo): rank and world come from the initialised process group.
        The reductions gather the 8-byte patterns and fold them here in the asked dtype (torch has no UInt64 reductions; integer sums wrap)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()

        def gather_rows(a: np.ndarray) -> np.ndarray:
            mine = torch.from_numpy(a.view(np.int64).copy())
            box = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(box, mine)
            return np.stack([b.numpy() for b in box])

        def allreduce(vals, dtype, op):
            rows = gather_rows(vals).view({ir.F64: np.float64, ir.U64: np.uint64}.get(dtype, np.int64))
            with np.errstate(over="ignore"):
                r = rows.sum(axis=0, dtype=rows.dtype) if op == N.AGG_SUM else (rows.min(axis=0) if op == N.AGG_MIN else rows.max(axis=0))
            vals[:] = np.ascontiguousarray(r).view(np.uint64)

        def allgather(send: bytes) -> bytes:
            a = np.frombuffer(send + b"\0" * (-len(send) % 8), np.uint8)
            return b"".join(bytes(row.view(np.uint8)[:len(send)]) for row in gather_rows(a))
        return cls.create_rank_callbacks(device, rank, world, allreduce, allgather, stream)

    def ctx(self, local: int = 0) -> api.Context:
        """borrowed Context of one local shard (profiling, options, device info)"""
        h = C.c_void_p()
        N.check(N.load().dfdb_group_ctx(self._h, local, C.byref(h)))
        c = api.Context.__new__(api.Context)
        c._h, c.device = h, -1
        c.close = lambda: None          # owned by the group
        return c

    def synchronize(self): N.check(N.load().dfdb_group_synchronize(self._h))

    def barrier(self): N.check(N.load().dfdb_group_barrier(self._h))

    def set_option(self, key: str, value: int): N.check(N.load().dfdb_group_set_option(self._h, key.encode(), value))

    def allreduce(self, vals: Sequence[Sequence[float]], op: int = N.AGG_SUM) -> List[List[float]]:
        """vals[local shard][k] -> reduced over every rank"""
        n = len(vals[0])
        a = np.ascontiguousarray(vals, np.float64).reshape(self.nlocal, n)
        N.check(N.load().dfdb_group_allreduce_f64(self._h, a.ctypes.data, n, op))
        return a.tolist()

    def close(self):
        if self._h:
            N.load().dfdb_group_destroy(self._h)
            self._h = C.c_void_p()


class _ShardTable(api.DFTable):
    """the ordinary handle of one shard, borrowed from the group table (metadata, per-shard queries); never closed from here"""

    def close(self):
        object.__setattr__(self, "_h", C.c_void_p())
        object.__setattr__(self, "is_opened", False)


class GroupTable:
    def __init__(self, handle, group: Group, path: str = ""):
        self._h, self.group, self.path = handle, group, path

    @classmethod
    def new(cls, group: Group, block_size: int = 65536) -> "GroupTable":
        h = C.c_void_p()
        N.check(N.load().dfdb_group_table_new(group._h, block_size, C.byref(h)))
        return cls(h, group)

    @classmethod
    def open(cls, group: Group, path: str, load: bool = True, columns: Optional[Sequence[str]] = None) -> "GroupTable":
        h = C.c_void_p()
        N.check(N.load().dfdb_group_table_open(group._h, path.encode(), C.byref(h)))
        t = cls(h, group, path)
        if load:
            t.load(columns)
        return t

    def shard(self, local: int = 0) -> api.DFTable:
        h = C.c_void_p()
        N.check(N.load().dfdb_group_table_shard(self._h, local, C.byref(h)))
        return _ShardTable(h, self.group.ctx(local), self.path)

    def load(self, columns: Optional[Sequence[str]] = None) -> dict:
        st = N.SizeStats()
        if columns is None:
            N.check(N.load().dfdb_group_table_load(self._h, None, 0, C.byref(st)))
        else:
            meta = self.shard(0)
            ords = (C.c_int32 * len(columns))(*[meta.ordinal(c) for c in columns])
            N.check(N.load().dfdb_group_table_load(self._h, ords, len(columns), C.byref(st)))
        return {"rows": st.rows, "compressed": st.compressed, "uncompressed": st.uncompressed}

    def add_generated(self, name: str, generator: int, seed: int, nrows_total: int):
        N.check(N.load().dfdb_group_table_add_generated(self._h, name.encode(), generator, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), nrows_total))

    def add_column(self, name: str, values, dtype: Optional[int] = None):
        """whole-table host column (numpy array, masked array or list of str / None); every shard uploads its block range"""
        L = N.load()
        if isinstance(values, (list, tuple)) and (len(values) == 0 or isinstance(values[0], (str, bytes, type(None)))):
            enc = [None if v is None else (v.encode() if isinstance(v, str) else bytes(v)) for v in values]
            sizes = np.array([-1 if e is None else len(e) for e in enc], np.int32)
            data = np.frombuffer(b"".join(e for e in enc if e), np.uint8).copy()
            dt = ir.STRING | (ir.NULLABLE if any(e is None for e in enc) or (dtype or 0) & ir.NULLABLE else 0)
            N.check(L.dfdb_group_table_add_column(self._h, name.encode(), dt, len(sizes), sizes.ctypes.data, data.ctypes.data if len(data) else None, len(data), None))
            return
        missing = None
        if isinstance(values, np.ma.MaskedArray):
            missing = np.ascontiguousarray(np.ma.getmaskarray(values), np.uint8)
            values = values.filled(0)
        arr = np.ascontiguousarray(values)
        if dtype is None:
            dtype = ir.dtype_of_numpy(arr.dtype)
        arr = np.ascontiguousarray(arr.astype(ir.numpy_of_dtype(dtype), copy=False))
        if missing is not None:
            dtype |= ir.NULLABLE
        N.check(L.dfdb_group_table_add_column(self._h, name.encode(), dtype, len(arr), arr.ctypes.data if len(arr) else None, None, 0,
                                              missing.ctypes.data if missing is not None else None))

    @classmethod
    def from_columns(cls, group: Group, columns: Dict[str, Any], block_size: int = 65536) -> "GroupTable":
        t = cls.new(group, block_size)
        for k, v in columns.items():
            t.add_column(k, v)
        return t

    @property
    def nrows(self) -> int:
        n = C.c_int64()
        N.check(N.load().dfdb_group_table_nrows(self._h, C.byref(n)))
        return n.value

    def view(self) -> api.DFView:
        """the lazy view algebra runs over shard 0's metadata; evaluation goes through gnrow / gmaterialize / ..."""
        t = self.shard(0)
        object.__setattr__(t, "_group_table", self)      # every view derived from this one finds the group through its table
        return api.DFView(t)

    def __getitem__(self, key):
        v = self.view()[key]
        return v

    def close(self):
        if self._h:
            N.load().dfdb_group_table_close(self._h)
            self._h = C.c_void_p()


class GroupQuery:
    """dfdb_gquery built from a DFView's SelectionQueue + Projection"""

    def __init__(self, gt: GroupTable, v: api.DFView):
        L = N.load()
        self.gt, self.view = gt, v
        self._h = C.c_void_p()
        N.check(L.dfdb_group_query_new(gt._h, C.byref(self._h)))
        try:
            for st in v.selection.queue:
                if isinstance(st, ir.Expr):
                    b = st.to_ir()
                    N.check(L.dfdb_group_query_add_predicate(self._h, b, len(b)))
                elif isinstance(st, api.JRange):
                    N.check(L.dfdb_group_query_add_range(self._h, int(st.start), int(st.step), int(st.stop)))
                elif isinstance(st, int):
                    N.check(L.dfdb_group_query_add_integer(self._h, st))
                else:
                    a = np.ascontiguousarray(st, np.int64)
                    N.check(L.dfdb_group_query_add_indices(self._h, a.ctypes.data if len(a) else None, len(a)))
            items = [(k, e.to_ir()) for k, e in v.projection.cols.items()]
            n = len(items)
            names = (C.c_char_p * max(n, 1))(*[k.encode() for k, _ in items])
            bufs = [C.create_string_buffer(b, len(b)) for _, b in items]
            irs = (C.c_void_p * max(n, 1))(*[C.cast(b, C.c_void_p) for b in bufs])
            lens = (C.c_size_t * max(n, 1))(*[len(b) for _, b in items])
            N.check(L.dfdb_group_query_set_projection(self._h, n, names, irs, lens))
        except Exception:
            L.dfdb_group_query_free(self._h)
            self._h = C.c_void_p()
            raise

    def close(self):
        if self._h:
            N.load().dfdb_group_query_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            import sys
            if self._h and not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def reset(self): N.check(N.load().dfdb_group_query_reset(self._h))

    def prepare(self) -> int:
        """dfdb_group_query_prepare: every shard loads its block range of exactly the columns this view needs if that fits (1; 0 = resident already), else the
        shards stay on disk and every call streams (3)"""
        how = C.c_int32(-1)
        N.check(N.load().dfdb_group_query_prepare(self._h, C.byref(how)))
        return how.value

    def count(self) -> int:
        n = C.c_int64()
        N.check(N.load().dfdb_group_count(self._h, C.byref(n)))
        return n.value

    def count_async(self): N.check(N.load().dfdb_group_count(self._h, None))

    def shard_counts(self) -> List[int]:
        a = (C.c_int64 * self.gt.group.world)()
        N.check(N.load().dfdb_group_shard_counts(self._h, a))
        return list(a)

    def coltype(self, i: int) -> int:
        q = C.c_void_p()
        N.check(N.load().dfdb_group_query_shard(self._h, 0, C.byref(q)))
        dt = C.c_int32()
        N.check(N.load().dfdb_query_coltype(q, i, C.byref(dt)))
        return dt.value

    def local_count(self) -> int:
        g = self.gt.group
        return sum(self.shard_counts()[g.first_rank:g.first_rank + g.nlocal])

    def indices(self) -> np.ndarray:
        n = self.local_count()
        out = np.empty(n, np.int64)
        got = C.c_int64()
        N.check(N.load().dfdb_group_select_indices(self._h, out.ctypes.data if n else None, n, C.byref(got)))
        assert got.value == n
        return out

    def indices_device(self, dev_ptrs: Sequence[int], caps: Sequence[int]):
        nl = self.gt.group.nlocal
        outs = (C.c_void_p * nl)(*dev_ptrs)
        cps = (C.c_int64 * nl)(*caps)
        N.check(N.load().dfdb_group_select_indices_device(self._h, outs, cps))

    def hint_aggregate(self, op: int, col: int = 0): N.check(N.load().dfdb_group_query_hint_aggregate(self._h, op, col))

    def aggregate(self, op: int, col: int = 0):
        oi, of = C.c_int64(), C.c_double()
        if op in (N.AGG_SUM, N.AGG_MIN, N.AGG_MAX):
            self.hint_aggregate(op, col)
        N.check(N.load().dfdb_group_aggregate(self._h, op, col, C.byref(oi), C.byref(of)))
        dt = self.coltype(col) & ir.DTYPE_MASK if op != N.AGG_COUNT else ir.I64
        return of.value if dt in (ir.F32, ir.F64) else oi.value

    def materialize(self) -> List[Any]:
        L = N.load()
        N.check(L.dfdb_group_query_hint_materialize(self._h, 1))
        n = self.local_count()
        ncols = len(self.view.projection)
        outs = (N.OutCol * max(ncols, 1))()
        keep = []
        for i in range(ncols):
            dt = self.coltype(i)
            o = outs[i]
            o.memkind = N.MEM_HOST
            if (dt & ir.DTYPE_MASK) == ir.STRING:
                nb = C.c_int64()
                N.check(L.dfdb_group_result_string_bytes(self._h, i, C.byref(nb)))
                sizes = np.empty(max(n, 1), np.int32)
                data = np.empty(max(nb.value, 1), np.uint8)
                keep.append((dt, sizes, data, None))
                o.data, o.bytes, o.bytes_cap = sizes.ctypes.data, data.ctypes.data, nb.value
            else:
                arr = np.empty(max(n, 1), ir.numpy_of_dtype(dt))
                miss = np.zeros(max(n, 1), np.uint8) if dt & ir.NULLABLE else None
                keep.append((dt, arr, None, miss))
                o.data = arr.ctypes.data
                if miss is not None:
                    o.missing = miss.ctypes.data
        if ncols:
            N.check(L.dfdb_group_materialize(self._h, outs, ncols))
        res = []
        for i, (dt, a, b, m) in enumerate(keep):
            if (dt & ir.DTYPE_MASK) == ir.STRING:
                res.append((a[:n].copy(), b[:outs[i].nbytes].copy()))
            elif m is not None:
                res.append(np.ma.masked_array(a[:n].copy(), mask=m[:n].astype(bool)))
            else:
                res.append(a[:n].copy())
        return res


def gmaterialize_device(v: api.DFView, alloc):
    """materialize(v) left sharded on the devices (dfdb_group_materialize_device): `alloc(local_shard, nbytes)` returns the device address of a
    buffer of that many bytes in that shard's HBM (e.g. a torch tensor's data_ptr(); the caller keeps it alive).  Returns, per local shard, a list of
    per-column dicts {dtype, count, data, bytes, nbytes, missing} holding device addresses — rank order = table order.  Nothing crosses PCIe."""
    L = N.load()
    gq = _gq(v)
    g = gq.gt.group
    N.check(L.dfdb_group_query_hint_materialize(gq._h, 1))
    counts = gq.shard_counts()[g.first_rank:g.first_rank + g.nlocal]
    ncols = len(v.projection)
    dts = [gq.coltype(i) for i in range(ncols)]
    sbytes = {}
    for i, dt in enumerate(dts):
        if (dt & ir.DTYPE_MASK) == ir.STRING:
            a = (C.c_int64 * g.nlocal)()
            N.check(L.dfdb_group_shard_string_bytes(gq._h, i, a))
            sbytes[i] = list(a)
    outs = (N.OutCol * max(g.nlocal * ncols, 1))()
    for l in range(g.nlocal):
        n = counts[l]
        for i, dt in enumerate(dts):
            o = outs[l * ncols + i]
            o.memkind = N.MEM_DEVICE
            if (dt & ir.DTYPE_MASK) == ir.STRING:
                o.data = alloc(l, max(n, 1) * 4); o.bytes = alloc(l, sbytes[i][l] + 64); o.bytes_cap = sbytes[i][l]
            else:
                o.data = alloc(l, max(n, 1) * np.dtype(ir.numpy_of_dtype(dt)).itemsize)
                if dt & ir.NULLABLE:
                    o.missing = alloc(l, max(n, 1))
    if ncols:
        N.check(L.dfdb_group_materialize_device(gq._h, outs, ncols))
    return [[dict(dtype=outs[l * ncols + i].dtype, count=outs[l * ncols + i].count, data=outs[l * ncols + i].data, bytes=outs[l * ncols + i].bytes,
                  nbytes=outs[l * ncols + i].nbytes, missing=outs[l * ncols + i].missing) for i in range(ncols)] for l in range(g.nlocal)]


def _gq(v) -> GroupQuery:
    if isinstance(v, api.DFColumn):
        v = v.view
    gt = getattr(v.table, "_group_table", None)
    if gt is None:
        raise ValueError("not a view of a GroupTable")
    q = getattr(v, "_gq", None)
    if q is None:
        q = v._gq = GroupQuery(gt, v)
    return q


def gnrow(v: api.DFView) -> int: return _gq(v).count()
def gindices(v: api.DFView) -> np.ndarray: return _gq(v).indices()
def gaggregate(v: api.DFView, op: int, col: int = 0): return _gq(v).aggregate(op, col)


def gmaterialize(v: api.DFView):
    import pandas as pd
    cols = _gq(v).materialize() if len(v.projection) else []
    lg = api._logicals(v)
    return pd.DataFrame({k: api._to_user(c, lg[i]) for i, (k, c) in enumerate(zip(v.projection.keys(), cols))})


# ---------------------------------------------------------------- unique / groupreduce over shards
# Behind the C ABI since round 3 (dfdb_group_query_unique / _groupreduce + their fetches, csrc/group.cpp): every shard reduces its own rows on its
# own device, one record per distinct key crosses to the other ranks (RCCL all-gather; nothing when one process holds every shard), and the
# records are merged by key in rank order — the table's row order — so the keys come out in order of first appearance over the whole table, as
# Base.unique / the reference's group_map numbering give them.  What is left here is sizing the caller-owned buffers.
def _fetch_keys(L, gq: GroupQuery, kdt: int, n: int, kbytes: int, fetch, logical: str = ""):
    out = N.OutCol()
    out.memkind = N.MEM_HOST
    is_str = (kdt & ir.DTYPE_MASK) == ir.STRING
    if is_str:
        ksz = np.empty(max(n, 1), np.int32); kby = np.empty(max(kbytes, 1), np.uint8)
        out.data, out.bytes, out.bytes_cap = ksz.ctypes.data, kby.ctypes.data, kbytes
    else:
        karr = np.empty(max(n, 1), ir.numpy_of_dtype(kdt))
        kmiss = np.zeros(max(n, 1), np.uint8) if kdt & ir.NULLABLE else None
        out.data = karr.ctypes.data
        if kmiss is not None:
            out.missing = kmiss.ctypes.data
    fetch(out)
    if is_str:
        return api._to_user((ksz[:n].copy(), kby[:out.nbytes].copy()), logical)
    if kdt & ir.NULLABLE:
        return api._to_user(np.ma.masked_array(karr[:n].copy(), mask=kmiss[:n].astype(bool)), logical)
    return api._to_user(karr[:n].copy(), logical)


def gunique(col: api.DFColumn):
    """unique(col) over every shard: the distinct values in order of first appearance in the whole table"""
    v = col.view
    gt = getattr(v.table, "_group_table", None)
    if gt is None:
        raise ValueError("not a column of a GroupTable")
    L = N.load()
    gq = GroupQuery(gt, v)
    try:
        n, kb = C.c_int64(), C.c_int64()
        N.check(L.dfdb_group_query_unique(gq._h, 0, C.byref(n), C.byref(kb)))
        return _fetch_keys(L, gq, gq.coltype(0), n.value, kb.value, lambda out: N.check(L.dfdb_group_query_unique_fetch(gq._h, C.byref(out))), api._logicals(v)[0])
    finally:
        gq.close()


def ggroupreduce(v: api.DFView, by: str, col: Optional[str] = None, stat: str = "count"):
    """groupreduce(view, (:by,); out = :col => Stat()) over every shard (dfdb_group_query_groupreduce): counts and sums add — Int64 sums wrap as on
    one device, Float64 sums are sums of the shards' sums —, min / max fold with Julia's NaN and signed-zero rules; same frame as dfdb.groupreduce."""
    sub, with_value = api._groupreduce_view(v, by, col, stat)
    gt = getattr(sub.table, "_group_table", None)
    if gt is None:
        raise ValueError("not a view of a GroupTable")
    L = N.load()
    gq = GroupQuery(gt, sub)
    try:
        ng, kb = C.c_int64(), C.c_int64()
        N.check(L.dfdb_group_query_groupreduce(gq._h, 0, 1 if with_value else -1, api._STATS[stat], C.byref(ng), C.byref(kb)))
        n = ng.value
        counts = np.zeros(max(n, 1), np.int64); vi = np.zeros(max(n, 1), np.int64); vf = np.zeros(max(n, 1), np.float64)
        keys = _fetch_keys(L, gq, gq.coltype(0), n, kb.value,
                           lambda out: N.check(L.dfdb_group_query_groupreduce_fetch(gq._h, C.byref(out), counts.ctypes.data, vi.ctypes.data, vf.ctypes.data)))
        vdt = gq.coltype(1) & ir.DTYPE_MASK if with_value else None
        return api._groupreduce_frame(by, stat, keys, counts[:n].copy(), vi[:n].copy(), vf[:n].copy(), vdt)
    finally:
        gq.close()
