"""Host-side mirror of DataFrameDBs.jl's lazy table algebra over the HIP engine.

Same names, argument meaning and error behaviour as the reference (paths under /root/reference):

    DFTable / open_table          src/tables/table.jl:9-56, src/tables/creators.jl:7-16
    DFView, selection, projection src/tables/view.jl:26-138
    DFColumn + broadcasting       src/tables/column.jl:30-126, src/tables/columnbroadcast.jl:1-72
    SelectionQueue composition    src/tables/selection.jl:4-60
    Projection                    src/tables/projection.jl:1-97
    materialize / nrow / size     src/tables/materialization.jl:27-56, src/tables/view.jl:192-232

Julia idioms map to Python as follows (rows stay 1-BASED, ranges inclusive, like the reference):

    t[:, :]                      t[ALL, ALL]           (or t[:, :])
    t[5:20, [:a, :c]]            t[jr(5, 20), ["a", "c"]]
    t[1:2:end, :a]               t[jr(1, 2, END), "a"]
    :a => f  /  (:a,:b) => f     ("a", f)  /  (("a", "b"), f)
    (e = :a, k = (:a,:c) => f)   {"e": "a", "k": (("a", "c"), f)}
    t.a .> 5  /  a .& b          t.a > 5   /  a & b          (DFColumn operators build the same trees)
    startswith.(t.b, "1")        startswith(t.b, "1")
    dest .= t.a .* t.c           (t.a * t.c).copyto(dest)

Everything data-parallel runs in libdfdb_hip.so; this module only builds queries.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _native as N
from . import ir


# ---------------------------------------------------------------- selectors
class _All:
    def __repr__(self): return ":"


ALL = _All()


class _End:
    """Julia's `end` inside an index expression: resolved with lastindex(v, dim) (view.jl:227-232)."""

    def __init__(self, offset: int = 0): self.offset = offset
    def __add__(self, k: int): return _End(self.offset + int(k))
    def __sub__(self, k: int): return _End(self.offset - int(k))
    def resolve(self, n: int) -> int: return n + self.offset
    def __repr__(self): return "end" + (f"{self.offset:+d}" if self.offset else "")


END = _End()


class JRange:
    """Julia `a:b` / `a:s:b`: 1-based, inclusive."""

    def __init__(self, start, step, stop):
        self.start, self.step, self.stop = start, step, stop

    def resolved(self, n: int) -> "JRange":
        f = lambda x: x.resolve(n) if isinstance(x, _End) else int(x)
        return JRange(f(self.start), int(self.step), f(self.stop))

    def has_end(self) -> bool:
        return isinstance(self.start, _End) or isinstance(self.stop, _End)

    def __len__(self):
        a, s, b = int(self.start), int(self.step), int(self.stop)
        return max(0, (b - a) // s + 1) if s > 0 else max(0, (a - b) // (-s) + 1)

    def __iter__(self):
        a, s = int(self.start), int(self.step)
        return (a + k * s for k in range(len(self)))

    def __eq__(self, o): return isinstance(o, JRange) and list(self) == list(o)
    def __hash__(self): return hash((self.start, self.step, self.stop))
    def __repr__(self): return f"{self.start}:{self.step}:{self.stop}" if self.step != 1 else f"{self.start}:{self.stop}"


def jr(a, b, c=None) -> JRange:
    """jr(a, b) == a:b ; jr(a, s, b) == a:s:b"""
    return JRange(a, 1, b) if c is None else JRange(a, b, c)


def _is_colon(x) -> bool:
    return x is ALL or x is Ellipsis or (isinstance(x, slice) and x == slice(None))


# ---------------------------------------------------------------- SelectionQueue (selection.jl:4-60)
class SelectionQueue:
    """Immutable tuple of stages: JRange | int | list[int] | ir.Expr (Bool)."""

    def __init__(self, queue: Tuple = ()):
        self.queue = tuple(queue)

    def __len__(self): return len(self.queue)
    def isempty(self): return not self.queue

    @staticmethod
    def _index(old, elem):
        """old[elem] with Julia bounds checking (range∘range collapse, selection.jl:40)."""
        def at(k: int):
            if isinstance(old, int):
                if k != 1:
                    raise IndexError("BoundsError: indexing a scalar selection")
                return old
            n = len(old)
            if k < 1 or k > n:
                raise IndexError(f"BoundsError: attempt to access {n}-element selection at index [{k}]")
            return (old.start + (k - 1) * old.step) if isinstance(old, JRange) else old[k - 1]
        if isinstance(elem, int):
            return at(elem)
        if isinstance(old, int):
            raise IndexError("BoundsError: indexing a scalar selection")
        if isinstance(old, JRange) and isinstance(elem, JRange):
            if len(elem) == 0:
                return JRange(old.start, old.step * elem.step, old.start - old.step * elem.step)
            return JRange(at(elem.start), old.step * elem.step, at(elem.start + (len(elem) - 1) * elem.step))
        return [at(k) for k in elem]

    def add(self, elem) -> "SelectionQueue":
        if _is_colon(elem):
            return self                                                    # add(q, ::Colon) = q (:37)
        if isinstance(elem, ir.Expr):
            if not self.queue or not isinstance(self.queue[-1], ir.Expr):
                return SelectionQueue(self.queue + (elem,))
            return SelectionQueue(self.queue[:-1] + (self.queue[-1] & elem,))   # fuse with & (:44-47)
        if not isinstance(elem, (JRange, int, list)):
            raise TypeError(f"unsupported selection element {elem!r}")
        if self.queue and not isinstance(self.queue[-1], ir.Expr):
            return SelectionQueue(self.queue[:-1] + (self._index(self.queue[-1], elem),))
        return SelectionQueue(self.queue + (elem,))

    def same(self, o: "SelectionQueue") -> bool:
        if len(self.queue) != len(o.queue):
            return False
        for a, b in zip(self.queue, o.queue):
            if isinstance(a, ir.Expr) != isinstance(b, ir.Expr):
                return False
            if isinstance(a, ir.Expr):
                if not _expr_equal(a, b):
                    return False
            elif type(a) is not type(b) or (list(a) if not isinstance(a, int) else a) != (list(b) if not isinstance(b, int) else b):
                return False
        return True

    def __repr__(self):
        return "Selection: " + " |> ".join(map(repr, self.queue))


def _expr_equal(a: ir.Expr, b: ir.Expr) -> bool:
    """View equality compares the stored objects (view.jl:34-38, quirk Q12): traces of two distinct
    closures differ even if structurally equal; the same function object compares equal."""
    fa, fb = getattr(a, "_origin", None), getattr(b, "_origin", None)
    if fa is not None or fb is not None:
        return fa is fb and a.same(b)
    return a.same(b)


class _TracedExpr(ir.Expr):
    __slots__ = ("_origin",)


def _trace(fn: Callable, leaves: Sequence[ir.Expr]) -> ir.Expr:
    e = ir.trace(fn, leaves)
    t = _TracedExpr(e.op, e.args, e.payload)
    t._origin = fn
    return t


# ---------------------------------------------------------------- Projection (projection.jl:1-97)
class Projection:
    """Ordered output-name -> ir.Expr (a plain column is `ir.col(k)`)."""

    def __init__(self, cols: Optional[Dict[str, ir.Expr]] = None):
        self.cols: Dict[str, ir.Expr] = dict(cols or {})

    def keys(self) -> List[str]: return list(self.cols.keys())
    def __len__(self): return len(self.cols)

    def add(self, el: Dict[str, ir.Expr]) -> "Projection":
        for k in el:
            if k in self.cols:
                raise ValueError(f"ArgumentError: Duplicated column {k}")      # projection.jl:25-28
        d = dict(self.cols)
        d.update(el)
        return Projection(d)

    def select_positions(self, idx: Sequence[int]) -> "Projection":            # getindex by Integer positions (:43-54)
        ks = self.keys()
        out = {}
        for i in idx:
            if i < 1 or i > len(ks):
                raise IndexError(f"BoundsError: projection has {len(ks)} columns, index [{i}]")
            out[ks[i - 1]] = self.cols[ks[i - 1]]
        return Projection(out)

    def select_names(self, names: Sequence[str]) -> "Projection":              # _get_indexes keeps PROJECTION order (:55-75, Q13)
        return Projection({k: v for k, v in self.cols.items() if k in names})

    def same(self, o: "Projection") -> bool:
        return self.keys() == o.keys() and all(_expr_equal(self.cols[k], o.cols[k]) for k in self.cols)

    def __repr__(self):
        return "Projection: " + "; ".join(f"{k}=>{v!r}" for k, v in self.cols.items())


# ---------------------------------------------------------------- engine handles
class Context:
    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._h = C.c_void_p()
        N.check(N.load().dfdb_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = device

    def synchronize(self): N.check(N.load().dfdb_ctx_synchronize(self._h))

    def device_info(self) -> dict:
        d = N.DeviceInfo()
        N.check(N.load().dfdb_ctx_device_info(self._h, C.byref(d)))
        return dict(name=d.name.decode(), compute_units=d.compute_units, wavefront_size=d.wavefront_size, hbm_bytes=d.hbm_bytes,
                    peak_hbm_gbps=d.peak_hbm_gbps)

    def set_option(self, key: str, value: int): N.check(N.load().dfdb_ctx_set_option(self._h, key.encode(), value))

    def timer_start(self): N.check(N.load().dfdb_ctx_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_double()
        N.check(N.load().dfdb_ctx_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def profile(self, on: bool): N.check(N.load().dfdb_ctx_profile_enable(self._h, 1 if on else 0))

    def profile_get(self, kernel: str) -> Tuple[int, float]:
        n, ms = C.c_int64(), C.c_double()
        N.check(N.load().dfdb_ctx_profile_get(self._h, kernel.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def close(self):
        if self._h:
            N.load().dfdb_ctx_destroy(self._h)
            self._h = C.c_void_p()


_default_ctx: Dict[int, Context] = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class ColumnMeta:
    def __init__(self, id: int, name: str, dtype: int, logical: str = ""):
        self.id, self.name, self.dtype, self.logical = id, name, dtype, logical

    @property
    def type(self) -> str:
        if not self.logical:
            return ir.dtype_name(self.dtype)
        return f"Missing({self.logical})" if self.dtype & ir.NULLABLE else self.logical     # Date / DateTime / Time / Char
    def __eq__(self, o): return isinstance(o, ColumnMeta) and (self.id, self.name, self.dtype, self.logical) == (o.id, o.name, o.dtype, o.logical)
    def __repr__(self): return f"ColumnMeta({self.id}, :{self.name}, {self.type})"


class DFTable:
    """A table whose columns are decoded and resident in HBM (struct DFTable: table.jl:9-15)."""

    def __init__(self, handle, ctx: Context, path: str = ""):
        object.__setattr__(self, "_h", handle)
        object.__setattr__(self, "ctx", ctx)
        object.__setattr__(self, "path", path)
        object.__setattr__(self, "is_opened", True)

    # -- construction
    @classmethod
    def new(cls, block_size: int = 65536, ctx: Optional[Context] = None) -> "DFTable":
        ctx = ctx or default_context()
        h = C.c_void_p()
        N.check(N.load().dfdb_table_new(ctx._h, C.c_int64(block_size), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def from_columns(cls, columns: Dict[str, Any], block_size: int = 65536, ctx: Optional[Context] = None) -> "DFTable":
        """In-memory table from host arrays (numpy / list[str|None] / masked arrays): the data path of
        create_table(path; from = df) (creators.jl:81-89) without the files."""
        t = cls.new(block_size, ctx)
        for name, v in columns.items():
            t.add_column(name, v)
        return t

    def add_column(self, name: str, values, dtype: Optional[int] = None, logical: Optional[str] = None):
        """values: numpy array (datetime64 / timedelta64 become Date / DateTime / Time columns), masked array, or a list of str / None.
        logical="Char" with 1-character strings makes a Char column."""
        L = N.load()
        if logical == "Char":
            values = np.array([ir.julia_char(c) for c in values], np.uint32)
        elif isinstance(values, np.ndarray) and values.dtype.kind in "Mm":
            fill = None
            if isinstance(values, np.ma.MaskedArray):
                fill = np.ma.getmaskarray(values); values = values.filled(values.dtype.type(0))
            if values.dtype.kind == "m":
                ints, logical = values.astype("timedelta64[ns]").astype(np.int64), "Time"
            elif np.datetime_data(values.dtype)[0] in ("D", "W", "M", "Y"):
                ints, logical = values.astype("datetime64[D]").astype(np.int64) + ir.RATA_DIE_DAYS, "Date"
            else:
                ints, logical = values.astype("datetime64[ms]").astype(np.int64) + ir.RATA_DIE_MS, "DateTime"
            values = np.ma.masked_array(ints, mask=fill) if fill is not None else ints
        if logical:
            self.add_column(name, values, dtype)
            N.check(L.dfdb_table_set_logical_type(self._h, self.ordinal(name), logical.encode()))
            return
        if isinstance(values, (list, tuple)) and (len(values) == 0 or isinstance(values[0], (str, bytes, type(None)))) and not isinstance(values, np.ndarray):
            enc = [None if v is None else (v.encode() if isinstance(v, str) else bytes(v)) for v in values]
            sizes = np.array([-1 if e is None else len(e) for e in enc], np.int32)
            data = np.frombuffer(b"".join(e for e in enc if e), np.uint8).copy()
            dt = ir.STRING | (ir.NULLABLE if any(e is None for e in enc) or (dtype or 0) & ir.NULLABLE else 0)
            N.check(L.dfdb_table_add_column(self._h, name.encode(), dt, len(sizes), sizes.ctypes.data, data.ctypes.data if len(data) else None,
                                            len(data), None))
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
        N.check(L.dfdb_table_add_column(self._h, name.encode(), dtype, len(arr), arr.ctypes.data if len(arr) else None, None, 0,
                                        missing.ctypes.data if missing is not None else None))

    def add_generated(self, name: str, generator: int, seed: int, nrows: int, row_first: int = 0):
        N.check(N.load().dfdb_table_add_generated(self._h, name.encode(), generator, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), row_first, nrows))

    def load(self, columns: Optional[Sequence[str]] = None, block_first: int = 0, block_last: int = -1) -> dict:
        """Decode blocks [block_first, block_last) of the listed columns into HBM (read_block! for every block)."""
        st = N.SizeStats()
        if columns is None:
            N.check(N.load().dfdb_table_load(self._h, None, 0, block_first, block_last, C.byref(st)))
        else:
            ords = (C.c_int32 * len(columns))(*[self.ordinal(c) for c in columns])
            N.check(N.load().dfdb_table_load(self._h, ords, len(columns), block_first, block_last, C.byref(st)))
        return dict(rows=st.rows, compressed=st.compressed, uncompressed=st.uncompressed)

    def load_image(self, column: str, image: bytes, block_first: int = 0, block_last: int = -1) -> dict:
        st = N.SizeStats()
        N.check(N.load().dfdb_table_load_image(self._h, self.ordinal(column), image, len(image), block_first, block_last, C.byref(st)))
        return dict(rows=st.rows, compressed=st.compressed, uncompressed=st.uncompressed)

    def decode_resident(self, column: str):
        """K7 over the LZ4 blocks the column kept in HBM (ctx option keep_compressed at load time), asynchronously"""
        N.check(N.load().dfdb_table_decode_resident(self._h, self.ordinal(column)))

    def decode_status(self, column: str) -> int:
        """blocks of the column's last resident decode that did not end with their stored size (0 = fine); waits for the stream"""
        bad = C.c_int64()
        N.check(N.load().dfdb_table_decode_status(self._h, self.ordinal(column), C.byref(bad)))
        return bad.value

    def compress_column(self, column: str, mode: int = 2) -> dict:
        """a resident plain fixed-width column -> compressed-resident (mode 1) or compressed-only (mode 2) in HBM, encoded on the device: dfdb_table_compress_column"""
        st = N.SizeStats()
        N.check(N.load().dfdb_table_compress_column(self._h, self.ordinal(column), mode, C.byref(st)))
        return dict(rows=st.rows, compressed=st.compressed, uncompressed=st.uncompressed)

    def resident_bytes(self, column=None) -> dict:
        """HBM bytes held right now by one column (or the whole table): {"decoded": ..., "compressed": ...} — dfdb_table_resident_bytes"""
        d, k = C.c_int64(), C.c_int64()
        N.check(N.load().dfdb_table_resident_bytes(self._h, -1 if column is None else self.ordinal(column), C.byref(d), C.byref(k)))
        return {"decoded": d.value, "compressed": k.value}

    def read_probe(self, column: str, repeats: int = 5):
        """(best_ms, avg_ms) of K1's read stream alone over a resident 8-byte column: dfdb_table_read_probe"""
        best, avg = C.c_double(), C.c_double()
        N.check(N.load().dfdb_table_read_probe(self._h, self.ordinal(column), repeats, C.byref(best), C.byref(avg)))
        return best.value, avg.value

    def build_dictionary(self, column: str, max_entries: int = 4096) -> int:
        """K9: 16-bit codes + the distinct strings of a resident String column (kept beside its flat form); returns the number of distinct
        strings, 0 when none was built (more than max_entries, nullable, a string over 4 KB).  Results of every query stay the same."""
        n = C.c_int64()
        N.check(N.load().dfdb_table_build_dictionary(self._h, self.ordinal(column), max_entries, C.byref(n)))
        return n.value

    def set_row_base(self, row_base: int): N.check(N.load().dfdb_table_set_row_base(self._h, row_base))

    def add_column_from(self, name: str, col) -> None:
        """add_column!(table, name, lazy_col) (table.jl:96-124): a DFColumn (or one-column DFView), plain or computed,
        filtered or not, becomes a new resident column of this table without leaving the device."""
        v = col.view if isinstance(col, DFColumn) else col
        if not isinstance(v, DFView) or len(v.projection) != 1:
            raise ValueError("ArgumentError: add_column_from needs a DFColumn or a one-column DFView")
        q = v._query()
        N.check(N.load().dfdb_table_add_from_query(self._h, name.encode(), q._h, 0))

    def save(self, path: str) -> dict:
        """create_table(path; from=table) (creators.jl:18-60): every resident column to `<path>/<id>.bin` in the reference's
        block format (bodies packed + LZ4-compressed on the device) and `<path>/meta.bin`.  Returns the SizeStats."""
        st = N.SizeStats()
        N.check(N.load().dfdb_table_save(self._h, os.fsencode(path), C.byref(st)))
        return {"rows": st.rows, "compressed": st.compressed, "uncompressed": st.uncompressed}

    def save_column(self, name: str, file: str) -> dict:
        st = N.SizeStats()
        N.check(N.load().dfdb_table_save_column(self._h, self.ordinal(name), os.fsencode(file), C.byref(st)))
        return {"rows": st.rows, "compressed": st.compressed, "uncompressed": st.uncompressed}

    def close(self):
        if self._h:
            N.load().dfdb_table_close(self._h)
            object.__setattr__(self, "_h", C.c_void_p())
            object.__setattr__(self, "is_opened", False)

    # -- metadata (table.jl:23-56)
    @property
    def ncols(self) -> int:
        n = C.c_int32()
        N.check(N.load().dfdb_table_ncols(self._h, C.byref(n)))
        return n.value

    def columns_meta(self) -> List[ColumnMeta]:
        out = []
        for i in range(self.ncols):
            ci = N.ColInfo()
            N.check(N.load().dfdb_table_colinfo(self._h, i, C.byref(ci)))
            out.append(ColumnMeta(ci.id, ci.name.decode(), ci.dtype, ci.logical.decode()))
        return out

    def names(self) -> List[str]: return [m.name for m in self.columns_meta()]

    def resident(self, ordinal: int) -> bool:
        ci = N.ColInfo()
        N.check(N.load().dfdb_table_colinfo(self._h, ordinal, C.byref(ci)))
        return bool(ci.resident)

    def ordinal(self, name: str) -> int:
        o = C.c_int32()
        N.check(N.load().dfdb_table_find_column(self._h, name.encode(), C.byref(o)))
        return o.value

    def getmeta(self, name: str) -> ColumnMeta: return self.columns_meta()[self.ordinal(name)]

    @property
    def blocksize(self) -> int:
        b = C.c_int64()
        N.check(N.load().dfdb_table_block_size(self._h, C.byref(b)))
        return b.value

    def expr_dtype(self, e: ir.Expr) -> int:
        dt = C.c_int32()
        b = e.to_ir()
        N.check(N.load().dfdb_expr_result_type(self._h, b, len(b), C.byref(dt)))
        return dt.value

    def __eq__(self, o):     # table.jl:18-21: path + column metas only (quirk Q12)
        return isinstance(o, DFTable) and (self is o or (self.path != "" and self.path == o.path and self.columns_meta() == o.columns_meta()))

    def __hash__(self): return id(self)

    # -- DFTable indexing proxies to the full view (view.jl:138,174-177)
    def view(self) -> "DFView": return DFView(self)
    def __getitem__(self, key): return DFView(self)[key]

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return DFView(self)[ALL, name]
        except (KeyError, ValueError):
            raise AttributeError(name) from None

    def __repr__(self): return f"DFTable({self.path!r}, {self.columns_meta()})"


def open_table(path: str, device: int = 0, ctx: Optional[Context] = None, load: bool = True, block_first: int = 0,
               block_last: int = -1) -> DFTable:
    """open_table(path) (creators.jl:7-16); with load=True all blocks in the range are decoded into HBM."""
    ctx = ctx or default_context(device)
    h = C.c_void_p()
    N.check(N.load().dfdb_table_open(ctx._h, path.encode(), C.byref(h)))
    t = DFTable(h, ctx, path)
    if load:
        t.load(None, block_first, block_last)
    return t


# ---------------------------------------------------------------- DFView (view.jl:26-232)
Selector = Any
Projector = Any


class DFView:
    def __init__(self, table: DFTable, projection: Optional[Projection] = None, selection: Optional[SelectionQueue] = None):
        self.table = table
        if projection is None:   # full_table_projection (view.jl:43-48)
            projection = Projection({m.name: ir.col(i) for i, m in enumerate(table.columns_meta())})
        self.projection = projection
        self.selection = selection or SelectionQueue()
        self._q = None

    # -- selection(v, el) (view.jl:60-72, column.jl:72-75)
    def _selection(self, el) -> "DFView":
        if _is_colon(el):
            return self
        if isinstance(el, DFColumn):
            if el.eltype != ir.BOOL:
                raise ValueError("ArgumentError: Function for selection must have Bool result type")
            if not self.selection.same(el.view.selection):
                raise ValueError("ArgumentError: col must have same selection as view")
            return DFView(self.table, self.projection, self.selection.add(el.expr))
        if isinstance(el, tuple) and len(el) == 2 and callable(el[1]):      # cols => f
            cols = [el[0]] if isinstance(el[0], str) else list(el[0])
            for c in cols:
                if c not in self.projection.cols:
                    raise ValueError(f"ArgumentError: view don't have column :{c}")
            args = list(self.projection.select_names(cols).cols.values())     # projection order (quirk Q13)
            e = _trace(el[1], args)
            if self.table.expr_dtype(e) != ir.BOOL:
                raise ValueError("ArgumentError: Function for selection must have Bool result type")   # selection.jl:52-55
            return DFView(self.table, self.projection, self.selection.add(e))
        if isinstance(el, ir.Expr):
            if self.table.expr_dtype(el) != ir.BOOL:
                raise ValueError("ArgumentError: Function for selection must have Bool result type")
            return DFView(self.table, self.projection, self.selection.add(el))
        if isinstance(el, (bool, np.bool_)):
            raise TypeError("Bool masks enter only as lazy DFColumn{Bool} (quirk Q4)")
        if isinstance(el, (int, np.integer)):
            return DFView(self.table, self.projection, self.selection.add(int(el)))
        if isinstance(el, _End):
            return DFView(self.table, self.projection, self.selection.add(el.resolve(nrow(self))))
        if isinstance(el, JRange):
            r = el.resolved(nrow(self)) if el.has_end() else el.resolved(0)
            if r.step == 0:
                raise ValueError("ArgumentError: step cannot be zero")
            return DFView(self.table, self.projection, self.selection.add(r))
        if isinstance(el, (list, np.ndarray)):
            arr = np.asarray(el)
            if arr.dtype == bool:
                raise TypeError("Bool masks enter only as lazy DFColumn{Bool} (quirk Q4)")
            return DFView(self.table, self.projection, self.selection.add(arr.astype(np.int64).tolist()))
        raise TypeError(f"unsupported selector {el!r}; use jr(a, b) for ranges (1-based, inclusive)")

    # -- projection(v, p) (view.jl:75-109)
    def _proj_elem(self, elem) -> ir.Expr:
        if isinstance(elem, str):
            if elem not in self.projection.cols:
                raise ValueError(f"ArgumentError: view don't have column :{elem}")
            return self.projection.cols[elem]
        if isinstance(elem, tuple) and len(elem) == 2 and callable(elem[1]):
            cols = [elem[0]] if isinstance(elem[0], str) else list(elem[0])
            return _trace(elem[1], [self._proj_elem(c) for c in cols])          # tuple order (quirk Q13)
        if isinstance(elem, DFColumn):
            return elem.expr
        raise TypeError(f"unsupported projection element {elem!r}")

    def _projection(self, p) -> "DFView":
        if _is_colon(p):
            return self
        if isinstance(p, dict):
            return DFView(self.table, Projection({k: self._proj_elem(v) for k, v in p.items()}), self.selection)
        if isinstance(p, JRange):
            return DFView(self.table, self.projection.select_positions(list(p.resolved(len(self.projection)))), self.selection)
        if isinstance(p, (list, tuple)):
            if all(isinstance(x, str) for x in p):
                if len(set(p)) != len(p):
                    raise ValueError(f"ArgumentError: Duplicated column")
                return self._projection({x: x for x in p})
            if all(isinstance(x, (int, np.integer)) for x in p):
                return DFView(self.table, self.projection.select_positions([int(x) for x in p]), self.selection)
            if all(isinstance(x, tuple) and len(x) == 2 and isinstance(x[0], str) for x in p):   # Vector{Pair}: [:a=>:a, :c=>:c=>f]
                return self._projection({k: v for k, v in p})
        raise TypeError(f"unsupported projector {p!r}")

    def __getitem__(self, key):
        if not isinstance(key, tuple) or len(key) != 2:
            raise TypeError("index a view as v[selector, projector]")
        s, p = key
        # getindex(v, s, p::Union{Number,Symbol}) -> DFColumn (view.jl:123-128)
        single = isinstance(p, (str, int, np.integer)) or (isinstance(p, _End)) or (isinstance(p, tuple) and len(p) == 2 and callable(p[1]))
        if single:
            if isinstance(p, _End):
                p = p.resolve(len(self.projection))
            base = self._selection(s)
            pv = base._projection({"a": p}) if isinstance(p, tuple) else base._projection([p])
            col = DFColumn(pv)
            if isinstance(s, (int, np.integer)) and not isinstance(s, bool):
                return col._scalar()                                          # view.jl:125
            return col
        if isinstance(s, (int, np.integer)) and not isinstance(s, bool):       # getindex(v, s::Number, p::Any) -> first row (view.jl:130-135)
            v = self._selection([int(s)])._projection(p)
            df = materialize(v)
            if len(df) == 0:
                raise IndexError(f"BoundsError: attempt to access view at index [{s}]")
            return {k: df[k].iloc[0] for k in df.columns}
        if _is_colon(s) and _is_colon(p):
            return self
        return self._selection(s)._projection(p)

    def __getattr__(self, name):   # getproperty(v, name) = v[:, name] (view.jl:167-170)
        if name.startswith("_") or name in ("table", "projection", "selection"):
            raise AttributeError(name)
        try:
            return self[ALL, name]
        except ValueError:
            raise AttributeError(name) from None

    def names(self) -> List[str]: return self.projection.keys()

    def __eq__(self, o):          # view.jl:34-38
        return isinstance(o, DFView) and self.table == o.table and self.projection.same(o.projection) and self.selection.same(o.selection)

    def __ne__(self, o): return not self.__eq__(o)
    def __hash__(self): return id(self)

    def required_columns(self) -> List[str]:   # view.jl:183-190
        ords: List[int] = []
        for e in self.projection.cols.values():
            for c in e.columns():
                if c not in ords:
                    ords.append(c)
        for st in self.selection.queue:
            if isinstance(st, ir.Expr):
                for c in st.columns():
                    if c not in ords:
                        ords.append(c)
        nm = self.table.names()
        return [nm[c] for c in ords]

    # -- engine query (built once per immutable view)
    def _query(self):
        if self._q is None:
            self._q = _Query(self)
        return self._q

    def __repr__(self):
        return f"View of table {self.table.path}\n{self.projection}\n{self.selection}"


def issameselection(a: DFView, b: DFView) -> bool:   # view.jl:180-181
    return a.table == b.table and a.selection.same(b.selection)


def selection(v: Union[DFView, DFTable], el) -> DFView:
    return (v if isinstance(v, DFView) else DFView(v))._selection(el)


def projection(v: Union[DFView, DFTable], p) -> DFView:
    return (v if isinstance(v, DFView) else DFView(v))._projection(p)


def selproj(v: DFView, select, project) -> DFView:   # view.jl:112-118
    return v._selection(select)._projection(project)


# ---------------------------------------------------------------- engine query wrapper
class _Query:
    def __init__(self, v: DFView):
        L = N.load()
        self.view = v
        self._h = C.c_void_p()
        N.check(L.dfdb_query_new(v.table._h, C.byref(self._h)))
        try:
            for st in v.selection.queue:
                if isinstance(st, ir.Expr):
                    b = st.to_ir()
                    N.check(L.dfdb_query_add_predicate(self._h, b, len(b)))
                elif isinstance(st, JRange):
                    N.check(L.dfdb_query_add_range(self._h, int(st.start), int(st.step), int(st.stop)))
                elif isinstance(st, int):
                    N.check(L.dfdb_query_add_integer(self._h, st))
                else:
                    a = np.ascontiguousarray(st, np.int64)
                    N.check(L.dfdb_query_add_indices(self._h, a.ctypes.data if len(a) else None, len(a)))
            items = [(k, e.to_ir()) for k, e in v.projection.cols.items()]
            n = len(items)
            names = (C.c_char_p * max(n, 1))(*[k.encode() for k, _ in items])
            bufs = [C.create_string_buffer(b, len(b)) for _, b in items]
            irs = (C.c_void_p * max(n, 1))(*[C.cast(b, C.c_void_p) for b in bufs])
            lens = (C.c_size_t * max(n, 1))(*[len(b) for _, b in items])
            N.check(L.dfdb_query_set_projection(self._h, n, names, irs, lens))
        except Exception:
            L.dfdb_query_free(self._h)
            self._h = C.c_void_p()
            raise

    def __del__(self):
        try:
            import sys
            if self._h and not sys.is_finalizing():    # never call into HIP while the interpreter tears down
                N.load().dfdb_query_free(self._h)
        except Exception:
            pass

    def execute(self): N.check(N.load().dfdb_query_execute(self._h))

    def reset(self): N.check(N.load().dfdb_query_reset(self._h))

    def count(self) -> int:
        n = C.c_int64()
        N.check(N.load().dfdb_count(self._h, C.byref(n)))
        return n.value

    def count_device(self, dev_ptr: int):
        N.check(N.load().dfdb_count_to(self._h, C.c_void_p(dev_ptr), N.MEM_DEVICE))

    def coltype(self, i: int) -> int:
        dt = C.c_int32()
        N.check(N.load().dfdb_query_coltype(self._h, i, C.byref(dt)))
        return dt.value

    def indices(self) -> np.ndarray:
        n = self.count()
        out = np.empty(n, np.int64)
        got = C.c_int64()
        N.check(N.load().dfdb_select_indices(self._h, out.ctypes.data if n else None, n, N.MEM_HOST, C.byref(got)))
        return out

    def indices_device(self, dev_ptr: int, cap: int, want_count: bool = False) -> Optional[int]:
        got = C.c_int64()
        N.check(N.load().dfdb_select_indices(self._h, C.c_void_p(dev_ptr), cap, N.MEM_DEVICE, C.byref(got) if want_count else None))
        return got.value if want_count else None

    def bitmap(self) -> np.ndarray:
        nrows = C.c_int64()
        N.check(N.load().dfdb_table_nrows(self.view.table._h, C.byref(nrows)))
        out = np.zeros((nrows.value + 63) // 64, np.uint64)
        if len(out):
            N.check(N.load().dfdb_select_bitmap(self._h, out.ctypes.data, N.MEM_HOST))
        return out

    def hint_materialize(self, on: bool = True):
        N.check(N.load().dfdb_query_hint_materialize(self._h, 1 if on else 0))

    def materialize(self) -> List[Any]:
        L = N.load()
        self.hint_materialize(True)       # count() below is the scan: let it keep projected predicate columns
        n = self.count()
        ncols = len(self.view.projection)
        outs = (N.OutCol * max(ncols, 1))()
        keep = []
        for i in range(ncols):
            dt = self.coltype(i)
            base = dt & ir.DTYPE_MASK
            o = outs[i]
            o.memkind = N.MEM_HOST
            if base == ir.STRING:
                nb = C.c_int64()
                N.check(L.dfdb_result_string_bytes(self._h, i, C.byref(nb)))
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
            N.check(L.dfdb_materialize(self._h, outs, ncols))
        res = []
        for i, (dt, a, b, m) in enumerate(keep):
            if (dt & ir.DTYPE_MASK) == ir.STRING:
                res.append((a[:n].copy(), b[:outs[i].nbytes].copy()))
            elif m is not None:
                res.append(np.ma.masked_array(a[:n].copy(), mask=m[:n].astype(bool)))
            else:
                res.append(a[:n].copy())
        return res

    def hint_aggregate(self, op: int, col: int = 0):
        N.check(N.load().dfdb_query_hint_aggregate(self._h, op, col))

    def aggregate(self, op: int, col: int = 0):
        oi, of = C.c_int64(), C.c_double()
        if op in (N.AGG_SUM, N.AGG_MIN, N.AGG_MAX):
            self.hint_aggregate(op, col)      # a first execution below lets the scan reduce the column while it holds it
        N.check(N.load().dfdb_aggregate(self._h, op, col, C.byref(oi), C.byref(of)))
        dt = self.coltype(col) & ir.DTYPE_MASK if op != N.AGG_COUNT else ir.I64
        if dt in (ir.F32, ir.F64):
            return of.value
        # Base.sum widens unsigned element types to UInt64 (Base.add_sum), minimum / maximum keep them: the 64 bits are an unsigned number
        return oi.value & 0xFFFFFFFFFFFFFFFF if dt in (ir.U8, ir.U16, ir.U32, ir.U64) else oi.value


class _ChunkQuery(_Query):
    """One chunk of a streamed view: the engine owns the handle (valid until the stream advances)."""

    def __init__(self, view: DFView, handle, chunk_rows: int, first_row: int):
        self.view, self._h, self.chunk_rows, self.first_row = view, handle, chunk_rows, first_row

    def __del__(self):
        pass

    def bitmap(self) -> np.ndarray:
        out = np.zeros((self.chunk_rows + 63) // 64, np.uint64)
        if len(out):
            N.check(N.load().dfdb_select_bitmap(self._h, out.ctypes.data, N.MEM_HOST))
        return out


DEFAULT_CHUNK_BLOCKS = 512      # blocks per chunk of the block-streamed consumers that open their own Stream (unique, aggregates, groupreduce over a table that is not resident)


class Stream:
    """Base.iterate(::BlocksIterator) in chunks of blocks (blocksiterator.jl:98-145) for a view over a table that was opened
    with load=False: `for part in dfdb.stream(v, chunk_blocks=256): part.count(), part.indices(), part.materialize()`.
    The next chunk is read, copied and LZ4-decoded on another HIP stream while the caller works on the current one."""

    def __init__(self, v: Union[DFView, DFTable], chunk_blocks: Optional[int] = None):
        chunk_blocks = DEFAULT_CHUNK_BLOCKS if chunk_blocks is None else chunk_blocks
        self.view = v if isinstance(v, DFView) else DFView(v)
        self._q = _Query(self.view)
        self._h = C.c_void_p()
        N.check(N.load().dfdb_stream_open(self._q._h, chunk_blocks, C.byref(self._h)))

    def stats(self) -> dict:
        st = N.SizeStats()
        N.check(N.load().dfdb_stream_stats(self._h, C.byref(st)))
        return {"rows": st.rows, "compressed": st.compressed, "uncompressed": st.uncompressed}

    def read_stats(self, column: Optional[str] = None) -> dict:
        """what the loaders have read so far of one column (None: of every required column): late materialization reads a projection-only
        column only for the blocks whose selection kept a row (blocksiterator.jl:111-113)"""
        st = N.SizeStats()
        o = -1 if column is None else self.view.table.names().index(column)
        N.check(N.load().dfdb_stream_read_stats(self._h, o, C.byref(st)))
        return {"rows": st.rows, "compressed": st.compressed, "uncompressed": st.uncompressed}

    def __iter__(self):
        return self

    def __next__(self) -> _ChunkQuery:
        if not self._h:
            raise StopIteration
        h, rows, first = C.c_void_p(), C.c_int64(), C.c_int64()
        N.check(N.load().dfdb_stream_next(self._h, C.byref(h), C.byref(rows), C.byref(first)))
        if not h:
            self.close()
            raise StopIteration
        return _ChunkQuery(self.view, h, rows.value, first.value)

    def close(self):
        if self._h:
            N.load().dfdb_stream_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def __del__(self):
        try:
            import sys
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass


def stream(v: Union[DFView, DFTable], chunk_blocks: int = 512) -> Stream:
    return Stream(v, chunk_blocks)


def _with_chunk_blocks(view: "DFView", chunk_blocks: Optional[int]):
    if chunk_blocks is not None:
        view.table.ctx.set_option("ooc_chunk_blocks", int(chunk_blocks))


def nrow_streamed(v: Union[DFView, DFTable], chunk_blocks: Optional[int] = None) -> int:
    """nrow(v) without holding the table in HBM (BlockRowsIterator, view.jl:192-206): ONE call — dfdb_count notices that the view's columns are not
    resident and streams the selection's columns (the first projection column when there is no predicate) inside the library (csrc/ooc.cpp)."""
    view = v if isinstance(v, DFView) else DFView(v)
    _with_chunk_blocks(view, chunk_blocks)
    return nrow(view)


def materialize_streamed(v: Union[DFView, DFTable], chunk_blocks: Optional[int] = None):
    """materialize(v) over a table that is not resident: dfdb_count + dfdb_materialize stream inside the library, every chunk's rows appended to the
    caller's buffers (materialization.jl:33-37)."""
    view = v if isinstance(v, DFView) else DFView(v)
    _with_chunk_blocks(view, chunk_blocks)
    return materialize(view)


def _flat_to_strings(sizes: np.ndarray, data: np.ndarray) -> List[Optional[str]]:
    out, o = [], 0
    raw = data.tobytes()
    for s in sizes.tolist():
        if s < 0:
            out.append(None)
        else:
            out.append(raw[o:o + s].decode(errors="replace"))
            o += s
    return out


# String columns come back from the engine as (Int32 sizes, byte arena) — a FlatStringsVector.  "object": one Python str per row like the
# reference's Vector{String} (projection.jl:99-100; ~0.2 us per row in the interpreter); "arrow": the same two buffers wrapped as a
# pyarrow large_string array inside a pandas ArrowExtensionArray — no per-row work, missing rows are nulls.
_STRING_OUTPUT = "object"


def set_string_output(kind: str) -> None:
    global _STRING_OUTPUT
    if kind not in ("object", "arrow"):
        raise ValueError("string output is 'object' or 'arrow'")
    _STRING_OUTPUT = kind


def _flat_to_arrow(sizes: np.ndarray, data: np.ndarray):
    import pyarrow as pa
    sizes = np.asarray(sizes, np.int32)
    off = np.zeros(len(sizes) + 1, np.int64)
    np.cumsum(np.where(sizes > 0, sizes, 0), out=off[1:])
    valid = np.packbits(sizes >= 0, bitorder="little")
    return pa.Array.from_buffers(pa.large_string(), len(sizes), [pa.py_buffer(valid), pa.py_buffer(off), pa.py_buffer(np.ascontiguousarray(data, np.uint8))])


# ---------------------------------------------------------------- nrow / size / materialize
def nrow(v: Union[DFView, DFTable, "DFColumn"]) -> int:      # view.jl:192-206
    if isinstance(v, DFTable):
        v = DFView(v)
    if isinstance(v, DFColumn):
        v = v.view
    if len(v.projection) == 0:
        return 0      # isempty(it.streams): nothing to read, zero rows (blocksiterator.jl:101)
    return v._query().count()      # (a view over columns that are not resident is block-streamed inside dfdb_count: csrc/ooc.cpp)


def _out_of_core(v: DFView) -> bool:
    """The view touches a column of a file-backed table that is not resident (open_table(path, load=False)): the library evaluates it
    block-streamed, the way the reference always does (blocksiterator.jl:98-145).  Informational: nothing here routes on it any more."""
    t = v.table
    if not t.path:
        return False
    need = set(v.required_columns()) or set(t.names()[:1])
    return any(not t.resident(t.ordinal(n)) for n in need)


def ncol(v: Union[DFView, DFTable]) -> int:
    return len((v if isinstance(v, DFView) else DFView(v)).projection)


def size(v, dim: Optional[int] = None):
    if isinstance(v, DFColumn):
        return (nrow(v),) if dim is None else (nrow(v) if dim == 1 else 1)
    if dim is None:
        return (nrow(v), ncol(v))
    if dim not in (1, 2):
        raise ValueError("ArgumentError: DFView have only 2 dimensions")
    return nrow(v) if dim == 1 else ncol(v)


def materialize(v: Union[DFView, DFTable, "DFColumn"]):
    """materialize(::DFView) -> pandas.DataFrame (stand-in for DataFrames.DataFrame); materialize(::DFColumn) -> array."""
    if isinstance(v, DFColumn):
        cols = v.view._query().materialize()
        return _to_user(cols[0], _logicals(v.view)[0])
    if isinstance(v, DFTable):
        v = DFView(v)
    import pandas as pd
    cols = v._query().materialize() if len(v.projection) else []
    lg = _logicals(v)
    return pd.DataFrame({k: _to_user(c, lg[i]) for i, (k, c) in enumerate(zip(v.projection.keys(), cols))})


def _logicals(v: DFView) -> List[str]:
    """per projection column: the bits type a plain column stands for ("" for ordinary dtypes and computed columns)"""
    metas = v.table.columns_meta()
    return [metas[e.payload].logical if e.op == ir.COL else "" for e in v.projection.cols.values()]


def _to_user(c, logical: str = ""):
    if isinstance(c, tuple):
        if _STRING_OUTPUT == "arrow":
            import pandas as pd
            return pd.arrays.ArrowExtensionArray(_flat_to_arrow(*c))
        return np.array(_flat_to_strings(*c), dtype=object)
    if logical:
        if isinstance(c, np.ma.MaskedArray):
            return np.ma.masked_array(ir.from_logical(np.asarray(c.data), logical), mask=np.ma.getmaskarray(c))
        return ir.from_logical(c, logical)
    return c


def head(v, rows: int = 10):      # materialization.jl:64-66
    v = v if isinstance(v, DFView) else DFView(v)
    return materialize(v[jr(1, rows), ALL])


# ---------------------------------------------------------------- DFColumn (column.jl:30-126, columnbroadcast.jl)
class DFColumn:
    def __init__(self, view: DFView):
        if len(view.projection) != 1:
            raise ValueError("ArgumentError: Column projection must contains singe element")
        self.view = view
        self._dtype: Optional[int] = None

    @property
    def expr(self) -> ir.Expr: return next(iter(self.view.projection.cols.values()))

    @property
    def eltype(self) -> int:
        if self._dtype is None:
            self._dtype = self.view.table.expr_dtype(self.expr)
        return self._dtype

    def _as_expr(self) -> ir.Expr: return self.expr

    def __len__(self): return nrow(self.view)
    def __eq_view__(self, o): return isinstance(o, DFColumn) and self.view == o.view

    def __getitem__(self, i):
        if isinstance(i, DFColumn):      # column.jl:63-67
            if not self.view.selection.same(i.view.selection):
                raise ValueError("ArgumentError: cols must have same selections")
            return DFColumn(self.view._selection(i))
        if isinstance(i, (int, np.integer)):
            return DFColumn(self.view._selection(int(i)))._scalar(i)
        if isinstance(i, _End):
            return self[i.resolve(len(self))]
        return DFColumn(self.view._selection(i))

    def _scalar(self, i=None):             # column.jl:93-99
        vals = self.view._query().materialize()[0]
        vals = _to_user(vals)
        if len(vals) == 0:
            raise IndexError(f"BoundsError: attempt to access column at index [{i}]")
        return vals[0]

    def materialize(self): return materialize(self)
    def collect(self): return materialize(self)
    def __iter__(self): return iter(materialize(self))

    def copyto(self, dest: np.ndarray):    # Base.copyto!(dest, src::DFColumn) (column.jl:83-91)
        vals = materialize(self)
        dest[:len(vals)] = vals
        return dest

    def unique(self):
        """unique(col) (docs/src/index.md:171-182): the distinct values in order of first appearance.  On the device the
        first occurrences are a selection (dfdb_query_unique); over a table that is not resident the same call reduces every chunk on the
        device and merges the (small) per-chunk results in order inside the library (csrc/ooc.cpp)."""
        def one(q):
            q.hint_materialize(True)
            q.execute()
            N.check(N.load().dfdb_query_unique(q._h, 0))
            return _to_user(q.materialize()[0], _logicals(self.view)[0])
        return one(self.view._query())

    def _aggregate(self, op: int, with_count: bool = False):
        """sum / min / max driven by Base.iterate(::DFColumn) in the reference (column.jl:102-126).  Over a table that is not resident
        every chunk is reduced on the device and the per-chunk results are combined in block order — inside dfdb_aggregate (csrc/ooc.cpp)."""
        # sum(f.(cols)) of a computed Bool column — `sum(ismissing.(t.col))`, docs/src/index.md:326-328 — is the number of selected rows for
        # which it holds: one more predicate on the selection (routed to the scan kernels like any other) instead of a materialised
        # Bool per row and a reduction over it
        if op == N.AGG_SUM and self.eltype == ir.BOOL and self.expr.op != ir.COL:
            cnt = nrow(DFView(self.view.table, self.view.projection, self.view.selection.add(self.expr)))
            return (cnt, nrow(self.view)) if with_count else cnt
        q = self.view._query()
        q.hint_aggregate(op)                           # before anything executes the selection
        r = q.aggregate(op)                            # (not resident: per-chunk reductions folded in block order inside the library)
        return (r, q.count()) if with_count else r

    def sum(self): return self._aggregate(N.AGG_SUM)
    def min(self): return self._aggregate(N.AGG_MIN)
    def max(self): return self._aggregate(N.AGG_MAX)

    def mean(self):
        s, n = self._aggregate(N.AGG_SUM, with_count=True)      # one evaluation of the selection gives both
        return float("nan") if n == 0 else s / n

    # -- broadcasting (columnbroadcast.jl:19-62): every DFColumn argument must share (table, selection)
    def _bc(self, op: int, other, swap: bool = False) -> "DFColumn":
        if isinstance(other, DFColumn):
            if not (self.view.table == other.view.table and self.view.selection.same(other.view.selection)):
                raise ValueError("ArgumentError: All columns in broadcast must have same selection and table")
        elif isinstance(other, (list, tuple, np.ndarray)):
            raise TypeError("broadcasting a DFColumn with an array falls back to array style in the reference (columnbroadcast.jl:16-17)")
        a, b = ir.wrap(self), ir.wrap(other)
        e = ir.Expr(op, (b, a) if swap else (a, b))
        return DFColumn(DFView(self.view.table, Projection({"a": e}), self.view.selection))

    def _un(self, op: int) -> "DFColumn":
        return DFColumn(DFView(self.view.table, Projection({"a": ir.Expr(op, (self.expr,))}), self.view.selection))

    def __add__(self, o): return self._bc(ir.ADD, o)
    def __radd__(self, o): return self._bc(ir.ADD, o, True)
    def __sub__(self, o): return self._bc(ir.SUB, o)
    def __rsub__(self, o): return self._bc(ir.SUB, o, True)
    def __mul__(self, o): return self._bc(ir.MUL, o)
    def __rmul__(self, o): return self._bc(ir.MUL, o, True)
    def __truediv__(self, o): return self._bc(ir.DIV, o)
    def __rtruediv__(self, o): return self._bc(ir.DIV, o, True)
    def __mod__(self, o): return self._bc(ir.REM, o)
    def __rmod__(self, o): return self._bc(ir.REM, o, True)
    def __neg__(self): return self._un(ir.NEG)
    def __abs__(self): return self._un(ir.ABS)
    def __invert__(self): return self._un(ir.NOT)
    def __and__(self, o): return self._bc(ir.AND, o)
    def __rand__(self, o): return self._bc(ir.AND, o, True)
    def __or__(self, o): return self._bc(ir.OR, o)
    def __ror__(self, o): return self._bc(ir.OR, o, True)
    def __xor__(self, o): return self._bc(ir.XOR, o)
    def __lt__(self, o): return self._bc(ir.LT, o)
    def __le__(self, o): return self._bc(ir.LE, o)
    def __gt__(self, o): return self._bc(ir.GT, o)
    def __ge__(self, o): return self._bc(ir.GE, o)
    def __eq__(self, o): return self._bc(ir.EQ, o)     # `.==` ; use col_equal(a, b) for DFColumn == DFColumn (column.jl:39)
    def __ne__(self, o): return self._bc(ir.NE, o)
    def __hash__(self): return id(self)
    def __bool__(self): raise TypeError("a DFColumn has no truth value: use & | ~ and split chained comparisons")

    def __repr__(self): return f"DFColumn{{{ir.dtype_name(self.eltype)}}}"


def col_equal(a: DFColumn, b: DFColumn) -> bool:
    """Base.:(==)(a::DFColumn, b::DFColumn) = a.view == b.view (column.jl:39)."""
    return a.view == b.view


def _fn_bc(maker):
    def f(c, *args):
        if isinstance(c, DFColumn):
            e = maker(c.expr, *args)
            return DFColumn(DFView(c.view.table, Projection({"a": e}), c.view.selection))
        return maker(c, *args)
    return f


startswith = _fn_bc(ir.startswith)
endswith = _fn_bc(ir.endswith)
ismissing = _fn_bc(ir.ismissing)
isin = _fn_bc(ir.isin)
sizeof = _fn_bc(ir.sizeof)
float64 = _fn_bc(ir.float64)


def coalesce(c, default):
    """coalesce.(col, default): DFColumn or Expr; `default` may itself be a column of the same view."""
    if isinstance(c, DFColumn):
        d = default.expr if isinstance(default, DFColumn) else default
        return DFColumn(DFView(c.view.table, Projection({"a": ir.coalesce(c.expr, d)}), c.view.selection))
    return ir.coalesce(c, default)


def view_from_columns(**cols: DFColumn) -> DFView:
    """DFView(a = col1, g = col2) (column.jl:143-164)."""
    first = None
    for c in cols.values():
        if first is None:
            first = c
        elif not (c.view.table == first.view.table and c.view.selection.same(first.view.selection)):
            raise ValueError("ArgumentError: All columns must have same selection and table")
    if first is None:
        raise ValueError("ArgumentError: no columns")
    return DFView(first.view.table, Projection({k: c.expr for k, c in cols.items()}), first.view.selection)


def table_stats(t: DFTable):
    """table_stats(table) (misc.jl:6-43): rows, uncompressed and compressed size and ratio per column + the table total, from the
    block headers of the column files (numbers, not the reference's pretty-printed strings)."""
    import pandas as pd
    rows = []
    tot = [0, 0, 0]
    for i, m in enumerate(t.columns_meta()):
        st = N.SizeStats()
        N.check(N.load().dfdb_table_column_stats(t._h, i, C.byref(st)))
        rows.append((m.name, m.type, st.rows, st.uncompressed, st.compressed, st.uncompressed / st.compressed if st.compressed else float("nan")))
        tot = [st.rows, tot[1] + st.uncompressed, tot[2] + st.compressed]
    if rows:
        rows.append(("Table total", "", tot[0], tot[1], tot[2], tot[1] / tot[2] if tot[2] else float("nan")))
    return pd.DataFrame(rows, columns=["column", "type", "rows", "uncompressed size", "compressed size", "compression ratio"])


def create_table(path: str, from_=None, block_size: int = 65536, ctx: Optional[Context] = None, **columns) -> "DFTable":
    """create_table(path; from=..., block_size=...) (creators.jl:18-60).  `from_`: a dict of host columns, a DFTable, or a
    DFView / DFColumn(s) to materialise (on the device) first.  Writes the table directory and returns the opened table."""
    src = from_ if from_ is not None else columns
    if isinstance(src, DFTable):
        t = src
    elif isinstance(src, (DFView, DFColumn)):
        v = src.view if isinstance(src, DFColumn) else src
        t = DFTable.new(block_size=block_size, ctx=v.table.ctx)
        for name in v.names():
            t.add_column_from(name, v[ALL, name])
    else:
        t = DFTable.from_columns(dict(src), block_size=block_size, ctx=ctx)
    t.save(path)
    return open_table(path, ctx=t.ctx)


def map_to_column(f: Callable, v: Union[DFView, DFTable]) -> DFColumn:   # view.jl:160-164
    v = v if isinstance(v, DFView) else DFView(v)
    return v[ALL, (tuple(v.names()), f)]


# ---------------------------------------------------------------- groupreduce (aggregate.jl:1-36, completed to its evident intent)
_STATS = {"count": N.AGG_COUNT, "sum": N.AGG_SUM, "min": N.AGG_MIN, "minimum": N.AGG_MIN, "max": N.AGG_MAX, "maximum": N.AGG_MAX, "mean": N.AGG_SUM}


def _groupreduce_view(v: Union[DFView, DFTable], by: str, col: Optional[str], stat: str):
    v = v if isinstance(v, DFView) else DFView(v)
    if stat not in _STATS:
        raise ValueError(f"ArgumentError: unknown statistic {stat}")
    names = [by] if col is None or stat == "count" else [by, col]
    return (v[ALL, names] if len(names) > 1 else DFView(v.table, Projection({by: v.projection.cols[by]}), v.selection)), len(names) > 1


def _groupreduce_raw(q: "_Query", with_value: bool, stat: str):
    """dfdb_query_groupreduce + _fetch on one query handle -> (keys, counts, values as Int64 bits, values as Float64, value dtype or None);
    keys are a numpy array, a masked array (nullable key) or a list of str / None."""
    L = N.load()
    ng, kb = C.c_int64(), C.c_int64()
    N.check(L.dfdb_query_groupreduce(q._h, 0, 1 if with_value else -1, _STATS[stat], C.byref(ng), C.byref(kb)))
    n = ng.value
    kdt = q.coltype(0)
    out = N.OutCol()
    out.memkind = N.MEM_HOST
    if (kdt & ir.DTYPE_MASK) == ir.STRING:
        ksz = np.empty(max(n, 1), np.int32); kby = np.empty(max(kb.value, 1), np.uint8)
        out.data, out.bytes, out.bytes_cap = ksz.ctypes.data, kby.ctypes.data, kb.value
    else:
        karr = np.empty(max(n, 1), ir.numpy_of_dtype(kdt))
        kmiss = np.zeros(max(n, 1), np.uint8) if kdt & ir.NULLABLE else None
        out.data = karr.ctypes.data
        if kmiss is not None:
            out.missing = kmiss.ctypes.data
    counts = np.zeros(max(n, 1), np.int64); vi = np.zeros(max(n, 1), np.int64); vf = np.zeros(max(n, 1), np.float64)
    N.check(L.dfdb_query_groupreduce_fetch(q._h, C.byref(out), counts.ctypes.data, vi.ctypes.data, vf.ctypes.data))
    if (kdt & ir.DTYPE_MASK) == ir.STRING:
        keys = _to_user((ksz[:n].copy(), kby[:out.nbytes].copy()))
    elif kdt & ir.NULLABLE:
        keys = np.ma.masked_array(karr[:n].copy(), mask=kmiss[:n].astype(bool))
    else:
        keys = karr[:n].copy()
    return keys, counts[:n].copy(), vi[:n].copy(), vf[:n].copy(), (q.coltype(1) & ir.DTYPE_MASK if with_value else None)


def _groupreduce_frame(by: str, stat: str, keys, counts, vi, vf, vdt):
    import pandas as pd
    res = {by: keys, "count": counts}
    if stat != "count":
        isf = vdt in (ir.F32, ir.F64)
        uns = vdt in (ir.U8, ir.U16, ir.U32, ir.U64)
        vals = vf if isf else (vi.astype(np.uint64) if uns and stat != "mean" else vi)
        if stat == "mean":
            vals = (vf if isf else (vi.astype(np.uint64).astype(np.float64) if uns else vi.astype(np.float64))) / np.maximum(counts, 1)
        res[stat] = vals
    return pd.DataFrame(res)


def groupreduce(v: Union[DFView, DFTable], by: str, col: Optional[str] = None, stat: str = "count"):
    """groupreduce(view, (:by,); out = :col => Stat()): one row per distinct value of `by` over the view's selected rows, in order of first
    appearance (the reference's group_map numbering), with the group's row count and stat(col) — stat in count / sum / min / max / mean.
    Returns a pandas.DataFrame with columns [by, "count", stat]."""
    sub, with_value = _groupreduce_view(v, by, col, stat)
    return _groupreduce_frame(by, stat, *_groupreduce_raw(_Query(sub), with_value, stat))
