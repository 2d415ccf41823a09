"""ctypes binding of oracle/liboracle.so — the CPU restatement of the reference path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (dataframedbs.jl_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

I8, I16, I32, I64, U8, U16, U32, U64, F32, F64, BOOL, STRING = range(1, 13)
NULLABLE = 0x80
_NP = {I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64, U8: np.uint8, U16: np.uint16, U32: np.uint32,
       U64: np.uint64, F32: np.float32, F64: np.float64, BOOL: np.bool_}
_ERRORS = {1: ValueError, 2: OSError, 3: OSError, 4: KeyError, 5: IndexError, 6: ZeroDivisionError, 7: NotImplementedError,
           9: MemoryError}


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (the Makefile in this directory)."""
    srcs = [os.path.join(_HERE, f) for f in ("orc_codec.c", "orc_expr.c", "orc_view.c", "oracle.h", "orc_internal.h")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


class _OutCol(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("count", C.c_int64), ("data", C.c_void_p), ("bytes", C.c_void_p),
                ("nbytes", C.c_int64), ("missing", C.c_void_p)]


class _Stats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("compressed", C.c_int64), ("uncompressed", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_gen_str_brands10.restype = C.c_int64
        L.orc_table_block_size.restype = C.c_int64
        L.orc_table_image.restype = C.c_void_p
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise _ERRORS.get(rc, RuntimeError)(lib().orc_last_error().decode())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def splitmix64(x: int) -> int:
    return lib().orc_splitmix64(C.c_uint64(x & 0xFFFFFFFFFFFFFFFF))


def gen_i64(seed: int, row_first: int, n: int) -> np.ndarray:
    out = np.empty(n, np.int64)
    lib().orc_gen_i64_mod1m(C.c_uint64(seed), C.c_int64(row_first), C.c_int64(n), _ptr(out))
    return out


def gen_f64(seed: int, row_first: int, n: int) -> np.ndarray:
    out = np.empty(n, np.float64)
    lib().orc_gen_f64_u2000(C.c_uint64(seed), C.c_int64(row_first), C.c_int64(n), _ptr(out))
    return out


def gen_str(seed: int, row_first: int, n: int) -> Tuple[np.ndarray, np.ndarray]:
    sizes = np.empty(n, np.int32)
    data = np.empty(9 * n + 16, np.uint8)
    tot = lib().orc_gen_str_brands10(C.c_uint64(seed), C.c_int64(row_first), C.c_int64(n), _ptr(sizes), _ptr(data))
    return sizes, data[:tot].copy()


def strings_to_flat(values: Sequence[Optional[str]]) -> Tuple[np.ndarray, np.ndarray]:
    """FlatStringsVector(source) (FlatStringsVectors.jl:29-51): sizes (-1 = missing) + byte arena."""
    enc = [None if v is None else (v.encode() if isinstance(v, str) else bytes(v)) for v in values]
    sizes = np.array([-1 if e is None else len(e) for e in enc], np.int32)
    data = np.frombuffer(b"".join(e for e in enc if e), np.uint8).copy() if any(enc) else np.zeros(0, np.uint8)
    return sizes, data


def flat_to_strings(sizes: np.ndarray, data: np.ndarray) -> List[Optional[str]]:
    out, o = [], 0
    raw = data.tobytes()
    for s in sizes.tolist():
        if s < 0:
            out.append(None)
        else:
            out.append(raw[o:o + s].decode())
            o += s
    return out


class Table:
    """DFTable restated: in-memory images of meta.bin / <id>.bin in the reference's on-disk format."""

    def __init__(self, block_size: int = 65536, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            _check(lib().orc_table_create(C.c_int64(block_size), C.byref(self._h)))
        self._keep: list = []

    @classmethod
    def open(cls, path: str) -> "Table":
        h = C.c_void_p()
        _check(lib().orc_table_open(path.encode(), C.byref(h)))
        return cls(_handle=h)

    def save(self, path: str):
        _check(lib().orc_table_save(self._h, path.encode()))

    def close(self):
        if self._h:
            lib().orc_table_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def block_size(self) -> int:
        return lib().orc_table_block_size(self._h)

    @property
    def ncols(self) -> int:
        return lib().orc_table_ncols(self._h)

    def colinfo(self, i: int):
        cid, dt = C.c_int64(), C.c_int32()
        name = C.create_string_buffer(128)
        _check(lib().orc_table_colinfo(self._h, i, C.byref(cid), name, C.c_size_t(128), C.byref(dt)))
        return cid.value, name.value.decode(), dt.value

    def logical(self, i: int) -> str:
        """"Date" / "DateTime" / "Time" / "Char" when the column is one of those bits types carried as an integer, else ""."""
        buf = C.create_string_buffer(32)
        _check(lib().orc_table_col_logical(self._h, i, buf, C.c_size_t(32)))
        return buf.value.decode()

    def names(self) -> List[str]:
        return [self.colinfo(i)[1] for i in range(self.ncols)]

    def find(self, name: str) -> int:
        i = lib().orc_table_find(self._h, name.encode())
        if i < 0:
            raise KeyError(name)
        return i

    def add_column(self, name: str, values, dtype: Optional[int] = None, missing: Optional[np.ndarray] = None, logical: Optional[str] = None):
        """Write a whole column block by block (write_column: columns.jl:30-53)."""
        if isinstance(values, tuple):                      # (sizes, bytes) flat strings
            sizes, data = values
            dt = STRING | (NULLABLE if (sizes < 0).any() or (dtype or 0) & NULLABLE else 0)
            sizes = np.ascontiguousarray(sizes, np.int32)
            data = np.ascontiguousarray(data, np.uint8)
            _check(lib().orc_table_add_column(self._h, name.encode(), dt, C.c_int64(len(sizes)), _ptr(sizes), _ptr(data), None))
            return
        if isinstance(values, (list, tuple)) and values and isinstance(values[0], (str, type(None))):
            return self.add_column(name, strings_to_flat(values), dtype)
        arr = np.ascontiguousarray(values)
        if dtype is None:
            from_np = {np.dtype(v): k for k, v in _NP.items()}
            dtype = from_np[arr.dtype]
        arr = arr.astype(_NP[dtype & 0x3F], copy=False)
        m = None
        if missing is not None:
            dtype |= NULLABLE
            m = np.ascontiguousarray(missing, np.uint8)
        _check(lib().orc_table_add_column_as(self._h, name.encode(), dtype, logical.encode() if logical else None, C.c_int64(len(arr)), _ptr(arr), None, _ptr(m)))

    def image(self, i: int) -> bytes:
        n = C.c_size_t()
        p = lib().orc_table_image(self._h, i, C.byref(n))
        return C.string_at(p, n.value)

    def column_stats(self, i: int):
        st, nb = _Stats(), C.c_int64()
        _check(lib().orc_table_column_stats(self._h, i, C.byref(st), C.byref(nb)))
        return dict(rows=st.rows, compressed=st.compressed, uncompressed=st.uncompressed, blocks=nb.value)

    def view(self) -> "View":
        return View(self)


class View:
    """DFView restated (projection + SelectionQueue)."""

    def __init__(self, table: Table):
        self.table = table
        self._h = C.c_void_p()
        _check(lib().orc_view_new(table._h, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                lib().orc_view_free(self._h)
        except Exception:
            pass

    # selection(v, …)
    def add_range(self, start: int, step: int, stop: int) -> "View":
        _check(lib().orc_view_add_range(self._h, C.c_int64(start), C.c_int64(step), C.c_int64(stop)))
        return self

    def add_integer(self, i: int) -> "View":
        _check(lib().orc_view_add_integer(self._h, C.c_int64(i)))
        return self

    def add_indices(self, idx) -> "View":
        a = np.ascontiguousarray(idx, np.int64)
        _check(lib().orc_view_add_indices(self._h, _ptr(a), C.c_int64(len(a))))
        return self

    def add_predicate(self, ir: bytes) -> "View":
        _check(lib().orc_view_add_predicate(self._h, ir, C.c_size_t(len(ir))))
        return self

    @property
    def nstages(self) -> int:
        return lib().orc_view_nstages(self._h)

    def stage(self, i: int):
        kind, a, s, b, n = C.c_int(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _check(lib().orc_view_stage(self._h, i, C.byref(kind), C.byref(a), C.byref(s), C.byref(b), C.byref(n)))
        return dict(kind=("range", "integer", "indices", "predicate")[kind.value], start=a.value, step=s.value, stop=b.value, n=n.value)

    def set_projection(self, items: Sequence[Tuple[str, bytes]]) -> "View":
        n = len(items)
        names = (C.c_char_p * n)(*[nm.encode() for nm, _ in items])
        irs = [C.create_string_buffer(ir, len(ir)) for _, ir in items]   # IR bytes contain NULs: keep real buffers alive
        irp = (C.c_void_p * n)(*[C.cast(b, C.c_void_p) for b in irs])
        lens = (C.c_size_t * n)(*[len(ir) for _, ir in items])
        self._keep = (names, irs, irp, lens)
        _check(lib().orc_view_set_projection(self._h, n, names, irp, lens))
        return self

    @property
    def ncols(self) -> int:
        return lib().orc_view_ncols(self._h)

    def coltype(self, i: int) -> int:
        dt = C.c_int32()
        _check(lib().orc_view_coltype(self._h, i, C.byref(dt)))
        return dt.value

    def required_columns(self) -> List[int]:
        buf = (C.c_int32 * 256)()
        n = lib().orc_view_required_columns(self._h, buf, 256)
        return list(buf[:n])

    def nrow(self) -> int:
        n = C.c_int64()
        _check(lib().orc_nrow(self._h, C.byref(n)))
        return n.value

    def materialize(self, count_pass: bool = True):
        """-> list of numpy arrays; String columns come back as (sizes, bytes)."""
        n = self.ncols
        outs = (_OutCol * n)()
        fn = lib().orc_materialize if count_pass else lib().orc_materialize_nocount
        _check(fn(self._h, outs, n))
        res = []
        try:
            for o in outs:
                base = o.dtype & 0x3F
                if base == STRING:
                    sizes = np.ctypeslib.as_array(C.cast(o.data, C.POINTER(C.c_int32)), (o.count,)).copy() if o.count else np.zeros(0, np.int32)
                    data = np.ctypeslib.as_array(C.cast(o.bytes, C.POINTER(C.c_uint8)), (o.nbytes,)).copy() if o.nbytes else np.zeros(0, np.uint8)
                    res.append((sizes, data))
                else:
                    npdt = np.dtype(_NP[base])
                    if o.count:
                        raw = C.string_at(o.data, o.count * npdt.itemsize)
                        arr = np.frombuffer(raw, npdt).copy()
                    else:
                        arr = np.zeros(0, npdt)
                    if o.dtype & NULLABLE:
                        m = np.frombuffer(C.string_at(o.missing, o.count), np.uint8).astype(bool) if o.count else np.zeros(0, bool)
                        res.append(np.ma.masked_array(arr, mask=m))
                    else:
                        res.append(arr)
        finally:
            lib().orc_outcols_free(outs, n)
        return res

    def select_indices(self) -> np.ndarray:
        n = C.c_int64()
        _check(lib().orc_select_indices(self._h, None, C.c_int64(0), C.byref(n)))
        out = np.empty(n.value, np.int64)
        _check(lib().orc_select_indices(self._h, _ptr(out), C.c_int64(n.value), C.byref(n)))
        return out

    def select_bitmap(self, nrows: int) -> np.ndarray:
        nw = (nrows + 63) // 64
        out = np.zeros(nw, np.uint64)
        _check(lib().orc_select_bitmap(self._h, _ptr(out), C.c_int64(nw)))
        return out

    def sum_f64(self, col: int = 0) -> float:
        v = C.c_double()
        _check(lib().orc_sum_f64(self._h, col, C.byref(v)))
        return v.value

    def sum_i64(self, col: int = 0) -> int:
        v = C.c_int64()
        _check(lib().orc_sum_i64(self._h, col, C.byref(v)))
        return v.value

    def bench_scan(self, cap: int):
        out = np.empty(max(cap, 1), np.int64)
        n, sec = C.c_int64(), C.c_double()
        _check(lib().orc_bench_scan(self._h, _ptr(out), C.c_int64(cap), C.byref(n), C.byref(sec)))
        return n.value, sec.value, out[:min(n.value, cap)]


class SelExec:
    """SelectionExecutor over caller blocks (test/selection.jl:40-106)."""

    def __init__(self, view: View):
        self.view = view
        self._h = C.c_void_p()
        _check(lib().orc_selexec_new(view._h, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                lib().orc_selexec_free(self._h)
        except Exception:
            pass

    def apply(self, rows: int, cols: Optional[dict] = None) -> np.ndarray:
        ncols = self.view.table.ncols
        arr = (C.c_void_p * max(ncols, 1))()
        keep = []
        for k, v in (cols or {}).items():
            a = np.ascontiguousarray(v)
            keep.append(a)
            arr[k] = a.ctypes.data
        mask = np.zeros(rows, np.uint8)
        n = C.c_int64()
        _check(lib().orc_selexec_apply(self._h, C.c_int64(rows), arr, _ptr(mask), C.byref(n)))
        assert n.value == int(mask.sum())
        return mask.astype(bool)

    def is_finished(self) -> bool:
        return bool(lib().orc_selexec_is_finished(self._h))

    def skip_if_can(self, size: int) -> bool:
        return bool(lib().orc_selexec_skip_if_can(self._h, C.c_int64(size)))


def expr_result_type(table: Table, ir: bytes) -> int:
    dt = C.c_int32()
    _check(lib().orc_expr_result_type(table._h, ir, C.c_size_t(len(ir)), C.byref(dt)))
    return dt.value


def expr_required_columns(table: Table, ir: bytes) -> List[int]:
    buf = (C.c_int32 * 256)()
    n = lib().orc_expr_required_columns(table._h, ir, C.c_size_t(len(ir)), buf, 256)
    if n < 0:
        _check(-n)
    return list(buf[:n])


def block_encode(body: bytes, rows: int) -> bytes:
    cap = len(body) + len(body) // 255 + 64
    out = C.create_string_buffer(cap)
    w = C.c_size_t()
    _check(lib().orc_block_encode(body, C.c_int64(len(body)), C.c_int32(rows), out, C.c_size_t(cap), C.byref(w)))
    return out.raw[:w.value]


def block_decode(buf: bytes, offset: int = 0):
    rows, origin, comp = C.c_int32(), C.c_int64(), C.c_int64()
    view = buf[offset:]
    _check(lib().orc_block_sizes(view, C.c_size_t(len(view)), C.byref(rows), C.byref(origin), C.byref(comp)))
    out = C.create_string_buffer(max(origin.value, 1))
    consumed = C.c_size_t()
    _check(lib().orc_block_decode(view, C.c_size_t(len(view)), out, C.c_size_t(origin.value), C.byref(rows), C.byref(origin), C.byref(consumed)))
    return rows.value, out.raw[:origin.value], consumed.value
