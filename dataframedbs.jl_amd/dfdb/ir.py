"""Expression IR front-end: the Python stand-in for Julia's `BlockBroadcasting` trees.

The reference builds a tree of `BlockBroadcasting(f, args)` over `ColRef{T}` leaves and 0-dim scalars
(reference: src/tables/broadcast.jl:2-31) either from `cols => closure` pairs (src/tables/view.jl:64-70)
or from `DFColumn` broadcasting (src/tables/columnbroadcast.jl:35-62).  Here the same tree is obtained by
calling the user's function on symbolic `Expr` leaves (operator overloading = the tracer SURVEY.md §7
item 9 describes) and serialised to the postfix byte stream of include/dfdb_ir.h.

Operators follow JULIA semantics, because the mirrored tests are Julia tests: `%` is `rem` (sign of the
dividend), `/` on integers yields Float64, `&`/`|` are non-short-circuit.  Use `mod()`/`div()` for Julia's
`mod`/`÷`.  Python's `and`/`or`/chained comparisons cannot be traced: `Expr.__bool__` raises.
"""
from __future__ import annotations

import struct
from typing import Any, Iterable, List, Sequence

import numpy as np

# ---- dtype enum (include/dfdb_ir.h) ------------------------------------------------------------
I8, I16, I32, I64, U8, U16, U32, U64, F32, F64, BOOL, STRING = range(1, 13)
NULLABLE = 0x80
DTYPE_MASK = 0x3F

_NP_TO_DT = {
    np.dtype("int8"): I8, np.dtype("int16"): I16, np.dtype("int32"): I32, np.dtype("int64"): I64,
    np.dtype("uint8"): U8, np.dtype("uint16"): U16, np.dtype("uint32"): U32, np.dtype("uint64"): U64,
    np.dtype("float32"): F32, np.dtype("float64"): F64, np.dtype("bool"): BOOL,
}
_DT_TO_NP = {v: k for k, v in _NP_TO_DT.items()}
_DT_NAME = {I8: "Int8", I16: "Int16", I32: "Int32", I64: "Int64", U8: "UInt8", U16: "UInt16", U32: "UInt32",
            U64: "UInt64", F32: "Float32", F64: "Float64", BOOL: "Bool", STRING: "String"}


def dtype_of_numpy(dt) -> int:
    return _NP_TO_DT[np.dtype(dt)]


def numpy_of_dtype(dt: int):
    return _DT_TO_NP[dt & DTYPE_MASK]


def dtype_name(dt: int) -> str:
    n = _DT_NAME[dt & DTYPE_MASK]
    return f"Missing({n})" if dt & NULLABLE else n


def dtype_width(dt: int) -> int:
    b = dt & DTYPE_MASK
    return 0 if b == STRING else np.dtype(_DT_TO_NP[b]).itemsize


# ---- opcodes (include/dfdb_ir.h) ------------------------------------------------------------------
COL, CONST, CONST_STR, CONST_SET = 0x01, 0x02, 0x03, 0x04
ADD, SUB, MUL, DIV, IDIV, REM, MOD, NEG, ABS, MIN, MAX = 0x10, 0x11, 0x12, 0x13, 0x14, 0x15, 0x16, 0x17, 0x18, 0x19, 0x1A
EQ, NE, LT, LE, GT, GE = 0x20, 0x21, 0x22, 0x23, 0x24, 0x25
AND, OR, XOR, NOT = 0x30, 0x31, 0x32, 0x33
IN_SET, STARTSWITH, ENDSWITH, ISMISSING, SIZEOF, COALESCE = 0x40, 0x41, 0x42, 0x43, 0x44, 0x45
CAST = 0x50


class Expr:
    """One node of the expression tree (a `BlockBroadcasting`, a `ColRef` or a scalar)."""

    __slots__ = ("op", "args", "payload")

    def __init__(self, op: int, args: Sequence["Expr"] = (), payload: Any = None):
        self.op, self.args, self.payload = op, tuple(args), payload

    # -- structural equality: the reference compares views by their projection/selection objects
    #    (view.jl:34-38); two traces of the same function give the same tree here.
    def same(self, other: "Expr") -> bool:
        if not isinstance(other, Expr) or self.op != other.op or len(self.args) != len(other.args):
            return False
        if self.op == CONST:
            a, b = self.payload, other.payload
            if a[0] != b[0] or _const_bytes(*a) != _const_bytes(*b):
                return False
        elif self.op == CONST_SET:
            if self.payload[0] != other.payload[0] or list(self.payload[1]) != list(other.payload[1]):
                return False
        elif self.payload != other.payload:
            return False
        return all(x.same(y) for x, y in zip(self.args, other.args))

    def __hash__(self):
        return hash((self.op, len(self.args)))

    def __bool__(self):
        raise TypeError("an Expr has no truth value: use & | ~ instead of and/or/not, and split chained "
                        "comparisons (a > 1) & (a < 5)")

    # arithmetic
    def __add__(self, o): return Expr(ADD, (self, wrap(o)))
    def __radd__(self, o): return Expr(ADD, (wrap(o), self))
    def __sub__(self, o): return Expr(SUB, (self, wrap(o)))
    def __rsub__(self, o): return Expr(SUB, (wrap(o), self))
    def __mul__(self, o): return Expr(MUL, (self, wrap(o)))
    def __rmul__(self, o): return Expr(MUL, (wrap(o), self))
    def __truediv__(self, o): return Expr(DIV, (self, wrap(o)))
    def __rtruediv__(self, o): return Expr(DIV, (wrap(o), self))
    def __mod__(self, o): return Expr(REM, (self, wrap(o)))       # Julia `%` == rem
    def __rmod__(self, o): return Expr(REM, (wrap(o), self))
    def __floordiv__(self, o): raise TypeError("`//` is ambiguous between Julia ÷ and fld: use dfdb.div(a, b)")
    def __neg__(self): return Expr(NEG, (self,))
    def __abs__(self): return Expr(ABS, (self,))
    # comparisons
    def __eq__(self, o): return Expr(EQ, (self, wrap(o)))       # type: ignore[override]
    def __ne__(self, o): return Expr(NE, (self, wrap(o)))       # type: ignore[override]
    def __lt__(self, o): return Expr(LT, (self, wrap(o)))
    def __le__(self, o): return Expr(LE, (self, wrap(o)))
    def __gt__(self, o): return Expr(GT, (self, wrap(o)))
    def __ge__(self, o): return Expr(GE, (self, wrap(o)))
    # logic
    def __and__(self, o): return Expr(AND, (self, wrap(o)))
    def __rand__(self, o): return Expr(AND, (wrap(o), self))
    def __or__(self, o): return Expr(OR, (self, wrap(o)))
    def __ror__(self, o): return Expr(OR, (wrap(o), self))
    def __xor__(self, o): return Expr(XOR, (self, wrap(o)))
    def __rxor__(self, o): return Expr(XOR, (wrap(o), self))
    def __invert__(self): return Expr(NOT, (self,))

    # serialisation
    def to_ir(self) -> bytes:
        out: List[bytes] = []
        _emit(self, out)
        return b"".join(out)

    def columns(self) -> List[int]:
        """Referenced column ordinals in first-appearance order (required_columns: broadcast.jl:33-35)."""
        seen: List[int] = []

        def walk(e: "Expr"):
            if e.op == COL:
                if e.payload not in seen:
                    seen.append(e.payload)
            for a in e.args:
                walk(a)
        walk(self)
        return seen

    def remap(self, mapping) -> "Expr":
        """Copy with every COL ordinal replaced through `mapping` (dict or callable)."""
        if self.op == COL:
            new = mapping(self.payload) if callable(mapping) else mapping[self.payload]
            return new if isinstance(new, Expr) else Expr(COL, (), new)
        return Expr(self.op, tuple(a.remap(mapping) for a in self.args), self.payload)

    def __repr__(self):
        if self.op == COL:
            return f"col({self.payload})"
        if self.op in (CONST, CONST_STR, CONST_SET):
            return repr(self.payload)
        return f"op{self.op:#04x}({', '.join(map(repr, self.args))})"


def _const_bytes(dt: int, value) -> bytes:
    b = dt & DTYPE_MASK
    if b == F64:
        return struct.pack("<d", float(value))
    if b == F32:
        return struct.pack("<f", float(value)) + b"\0\0\0\0"
    if b == BOOL:
        return struct.pack("<q", 1 if value else 0)
    if b in (U8, U16, U32, U64):
        return struct.pack("<Q", int(value) & 0xFFFFFFFFFFFFFFFF)
    return struct.pack("<q", int(value))


def _emit(e: Expr, out: List[bytes]):
    for a in e.args:
        _emit(a, out)
    if e.op == COL:
        out.append(struct.pack("<BI", COL, e.payload))
    elif e.op == CONST:
        dt, v = e.payload
        out.append(struct.pack("<BB", CONST, dt) + _const_bytes(dt, v))
    elif e.op == CONST_STR:
        out.append(struct.pack("<BI", CONST_STR, len(e.payload)) + e.payload)
    elif e.op == CONST_SET:
        dt, vals = e.payload
        out.append(struct.pack("<BBI", CONST_SET, dt, len(vals)) + b"".join(_const_bytes(dt, v) for v in vals))
    elif e.op == CAST:
        out.append(struct.pack("<BB", CAST, e.payload))
    else:
        out.append(struct.pack("<B", e.op))


def col(ordinal: int) -> Expr:
    return Expr(COL, (), int(ordinal))


def const(value, dtype: int | None = None) -> Expr:
    if isinstance(value, Expr):
        return value
    if isinstance(value, (bytes, str)):
        return Expr(CONST_STR, (), value.encode() if isinstance(value, str) else bytes(value))
    if dtype is None and julia_instant(value) is not None:      # Date / DateTime / Time constants compare as their Int64 instants
        value, dtype = julia_instant(value)[0], I64
    if dtype is None:
        if isinstance(value, (bool, np.bool_)):
            dtype = BOOL
        elif isinstance(value, np.generic):
            dtype = dtype_of_numpy(value.dtype)
        elif isinstance(value, int):
            dtype = I64 if -(1 << 63) <= value < (1 << 63) else U64
        elif isinstance(value, float):
            dtype = F64
        else:
            # arrays are rejected exactly like the reference (broadcast.jl:23-29)
            raise ValueError("Cannot do BlockBroadcasting with arrays")
    return Expr(CONST, (), (dtype, value))


# ---- Julia bits types carried as integers (Date / DateTime / Time / Char): include/dfdb.h dfdb_colinfo.logical -------------
RATA_DIE_DAYS = 719163               # Date(1970,1,1).instant: days since 0000-12-31
RATA_DIE_MS = RATA_DIE_DAYS * 86_400_000


def julia_instant(x):
    """numpy / datetime value -> (Int64 representation, logical type) as Julia's Dates stores it, or None."""
    import datetime as _dt
    if isinstance(x, np.datetime64):
        unit = np.datetime_data(x.dtype)[0]
        if unit in ("D", "W", "M", "Y"):
            return int(x.astype("datetime64[D]").astype(np.int64)) + RATA_DIE_DAYS, "Date"
        return int(x.astype("datetime64[ms]").astype(np.int64)) + RATA_DIE_MS, "DateTime"
    if isinstance(x, np.timedelta64):
        return int(x.astype("timedelta64[ns]").astype(np.int64)), "Time"
    if isinstance(x, _dt.datetime):
        return julia_instant(np.datetime64(x, "ms"))
    if isinstance(x, _dt.date):
        return julia_instant(np.datetime64(x, "D"))
    return None


def julia_char(ch: str) -> int:
    """reinterpret(UInt32, c::Char): the UTF-8 bytes left-aligned in a big-endian word."""
    b = ch.encode("utf8")
    if len(ch) != 1 or len(b) > 4:
        raise ValueError("a Char is one Unicode scalar")
    return int.from_bytes(b.ljust(4, b"\0"), "big")


def from_logical(arr: np.ndarray, logical: str):
    """the integer representation back to numpy datetime64 / timedelta64 / 1-character strings"""
    if logical == "Date":
        return (arr.astype(np.int64) - RATA_DIE_DAYS).astype("datetime64[D]")
    if logical == "DateTime":
        return (arr.astype(np.int64) - RATA_DIE_MS).astype("datetime64[ms]")
    if logical == "Time":
        return arr.astype(np.int64).astype("timedelta64[ns]")
    if logical == "Char":
        return np.array([int(x).to_bytes(4, "big").rstrip(b"\0").decode("utf8") for x in arr.tolist()], dtype=object)
    return arr


def wrap(x) -> Expr:
    if hasattr(x, "_as_expr"):
        return x._as_expr()
    if isinstance(x, (list, tuple, np.ndarray)):
        raise ValueError("Cannot do BlockBroadcasting with arrays")
    return const(x)


# ---- named functions used by the reference's tests/docs ---------------------
def isin(a, values: Iterable, dtype: int | None = None) -> Expr:
    """`in.(a, Ref(values))` (test/broadcast.jl:63-71)."""
    vals = list(values)
    if vals and all(isinstance(v, (str, bytes)) for v in vals):
        # a set of strings: `any(==(x), values)` spelled out — (a == v1) | (a == v2) | ... (three-valued like Base.in: missing unless some value
        # matches).  Over a dictionary-coded column the engine folds the whole chain into one bit-table lookup of the codes (K9).
        a = wrap(a)
        out = a == vals[0]
        for v in vals[1:]:
            out = out | (a == v)
        return out
    if dtype is None:
        dtype = F64 if any(isinstance(v, (float, np.floating)) for v in vals) else I64
    return Expr(IN_SET, (wrap(a), Expr(CONST_SET, (), (dtype, vals))))


def startswith(a, prefix) -> Expr:
    return Expr(STARTSWITH, (wrap(a), const(prefix)))


def endswith(a, suffix) -> Expr:
    return Expr(ENDSWITH, (wrap(a), const(suffix)))


def ismissing(a) -> Expr:
    return Expr(ISMISSING, (wrap(a),))


def sizeof(a) -> Expr:
    return Expr(SIZEOF, (wrap(a),))


def coalesce(a, b) -> Expr:
    """coalesce(a, b): a where it is not missing, else b (the way a Union{T,Missing} expression becomes a predicate)."""
    return Expr(COALESCE, (wrap(a), wrap(b)))


def rem(a, b) -> Expr: return Expr(REM, (wrap(a), wrap(b)))
def mod(a, b) -> Expr: return Expr(MOD, (wrap(a), wrap(b)))
def div(a, b) -> Expr: return Expr(IDIV, (wrap(a), wrap(b)))
def minimum(a, b) -> Expr: return Expr(MIN, (wrap(a), wrap(b)))
def maximum(a, b) -> Expr: return Expr(MAX, (wrap(a), wrap(b)))
def cast(a, dtype: int) -> Expr: return Expr(CAST, (wrap(a),), dtype)
def float64(a) -> Expr: return cast(a, F64)


def trace(fn, leaves: Sequence[Expr]) -> Expr:
    """Call a user function on symbolic leaves: `(:a,:b) => (a,b) -> …` (view.jl:64-68)."""
    r = fn(*leaves)
    return wrap(r)
