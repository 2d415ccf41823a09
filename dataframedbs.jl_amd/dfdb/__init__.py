"""dfdb — MI355X-native drop-in for DataFrameDBs.jl's scan/filter hot path (Python host mirror).

The data path is libdfdb_hip.so (hand-written gfx950 HIP kernels behind the C ABI of include/dfdb.h);
this package only mirrors the reference's lazy DFTable / DFView / DFColumn algebra and lowers it to the
expression IR.  Importing the package does not need a GPU; creating a Context does.
"""
from . import ir
from ._native import (AGG_COUNT, AGG_MAX, AGG_MIN, AGG_SUM, GEN_F64_U2000, GEN_I64_IOTA, GEN_I64_MOD1M, GEN_STR_BRANDS10, GEN_STR_BRANDS10_MISSING,
                      LIB_PATH, MEM_DEVICE, MEM_HOST, SYMBOLS, DfdbError, load)
from .api import (ALL, END, groupreduce, ColumnMeta, Context, DFColumn, DFTable, DFView, JRange, Projection, SelectionQueue, coalesce, col_equal,
                  create_table, default_context, endswith, float64, head, isin, ismissing, issameselection, jr, map_to_column, materialize,
                  materialize_streamed, ncol, nrow, nrow_streamed, open_table, projection, selection, selproj, set_string_output, size, sizeof, startswith, stream, table_stats, view_from_columns)
from .ir import div, maximum, minimum, mod, rem

__all__ = [n for n in dir() if not n.startswith("_")]
