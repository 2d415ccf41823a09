"""ctypes binding of libdfdb_hip.so (include/dfdb.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is visible, loading or
creating a context raises.  The same symbols are what the Julia `ccall` shim binds.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_PKG), "libdfdb_hip.so")

# status codes -> the Python exception standing in for the Julia one (include/dfdb.h)
OK, ERR_ARGUMENT, ERR_IO, ERR_FORMAT, ERR_KEY, ERR_BOUNDS, ERR_DIVIDE, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NOMEM = range(10)
MEM_HOST, MEM_DEVICE = 0, 1
GEN_I64_MOD1M, GEN_F64_U2000, GEN_STR_BRANDS10, GEN_I64_IOTA, GEN_STR_BRANDS10_MISSING = 1, 2, 3, 4, 5
AGG_COUNT, AGG_SUM, AGG_MIN, AGG_MAX = 0, 1, 2, 3


class DfdbError(RuntimeError):
    """ErrorException (bad/missing file, header mismatch, decompression error, device failure)."""


_EXC = {ERR_ARGUMENT: ValueError, ERR_IO: DfdbError, ERR_FORMAT: DfdbError, ERR_KEY: KeyError, ERR_BOUNDS: IndexError,
        ERR_DIVIDE: ZeroDivisionError, ERR_UNSUPPORTED: NotImplementedError, ERR_DEVICE: DfdbError, ERR_NOMEM: MemoryError}


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("compute_units", C.c_int32), ("wavefront_size", C.c_int32),
                ("hbm_bytes", C.c_int64), ("peak_hbm_gbps", C.c_double)]


class ColInfo(C.Structure):
    _fields_ = [("id", C.c_int64), ("name", C.c_char * 128), ("dtype", C.c_int32), ("resident", C.c_int32), ("logical", C.c_char * 32)]


class SizeStats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("compressed", C.c_int64), ("uncompressed", C.c_int64)]


class OutCol(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bytes", C.c_void_p), ("missing", C.c_void_p), ("bytes_cap", C.c_int64),
                ("memkind", C.c_int32), ("dtype", C.c_int32), ("count", C.c_int64), ("nbytes", C.c_int64)]


# every exported symbol of include/dfdb.h: (restype is always int32 status)
SYMBOLS = [
    "dfdb_version", "dfdb_shutdown", "dfdb_jit_cache_dir", "dfdb_selftest", "dfdb_device_count", "dfdb_last_error",
    "dfdb_ctx_create", "dfdb_ctx_destroy", "dfdb_ctx_synchronize", "dfdb_ctx_device_info", "dfdb_ctx_timer_start",
    "dfdb_ctx_timer_stop", "dfdb_ctx_set_option", "dfdb_ctx_profile_enable", "dfdb_ctx_profile_get",
    "dfdb_table_open", "dfdb_table_new", "dfdb_table_close", "dfdb_table_ncols", "dfdb_table_nrows", "dfdb_table_block_size",
    "dfdb_table_colinfo", "dfdb_table_find_column", "dfdb_table_load", "dfdb_table_load_image", "dfdb_table_add_column",
    "dfdb_table_add_generated", "dfdb_table_decode_resident", "dfdb_table_decode_status", "dfdb_table_resident_bytes", "dfdb_table_compress_column", "dfdb_table_read_probe", "dfdb_table_build_dictionary", "dfdb_table_set_row_base", "dfdb_table_set_logical_type", "dfdb_table_column_stats", "dfdb_table_add_from_query", "dfdb_table_save", "dfdb_table_save_column", "dfdb_query_hint_materialize", "dfdb_query_hint_aggregate", "dfdb_query_unique", "dfdb_query_groupreduce", "dfdb_query_groupreduce_fetch", "dfdb_stream_open", "dfdb_stream_next", "dfdb_stream_stats", "dfdb_stream_read_stats", "dfdb_stream_close",
    "dfdb_query_new", "dfdb_query_free", "dfdb_query_add_range", "dfdb_query_add_indices", "dfdb_query_add_integer",
    "dfdb_query_add_predicate", "dfdb_query_nstages", "dfdb_query_set_projection", "dfdb_query_ncols", "dfdb_query_coltype",
    "dfdb_expr_result_type", "dfdb_query_set_stage_base", "dfdb_query_count_prefix",
    "dfdb_query_execute", "dfdb_query_reset", "dfdb_count", "dfdb_count_to", "dfdb_select_bitmap", "dfdb_select_indices", "dfdb_result_string_bytes",
    "dfdb_materialize", "dfdb_aggregate", "dfdb_query_prepare", "dfdb_table_unload", "dfdb_query_read_stats",
    # multi-GPU groups (block-range shards + RCCL)
    "dfdb_group_create", "dfdb_group_unique_id", "dfdb_group_create_rank", "dfdb_group_create_rank_callbacks", "dfdb_group_destroy", "dfdb_group_info", "dfdb_group_ctx",
    "dfdb_group_synchronize", "dfdb_group_barrier", "dfdb_group_set_option", "dfdb_group_allreduce_f64",
    "dfdb_group_table_open", "dfdb_group_table_new", "dfdb_group_table_close", "dfdb_group_table_unload", "dfdb_group_table_load", "dfdb_group_table_add_generated",
    "dfdb_group_table_add_column", "dfdb_group_table_nrows", "dfdb_group_table_shard",
    "dfdb_group_query_new", "dfdb_group_query_free", "dfdb_group_query_prepare", "dfdb_group_query_add_range", "dfdb_group_query_add_indices", "dfdb_group_query_add_integer",
    "dfdb_group_query_add_predicate", "dfdb_group_query_set_projection", "dfdb_group_query_hint_aggregate", "dfdb_group_query_hint_materialize",
    "dfdb_group_query_reset", "dfdb_group_query_shard", "dfdb_group_count", "dfdb_group_shard_counts", "dfdb_group_aggregate",
    "dfdb_group_select_indices_device", "dfdb_group_select_indices", "dfdb_group_result_string_bytes", "dfdb_group_materialize",
    "dfdb_group_shard_string_bytes", "dfdb_group_materialize_device",
    "dfdb_group_query_unique", "dfdb_group_query_unique_fetch", "dfdb_group_query_groupreduce", "dfdb_group_query_groupreduce_fetch",
]
EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_HOST, EXCHANGE_CALLBACK = 0, 1, 2, 3
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32)      # (user, vals, n, dtype, op)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)               # (user, send, recv, nbytes)


class ExchangeFns(C.Structure):      # dfdb_exchange_fns
    _fields_ = [("user", C.c_void_p), ("allreduce", ALLREDUCE_FN), ("allgather", ALLGATHER_FN)]
GROUP_ID_BYTES = 128

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree HIP library; loud failure if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C dataframedbs.jl_amd/csrc` "
                              "(__graft_entry__.build()).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            getattr(lib, s).restype = C.c_int32
        import atexit
        atexit.register(lib.dfdb_shutdown)          # before the C runtime's own exit processing: the background compiler must not be inside LLVM by then (dfdb.h)
        lib.dfdb_table_add_generated.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_uint64, C.c_int64, C.c_int64]
        lib.dfdb_table_add_column.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dfdb_table_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]
        lib.dfdb_table_load_image.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_void_p]
        lib.dfdb_table_set_row_base.argtypes = [C.c_void_p, C.c_int64]
        lib.dfdb_table_decode_resident.argtypes = [C.c_void_p, C.c_int32]
        lib.dfdb_table_decode_status.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
        lib.dfdb_jit_cache_dir.argtypes = [C.c_char_p, C.c_size_t]
        lib.dfdb_selftest.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int64), C.c_int32]
        lib.dfdb_table_compress_column.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SizeStats)]
        lib.dfdb_table_resident_bytes.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.dfdb_table_read_probe.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.dfdb_table_build_dictionary.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]
        lib.dfdb_table_set_logical_type.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
        lib.dfdb_table_column_stats.argtypes = [C.c_void_p, C.c_int32, C.POINTER(SizeStats)]
        lib.dfdb_table_add_from_query.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32]
        lib.dfdb_table_save.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(SizeStats)]
        lib.dfdb_table_save_column.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.POINTER(SizeStats)]
        lib.dfdb_query_hint_materialize.argtypes = [C.c_void_p, C.c_int32]
        lib.dfdb_query_hint_aggregate.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.dfdb_query_unique.argtypes = [C.c_void_p, C.c_int32]
        lib.dfdb_query_groupreduce.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        lib.dfdb_query_groupreduce_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dfdb_query_prepare.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        lib.dfdb_table_unload.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        lib.dfdb_query_read_stats.argtypes = [C.c_void_p, C.POINTER(SizeStats)]
        lib.dfdb_stream_open.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        lib.dfdb_stream_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.dfdb_stream_stats.argtypes = [C.c_void_p, C.POINTER(SizeStats)]
        lib.dfdb_stream_read_stats.argtypes = [C.c_void_p, C.c_int32, C.POINTER(SizeStats)]
        lib.dfdb_stream_close.argtypes = [C.c_void_p]
        lib.dfdb_query_add_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
        lib.dfdb_query_add_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        lib.dfdb_query_add_integer.argtypes = [C.c_void_p, C.c_int64]
        lib.dfdb_query_add_predicate.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.dfdb_expr_result_type.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p]
        lib.dfdb_query_set_stage_base.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        lib.dfdb_select_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        lib.dfdb_count_to.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        lib.dfdb_select_bitmap.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        lib.dfdb_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.dfdb_ctx_create.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
        lib.dfdb_group_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        lib.dfdb_group_unique_id.argtypes = [C.c_void_p]
        lib.dfdb_group_create_rank.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        lib.dfdb_group_create_rank_callbacks.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        lib.dfdb_group_info.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        lib.dfdb_group_ctx.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.dfdb_group_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.dfdb_group_allreduce_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        lib.dfdb_group_table_open.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        lib.dfdb_group_table_new.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.dfdb_group_table_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        lib.dfdb_group_table_add_generated.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_uint64, C.c_int64]
        lib.dfdb_group_table_add_column.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dfdb_group_table_nrows.argtypes = [C.c_void_p, C.c_void_p]
        lib.dfdb_group_table_shard.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.dfdb_group_query_new.argtypes = [C.c_void_p, C.c_void_p]
        lib.dfdb_group_table_unload.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        lib.dfdb_group_query_prepare.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        lib.dfdb_group_query_add_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
        lib.dfdb_group_query_add_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        lib.dfdb_group_query_add_integer.argtypes = [C.c_void_p, C.c_int64]
        lib.dfdb_group_query_add_predicate.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.dfdb_group_query_set_projection.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dfdb_group_query_hint_aggregate.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.dfdb_group_query_hint_materialize.argtypes = [C.c_void_p, C.c_int32]
        lib.dfdb_group_query_shard.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.dfdb_group_count.argtypes = [C.c_void_p, C.c_void_p]
        lib.dfdb_group_shard_counts.argtypes = [C.c_void_p, C.c_void_p]
        lib.dfdb_group_aggregate.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        lib.dfdb_group_select_indices_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dfdb_group_select_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dfdb_group_result_string_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.dfdb_group_materialize.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        lib.dfdb_group_shard_string_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.dfdb_group_materialize_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        lib.dfdb_group_query_unique.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        lib.dfdb_group_query_unique_fetch.argtypes = [C.c_void_p, C.c_void_p]
        lib.dfdb_group_query_groupreduce.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        lib.dfdb_group_query_groupreduce_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for f in ("dfdb_group_destroy", "dfdb_group_synchronize", "dfdb_group_barrier", "dfdb_group_table_close", "dfdb_group_query_free", "dfdb_group_query_reset"):
            getattr(lib, f).argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    load().dfdb_last_error(buf, C.c_size_t(1024))
    return buf.value.decode(errors="replace")


def check(rc: int):
    if rc != OK:
        raise _EXC.get(rc, DfdbError)(last_error())
