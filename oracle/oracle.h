/* oracle.h — CPU restatement of DataFrameDBs.jl's block-streamed scan path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (dataframedbs.jl_amd/) may
 * include, link, dlopen or call this library; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do, and only as the checker / reported baseline.
 *
 * The reference is pure Julia and `julia` is not installed in the build image, so
 * the reference cannot be executed here (no oracle/_ref).  This file restates its
 * algorithm in plain C, single-threaded, with the same pass structure
 * (LZ4 decode -> block body -> mask fill -> gather -> evaluate -> write-back ->
 * count -> projection gather -> append).  It is pinned against the known-answer
 * values of the reference's own tests (tests/golden/) and against
 * numpy in tests/.  File-format parity is pinned by spec + liblz4 cross-decode
 * only: no reference-written file is available (SURVEY.md §8c last row).
 *
 * Third-party arithmetic: the LZ4 block codec.  The reference reaches liblz4
 * through the Julia package CodecLz4 (Project.toml:8, compat >= 0.3.0, no
 * Manifest -> unpinned) at src/io/BlockStreams.jl:39,42-48,110-111.  The oracle
 * calls the same three liblz4 entry points of the system liblz4.so.1 (1.9.3).
 *
 * Citations are file:line under /root/reference.
 */
#ifndef DFDB_ORACLE_H
#define DFDB_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/dfdb_ir.h"

#ifdef __cplusplus
extern "C" {
#endif

/* status codes mirror include/dfdb.h (same exception mapping) */
enum {
  ORC_OK = 0, ORC_ERR_ARGUMENT = 1, ORC_ERR_IO = 2, ORC_ERR_FORMAT = 3, ORC_ERR_KEY = 4,
  ORC_ERR_BOUNDS = 5, ORC_ERR_DIVIDE = 6, ORC_ERR_UNSUPPORTED = 7, ORC_ERR_NOMEM = 9
};

#define ORC_DEFAULT_BLOCK_SIZE 65536 /* src/DataFrameDBs.jl:4 */
#define ORC_FORMAT_VERSION 1         /* src/DataFrameDBs.jl:5 */
#define ORC_COMPRESSION_LEVEL 2      /* src/io/BlockStreams.jl:3 (LZ4 acceleration) */

typedef struct orc_table orc_table; /* DFTable: in-memory images of meta.bin and <id>.bin */
typedef struct orc_view orc_view;   /* DFView: table + Projection + SelectionQueue */
typedef struct orc_selexec orc_selexec; /* SelectionExecutor (selection.jl:87-92) */

typedef struct orc_sizestats { int64_t rows, compressed, uncompressed; } orc_sizestats;

typedef struct orc_outcol {  /* one materialized column (make_materialization: projection.jl:32-40) */
  int32_t dtype;
  int64_t count;
  void*    data;    /* fixed width values, or int32 sizes for strings */
  uint8_t* bytes;   /* strings: concatenated bytes */
  int64_t  nbytes;
  uint8_t* missing; /* nullable: 1 = missing */
} orc_outcol;

const char* orc_last_error(void);

/* ---- synthetic data (not part of the reference) ---- */
uint64_t orc_splitmix64(uint64_t x);
void orc_gen_i64_mod1m(uint64_t seed, int64_t row_first, int64_t n, int64_t* out);
void orc_gen_f64_u2000(uint64_t seed, int64_t row_first, int64_t n, double* out);
/* sizes[n], bytes (cap >= 9*n); returns total bytes */
int64_t orc_gen_str_brands10(uint64_t seed, int64_t row_first, int64_t n, int32_t* sizes, uint8_t* bytes);

/* ---- tables: create_table / insert / open_table (creators.jl:7-44, columns.jl:130-181) ---- */
int orc_table_create(int64_t block_size, orc_table** out);
int orc_table_open(const char* path, orc_table** out);   /* meta + header validation: filesystem.jl:47-54 */
int orc_table_save(orc_table* t, const char* path);      /* writes meta.bin and <id>.bin */
void orc_table_free(orc_table* t);
int orc_table_ncols(orc_table* t);
int64_t orc_table_block_size(orc_table* t);
int orc_table_colinfo(orc_table* t, int i, int64_t* id, char* name, size_t cap, int32_t* dtype);
int orc_table_find(orc_table* t, const char* name);      /* ordinal or -1 */
/* append a whole column in blocks of block_size (write_column: columns.jl:30-53; one
 * prepare_block_write!/commit_block_write! per block: BlockStreams.jl:31-60).
 * strings: data = int32 sizes (-1 missing), bytes = arena. missing: n bytes or NULL */
int orc_table_add_column(orc_table* t, const char* name, int32_t dtype, int64_t nrows,
                         const void* data, const uint8_t* bytes, const uint8_t* missing);
/* the same for a Julia bits type stored as an integer: logical = "Date" | "DateTime" | "Time" (Int64) | "Char" (UInt32) */
int orc_table_add_column_as(orc_table* t, const char* name, int32_t dtype, const char* logical, int64_t nrows,
                            const void* data, const uint8_t* bytes, const uint8_t* missing);
int orc_table_col_logical(orc_table* t, int i, char* buf, size_t cap);
/* raw image of column i's file (header + blocks) */
const uint8_t* orc_table_image(orc_table* t, int i, size_t* nbytes);
/* table_stats-style pass (skip_block over every block: misc.jl:6-42) */
int orc_table_column_stats(orc_table* t, int i, orc_sizestats* out, int64_t* nblocks);

/* ---- block codec unit level (test/block_streams.jl) ---- */
/* encode one block (header + LZ4 body) of `rows` rows whose uncompressed body is `body` */
int orc_block_encode(const uint8_t* body, int64_t body_bytes, int32_t rows, uint8_t* out, size_t cap, size_t* written);
/* read_sizes (BlockStreams.jl:68-72) */
int orc_block_sizes(const uint8_t* p, size_t avail, int32_t* rows, int64_t* origin, int64_t* compressed);
/* read_block: decode body into out (cap >= origin) */
int orc_block_decode(const uint8_t* p, size_t avail, uint8_t* out, size_t cap, int32_t* rows, int64_t* origin, size_t* consumed);

/* ---- views ---- */
int orc_view_new(orc_table* t, orc_view** out);          /* DFView(table): view.jl:50 */
void orc_view_free(orc_view* v);
int orc_view_add_range(orc_view* v, int64_t start, int64_t step, int64_t stop); /* selection(v, a:s:b) */
int orc_view_add_integer(orc_view* v, int64_t i);
int orc_view_add_indices(orc_view* v, const int64_t* idx, int64_t n);
int orc_view_add_predicate(orc_view* v, const uint8_t* ir, size_t len);
int orc_view_nstages(orc_view* v);
/* stage introspection for the composition tests (test/selection.jl:16-33):
 * kind 0=range 1=integer 2=indices 3=predicate */
int orc_view_stage(orc_view* v, int i, int* kind, int64_t* start, int64_t* step, int64_t* stop, int64_t* n);
int orc_view_set_projection(orc_view* v, int n, const char* const* names, const uint8_t* const* irs, const size_t* lens);
int orc_view_ncols(orc_view* v);
int orc_view_coltype(orc_view* v, int i, int32_t* dtype);
/* required columns in reference order (view.jl:183-190); returns count */
int orc_view_required_columns(orc_view* v, int32_t* ordinals, int cap);

int orc_nrow(orc_view* v, int64_t* n);                   /* view.jl:192-206 via BlockRowsIterator */
/* materialize(v) incl. the count pre-pass (materialization.jl:27-40); outs[] malloc'ed, free with orc_outcols_free */
int orc_materialize(orc_view* v, orc_outcol* outs, int ncols);
/* materialize(::DFColumn)-style: no count pre-pass (materialization.jl:46-52) */
int orc_materialize_nocount(orc_view* v, orc_outcol* outs, int ncols);
void orc_outcols_free(orc_outcol* outs, int ncols);
/* 1-based table row numbers of the selected rows = block start + LogicalIndex positions of
 * apply() (selection.jl:161-167).  out may be NULL to only count. */
int orc_select_indices(orc_view* v, int64_t* out, int64_t cap, int64_t* n);
/* packed mask (bit i of word i/64 = row i selected) over all rows */
int orc_select_bitmap(orc_view* v, uint64_t* out, int64_t nwords);
/* sum over Base.iterate(::DFColumn) order (column.jl:102-126): strictly left to right */
int orc_sum_f64(orc_view* v, int col, double* out);
int orc_sum_i64(orc_view* v, int col, int64_t* out);

/* ---- selection executor on caller blocks (test/selection.jl:40-106) ---- */
int orc_selexec_new(orc_view* v, orc_selexec** out);
void orc_selexec_free(orc_selexec* e);
/* apply(exe, rows, block): cols[i] is the decoded block of table column ordinal i (NULL if unused);
 * writes rows mask bytes; returns count via *n */
int orc_selexec_apply(orc_selexec* e, int64_t rows, const void* const* cols, uint8_t* mask, int64_t* n);
int orc_selexec_is_finished(orc_selexec* e);
int orc_selexec_skip_if_can(orc_selexec* e, int64_t size);

/* ---- expression level (test/broadcast.jl): eval IR over in-memory columns on rows given by mask ---- */
int orc_expr_result_type(orc_table* t, const uint8_t* ir, size_t len, int32_t* dtype);
int orc_expr_required_columns(orc_table* t, const uint8_t* ir, size_t len, int32_t* ordinals, int cap);

/* ---- cpu_baseline leg: whole headline job on one thread ----
 * `x OP c` over column `col` -> row indices into out (cap rows); returns rows selected and seconds */
int orc_bench_scan(orc_view* v, int64_t* out, int64_t cap, int64_t* nsel, double* seconds);

#ifdef __cplusplus
}
#endif
#endif
