/* dfdb.h — C ABI of libdfdb_hip.so, the MI355X (gfx950) scan/filter engine that
 * replaces DataFrameDBs.jl's block-streamed decode + selection + projection +
 * materialize hot path.
 *
 * The reference has no FFI seam of its own (it is pure Julia; its only ccalls are
 * the three liblz4 entry points in src/io/BlockStreams.jl:39,42,110).  Each entry
 * point below therefore names the Julia method it stands in for (paths relative
 * to /root/reference).  A Julia `ccall` shim and the Python ctypes mirror bind
 * exactly these symbols; see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int32 status (0 = DFDB_OK); no exception crosses the ABI
 *   - dfdb_last_error() gives the thread-local message of the last failure
 *   - handles are opaque; one host thread per handle at a time (the reference's
 *     iterator/executors are not thread-safe either: selection.jl:68-75)
 *   - row numbers are 1-based like Julia; column ordinals are 0-based positions
 *   - the engine is HIP-only: there is no CPU fallback behind any entry point
 */
#ifndef DFDB_H
#define DFDB_H
#include <stddef.h>
#include <stdint.h>
#include "dfdb_ir.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DFDB_ABI_VERSION 1

/* status codes; the Julia exception each maps to */
enum {
  DFDB_OK = 0,
  DFDB_ERR_ARGUMENT = 1,   /* ArgumentError  (selection.jl:54, projection.jl:27, columnbroadcast.jl:22) */
  DFDB_ERR_IO = 2,         /* ErrorException (filesystem.jl:16,31,50-57,70) */
  DFDB_ERR_FORMAT = 3,     /* header mismatch / "decompression error" (BlockStreams.jl:112) */
  DFDB_ERR_KEY = 4,        /* KeyError       (table.jl:54) */
  DFDB_ERR_BOUNDS = 5,     /* BoundsError    (view.jl:134, column.jl:97, selection.jl:40 range[range]) */
  DFDB_ERR_DIVIDE = 6,     /* DivideError raised by `%` / `÷` inside a predicate */
  DFDB_ERR_UNSUPPORTED = 7,/* outside the IR op set: caller falls back to the Julia path */
  DFDB_ERR_DEVICE = 8,     /* HIP runtime failure (message carries hipGetErrorString) */
  DFDB_ERR_NOMEM = 9
};

/* where a caller-provided output buffer lives */
enum { DFDB_MEM_HOST = 0, DFDB_MEM_DEVICE = 1 };

/* synthetic generators (SURVEY.md §8d "Data generation"): value of 0-based row i is a
 * function of h = splitmix64(seed + i) */
enum {
  DFDB_GEN_I64_MOD1M = 1,   /* Int64   : h mod 1 000 000 */
  DFDB_GEN_F64_U2000 = 2,   /* Float64 : (h >> 11) * 2^-53 * 2000.0 */
  DFDB_GEN_STR_BRANDS10 = 3,/* String  : brands10[h mod 10] */
  DFDB_GEN_I64_IOTA = 4,    /* Int64   : i + 1  (the reference tests' 1:N columns) */
  DFDB_GEN_STR_BRANDS10_MISSING = 5 /* Union{String,Missing} : missing when (h >> 32) mod 8 == 7, else brands10[h mod 10] (docs/src/index.md:264-272: the
                               docs' real data set is all Union{Missing,String}) */
};

/* aggregates over a filtered view (Base.iterate(::DFColumn) column.jl:102-126) */
enum { DFDB_AGG_COUNT = 0, DFDB_AGG_SUM = 1, DFDB_AGG_MIN = 2, DFDB_AGG_MAX = 3 };

typedef struct dfdb_ctx dfdb_ctx;     /* device + stream + workspace */
typedef struct dfdb_table dfdb_table; /* DFTable whose columns are decoded and resident in HBM */
typedef struct dfdb_query dfdb_query; /* DFView: projection + SelectionQueue over one table */
typedef struct dfdb_stream dfdb_stream; /* block-streamed execution of a query over a non-resident table */

typedef struct dfdb_device_info {
  char name[128];
  int32_t compute_units;
  int32_t wavefront_size;
  int64_t hbm_bytes;
  double  peak_hbm_gbps;   /* HBM3E spec peak of the part: 8000 on MI355X / gfx950 (what bench.py prices its roofline against) */
} dfdb_device_info;

typedef struct dfdb_colinfo {   /* ColumnMeta(id,name,type): src/tables/meta.jl:2-10 */
  int64_t id;
  char    name[128];
  int32_t dtype;      /* DFDB_* | DFDB_NULLABLE */
  int32_t resident;   /* 1 once decoded blocks are in HBM */
  char    logical[32]; /* "Date" / "DateTime" / "Time" (dtype Int64: days / ms / ns) or "Char" (dtype UInt32): a Julia bits type the
                        * hot path carries as its integer representation (read_block_body! is a memcpy: blocks.jl:37-44); "" otherwise */
} dfdb_colinfo;

typedef struct dfdb_sizestats { /* SizeStats: src/io/sizestats.jl:2-10 (compressed adds 24 B/block, quirk Q10) */
  int64_t rows, compressed, uncompressed;
} dfdb_sizestats;

/* one output column of materialize(); the caller allocates after dfdb_count()/dfdb_result_layout() */
typedef struct dfdb_outcol {
  void*    data;      /* fixed width: count*width bytes ; String: count * int32 sizes (-1 = missing) */
  uint8_t* bytes;     /* String only: concatenated UTF-8 of the selected rows */
  uint8_t* missing;   /* nullable only: count bytes, 1 = missing (may be NULL for non-nullable) */
  int64_t  bytes_cap; /* capacity of `bytes` */
  int32_t  memkind;   /* DFDB_MEM_HOST | DFDB_MEM_DEVICE */
  int32_t  dtype;     /* filled by the engine */
  int64_t  count;     /* filled: rows written */
  int64_t  nbytes;    /* filled: string bytes written */
} dfdb_outcol;

/* ------------------------------------------------------------------ misc */
int32_t dfdb_version(void);
/* Call from the host language's exit hook (Python atexit, Julia atexit: the shipped bindings do) before the process ends: stops the background compiler
 * of run-time expression kernels (a compile in flight is waited for: 0.1-0.3 s as a rule, two minutes at most), after which new expressions are interpreted.  Idempotent; handles stay valid.
 * Without it a process that exits while hipRTC is compiling can crash inside LLVM's static destructors (there is no Julia method this replaces). */
int32_t dfdb_shutdown(void);
/* Where the run-time compiler keeps its code objects between processes ($DFDB_JIT_CACHE_DIR, else $XDG_CACHE_HOME/dfdb-jit, else ~/.cache/dfdb-jit;
 * DFDB_JIT_CACHE=0 turns it off), or "" when there is none.  The directory is TRUSTED INPUT — a code object read from it runs in this process's GPU
 * context — so it is used only if it is a real directory (not a symbolic link) owned by the effective user and not writable by group or others; files in
 * it are opened with O_NOFOLLOW and must be regular files of the same owner.  A directory that fails a check is never read or written: buf receives ""
 * and dfdb_last_error says which check failed (the status is still 0: kernels are then compiled per process, as without a cache). */
int32_t dfdb_jit_cache_dir(char* buf, size_t cap);
/* Host-side self-tests of library internals that need neither a GPU nor RCCL (there is no Julia method this replaces; tests/test_host_cpu.py calls it).
 *   "rccl_bracket": the ncclGroupStart / ncclGroupEnd bracket of every RCCL exchange (csrc/group.cpp) driven by a stub collective table: three all-reduces of
 *   which number `arg` fails (0 = none, -1 = ncclGroupStart itself, -2 = ncclGroupEnd), then a second exchange on the same communicator state.
 *   out[0..5] = GroupStart calls, GroupEnd calls, collectives issued, status of the first exchange, status of the second, communicator marked dead.
 * An unknown name is ArgumentError. */
int32_t dfdb_selftest(const char* name, int64_t arg, int64_t* out, int32_t nout);
int32_t dfdb_device_count(int32_t* n);   /* visible HIP devices (0 without a GPU: no error); the Julia shim forms a group when n > 1 */
int32_t dfdb_last_error(char* buf, size_t cap);

/* ------------------------------------------------------------------ context */
/* hip_stream: a hipStream_t to launch on (e.g. torch's current stream) or NULL for a private one */
int32_t dfdb_ctx_create(int32_t device_id, void* hip_stream, dfdb_ctx** out);
int32_t dfdb_ctx_destroy(dfdb_ctx* ctx);
int32_t dfdb_ctx_synchronize(dfdb_ctx* ctx);
int32_t dfdb_ctx_device_info(dfdb_ctx* ctx, dfdb_device_info* out);
/* Context options (free-form keys; an unknown key is stored and ignored).  These TWENTY are the supported surface; defaults are what the benchmarks run with.
 * A/B switches of measured alternatives and switches that force a code path in tests are not part of it: they are listed in
 * dataframedbs.jl_amd/csrc/KNOBS.md, results never depend on them, and they may go without notice (round 6 removed ten).
 *  residency
 *   "keep_compressed"  0 (default) = a loaded column is its decoded array.  1 = plain fixed-width columns also keep their LZ4 blocks in HBM beside it
 *                      (dfdb_table_decode_resident re-decodes from them).  2 = COMPRESSED-ONLY: the LZ4 blocks + their sequence-start index and NO decoded array —
 *                      like the reference, whose iterator decodes every block into two reusable buffers and keeps nothing (BlockStreams.jl:9-15,101-119; "memory
 *                      use is O(block)", docs/src/index.md:182,192).  `col OP const` conjuncts (and intervals) over an 8-byte column are evaluated by the decoder
 *                      itself (bitmap and tile counts are the only output; from the second mask on, blocks without a survivor are not decoded); a projection
 *                      gathers from the blocks that KEPT A ROW, decoded into an arena the query owns (blocksiterator.jl:111-113); every other consumer gets a
 *                      whole-column decode that lives for the one ABI call.  Nullable and String columns load as with 0.
 *   "lz4_index"        1 (default) = the first decode of a column's resident blocks records where its LZ4 sequences start (one bit per compressed byte) and every
 *                      later decode reads that instead of parsing again: 450-490 -> 590-630 GB/s decoded on 8-byte integer columns; same bytes out
 *   "decode_on_scan"   1 = a fresh-mask scan of one `col OP const` term over an 8-byte column that holds its blocks (keep_compressed = 1) decodes and filters in
 *                      ONE pass instead of trusting the decoded array (default 0)
 *   "string_dictionary" N > 0 = a String column that becomes resident gets a dictionary when it has at most N distinct values (dfdb_table_build_dictionary; default 0)
 *   "hbm_budget_mb"    what dfdb_query_prepare / dfdb_group_query_prepare let a TABLE hold in HBM (per shard for a group), in MB; 0 (default) = no bound of its own:
 *                      80 % of the HBM that is free at the call (a group: of the device's HBM) decides alone
 *  out of core (see "out of core behind the ordinary entry points" and "block-streamed execution" below)
 *   "ooc_chunk_blocks" blocks per chunk of the streams the query entry points run internally over columns that are not resident (default 512)
 *   "stream_slots"     chunks a stream holds in HBM at once, 2 .. 8 (default 8): one is the caller's, the others are being read, copied and decoded
 *   "stream_readers"   loaders of a stream that may READ (page cache -> pinned memory) at the same time, the others wait for their copies and decode (default 3)
 *   "stream_piece_mb"  a loader reads and copies a chunk's bytes this many MB at a time through a three-piece pinned ring, 1 .. 512 (default 64)
 *   "stream_late_materialize"  1 (default) = a streamed chunk loads its projection-only columns only for the blocks whose selection kept a row
 *                      (blocksiterator.jl:111-113); 0 = every required column of every chunk whole
 *   "stream_cache"     1 (default) = dfdb_stream_close parks the stream (slot contexts, pinned buffers, loader threads) on its context for the next open (~40 ms saved)
 *  host I/O
 *   "io_threads"       concurrent preads a byte range of a column file is split into, 1 .. 64 (default 8; process-wide, read when a stream is opened)
 *   "numa_bind"        1 (default) = the host threads that move file bytes run on the CPUs of the NUMA node the GPU hangs off, pinned buffers come from there
 *   "load_progressive" 1 (default) = dfdb_table_load decodes a plain fixed-width column batch by batch while the rest of its file is still being read
 *   "save_fsync"       1 = dfdb_table_save / _save_column fdatasync every file before closing it, column files before meta.bin (default 0, like the reference)
 *  kernels
 *   "placement_calibrate" 1 = the first fresh-mask scan of a column of >= 2^26 rows times itself on a few fresh allocations of the column and of its bitmap and keeps
 *                      the fastest (~60 scans and 0.03-1.4 s once per column buy ~3.5 % of K1 and halve its spread; default 0; `bench.py --placement`)
 *   "scan_capture"     how many projected 8-byte predicate columns the scan that produces a query's final mask keeps for dfdb_materialize
 *                      (dfdb_query_hint_materialize): 2 (default), 1, or 0 = gather everything
 *   "lz4_enc_near"     > 0: the writer's compressor gives up a match whose source lies further back than this many bytes when a nearer one ends as late — K7 then copies
 *                      out of its on-chip history instead of fetching a line (at 1984, K7's reach: ratio - 8 %, indexed decode + 3 %; default 0: file bytes are what
 *                      the cold path waits for)
 *   "unique_radix"     1 (default) = dfdb_query_unique over a fixed-width key whose hash table would outgrow the L2s (an estimated 131 072 .. ~5 M distinct values among
 *                      at least 4 M selected rows) takes the radix-partitioned form (k_radix.hip: every selected row written once as a {key, row} record into one of
 *                      256 .. 1024 partitions, each reduced through a table in LDS; 12 bytes of scratch per selected row (20 for groupreduce) + 0.4-0.8 GB of part-filled pages, kept by the context between calls
 *                      unless it is more than a quarter of the device's memory; no room for it, or a
 *                      partition that outgrows its table: the hash table answers; a value that a large part of the column holds is kept out of the records by the
 *                      partition pass itself — LDS slots for hot keys).  8.8 ms against the hash table's 19.8 per 1e9 rows of 1e6 values
 *                      (profiles/r6_unique_radix.txt).  dfdb_query_groupreduce over more groups than a workgroup's LDS accumulators hold (9216; a fixed-width key,
 *                      nullable or not) goes the same way: 14.5 ms where global atomics took 89 per 1e9 rows in 50 000 groups (19.6 ms against 3.6 s when one key holds 30 % of the rows)
 *                      (profiles/r6_groupreduce_radix.txt).  0 = always the hash table / the global atomics
 *   "jit"              1 (default) = an expression the device INTERPRETER evaluates (outside `col OP const` terms, pairs and string matches: configs 2-5 never get here)
 *                      over at least 4 M rows is also compiled by hipRTC in the background — the interpreter's own source specialised for the program's shape — and later
 *                      executions run the compiled kernel; until it is ready, or when libhiprtc is absent, the interpreter answers.  0 = interpret always; 2 = wait for
 *                      the compiler (tests, benchmarks).  Results are identical by construction.  dfdb_shutdown stops the compiler; dfdb_jit_cache_dir is its disk cache */
int32_t dfdb_ctx_set_option(dfdb_ctx* ctx, const char* key, int64_t value);
/* HIP-event timing on the engine's own stream (bench.py's roofline leg) */
int32_t dfdb_ctx_timer_start(dfdb_ctx* ctx);
int32_t dfdb_ctx_timer_stop(dfdb_ctx* ctx, double* elapsed_ms);
/* per-kernel-family accumulated device time since the last reset (HIP events around each launch
 * when profiling is on): name -> (launches, total_ms) */
int32_t dfdb_ctx_profile_enable(dfdb_ctx* ctx, int32_t on);
int32_t dfdb_ctx_profile_get(dfdb_ctx* ctx, const char* kernel, int64_t* launches, double* total_ms);

/* ------------------------------------------------------------------ tables */
/* open_table(path): creators.jl:7-16 -> read_table_meta table_io.jl:21-33 + check_column_head filesystem.jl:47-54 */
int32_t dfdb_table_open(dfdb_ctx* ctx, const char* path, dfdb_table** out);
/* in-memory table for synthetic / caller-supplied columns (no files) */
int32_t dfdb_table_new(dfdb_ctx* ctx, int64_t block_size, dfdb_table** out);
int32_t dfdb_table_close(dfdb_table* t);
int32_t dfdb_table_ncols(dfdb_table* t, int32_t* n);
int32_t dfdb_table_nrows(dfdb_table* t, int64_t* n);      /* rows resident on this device */
int32_t dfdb_table_block_size(dfdb_table* t, int64_t* bs); /* blocksize(t): table.jl:47 */
int32_t dfdb_table_colinfo(dfdb_table* t, int32_t ordinal, dfdb_colinfo* out); /* getmeta: table.jl:52-56 */
int32_t dfdb_table_find_column(dfdb_table* t, const char* name, int32_t* ordinal); /* KeyError if absent */

/* BlockStream.read_block + read_block_body! for every block in [block_first, block_last) of the
 * listed columns (BlockStreams.jl:101-119, blocks.jl:37-71): file -> compressed bytes -> HBM ->
 * device LZ4 decode -> decoded column resident.  block_last < 0 means "to EOF".  This is the
 * block-range shard of SURVEY.md §8e: rank g loads only its blocks. */
int32_t dfdb_table_load(dfdb_table* t, const int32_t* ordinals, int32_t ncols,
                        int64_t block_first, int64_t block_last, dfdb_sizestats* stats);
/* same from a caller-held image of one column file (header + blocks), e.g. an mmap */
int32_t dfdb_table_load_image(dfdb_table* t, int32_t ordinal, const uint8_t* image, size_t nbytes,
                              int64_t block_first, int64_t block_last, dfdb_sizestats* stats);

/* caller-supplied decoded column (host memory).  Fixed width: data = nrows*width bytes.
 * String: data = nrows int32 sizes (-1 = missing), bytes = arena (FlatStringsVector: FlatStringsVectors.jl:5-52).
 * missing = nrows bytes (1 = missing) or NULL. */
int32_t dfdb_table_add_column(dfdb_table* t, const char* name, int32_t dtype, int64_t nrows,
                              const void* data, const uint8_t* bytes, int64_t nbytes,
                              const uint8_t* missing);
/* device-side synthetic fill (no PCIe): rows [row_first, row_first+nrows) of the global column */
int32_t dfdb_table_add_generated(dfdb_table* t, const char* name, int32_t generator,
                                 uint64_t seed, int64_t row_first, int64_t nrows);
/* add_column!(table, name, lazy_col) (src/tables/table.jl:96-124, src/tables/columns.jl:65-84): column `proj_col` of the
 * view behind `q` is materialised into a new RESIDENT column of `dst` without leaving the device.  dst may be the view's
 * own table; the column must have exactly as many rows as dst (ArgumentError otherwise). */
int32_t dfdb_table_add_from_query(dfdb_table* dst, const char* name, dfdb_query* q, int32_t proj_col);

/* create_table(path; from=...) / write_column (src/tables/creators.jl:18-60, src/tables/columns.jl:39-53): every resident
 * column -> `<path>/<id>.bin` in the reference's block format (write_column_head filesystem.jl:14-23, write_block_body
 * blocks.jl:2-33, commit_block_write! BlockStreams.jl:36-60) + `<path>/meta.bin` (write_table_meta table_io.jl:9-19).
 * Block bodies are packed and LZ4-compressed on the device; only compressed bytes cross PCIe.  The files open with the
 * reference's open_table and with dfdb_table_open.  ErrorException (DFDB_ERR_IO) if `path` exists at all (table_exists = isdir:
 * filesystem.jl:31,38) or a column file does (make_column_file :16); a full disk is DFDB_ERR_IO too (space is reserved before it is written). */
int32_t dfdb_table_save(dfdb_table* t, const char* path, dfdb_sizestats* stats);
/* one column file (header + blocks) */
int32_t dfdb_table_save_column(dfdb_table* t, int32_t ordinal, const char* file, dfdb_sizestats* stats);

/* table_stats(table) (src/tables/misc.jl:6-43): SizeStats of one column file from its block headers alone (skip_block,
 * BlockStreams.jl:74-78); nothing is read or decoded beyond the 20-byte headers */
int32_t dfdb_table_column_stats(dfdb_table* t, int32_t ordinal, dfdb_sizestats* stats);

/* compressed-resident columns.  With ctx option "keep_compressed" = 1 at dfdb_table_load time a plain fixed-width column keeps its
 * LZ4 blocks (as they sit in the file) in HBM beside the decoded array; dfdb_table_decode_resident runs the block decoder (K7,
 * read_block: BlockStreams.jl:101-119) over all of them again, asynchronously, into the resident array.  It is the device-side
 * equivalent of re-reading the column through BlockStream and what bench.py's decode-inclusive figure times. */
int32_t dfdb_table_decode_resident(dfdb_table* t, int32_t ordinal);
/* What the last decode of the column's resident blocks said (dfdb_table_decode_resident, or a decode_on_scan execution): waits for the context's stream and
 * counts the blocks whose decode did not end with exactly the stored size — the reference's `@assert size == sizes.origin "decompression error"`
 * (BlockStreams.jl:112).  *bad_blocks receives the count; with bad_blocks == NULL a count > 0 is DFDB_ERR_FORMAT ("decompression error").  The blocks
 * were validated when the column was loaded, so anything but 0 means the resident copy (or the sequence-start index beside it) was damaged; the index is
 * dropped then, and the next decode parses the blocks for itself and records a new one. */
int32_t dfdb_table_decode_status(dfdb_table* t, int32_t ordinal, int64_t* bad_blocks);
/* Measurement aid (there is no Julia method this replaces): K1's READ STREAM alone over a resident 8-byte column — the predicate scan's load shape (four
 * 1024-row tiles per wave trip, sixteen nontemporal 512-byte wave loads in flight, the scan's grid) with no ballot, no bitmap and no tile count written.
 * `repeats` (1 .. 64) launches back to back on the context's stream, each bracketed by HIP events: *best_ms / *avg_ms (either may be NULL).  bench.py prints
 * rows * 8 / best as `roofline.box_read_ceiling_GBps`: what this box and this allocation give the scan before it writes anything. */
int32_t dfdb_table_read_probe(dfdb_table* t, int32_t ordinal, int32_t repeats, double* best_ms, double* avg_ms);
/* A resident plain fixed-width column -> its compressed forms WITHOUT a file in between: every block of block_size rows is LZ4-encoded on the device
 * (write_block_body + commit_block_write!, blocks.jl:2-7, BlockStreams.jl:36-60: the very bytes dfdb_table_save would write) and kept in HBM.
 * mode 1 = beside the decoded array (what ctx option "keep_compressed" = 1 leaves after a load), mode 2 = COMPRESSED-ONLY: the decoded array is released and
 * the column answers queries as described under "keep_compressed" = 2.  *stats (may be NULL): rows, body bytes, compressed bytes incl. the 20-byte headers.
 * Nullable and String columns: DFDB_ERR_UNSUPPORTED.  A column that is compressed-only already is left as it is. */
int32_t dfdb_table_compress_column(dfdb_table* t, int32_t ordinal, int32_t mode, dfdb_sizestats* stats);
/* HBM bytes a column holds right now (ordinal < 0: the whole table): *decoded = its decoded arrays (values, string sizes / bytes / tile offsets, missing
 * bitmap, dictionary codes), *compressed = its LZ4 blocks, their descriptors and statuses and the sequence-start index (ctx option "keep_compressed").
 * A compressed-only column (keep_compressed = 2) reports decoded = 0.  Either pointer may be NULL.  (There is no Julia method this replaces.) */
int32_t dfdb_table_resident_bytes(dfdb_table* t, int32_t ordinal, int64_t* decoded, int64_t* compressed);
/* Dictionary form of a resident, non-nullable String column with at most max_entries (<= 65535) distinct values: one 16-bit code per row and the
 * distinct strings once, kept BESIDE the FlatStringsVector form (the reference has no such form: docs/src/index.md lists dictionary encoding under
 * "Future plans").  From then on `col == / != / startswith / endswith "const"` is decided once per distinct string and becomes a bit-table lookup of
 * the codes (2 B per row instead of 4 + L), and materialize copies the selected rows' strings out of the dictionary; results are unchanged.
 * *entries = the dictionary's size, 0 when none was built (too many distinct values, a nullable column, a string over 4 KB).  ctx option
 * "string_dictionary" = N > 0 builds one automatically, with max_entries = N, whenever a String column becomes resident. */
int32_t dfdb_table_build_dictionary(dfdb_table* t, int32_t ordinal, int64_t max_entries, int64_t* entries);

/* declare a caller-supplied Int64 / UInt32 column to be one of the bits types above, so that dfdb_table_save writes that type
 * string and the reference's open_table reads the column back as Date / DateTime / Time / Char */
int32_t dfdb_table_set_logical_type(dfdb_table* t, int32_t ordinal, const char* logical);

/* global row number (0-based) of this shard's first row, so that selection indices and leading
 * range stages refer to table rows when a table is block-range sharded over ranks */
int32_t dfdb_table_set_row_base(dfdb_table* t, int64_t row_base);

/* ------------------------------------------------------------------ queries (DFView) */
/* DFView(table): full projection, empty SelectionQueue (view.jl:50) */
int32_t dfdb_query_new(dfdb_table* t, dfdb_query** out);
int32_t dfdb_query_free(dfdb_query* q);
/* selection(v, range): stages are appended with the composition rules of selection.jl:37-60
 * (range∘range collapses, predicate∘predicate fuses with &). */
int32_t dfdb_query_add_range(dfdb_query* q, int64_t start, int64_t step, int64_t stop);
int32_t dfdb_query_add_indices(dfdb_query* q, const int64_t* idx, int64_t n);
/* selection(v, i::Integer) (view.jl:125,130; column.jl:93-99): a one-element selector that composes like a Number */
int32_t dfdb_query_add_integer(dfdb_query* q, int64_t i);
/* selection(v, cols => f) / selection(v, ::DFColumn{Bool}): IR must type to Bool else ArgumentError */
int32_t dfdb_query_add_predicate(dfdb_query* q, const uint8_t* ir, size_t len);
int32_t dfdb_query_nstages(dfdb_query* q, int32_t* n);
/* projection(v, …): replaces the projection by `n` named expressions (plain column = IR "COL k") */
int32_t dfdb_query_set_projection(dfdb_query* q, int32_t n, const char* const* names,
                                  const uint8_t* const* irs, const size_t* lens);
int32_t dfdb_query_ncols(dfdb_query* q, int32_t* n);                 /* ncol(v): view.jl:207-209 */
int32_t dfdb_query_coltype(dfdb_query* q, int32_t i, int32_t* dtype); /* coltype: projection.jl:80-81 */
/* result dtype of an expression over the table's columns: Base._return_type in BlockBroadcasting (broadcast.jl:13) */
int32_t dfdb_expr_result_type(dfdb_table* t, const uint8_t* ir, size_t len, int32_t* dtype);

/* multi-GPU: survivors of the stages before range stage `stage` that live on lower ranks
 * (exclusive scan of per-shard counts); default 0 */
int32_t dfdb_query_set_stage_base(dfdb_query* q, int32_t stage, int64_t survivors_before);
/* evaluate the queue only up to (not including) stage `nstages` and count: lets the host run the
 * all-gather between a predicate stage and a following range stage */
int32_t dfdb_query_count_prefix(dfdb_query* q, int32_t nstages, int64_t* n);

/* ------------------------------------------------------------------ execution */
/* SelectionExecutor.apply over every resident block (selection.jl:161-167, blocksiterator.jl:98-145):
 * leaves the selection bitmap + per-tile counts + their prefix in HBM.  Asynchronous. */
int32_t dfdb_query_execute(dfdb_query* q);
/* materialize(::DFView) evaluates the selection and then copies the projection (materialization.jl:27-40).  Telling the
 * engine BEFORE the first execution (dfdb_count / dfdb_query_execute) that the projection will be materialised lets a
 * single-stage scan of simple terms keep the selected values of a projected 8-byte predicate column while it has them in
 * registers; dfdb_materialize then copies them instead of gathering (re-reading) the column.  The same holds for a String column
 * filtered by ONE string term (== / != / startswith / endswith, pattern <= 64 bytes) and projected itself: the match pass keeps
 * the selected rows' sizes and bytes.  Results are identical; while the hint is on the query holds an extra nrows*8-byte buffer
 * (numeric capture) or nrows*4 bytes + a copy-sized byte arena (String capture). */
int32_t dfdb_query_hint_materialize(dfdb_query* q, int32_t on);
/* sum(col) / mean(col) / minimum / maximum over a filtered view (Base.iterate(::DFColumn), src/tables/column.jl:102-126;
 * docs/src/index.md:503-509) evaluate the selection and then reduce the column.  Telling the engine BEFORE the first execution that
 * dfdb_aggregate(q, op, proj_col) will follow (op = DFDB_AGG_SUM / _MIN / _MAX) lets the scan that produces the final mask reduce the
 * selected values of that column while it holds them (the column must
 * be a simple `col OP const` term of the last predicate stage, Int64 / UInt64 / Float64); dfdb_aggregate then only reduces one partial
 * per 1024-row tile.  op = 0 clears the hint.  Results: Int sums identical (wrapping), Float64 within the stated tolerance. */
int32_t dfdb_query_hint_aggregate(dfdb_query* q, int32_t op, int32_t proj_col);

/* unique(col) (Base.unique over Base.iterate(::DFColumn), src/tables/column.jl:102-126; docs/src/index.md:171-182,479-486):
 * narrows the CURRENT selection of q to the rows holding the first occurrence of their value in projection column proj_col
 * (a plain column; isequal semantics: NaN == NaN, 0.0 != -0.0, missing == missing), so dfdb_count / dfdb_materialize
 * afterwards give the distinct values in order of first appearance, like Julia.  dfdb_query_reset / _execute restore the
 * full selection. */
int32_t dfdb_query_unique(dfdb_query* q, int32_t proj_col);

/* groupreduce(view, (:key,); out = :val => Stat()) (src/tables/aggregate.jl:1-36, exported by the reference but unfinished there: it numbers the
 * groups in order of first appearance of the key — group_map[elem] = length(group_map) + 1 — and stops).  Completed to that intent: one group per
 * distinct value of projection column key_col (a plain column; isequal, missing is a group), groups in order of FIRST APPEARANCE, per group the
 * row count and stat(val_col) with stat = DFDB_AGG_COUNT / _SUM / _MIN / _MAX over a plain numeric column (Mean() = sum / count; integer sums wrap
 * like Julia's, Float64 sums are atomic adds in no fixed order: tolerance of DESIGN.md section 5; minimum / maximum propagate NaN).
 * Call 1 evaluates on the device and returns the number of groups (and the key column's string bytes); call 2 copies keys (a dfdb_outcol sized
 * for ngroups rows), counts and values (values_i for integer columns, values_f for floats; either may be NULL) to the caller and puts the query's
 * full selection back (between the two calls the selection is narrowed to the first occurrences, as after dfdb_query_unique). */
int32_t dfdb_query_groupreduce(dfdb_query* q, int32_t key_col, int32_t val_col, int32_t stat, int64_t* ngroups, int64_t* key_string_bytes);
int32_t dfdb_query_groupreduce_fetch(dfdb_query* q, dfdb_outcol* keys, int64_t* counts, int64_t* values_i, double* values_f);

/* forget the cached execution so the next count/indices/materialize re-evaluates the selection (a new
 * BlocksIterator in the reference: blocksiterator.jl:20-44). */
int32_t dfdb_query_reset(dfdb_query* q);
/* nrow(v) / size(v,1) / length(col): view.jl:192-206, column.jl:46-52 */
int32_t dfdb_count(dfdb_query* q, int64_t* n);
/* same total written to a caller buffer without a host sync when memkind == DFDB_MEM_DEVICE (the operand of
 * the multi-GPU all-reduce) */
int32_t dfdb_count_to(dfdb_query* q, int64_t* out, int32_t memkind);
/* 1 bit per resident row, LSB-first in uint64 words (the mask `apply` returns, packed) */
int32_t dfdb_select_bitmap(dfdb_query* q, uint64_t* out, int32_t memkind);
/* ascending 1-based table row numbers of the selected rows; n may be NULL (no host sync) */
int32_t dfdb_select_indices(dfdb_query* q, int64_t* out, int64_t cap, int32_t memkind, int64_t* n);
/* string bytes the i-th projection column will need (0 for fixed width) */
int32_t dfdb_result_string_bytes(dfdb_query* q, int32_t i, int64_t* nbytes);
/* materialize(v): materialization.jl:27-40 (projection gather projection.jl:128-154 + append).  When computed columns raise (DivideError /
 * InexactError on a selected row) the error returned is the one the reference's iteration meets first: first block, then first column of
 * the projection, then first row. */
int32_t dfdb_materialize(dfdb_query* q, dfdb_outcol* outs, int32_t ncols);
/* sum/min/max/count of projection column i over the selected rows; Float64 sums are pairwise
 * (tolerance documented in DESIGN.md), integer results exact */
int32_t dfdb_aggregate(dfdb_query* q, int32_t op, int32_t i, int64_t* out_i, double* out_f);

/* ---- out of core behind the ordinary entry points (round 6) ----
 * The reference never holds more than one block per column (src/io/blocksiterator.jl:98-121, src/io/BlockStreams.jl:9-15; "memory use is O(block)",
 * docs/src/index.md:182,192) and opens exactly required_columns(v) (src/tables/view.jl:183-190, blocksiterator.jl:20-33).  A query over a table opened
 * with dfdb_table_open whose required columns are NOT all resident is answered the same way by the SAME entry points — dfdb_count, dfdb_count_to,
 * dfdb_select_indices, dfdb_result_string_bytes, dfdb_materialize, dfdb_aggregate, dfdb_query_unique, dfdb_query_groupreduce (+ _fetch): each runs the
 * block stream below internally, chunk by chunk (ctx option "ooc_chunk_blocks", default 512 blocks per chunk), and merges the chunks' results inside the
 * library; HBM holds the stream's chunks, never the table.
 *   dfdb_count            reads the selection's columns only — the FIRST projection column when the queue holds no predicate — like the reference's row
 *                         counter (BlockRowsIterator, blocksiterator.jl:46-66); with dfdb_query_hint_materialize on it also sizes the projected String
 *                         columns (for the blocks that kept a row), so count + materialize stay the reference's two passes (materialization.jl:29-37)
 *   dfdb_materialize      appends every chunk's rows to the caller's buffers (append!(res, bl), materialization.jl:33-37); more rows than dfdb_count
 *                         counted (the files changed in between) is BoundsError, nothing is written past the count
 *   dfdb_aggregate        per-chunk reductions folded in chunk order: Int sums wrap, Float64 sums are sums of the chunks' sums (tolerance of DESIGN.md
 *                         section 5), minimum / maximum with Julia's NaN and signed-zero rules; empty -> ArgumentError for minimum / maximum
 *   dfdb_query_unique     per-chunk first occurrences merged by key (isequal) in chunk order = order of first appearance in the table; the query is
 *                         then NARROWED to those rows: dfdb_count = the number of distinct values, dfdb_select_indices = their table rows,
 *                         dfdb_materialize = the projection at those rows (the key column straight out of the merge), until dfdb_query_reset
 *   dfdb_query_groupreduce  per-chunk groups merged by key in chunk order; counts and sums add, minimum / maximum fold (as dfdb_group_query_groupreduce)
 *   dfdb_select_bitmap    DFDB_ERR_UNSUPPORTED (the bitmap exists one chunk at a time: dfdb_stream_next)
 *   dfdb_table_add_from_query  the new resident column is filled from the stream, chunk by chunk (add_column!(t, name, lazy_col) / create_table(path; from = view)
 *                         over a view of a table that does not fit: table.jl:96-124, creators.jl:18-60)
 * dfdb_query_execute is a no-op there (nothing is left in HBM), dfdb_query_reset forgets what the passes learnt.
 *
 * dfdb_query_prepare(q, &how): "only the required columns are opened".  Brings the columns the view needs — and no others — into HBM when they fit
 * ctx option "hbm_budget_mb" (an explicit budget bounds what the TABLE holds; in any case at most 80 % of the HBM that is free at the call): *how = 0 they were
 * resident already, 1 loaded decoded (dfdb_table_load of exactly those ordinals), 2 loaded COMPRESSED-ONLY (as ctx option "keep_compressed" = 2: every
 * missing column is a plain fixed-width one and the LZ4 blocks fit where the decoded arrays do not), 3 left on disk: the entry points above stream.
 * DFDB_ERR_NOMEM inside a load falls through to the next form instead of failing.  A binding calls it once per view before it asks for results
 * (julia/DataFrameDBsAMD.jl: device_query).  dfdb_table_unload releases resident columns of a file-backed table again (NULL = all). */
int32_t dfdb_query_prepare(dfdb_query* q, int32_t* how);
int32_t dfdb_table_unload(dfdb_table* t, const int32_t* ordinals, int32_t ncols);
/* what the internal streams of this query have read so far: rows of the blocks read, their compressed bytes (+ 24 per block, quirk Q10), their decoded
 * bytes — summed over the passes (a count reads one column; a materialize after it reads the projection's blocks that kept a row) */
int32_t dfdb_query_read_stats(dfdb_query* q, dfdb_sizestats* stats);

/* ---- block-streamed execution: Base.iterate(::BlocksIterator) (src/io/blocksiterator.jl:98-145) in chunks of blocks ----
 * For a query over a table opened with dfdb_table_open but NOT loaded (tables larger than HBM, one-off scans).  Each
 * dfdb_stream_next yields a query over the next chunk of `chunk_blocks` blocks of every required column, decoded in HBM:
 * use dfdb_count / dfdb_select_indices (global 1-based row numbers) / dfdb_materialize / dfdb_aggregate on it; it is
 * owned by the stream and valid until the next dfdb_stream_next / dfdb_stream_close (never dfdb_query_free it) — the
 * contract of the NamedTuple an iteration of the reference yields.  *chunk == NULL marks the end.  While the caller works
 * on chunk i a loader thread reads, copies and LZ4-decodes chunk i+1 on its own HIP stream.  The per-stage running offsets
 * of RangeToProcess (selection.jl:68-75,107), skip_if_can and is_finished (:177-196) carry over between chunks, so
 * t[pred, :][1:100, :] or head(t) read only the chunks they need.  With chunk_blocks >= 256 the first two chunks are shorter (a quarter and half
 * of chunk_blocks): the first rows arrive after a quarter of a full chunk's latency. */
int32_t dfdb_stream_open(dfdb_query* q, int64_t chunk_blocks, dfdb_stream** out);
int32_t dfdb_stream_next(dfdb_stream* s, dfdb_query** chunk, int64_t* chunk_rows, int64_t* first_row);
int32_t dfdb_stream_stats(dfdb_stream* s, dfdb_sizestats* stats);   /* table_stats over the required columns (headers only) */
/* Late materialization at block granularity (blocksiterator.jl:111-113: `if rows > 0` guards read_block! of the projection-only columns;
 * skip_block, BlockStreams.jl:74-78): a loader reads, copies and decodes the columns the selection reads, evaluates the
 * chunk's selection, and reads the projection-only columns ONLY for the blocks that kept a row — the other blocks' bytes never leave the file.
 * This reports what the loaders have read so far of table column `ordinal` (-1: of every required column): rows of the blocks read, their
 * compressed bytes (+ 24 per block like dfdb_stream_stats, quirk Q10) and their decoded bytes.  The loaders run ahead of the caller by up
 * to three chunks; after the last dfdb_stream_next it is the whole scan's figure. */
int32_t dfdb_stream_read_stats(dfdb_stream* s, int32_t ordinal, dfdb_sizestats* stats);
int32_t dfdb_stream_close(dfdb_stream* s);

/* ---- multi-GPU groups: ONE table block-range sharded over the GPUs of a node (§8b backend HIP_N) ----
 * Blocks are independent units every column shares (check_column_head, filesystem.jl:47-54): rank g of G owns blocks
 * [g*ceil(nb/G), (g+1)*ceil(nb/G)) of every column and runs the ordinary engine over them.  The shards meet only in
 * nrow(v) (view.jl:192-206: one RCCL all-reduce of an Int64), sum / minimum / maximum (Base.iterate(::DFColumn), column.jl:102-126:
 * one all-reduce of {value, count}) and in a range stage that follows a predicate stage, which numbers the GLOBAL survivor
 * stream (RangeToProcess.offset, selection.jl:68-75,94-111: all-gather of one Int64 per rank + exclusive scan, done inside).
 * No column data crosses xGMI.  Rank order = table order.
 *   dfdb_group_create        one process drives n GPUs (a host thread per GPU, ncclCommInitAll): what a Julia session gets
 *   dfdb_group_create_rank   one process per GPU (ncclCommInitRank; the id comes from dfdb_group_unique_id on rank 0 and is
 *                            handed round by the launcher: MPI, Distributed.jl, a torch.distributed store)
 *   dfdb_group_create_rank_callbacks   one process per GPU, the exchanges through the caller's own collectives (dfdb_exchange_fns)
 * RCCL is dlopen'ed on first use.  DFDB_EXCHANGE_HOST does the same exchanges through host memory and exists for one-process
 * groups whose shards share a physical GPU (functional tests on a 1-GPU box; RCCL refuses duplicate devices).
 * Failures: when the per-shard half of a collective call fails on one rank only (a DivideError / InexactError that only its rows reach, out of
 * memory, a column that is not resident) that rank still takes part in the exchange, and every exchange carries an 8-byte fault key reduced
 * with MIN: all ranks return the SAME error from the same call — the one of the lowest table row when the error knows its row, which is the
 * one the reference's serial block iteration meets first.  An enqueue-only call (dfdb_group_count(gq, NULL)) returns a local failure at once
 * and the other ranks meet it at their next call that reads a result back.  A rank that dies outright must take its process down with a
 * non-zero exit (the launcher then stops the others); it must never re-exec. */
typedef struct dfdb_group dfdb_group;
typedef struct dfdb_gtable dfdb_gtable;   /* a DFTable, sharded */
typedef struct dfdb_gquery dfdb_gquery;   /* a DFView over it */
enum { DFDB_EXCHANGE_AUTO = 0, DFDB_EXCHANGE_RCCL = 1, DFDB_EXCHANGE_HOST = 2, DFDB_EXCHANGE_CALLBACK = 3 };
/* DFDB_EXCHANGE_CALLBACK: one process per GPU whose host brings its OWN collectives (MPI, Distributed.jl, torch.distributed over gloo): the two
 * exchanges a group needs, over HOST memory, blocking, called by every rank in the same order.  Both return 0 on success.
 *   allreduce  vals[n] (8-byte values: dtype DFDB_I64 / DFDB_U64 / DFDB_F64; op DFDB_AGG_SUM / _MIN / _MAX) := the reduction over all ranks, in place
 *              (integer sums wrap; the library never asks for a Float64 MIN / MAX: it gathers and folds with Julia's NaN rules itself)
 *   allgather  recv[world * nbytes] := every rank's send[nbytes], rank order */
typedef struct dfdb_exchange_fns {
  void* user;
  int32_t (*allreduce)(void* user, void* vals, int32_t n, int32_t dtype, int32_t op);
  int32_t (*allgather)(void* user, const void* send, void* recv, int64_t nbytes);
} dfdb_exchange_fns;
#define DFDB_GROUP_ID_BYTES 128

int32_t dfdb_group_create(const int32_t* device_ids, int32_t n, int32_t exchange, dfdb_group** out);
int32_t dfdb_group_unique_id(uint8_t id[DFDB_GROUP_ID_BYTES]);
int32_t dfdb_group_create_rank(int32_t device_id, void* hip_stream, const uint8_t id[DFDB_GROUP_ID_BYTES], int32_t rank, int32_t world,
                               dfdb_group** out);
/* the same as dfdb_group_create_rank with the caller's collectives instead of RCCL (the function table is copied; `user` must outlive the group) */
int32_t dfdb_group_create_rank_callbacks(int32_t device_id, void* hip_stream, int32_t rank, int32_t world, const dfdb_exchange_fns* fns, dfdb_group** out);
int32_t dfdb_group_destroy(dfdb_group* g);
int32_t dfdb_group_info(dfdb_group* g, int32_t* world, int32_t* nlocal, int32_t* first_rank, int32_t* exchange);
int32_t dfdb_group_ctx(dfdb_group* g, int32_t local, dfdb_ctx** ctx);          /* borrowed */
int32_t dfdb_group_synchronize(dfdb_group* g);                                 /* the local engine streams have drained */
int32_t dfdb_group_barrier(dfdb_group* g);                                     /* ... on every rank */
int32_t dfdb_group_set_option(dfdb_group* g, const char* key, int64_t value);  /* dfdb_ctx_set_option on every local shard */
/* all-reduce (DFDB_AGG_SUM / _MIN / _MAX) of n <= 12 caller scalars per local shard, vals[nlocal][n], in place */
int32_t dfdb_group_allreduce_f64(dfdb_group* g, double* vals, int32_t n, int32_t op);

int32_t dfdb_group_table_open(dfdb_group* g, const char* path, dfdb_gtable** out);       /* open_table: creators.jl:7-16 */
int32_t dfdb_group_table_new(dfdb_group* g, int64_t block_size, dfdb_gtable** out);
int32_t dfdb_group_table_close(dfdb_gtable* gt);
/* dfdb_table_unload on every shard (NULL = all columns): the files stay, the group's entry points stream them from then on */
int32_t dfdb_group_table_unload(dfdb_gtable* gt, const int32_t* ordinals, int32_t ncols);
/* every shard loads ITS block range of the listed columns (NULL = all): dfdb_table_load(block range of rank) */
int32_t dfdb_group_table_load(dfdb_gtable* gt, const int32_t* ordinals, int32_t ncols, dfdb_sizestats* stats);
/* columns of the WHOLE table (nrows_total rows); every shard generates / uploads the rows of its block range */
int32_t dfdb_group_table_add_generated(dfdb_gtable* gt, const char* name, int32_t generator, uint64_t seed, int64_t nrows_total);
int32_t dfdb_group_table_add_column(dfdb_gtable* gt, const char* name, int32_t dtype, int64_t nrows_total, const void* data,
                                    const uint8_t* bytes, int64_t nbytes, const uint8_t* missing);
int32_t dfdb_group_table_nrows(dfdb_gtable* gt, int64_t* total);                         /* rows of the whole table */
int32_t dfdb_group_table_shard(dfdb_gtable* gt, int32_t local, dfdb_table** t);          /* borrowed: the ordinary handle of one shard */

int32_t dfdb_group_query_new(dfdb_gtable* gt, dfdb_gquery** out);                        /* DFView(table): view.jl:50 */
int32_t dfdb_group_query_free(dfdb_gquery* gq);
/* dfdb_query_prepare for a sharded table: every shard loads ITS block range of exactly the columns the view needs when its share fits (*how = 1; 0: they
 * were resident already) — group option "hbm_budget_mb", 0 = 80 % of the device's HBM; decided alike on every rank from the files' headers, never from a
 * rank's momentary free memory.  Otherwise (*how = 3) the shards keep nothing: every dfdb_group_* entry point then has each shard STREAM its block range of
 * the column files (the block windows are laid out by dfdb_group_table_open) and the per-rank values meet in the same exchanges as resident shards' do.
 * A sharded table that was opened and never loaded or prepared behaves like *how = 3. */
int32_t dfdb_group_query_prepare(dfdb_gquery* gq, int32_t* how);
int32_t dfdb_group_query_add_range(dfdb_gquery* gq, int64_t start, int64_t step, int64_t stop);   /* as dfdb_query_add_*, on every shard */
int32_t dfdb_group_query_add_indices(dfdb_gquery* gq, const int64_t* idx, int64_t n);
int32_t dfdb_group_query_add_integer(dfdb_gquery* gq, int64_t i);
int32_t dfdb_group_query_add_predicate(dfdb_gquery* gq, const uint8_t* ir, size_t len);
int32_t dfdb_group_query_set_projection(dfdb_gquery* gq, int32_t n, const char* const* names, const uint8_t* const* irs, const size_t* lens);
int32_t dfdb_group_query_hint_aggregate(dfdb_gquery* gq, int32_t op, int32_t proj_col);
int32_t dfdb_group_query_hint_materialize(dfdb_gquery* gq, int32_t on);
int32_t dfdb_group_query_reset(dfdb_gquery* gq);
int32_t dfdb_group_query_shard(dfdb_gquery* gq, int32_t local, dfdb_query** q);          /* borrowed */
/* nrow(v) over the whole table (view.jl:192-206).  n == NULL only enqueues: the reduced count stays on the devices, no host wait */
int32_t dfdb_group_count(dfdb_gquery* gq, int64_t* n);
int32_t dfdb_group_shard_counts(dfdb_gquery* gq, int64_t* counts /* world values, rank order */);
/* sum / minimum / maximum / count over the whole table; integers exact, Float64 sums within the tolerance of DESIGN.md */
int32_t dfdb_group_aggregate(dfdb_gquery* gq, int32_t op, int32_t i, int64_t* out_i, double* out_f);
/* 1-based TABLE row numbers: each local shard into its own device buffer (asynchronous) ... */
int32_t dfdb_group_select_indices_device(dfdb_gquery* gq, int64_t* const* outs, const int64_t* caps);
/* ... or the local shards concatenated in rank order into ONE host buffer (*n = rows written by this process) */
int32_t dfdb_group_select_indices(dfdb_gquery* gq, int64_t* out, int64_t cap, int64_t* n);
int32_t dfdb_group_result_string_bytes(dfdb_gquery* gq, int32_t i, int64_t* nbytes);
/* materialize(v) (materialization.jl:27-40) into caller-owned HOST buffers: the local shards' rows in rank order = table order */
int32_t dfdb_group_materialize(dfdb_gquery* gq, dfdb_outcol* outs, int32_t ncols);
/* ... or left SHARDED on the devices ("results stay sharded per GPU"): outs[l * ncols + p] is output column p of local shard l,
 * buffers in that shard's own HBM (memkind DFDB_MEM_DEVICE), sized from dfdb_group_shard_counts (rows of rank first_rank + l) and
 * dfdb_group_shard_string_bytes (String columns).  Asynchronous on the shards' engine streams; nothing crosses PCIe or xGMI.  Rank order = table
 * order, so the caller's lazy concatenation of the shards IS the materialised view. */
int32_t dfdb_group_shard_string_bytes(dfdb_gquery* gq, int32_t i, int64_t* nbytes /* nlocal values */);
int32_t dfdb_group_materialize_device(dfdb_gquery* gq, dfdb_outcol* outs /* [nlocal][ncols] */, int32_t ncols);

/* unique(col) over the WHOLE table (Base.unique over Base.iterate(::DFColumn), column.jl:102-126; docs/src/index.md:171-182,479-487).  Every shard
 * reduces its own rows on its own device (dfdb_query_groupreduce's kernels), ONE record per distinct key crosses to the other ranks (all-gather over
 * RCCL; nothing at all when one process holds every shard) and the records are merged by key in rank order — first appearance = lowest rank, then
 * lowest row — so the distinct values come out in the order Julia's unique gives over the whole column (isequal: NaN == NaN, -0.0 != 0.0, missing
 * is a value).  Call 1 returns how many there are and the string bytes they take; the fetch copies them into a caller-owned HOST column and
 * forgets them.  Every rank receives the whole answer. */
int32_t dfdb_group_query_unique(dfdb_gquery* gq, int32_t proj_col, int64_t* ndistinct, int64_t* string_bytes);
int32_t dfdb_group_query_unique_fetch(dfdb_gquery* gq, dfdb_outcol* keys);
/* groupreduce(view, (:key,); out = :val => Stat()) over the whole table (aggregate.jl:1-36, completed as dfdb_query_groupreduce completes it): the
 * shards' groups merged by key in rank order = groups in order of first appearance in the table; counts and sums add (Int sums wrap as on one device,
 * Float64 sums are the sum of the shards' sums: tolerance of DESIGN.md section 5), minimum / maximum fold with Julia's NaN and signed-zero rules. */
int32_t dfdb_group_query_groupreduce(dfdb_gquery* gq, int32_t key_col, int32_t val_col, int32_t stat, int64_t* ngroups, int64_t* key_string_bytes);
int32_t dfdb_group_query_groupreduce_fetch(dfdb_gquery* gq, dfdb_outcol* keys, int64_t* counts, int64_t* values_i, double* values_f);

#ifdef __cplusplus
}
#endif
#endif
