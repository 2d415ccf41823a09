// engine.hpp — host-side objects behind the opaque C-ABI handles.
#pragma once
#include "common.hpp"
#include "expr.hpp"
#include "kernels.hpp"
#include <memory>
#include <mutex>

namespace dfdb {

// where the blocks of a column file lie (read_sizes without the bodies: BlockStreams.jl:68-78, header then skip(io, compressed)).  Walked lazily, a
// chunk's worth of headers at a time, and kept with the table's column: a second stream over the same file finds it done.  30 000 headers are 30 000
// preads (~20 ms), which a block-streamed scan of 2e9 rows used to pay before its first byte moved.
struct BlockLoc { int64_t off; int32_t rows; int64_t origin, compressed; };
struct BlockIndex {
  std::string file; size_t data_off = 0;
  int64_t file_size = -1, file_mtime_ns = 0;   // what the file looked like when the walk began (a rewritten file gets a fresh index)
  std::mutex mu;
  std::vector<BlockLoc> v;                     // guarded by mu: readers copy the slice they need
  int64_t next_pos = -1;                       // where the next header lies (-1: not started)
  bool complete = false;
};

// one column of a table, decoded and resident in HBM as ONE contiguous array over all resident blocks
// (every block holds block_size rows except the last, so block b starts at row b*block_size)
struct Column {
  std::string name;
  int64_t id = 0;
  int32_t dtype = 0;
  std::string logical;   // "Date" / "DateTime" / "Time" / "Char": a bits type carried as its integer representation ("" otherwise)
  bool resident = false;
  int64_t nrows = 0;
  DevBuf data;       // fixed width: nrows*width bytes (+ pad); String: int32 sizes[nrows] (-1 = missing)
  DevBuf bytes;      // String: byte arena (+16 B pad so 8-byte probes never fault)
  int64_t nbytes = 0;
  DevBuf tile_off;   // String: u64[nstrtiles+1] byte offset of each 1024-row tile (K4)
  uint32_t max_tile_bytes = 0;   // String: the most bytes any 1024-row tile holds (K4; K5 stages a tile's bytes in LDS when every tile fits)
  DevBuf missing;    // nullable fixed width: bitmap, 1 = missing, padded like the selection bitmap
  // compressed-resident form (ctx option "keep_compressed" at load time; plain fixed-width columns): the column's LZ4 blocks as they
  // sit in the file stay in HBM with their descriptors, and dfdb_table_decode_resident re-runs K7 from them into `data`
  DevBuf comp, comp_blocks, comp_status;
  int64_t comp_nblocks = 0;
  DevBuf comp_index;            // one bit per byte of comp: where an LZ4 sequence starts (recorded by the first resident decode, read by the later ones)
  int comp_index_state = 0;     // 0: not recorded yet, 1: recorded
  // COMPRESSED-ONLY form (ctx option "keep_compressed" = 2 at load time; round 5): the LZ4 blocks + the index are all the column holds —
  // `data` is empty.  `col OP const` conjuncts are evaluated by the decoder itself (K7 HIST + SCAN: no decoded byte reaches a column array), gathers decode
  // the blocks that kept a row into a per-query arena, and anything else gets a whole-column decode that lives for one ABI call (`transient`).
  bool comp_only = false;
  bool transient = false;       // `data` is such a whole-column decode: released when the ABI call that needed it returns (TransientScope)
  std::vector<Lz4Block> comp_blocks_host;   // the descriptors of comp_blocks, for decoding subsets of the blocks
  // placement calibration (query.cpp: place_mask): the selection bitmap this column's scans run fastest against, found once by timing
  // the scan against a few candidate allocations; lent to one query at a time
  DevBuf mask_pref;
  bool mask_calibrated = false, mask_lent = false;
  float mask_ms_best = 0, mask_ms_worst = 0;   // per scanned sample, for dfdb_ctx_profile / the bench's JSON
  // dictionary form of a low-cardinality String column (dfdb_table_build_dictionary / ctx option "string_dictionary"; k_dict.hip): one 16-bit code per
  // row beside the flat form, the distinct strings once on the device and on the host (predicates are evaluated on the host copy)
  DevBuf dict_codes, dict_len, dict_off, dict_bytes;
  std::vector<std::string> dict_host;
  int32_t dict_n = 0;           // 0 = no dictionary
  // on-disk source (tables opened from files)
  std::string file;
  size_t data_off = 0;  // first block inside the file
  std::shared_ptr<BlockIndex> bix;   // block locations of `file`, as far as a stream or table_column_stats has walked them
};

enum StageKind { ST_RANGE = 0, ST_INTEGER = 1, ST_INDICES = 2, ST_PRED = 3 };

struct Stage {
  int kind = ST_RANGE;
  int64_t start = 0, step = 1, stop = 0, n = 0;  // ST_RANGE (stop normalised to the last element)
  std::vector<int64_t> idx;                      // ST_INDICES / ST_INTEGER, caller order
  NodePtr pred;                                  // ST_PRED
  int64_t stage_base = 0;                        // survivors on lower ranks (multi-GPU)
  int64_t first() const;                         // minimum (RangeToProcess.first)
  int64_t last() const;                          // maximum (RangeToProcess.last)
  int64_t elem(int64_t k) const;                 // 1-based element with Julia bounds checking
};

struct ProjCol { std::string name; NodePtr expr; };

}  // namespace dfdb

namespace dfdb { struct OocState; }
struct dfdb_query;
struct dfdb_table {
  dfdb_ctx* ctx = nullptr;
  std::string path;            // empty for in-memory tables
  int64_t block_size = 65536;
  int64_t format_version = 1;
  std::vector<dfdb::Column> cols;
  int64_t nrows = -1;          // rows resident (all resident columns agree); -1 = nothing resident yet
  int64_t row_base = 0;        // global 0-based row of local row 0 (block-range shard)
  int64_t block_first = 0;     // first resident block
  // block window of the table's out-of-core streams: a shard of a multi-GPU group that is NOT resident streams only ITS block range [win_first, win_last)
  // of the column files (win_last < 0: to the end of the file); rows keep their table numbers (row_base of a chunk = its first block * block_size)
  int64_t win_first = 0, win_last = -1;
  std::vector<dfdb_query*> queries;   // live queries over this table (orphaned, not dangling, when the table closes)
  // block-loading scratch (compressed bytes staged in HBM, packed bodies, block descriptors).  Freed after a load unless
  // the table is a stream slot that reloads a new block range every few milliseconds (hipMalloc/hipFree would dominate).
  dfdb::DevBuf ld_staged, ld_bodies, ld_blocks, ld_status, ld_aux;
  bool keep_load_scratch = false;
  bool ld_prestaged = false;      // the caller has already queued the image's compressed byte range into ld_staged on this table's stream (stream.cpp)
};

struct dfdb_query {
  dfdb_table* t = nullptr;
  std::vector<dfdb::Stage> stages;
  std::vector<dfdb::ProjCol> proj;
  // device state of the last execution
  dfdb::DevBuf bitmap, tile_counts, prefix, scan_scratch, idx_sorted, red_scratch, red_result, tmp_a, tmp_b, tmp_c,
      str_sizes, str_toff, str_bytes, str_scratch;
  // dfdb_query_hint_materialize: the projection WILL be materialised after the scan, so a single-stage scan of simple terms
  // keeps the selected values of one projected 8-byte predicate column (per-tile compact, cap_buf) and that column's
  // projection becomes a contiguous copy instead of a gather
  bool hint_materialize = false;
  int cap_col = -1;            // table ordinal captured by the last execution (-1: none)
  int cap_col2 = -1;           // a second captured column (k_scan_terms EXTRA = 5: the term before the last), in cap_buf2
  dfdb::DevBuf cap_buf2;
  int decoded_col = -1;        // table ordinal whose resident LZ4 blocks the last execution decoded on its way (decode_on_scan; -1: none)
  std::vector<int> comp_scanned;   // compressed-only columns whose blocks the last execution decoded inside its scan (their statuses are read with the count)
  // compressed-only projection columns: the blocks that kept a row, decoded for THIS query's gathers at their natural offsets inside the span
  // [first such block, last such block] — the iterator owns its decode buffers, like the reference's (blocksiterator.jl:98-121)
  struct Arena { dfdb::DevBuf buf, blocks, status; int64_t first_row = 0; int64_t nblocks = 0; bool valid = false; const void* from = nullptr; };   // from: the blocks (Column::comp) it was decoded out of
  std::map<int, Arena> arenas;
  dfdb::DevBuf cap_buf;
  // the same for a projected String column filtered by ONE short-pattern string term (K5 CAP): sizes per tile, bytes at the tile's arena
  // offset, selected byte totals per tile
  int cap_str_col = -1;
  // a conjunct `strcol == "const"` of some stage holds for every finally selected row: the projection of that column is `const` repeated
  int const_str_col = -1; std::string const_str;
  // dfdb_query_hint_aggregate: sum(projection column) WILL be asked for: when that column is a simple term of the launch that produces
  // the final mask, the scan adds up the selected values while it holds them (one partial per 1024-row tile) and dfdb_aggregate only
  // reduces the partials
  int hint_agg_op = 0, hint_agg_proj = -1;
  // the smallest rows on which a predicate of the current execution hit DivideError [0] / InexactError [1] (~0: none); decided at the end of query_execute
  uint64_t err_row[2] = {~0ull, ~0ull};
  bool err_checking = false;   // inside error_is_reached's partial executions: errors are not raised
  uint64_t* proj_err = nullptr; // query_materialize: the first erroring row per kind of the projection column being computed lands here instead of raising
  int agg_col = -1;            // table ordinal whose per-tile sums agg_partials holds (-1: none)
  int agg_dtype = 0, agg_op = 0;
  dfdb::DevBuf agg_partials, agg_ones;
  int64_t agg_ones_tiles = -1;
  dfdb::DevBuf cap_str_sizes, cap_str_bytes, cap_str_tb;
  bool stream_owned = false;   // a chunk query handed out by dfdb_stream_next: owned by the stream
  dfdb::DevBuf dict_sel;       // K9: the selected rows' dictionary codes (materialize of a dictionary column)
  // dfdb_query_groupreduce: the full selection set aside, per-group counts / values, what the fetch needs
  dfdb::DevBuf gr_sel, gr_cnt, gr_val, gr_keys;     // (gr_keys: the key column's values at the groups' first rows, group order — k_group_acc_hash_lds)
  dfdb::DevBuf du_first, du_rows, du_rank;   // dict_unique's scratch (first row per code, the group rows, rank of every code)
  int64_t gr_n = 0; int gr_key = -1, gr_op = 0, gr_kind = 0, gr_state = 0;   // state 0: none, 1: empty result, 2: results + narrowed selection pending
  int mask_from = -1;          // table ordinal of the column whose calibrated bitmap this query has borrowed (-1: its own)
  int64_t bitmap_rows = -1;    // rows the bitmap was sized (and zero-padded) for
  int executed_stages = -1;    // how many stages the current bitmap reflects (-1 = none)
  bool prefix_valid = false;
  int64_t count = -1;          // host copy of the total (valid when >= 0)
  // what the query has learnt by block-streaming a table whose required columns are not resident (ooc.hpp; null until it does)
  std::shared_ptr<dfdb::OocState> ooc;
};

namespace dfdb {
// engine entry points used by c_api.cpp
void table_open(dfdb_ctx* ctx, const char* path, dfdb_table** out);
void table_add_column(dfdb_table* t, const char* name, int32_t dtype, int64_t nrows, const void* data, const uint8_t* bytes,
                      int64_t nbytes, const uint8_t* missing);
void table_add_generated(dfdb_table* t, const char* name, int32_t gen, uint64_t seed, int64_t row_first, int64_t nrows);
void table_load(dfdb_table* t, const int32_t* ordinals, int32_t ncols, int64_t block_first, int64_t block_last, dfdb_sizestats* stats);
void table_load_image(dfdb_table* t, int32_t ordinal, const uint8_t* image, size_t nbytes, int64_t block_first, int64_t block_last,
                      dfdb_sizestats* stats);

// one block of a chunk whose compressed body sits in the table's load staging buffer (stream.cpp -> table.cpp)
struct StagedBlock { int32_t rows; int64_t origin, compressed, staged_off, row_pos; };
void table_decode_staged_blocks(dfdb_table* t, int32_t ordinal, const StagedBlock* bl, int64_t n, int64_t total_rows, dfdb_sizestats* stats);
// survivors of the query's current execution per block of `block_size` rows (query.cpp; synchronises)
void query_block_counts(dfdb_query* q, int64_t block_size, std::vector<int64_t>& counts);

// jit.cpp: the interpreter's source compiled by hipRTC for one program shape (everything below is a literal of the generated kernel = the cache key)
struct JitShape {
  int mode = 0, str = 0, nul = 0, and_existing = 0, stack_levels = 0, result_dtype = 0, nstr = 0;
  std::vector<uint32_t> w0, w1, w2; std::vector<int32_t> slot, aslot, col_dtype;
  int32_t str_slot[4] = {0, 0, 0, 0};
};
struct JitKernel;
std::shared_ptr<JitKernel> jit_request(dfdb_ctx* ctx, const JitShape& sh, bool wait);
bool jit_launch(JitKernel& k, dfdb_ctx* ctx, unsigned grid, size_t lds_bytes, void** args);
void jit_shutdown();
std::string jit_cache_dir(std::string* why);   // "" = off or refused
void jit_stats(int64_t* compiled, int64_t* failed, int64_t* pending, int64_t* from_disk = nullptr);

void query_add_stage(dfdb_query* q, Stage&& s);   // composition rules of selection.jl:39-49
void query_execute(dfdb_query* q, int nstages);   // evaluate stages [0, nstages) -> bitmap + counts + prefix
int64_t query_count(dfdb_query* q, int nstages);
void query_select_bitmap(dfdb_query* q, uint64_t* out, int32_t memkind);
void query_select_indices(dfdb_query* q, int64_t* out, int64_t cap, int32_t memkind, int64_t* n);
int64_t query_string_bytes(dfdb_query* q, int i);
void query_materialize(dfdb_query* q, dfdb_outcol* outs, int32_t ncols);
void query_aggregate(dfdb_query* q, int32_t op, int32_t i, int64_t* out_i, double* out_f);
void query_unique(dfdb_query* q, int32_t p);
void query_groupreduce(dfdb_query* q, int32_t key_p, int32_t val_p, int32_t op, int64_t* ngroups, int64_t* key_bytes);
void query_groupreduce_fetch(dfdb_query* q, dfdb_outcol* keys, int64_t* counts, int64_t* vals_i, double* vals_f);
void query_return_mask(dfdb_query* q);            // give a borrowed calibrated bitmap back to its column
void set_string_tile_offsets(dfdb_ctx* ctx, Column& c);   // K4 over a resident string column
bool read_file_range(const std::string& file, uint8_t* dst, int64_t lo, int64_t hi);   // table.cpp: parallel pread
bool read_file_range_fd(int fd, uint8_t* dst, int64_t lo, int64_t hi);
void set_io_threads(int64_t n);                  // table.cpp: concurrent preads per byte range (ctx option "io_threads")
// stream.cpp: block-streamed execution over a non-resident table
void stream_open(dfdb_query* q, int64_t chunk_blocks, dfdb_stream** out);
dfdb_query* stream_next(dfdb_stream* s, int64_t* chunk_rows, int64_t* first_row);
void stream_close(dfdb_stream* s);
void stream_drop_parked(dfdb_ctx* ctx);
// K9: codes + dictionary for String column `ordinal` if it has at most max_entries distinct values; returns the number of entries (0: none built)
int64_t table_build_dictionary(dfdb_table* t, int32_t ordinal, int64_t max_entries);   // dfdb_ctx_destroy: the parked stream dies with its context
void stream_stats(dfdb_stream* s, dfdb_sizestats* st);
void stream_read_stats(dfdb_stream* s, int32_t ordinal, dfdb_sizestats* st);   // what the loaders have read so far of one column (-1: of every required column)
void table_column_stats(dfdb_table* t, int32_t ordinal, dfdb_sizestats* st);
void table_decode_resident(dfdb_table* t, int32_t ordinal);   // table.cpp
// compressed-only columns (table.cpp).  column_data: the decoded array of a fixed-width column — for a compressed-only one a whole-column decode made now
// (asynchronously, on the context's stream) and kept until table_drop_transient; TransientScope drops them when the ABI call that made them returns.
const void* column_data(dfdb_table* t, Column& c);
void table_drop_transient(dfdb_table* t);
struct TransientScope { dfdb_table* t; explicit TransientScope(dfdb_table* t_) : t(t_) {} ~TransientScope() { if (t) table_drop_transient(t); } };
uint8_t* ctx_hist_scratch(dfdb_ctx* ctx, int* waves);       // the history rings of K7's HIST forms (one buffer per context, made on first use)
void table_resident_bytes(dfdb_table* t, int32_t ordinal, int64_t* decoded, int64_t* compressed);
int64_t table_decode_status(dfdb_table* t, int32_t ordinal);   // table.cpp: blocks of the last resident decode whose status is not 0 (synchronises)
int column_lz4_index(dfdb_ctx* ctx, Column& c, bool form_takes_index);   // table.cpp: 0 / 1 (record) / 2 (use) for launch_lz4_decode*
int32_t ctx_create_like(const dfdb_ctx* like, dfdb_ctx** out);   // c_api.cpp
void ctx_destroy(dfdb_ctx* c);
void table_save(dfdb_table* t, const char* path, dfdb_sizestats* stats);                       // writer.cpp
void table_save_column(dfdb_table* t, int32_t ordinal, const char* file, dfdb_sizestats* stats);
void table_compress_column(dfdb_table* t, int32_t ordinal, int32_t mode, dfdb_sizestats* stats);   // writer.cpp: resident -> compressed-resident (1) / compressed-only (2) in HBM
void table_add_from_query(dfdb_table* dst, const char* name, dfdb_query* q, int32_t p);       // add_column!(t, name, lazy column)
}  // namespace dfdb
