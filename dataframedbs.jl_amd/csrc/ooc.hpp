// ooc.hpp — out-of-core execution behind the ordinary query entry points, and the by-key merge it shares with the multi-GPU groups.
//
// The reference never holds more than one block per column (src/io/blocksiterator.jl:98-121, src/io/BlockStreams.jl:9-15; "memory use is O(block)",
// docs/src/index.md:182,192) and opens exactly required_columns(v) (src/tables/view.jl:183-190, blocksiterator.jl:20-33).  A query over a table that was
// opened from files and whose required columns are NOT resident is answered the same way: dfdb_count / dfdb_select_indices / dfdb_result_string_bytes /
// dfdb_materialize / dfdb_aggregate / dfdb_query_unique / dfdb_query_groupreduce notice it (query_out_of_core) and run the block stream of stream.cpp
// internally, chunk by chunk, merging the per-chunk results HERE — the loops a binding used to have to write (round 5: dfdb/api.py).
#pragma once
#include "engine.hpp"
#include <unordered_map>

namespace dfdb {

// Julia's min / max over Float64: NaN propagates, and -0.0 orders below 0.0 (Base.min / Base.max); integer sums wrap like Julia's
double fold_f64(double x, double y, int op);
uint64_t fold_bits(uint64_t a, uint64_t b, int dt /* DFDB_I64 / DFDB_U64 / DFDB_F64 */, int op);

// the groups of ONE part of a table (a shard of a multi-GPU group, a chunk of a block stream), in order of first appearance inside the part
struct GroupPart {
  int64_t ng = 0;
  std::vector<uint8_t> key_data, key_missing, key_bytes;
  std::vector<int64_t> counts; std::vector<uint64_t> vals;
  std::vector<int64_t> first_rows;                   // (block streams only) 1-based table row of each group's first occurrence
};
// the parts merged by key in part order (= table order): one record per distinct key, a key keeps the place of its first appearance
struct GroupMerged {
  bool valid = false, with_stats = false;
  int32_t key_dtype = 0; int kind = 0, op = 0;       // kind of the value column: 0 signed, 1 unsigned, 2 float (dfdb_query::gr_kind)
  int64_t ng = 0;
  std::vector<uint8_t> key_data;                     // fixed width: ng * width bytes; String: ng int32 sizes (-1 = missing)
  std::vector<uint8_t> key_missing;                  // ng flags (1 = the key is missing)
  std::vector<uint8_t> key_bytes;                    // String keys: their bytes, concatenated
  std::vector<int64_t> counts; std::vector<uint64_t> vals;   // vals: Int64 / UInt64 / Float64 bits
  std::vector<int64_t> first_rows;                   // parallel to the keys when the parts carried them
};
struct GroupMerger {
  std::unordered_map<std::string, int64_t> slot;     // isequal image of a key -> its place in the merged result
  // `part` appended to `m` (m.key_dtype / kind / op set by the caller): counts and sums add (Int sums wrap, Float64 sums are sums of the parts' sums),
  // minimum / maximum fold with Julia's NaN and signed-zero rules
  void add(GroupMerged& m, const GroupPart& part);
};
// one part out of a query on which query_groupreduce(q, key_p, val_p, op) has just returned ng groups and kb key string bytes: the fetch (which puts the
// query's full selection back) into host vectors; with_rows also records the first occurrences' table rows (taken before the fetch, while q is narrowed)
void fetch_group_part(dfdb_query* q, int32_t key_p, int64_t ng, int64_t kb, bool with_rows, GroupPart& part);
// the merged result -> caller buffers (HOST)
void merged_fetch(const GroupMerged& m, dfdb_outcol* keys, int64_t* counts, int64_t* vals_i, double* vals_f);

// ---- out-of-core state of a query (dfdb_query::ooc)
struct OocState {
  int64_t count = -1;                                // rows of the view (-1: not counted yet)
  std::vector<int64_t> str_bytes;                    // per projection column: string bytes of the selected rows (-1: unknown)
  // dfdb_query_unique narrowed the selection to the first occurrences of column `merged_col`: merged.first_rows are the selected rows now
  bool narrowed = false;
  int merged_col = -1;
  GroupMerged merged;                                // unique: keys + rows; groupreduce: the groups until their fetch
  bool gr_pending = false;
  dfdb_sizestats read{0, 0, 0};                      // what the streams this query ran have read (dfdb_query_read_stats)
};

bool query_out_of_core(const dfdb_query* q);         // the table has files behind it and a required column is not resident
int64_t ooc_count(dfdb_query* q);
void ooc_select_indices(dfdb_query* q, int64_t* out, int64_t cap, int32_t memkind, int64_t* n);
int64_t ooc_string_bytes(dfdb_query* q, int32_t i);
void ooc_materialize(dfdb_query* q, dfdb_outcol* outs, int32_t ncols);
void ooc_aggregate(dfdb_query* q, int32_t op, int32_t i, int64_t* out_i, double* out_f);
void ooc_unique(dfdb_query* q, int32_t p);
void ooc_groupreduce(dfdb_query* q, int32_t key_p, int32_t val_p, int32_t op, int64_t* ngroups, int64_t* key_bytes);
void ooc_groupreduce_fetch(dfdb_query* q, dfdb_outcol* keys, int64_t* counts, int64_t* vals_i, double* vals_f);
void ooc_reset(dfdb_query* q);
// projection column p alone into ONE caller buffer (dfdb_table_add_from_query over a view whose columns are not resident)
void ooc_materialize_column(dfdb_query* q, int32_t p, dfdb_outcol* o);
// what a multi-GPU group asks of a shard whose columns are not resident (group.cpp): the shard streams its own block window (dfdb_table::win_first / win_last)
int64_t ooc_count_prefix(dfdb_query* q, int nstages);
int ooc_aggregate_bits(dfdb_query* q, int32_t op, int32_t i, uint64_t out[2]);
int ooc_group_part(dfdb_query* q, int32_t key_p, int32_t val_p, int32_t op, GroupPart& part);
// dfdb_query_prepare: bring the query's required columns into HBM if they fit the budget — decoded, else compressed-only, else leave them to the stream
int32_t query_prepare(dfdb_query* q);

}  // namespace dfdb
