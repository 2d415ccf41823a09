// kernels.hpp — host-callable launchers of the gfx950 kernels (definitions in k_*.hip).
// Every launcher enqueues on `s` and returns immediately; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/dfdb.h"

namespace dfdb {

enum CmpOp : int { CMP_EQ = 0, CMP_NE = 1, CMP_LT = 2, CMP_LE = 3, CMP_GT = 4, CMP_GE = 5 };

// one simple term `col OP const` of a conjunction/disjunction (K1 multi-column form).  op2 >= 0: the term is the INTERVAL
// `col OP const  &  col OP2 const2` — two AND-ed comparisons of the same column (`65 > x > 34`, test/selection.jl:53) read it once
struct ScanTerm {
  const void* col;
  int32_t dtype;     // DFDB_* base dtype of the column
  int32_t op;        // CmpOp
  uint64_t cbits;    // constant already converted to the column's own type (bit pattern)
  int32_t op2 = -1;  // CmpOp of the second comparison, -1 = none
  uint64_t cbits2 = 0;
  // pre = 1: the compared value is rem(col, m) for a signed integer column and a constant m (`a % 50 == 0`, test/selection.jl:21): computed in
  // Int64 as sign(x) * (|x| mod |m|), |x| / |m| by multiplication with a precomputed magic number (pre_magic, pre_shift); cbits / cbits2 are Int64
  // pre = 2: the compared value is col * pre_magic + pre_d in wrapping Int64 (`a * 2 + 1 > c`; a signed integer column, integer constants);
  // pre = 3: the same in Float64 — fl(fl(Float64(x) * k) + d), two roundings, never fused (pre_magic / pre_d hold the doubles' bits);
  // pre = 4: Float64(x) / k (Julia's `/`: one correctly rounded division)
  int32_t pre = 0, pre_shift = 0;
  uint64_t pre_magic = 0, pre_d = 0;
};
constexpr int kMaxTerms = 6;
struct ScanTerms {
  ScanTerm t[kMaxTerms];
  int32_t n;
  int32_t combine_or;  // 0 = AND of all terms, 1 = OR
};

// ---- K1: predicate scan -> bitmap (+ per-1024-row tile counts) ----------------------------------
// single column `x OP c`; and_existing: bitmap &= result (a predicate stage after a range stage)
void launch_scan_cmp(hipStream_t s, const void* col, int32_t dtype, int op, uint64_t cbits, uint64_t* bitmap,
                     uint32_t* tile_counts, int64_t nrows, bool and_existing, bool nt = true, void* cap = nullptr,
                     int wt_store = 3 /* bit 0: the bitmap leaves with write-through stores (ctx option "scan_wt_store"); bits 1-2: ctx option "scan_narrow" — which
                                         narrow columns take k_scan_cmp_narrow (16 bytes per lane): 1 = 1-byte (default), 2 = 1-, 2- and 4-byte, 0 = none */);
// K1's read stream alone (k_read_probe): an 8-byte column of nrows rows, nothing written (sink: one device word that is never stored to)
void launch_read_probe(hipStream_t s, const void* col, int64_t nrows, uint64_t* sink);
// extra = 1 (capture): the LAST term's 8-byte column at the finally selected rows, compacted per tile at extra_out[tile*1024 + rank];
// extra = 2 (sum): one partial sum of that column per 1024-row tile in extra_out[tile] (double, or wrapping 64-bit integer).  AND only.
// extra = 5 (capture two): the last term's column like extra = 1, and the 8-byte column of the term before it into extra_out2, same layout.
void launch_scan_terms(hipStream_t s, const ScanTerms& terms, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows,
                       bool and_existing, int extra = 0, void* extra_out = nullptr,
                       int pair = 1 /* ctx option "scan_pair": two plain 8-byte terms go to the pipelined k_scan_pair */, void* extra_out2 = nullptr);
// whether launch_scan_terms hands these terms to k_scan_pair (pair != 0): an AND of exactly two plain comparisons / intervals on Int64 / Float64 columns, fresh mask
bool scan_pair_applies(const ScanTerms& terms, bool and_existing);
// captured values (per-tile compact) -> the output column: out[prefix[tile] + k] = cap[tile*1024 + k]
void launch_compact_captured(hipStream_t s, const uint64_t* cap, const uint64_t* prefix, uint64_t* out, int64_t nrows, int64_t out_cap);
// the same with the transform of launch_gather_transform applied to the captured values (their column's dtype: Int64 / UInt64 / Float64)
void launch_compact_captured_transform(hipStream_t s, const uint64_t* cap, const uint64_t* prefix, int32_t src_dtype, const ScanTerm& tf, uint64_t* out, int64_t nrows, int64_t out_cap);

// ---- tile-count scan: u32 counts[ntiles] -> u64 prefix[ntiles+1] (prefix[ntiles] = total) -------
// scratch: >= (ceil(ntiles/4096)+1) * 8 bytes
// carry_in/carry_out (device, optional): continue the scan of the previous piece of the same column
void launch_scan_counts(hipStream_t s, const uint32_t* counts, uint64_t* prefix, int64_t ntiles, uint64_t* scratch,
                        const uint64_t* carry_in = nullptr, uint64_t* carry_out = nullptr);
size_t scan_counts_scratch_bytes(int64_t ntiles);

// ---- range stages (selection.jl:94-111 on the packed mask) --------------------------------------
struct RangeSpec {
  int32_t kind;        // 0 = a:s:b (first/last/step normalised ascending), 1 = sorted unique index list
  int64_t first, last, step;
  const int64_t* sorted; int64_t nsorted;
};
// implicit_ones: the incoming mask is all ones and rank = rank_base + local row + 1 (a leading range stage);
// otherwise rank = rank_base + prefix[tile] + position among the survivors.
void launch_range_stage(hipStream_t s, const RangeSpec& r, uint64_t* bitmap, const uint64_t* prefix, uint32_t* tile_counts,
                        int64_t nrows, int64_t rank_base, bool implicit_ones);
// ismissing(col) / !ismissing(col): the column's missing bitmap (words padded like the selection bitmap) is the mask
void launch_missing_mask(hipStream_t s, const uint64_t* missing, bool negate, bool and_existing, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows);
// all-ones mask (empty SelectionQueue): bitmap + counts
void launch_fill_ones(hipStream_t s, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows);

// ---- K2: bitmap -> ascending 1-based row numbers -------------------------------------------------
// store (ctx option "compact_store"): 0 / 1 / 2 = one ctile per wave step, plain / nontemporal / write-through 8-byte stores (round 2: 1);
// 3 / 4 = wide: two ctiles per wave, nontemporal / plain 16-byte stores (round 3 default: 3); 5 / 6 = wide with 4 KB of LDS per wave.
// grid_cap (ctx option "compact_grid_cap"; wide forms): most workgroups launched, 0 = the shipped 65 536
void launch_compact_indices(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, int64_t* out, int64_t nrows,
                            int64_t row_base, int64_t out_cap, int store = 3, int grid_cap = 0);
// ---- K3: projection gather of a fixed-width column (width 1,2,4,8 bytes) -------------------------
void launch_gather(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const void* src, void* dst, int width,
                   int64_t nrows, int64_t out_cap);
// K3 with a column transform (ScanTerm::pre = 1..4: rem / col * k + d / col / k) applied to the gathered values; dst holds 8-byte results (Int64 or Float64)
void launch_gather_transform(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const void* src, int32_t src_dtype, const ScanTerm& tf, void* dst,
                             int64_t nrows, int64_t out_cap);
// bitmap (1 = missing) of the source gathered into one byte per selected row
void launch_gather_bits(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const uint64_t* srcbits, uint8_t* dst,
                        int64_t nrows, int64_t out_cap);

// ---- synthetic generators --------------------------------------------------------
void launch_gen_i64_mod1m(hipStream_t s, int64_t* out, uint64_t seed, int64_t row_first, int64_t n);
void launch_gen_i64_iota(hipStream_t s, int64_t* out, int64_t row_first, int64_t n);
void launch_gen_f64_u2000(hipStream_t s, double* out, uint64_t seed, int64_t row_first, int64_t n);
void launch_gen_brand_sizes(hipStream_t s, int32_t* sizes, uint64_t seed, int64_t row_first, int64_t n, bool with_missing = false);
void launch_gen_brand_bytes(hipStream_t s, const int32_t* sizes, const int64_t* tile_off, uint8_t* bytes, uint64_t seed,
                            int64_t row_first, int64_t n);

// ---- strings (K4-K6) -------------------------------------------------------------------------------
// per-1024-row byte totals of max(size,0)  (first half of unsafe_remake_offsets!)
void launch_str_tile_bytes(hipStream_t s, const int32_t* sizes, uint32_t* tile_bytes, int64_t nrows, uint32_t* max_tile_bytes = nullptr);   // (max: one zeroed device word)
// K5: s OP "const" (EQ / NE / STARTSWITH / ENDSWITH) -> bitmap + counts.  mode: 0 EQ, 1 NE, 2 STARTSWITH, 3 ENDSWITH
// pat_host: the pattern in host memory (<= 64 bytes travel as kernel arguments); pat_dev: device copy for longer ones
struct StrCapture { int32_t* sizes; uint8_t* bytes; uint32_t* tile_bytes; };   // K5 CAP outputs (see k_str_match_short)
void launch_str_match(hipStream_t s, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const uint8_t* pat_host,
                      const uint8_t* pat_dev, int32_t patlen, int mode, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows,
                      bool and_existing, const StrCapture* cap = nullptr, uint32_t max_tile_bytes = 0);   // max_tile_bytes: Column::max_tile_bytes (0 = not known: probes straight from memory)
void launch_str_compact_captured(hipStream_t s, const StrCapture& cap, const uint64_t* prefix, const int64_t* tile_off, const uint64_t* out_tile_off,
                                 int32_t* out_sizes, uint8_t* out_bytes, int64_t nrows, int64_t out_rows, int64_t out_bytes_cap);
// K6: selected sizes -> out sizes (+ per-ctile selected byte totals); then bytes
void launch_str_gather_sizes(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const int32_t* sizes, int32_t* out_sizes,
                             uint32_t* sel_tile_bytes, int64_t nrows, int64_t out_cap);
void launch_str_gather_bytes(hipStream_t s, const uint64_t* bitmap, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes,
                             const uint64_t* out_tile_off, uint8_t* out_bytes, int64_t nrows, int64_t out_bytes_cap);

// n rows that all hold the same string: sizes[i] = plen, bytes = the pattern (device memory) n times
void launch_fill_const_strings(hipStream_t s, int32_t* out_sizes, uint8_t* out_bytes, int64_t n, const uint8_t* pat_dev, int32_t plen);

// ---- K9: dictionary codes of a low-cardinality String column (k_dict.hip) ------------------------------
struct DictSlot { uint64_t key8; uint32_t len, code, off, pad; };      // open addressing; len = 0xffffffff: empty.  key8 = the first min(len, 8) bytes
struct DictMiss {                                                     // rows whose string is not in the table yet report here
  unsigned long long* count; unsigned long long* bytes_used;          // rows reported in all; bytes of `arena` handed out
  uint32_t* rec_off; int32_t* rec_len; uint8_t* arena;                // per reported row (the first max_records): its bytes in the arena (len -1: no room, -2: too long)
  int64_t max_records, bytes_cap; int32_t max_len;
};
uint64_t dict_hash_host(uint64_t key8, uint32_t len);
void launch_dict_encode(hipStream_t s, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const DictSlot* slots, uint32_t nslots,
                        const uint8_t* dict_bytes, uint16_t* codes, int64_t nrows, const DictMiss& miss);
// lut: one bit per code, 1 = the row is selected; lut_words = ceil(entries / 32) <= 2048
void launch_dict_scan(hipStream_t s, const uint16_t* codes, const uint32_t* lut, int32_t lut_words, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows,
                      bool and_existing);
// the projection of a dictionary column over n selected rows whose codes K3 has compacted: sizes + per-1024-output-row byte totals, then the bytes
void launch_dict_expand_sizes(hipStream_t s, const uint16_t* codes, int64_t n, const int32_t* dict_len, int32_t* out_sizes, uint32_t* out_tile_bytes);
void launch_dict_expand_bytes(hipStream_t s, const uint16_t* codes, int64_t n, const int32_t* dict_len, const uint32_t* dict_off, const uint8_t* dict_bytes,
                              const uint64_t* out_tile_off, uint8_t* out_bytes, int64_t out_bytes_cap);

// ---- reductions ------------------------------------------------------------------------------------
// sum/min/max of a fixed-width column over the selected rows -> partials then final (2 launches)
void launch_reduce(hipStream_t s, const uint64_t* bitmap, const void* col, int32_t dtype, int op, int64_t nrows, void* partials,
                   void* result /* 16 bytes: i64/u64 or f64 result + count */);
size_t reduce_scratch_bytes();

// ---- unique(col) as a selection of first occurrences (k_unique.hip: hash table of {key, smallest row}; k_unique_dense.hip: integer keys of a small range)
struct UniqueEntry { uint64_t key, row; };
void launch_unique_insert(hipStream_t s, const uint64_t* bitmap, const void* col, int dtype, const uint64_t* missing, int64_t row0, int64_t row1,
                          UniqueEntry* ent, uint64_t mask, uint64_t* aux);
void launch_unique_mark(hipStream_t s, uint64_t* bitmap, uint32_t* tile_counts, const void* col, int dtype, const uint64_t* missing, int64_t nrows,
                        const UniqueEntry* ent, uint64_t mask, const uint64_t* aux);
void launch_unique_migrate(hipStream_t s, const UniqueEntry* from, const uint64_t* from_off, const uint32_t* from_len, uint64_t from_cap, UniqueEntry* ent,
                           uint64_t* rep_off, uint32_t* rep_len, uint64_t mask, uint64_t* aux);
void launch_unique_scatter(hipStream_t s, const UniqueEntry* ent, uint64_t cap, const uint64_t* aux, uint64_t* bitmap, uint32_t* tile_counts);
// pass 0: insert the tiles [tile0, tile1), 1: verify, 2: mark
void launch_unique_str(hipStream_t s, int pass, uint64_t* bitmap, uint32_t* tile_counts, const int32_t* sizes, const int64_t* tile_off,
                       const uint8_t* bytes, int64_t nrows, int64_t tile0, int64_t tile1, UniqueEntry* ent, uint64_t* rep_off, uint32_t* rep_len, uint64_t mask,
                       uint64_t* aux, uint64_t salt);
// dense form.  aux words: 1 = smallest missing row, 5 = a key outside [lo, lo + range) was met, 6 = distinct keys, 7 = keys whose first row is known, 8 / 9 = min / max image
// radix-partitioned form of the hash-table unique (k_radix.hip): partition into a pool of pages -> one LDS table per partition.  false: the launch is
// not possible (LDS attribute refused, too many partition bits): the caller stays with the hash table.  The pool: `front` [2^kbits x radix_share()] running
// positions of the streams (zero before the pass), `pt` [2^kbits x radix_share()][maxv] the streams' pages (all ones before the pass), `next_page` the pool's
// counter (zero), `dump_page` the page nobody owns (radix_pool_pages() - 1); the records' buffer holds radix_pool_record_bytes().
struct RadixPool { uint32_t* front; uint32_t* pt; uint32_t* next_page; uint32_t maxv; uint32_t dump_page;
                   uint64_t* hot /* [hot_cap x 3]: the hot keys' list {key, value, rows << 32 | first row}: chunks x radix_hot_slots() entries */; uint32_t* hot_n /* zero before */; uint32_t hot_cap; };
int64_t radix_rows_per_chunk(int64_t nrows, int chunks);
int radix_share();
int64_t radix_pool_pages(int64_t cnt, int kbits);
int64_t radix_pool_record_bytes(int64_t cnt, int kbits, bool with_values);
int radix_group_slots();
uint32_t radix_pool_maxv(int64_t cnt, int kbits);
bool launch_radix_sample(hipStream_t s, const uint64_t* sel, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int kbits, int chunks, int step,
                         uint32_t* counts /* [2^kbits], zero before */);
// groupreduce by radix: the records carry the row's 8-byte value (valcol; null: count only); gop 0 count only, 1 wrapping integer sum, 2 double sum, 3 min, 4 max
// (of order images: vkind 0 signed, 1 unsigned, 2 double); results [<= groups] {first row, rows, value} + their count nres; gspec {rows, value} of the unstorable key
// and of the missing key (their first rows: aux[0], aux[1])
struct RadixGroup { const void* valcol; int valdt; int gop; int vkind; void* results; uint32_t* nres; uint64_t* gspec /* [4]: the unstorable key's, the missing key's */; };
int radix_hot_slots();
bool launch_radix_partition(hipStream_t s, const uint64_t* sel, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int kbits, int chunks,
                            const RadixPool& pool, uint32_t* recs_out /* 12 bytes per record: key image, row; with a group: 20, + the value */, uint64_t* aux,
                            const RadixGroup* group = nullptr, bool hot = false /* the kernels that keep hot keys out of the records (values narrower than 8 bytes: always) */);
bool launch_radix_group(hipStream_t s, const uint32_t* recs, const RadixPool& pool, int kbits, bool mark, uint64_t* bitmap, uint32_t* tile_counts, uint64_t* aux,
                        const RadixGroup& group, int cus);
void launch_radix_group_finish(hipStream_t s, const RadixGroup& group, const uint64_t* ubits, const uint64_t* uprefix, const uint64_t* aux, uint64_t* cnt, uint64_t* val);
bool launch_radix_unique(hipStream_t s, const uint32_t* recs, const RadixPool& pool, int kbits, uint64_t* bitmap, uint32_t* tile_counts, uint64_t* aux, int cus);
int64_t unique_dense_max_range();
bool unique_dense_dtype(int dtype);
void launch_dense_minmax(hipStream_t s, const uint64_t* bitmap, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int64_t tile_step, uint64_t* aux);
void launch_dense_presence(hipStream_t s, const uint64_t* bitmap, const void* col, int dtype, const uint64_t* missing, int64_t nrows, uint64_t lo, uint32_t range,
                           uint32_t* present, uint64_t* aux);
void launch_dense_first(hipStream_t s, const uint64_t* bitmap, const void* col, int dtype, const uint64_t* missing, int64_t row0, int64_t row1, uint64_t lo,
                        uint32_t range, uint64_t distinct, uint64_t* first, uint64_t* aux);
void launch_dense_scatter(hipStream_t s, const uint64_t* first, uint32_t range, const uint64_t* aux, uint64_t* bitmap, uint32_t* tile_counts);
void launch_dense_group_ids(hipStream_t s, uint64_t* first, uint32_t range, uint64_t* aux, const uint64_t* ubits, const uint64_t* uprefix);
int launch_group_accumulate_dense(hipStream_t s, const uint64_t* sel, const void* keycol, int keydt, const uint64_t* missing, const void* valcol, int valdt, int op,
                                   int64_t nrows, uint64_t lo, uint32_t range, uint64_t span_lo, uint64_t span_hi, const uint64_t* gids, const uint64_t* aux, uint64_t* cnt, uint64_t* val, int64_t ngroups, uint64_t val_init, uint64_t* unknown_flag = nullptr);

// ---- K7: LZ4 block decode, K8: missing bitmaps, block bodies ---------------------------------------
// (dst: the decoders may READ up to 32 bytes past the end of the last block's output — far-match sources are fetched 24 bytes at a time — so the
//  destination needs that much slack behind it; table.cpp's column arrays and body arenas carry 64-256)
struct Lz4Block {      // one (column, block) unit of work
  int64_t src_off;     // offset of the compressed bytes inside the staged image
  int32_t src_len;     // compressed bytes
  int32_t dst_len;     // expected uncompressed bytes (origin)
  int64_t dst_off;     // where the decoded body goes inside the body arena
};
void launch_lz4_decode(hipStream_t s, const uint8_t* src, uint8_t* dst, const Lz4Block* blocks, int32_t nblocks, int32_t* status,
                       int pipe = -1 /* -1: two waves per block when there are fewer blocks than wave slots, 0: never, 1: always (ctx option "lz4_pipeline") */,
                       uint32_t* index = nullptr, int index_mode = 0 /* 1: record where the sequences start (one bit per byte of src, zeroed by the caller), 2: decode with it */);
bool lz4_decode_takes_index(int32_t nblocks, int pipe);
// K7 fused with the first predicate of a scan (decode -> scan fusion): 8-byte columns whose blocks start on 1024-row tiles
// op2 >= 0: the interval `col OP c & col OP2 c2`; and_existing: the mask already holds survivors — words are AND-ed into it and a block none of whose tiles
// kept a row is not decoded at all; ticket: the history-ring form's block counter (set by its launcher)
struct LzScan { uint64_t* bitmap; uint32_t* counts; uint64_t cbits; int32_t dtype; int32_t op; int32_t op2 = -1; int32_t and_existing = 0; uint64_t cbits2 = 0; uint32_t* ticket = nullptr; };
void launch_lz4_decode_scan(hipStream_t s, const uint8_t* src, uint8_t* dst, const Lz4Block* blocks, int32_t nblocks, int32_t* status, const LzScan& sc,
                            uint32_t* index = nullptr, int index_mode = 0);

// K7 without a decoded column (round 5): what leaves the decoder's LDS ring goes to a 64-KB history ring per resident wave inside `scratch`
// (lz4_hist_scratch_bytes(waves) bytes: a ticket word, then the rings) — all an LZ4 match can reach — never to a column array.  sc != nullptr: the predicate
// term is evaluated on the way (bitmap + tile counts are the only output; 8-byte columns whose blocks start on 1024-row tiles); sc == nullptr: a validating
// decode (statuses, and the sequence-start index when index_mode = 1).  waves: workgroups launched = rings used (0: lz4_hist_default_waves(device CUs)).
size_t lz4_hist_scratch_bytes(int waves);
int lz4_hist_default_waves(int compute_units);
void launch_lz4_decode_hist(hipStream_t s, const uint8_t* src, uint8_t* scratch, int waves, const Lz4Block* blocks, int32_t nblocks, int32_t* status, const LzScan* sc,
                            uint32_t* index = nullptr, int index_mode = 0);

}  // namespace dfdb
