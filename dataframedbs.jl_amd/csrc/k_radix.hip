// k_radix.hip — unique(col) by RADIX PARTITION (round 6): the general (hash-table) form of k_unique.hip for many distinct values.
//
// Replaces Base.unique driven by Base.iterate(::DFColumn) (src/tables/column.jl:102-126; docs/src/index.md:479-487), like k_unique.hip.  The open-addressing
// table of {key, smallest row} in HBM costs one random 128-byte line per selected row once it outgrows the L2s (1e6 distinct Int64 / Float64 values: 32 MB of
// table, 1e9 probes, 19.8 ms per 1e9 rows = 0.05 of the HBM roofline on the 8 bytes per row it needs).  Here every byte moves in STREAMS:
//   partition  ONE pass over the key column: a workgroup sorts 8192 rows at a time by partition (the top k bits of a 32-bit hash of the key image) in LDS,
//              reserves each partition's run behind the running position of its STREAM — (partition, share), a share being the chunks (= workgroups) whose
//              number is equal mod 8: one XCD's, as the dispatcher deals them out — with one global atomicAdd per tile and partition, and writes it: 12-byte
//              records {key image, row}, every run one contiguous piece.  The workgroups of an XCD append to the SAME 512 places, so their runs follow each
//              other and complete their 128-byte lines in that XCD's L2.  A stream's records live in PAGES of 8192 taken from a pool as the pass goes (the
//              thread whose run holds a page's first record takes it and publishes it in the stream's page table): nothing has to be counted first
//   unique     one workgroup per partition: the records of its streams' pages go through a table that lives in LDS (8192 slots: 64-bit compare-and-swap
//              claims a slot, a 32-bit atomic minimum keeps the smallest row), and the occupied slots leave one bit each in the bitmap of first occurrences
//              (+ per-tile counts)
// 8 + 12 + 12 bytes per selected row, all streams.  The order of a partition's records depends on timing; the result (smallest row per key) does not.
// A key that a large part of the rows hold (one partition = one workgroup's work) is kept out of the records: the partition pass gives it one of a workgroup's
// 256 LDS slots (HOT KEYS, below) and the table pass gets one list entry per workgroup and key instead.
// A partition that holds more distinct keys than its table takes raises a flag and the host runs the hash-table form instead (query.cpp: unique_hashed); keys
// are isequal images (one NaN, -0.0 apart from 0.0), a missing key and the one image that cannot be stored (all ones) are kept aside in aux[1] / aux[0]
// exactly as k_unique_insert keeps them.  Measured history: profiles/r6_unique_radix.txt; what the instruction and store choices rest on: tools/ubench/.
#include <cstdlib>
#include "device_utils.hpp"
#include "kernels.hpp"
#include "../../include/dfdb_ir.h"

namespace dfdb {

namespace {
constexpr int kRBlock = 1024;                 // threads per workgroup of all three passes
constexpr int kRTile = 8192;                  // rows sorted at a time by the partition pass (8 per thread)
constexpr int kRSlots = 8192;                 // slots of a partition's table in LDS
constexpr uint64_t kREmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int kRPage = 8192;                   // records per page of the record pool (a tile's run of a partition — at most 8192 records — lies in at most two)
constexpr int kRShare = 8;                    // the workgroups whose number is equal mod 8 — one XCD's, as the dispatcher deals them out — share their running positions
// 32 bits of hash per key, three 32-bit multiplies (k_unique.hip lds_slot_of's mix; splitmix64's two 64-bit multiplies are eight quarter-rate instructions each and
// the partition pass hashed every row twice: 3.4e9 vector instructions per 1e9 rows).  The partition is its TOP bits, a table slot its low bits.
__device__ __forceinline__ uint32_t rhash(uint64_t key) {
  uint32_t h = ((uint32_t)key ^ (uint32_t)(key >> 32) * 0x85EBCA77u) * 0x9E3779B1u;
  h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13;
  return h;
}

__device__ __forceinline__ uint64_t rkey_fixed(const void* col, int dtype, int64_t row) {      // = k_unique.hip key_fixed: the isequal image
  switch (dtype) {
    case DFDB_I8:  return (uint64_t)(int64_t)((const int8_t*)col)[row];
    case DFDB_I16: return (uint64_t)(int64_t)((const int16_t*)col)[row];
    case DFDB_I32: return (uint64_t)(int64_t)((const int32_t*)col)[row];
    case DFDB_U8: case DFDB_BOOL: return ((const uint8_t*)col)[row];
    case DFDB_U16: return ((const uint16_t*)col)[row];
    case DFDB_U32: return ((const uint32_t*)col)[row];
    case DFDB_F32: { const float f = ((const float*)col)[row]; return f != f ? 0x7fc00000ull : (uint64_t)__float_as_uint(f); }
    case DFDB_F64: { const double d = ((const double*)col)[row]; return d != d ? 0x7ff8000000000000ull : (uint64_t)__double_as_longlong(d); }
    default: return ((const uint64_t*)col)[row];
  }
}
struct __attribute__((packed, aligned(4))) Rec12 { uint32_t lo, hi, row; };
struct __attribute__((packed, aligned(4))) Rec20 { uint32_t lo, hi, row, vlo, vhi; };
enum { kKindRaw8 = 0, kKindF64 = 1, kKindAny = 2 };       // what a key load is: 8 raw bytes (Int64 / UInt64), 8 bytes + isequal's one NaN (Float64), anything narrower (rkey_fixed)
__device__ __forceinline__ uint64_t wave_uniform(uint64_t v) {     // a value every lane of the wave holds, into scalar registers
  return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
}
__device__ __forceinline__ void aux_min(uint64_t* a, uint64_t row) {
  if (__atomic_load_n(a, __ATOMIC_RELAXED) > row) atomicMin((unsigned long long*)a, (unsigned long long)row);
}

// ---- one 8192-row tile, as the sample and the partition pass read it -----------------------------------------------------------------------------------------
// Wave w of the 16 takes the tile's rows [512 w, 512 w + 512): EIGHT WHOLE WORDS of the selection (and of the missing bits) — wave-uniform, scalar loads —, and
// lane l's j-th row is 512 w + 64 j + l: bit l of word j.  Who takes part is decided on the scalar unit (word & ~missing word, handed to the lanes as an execution
// mask: __builtin_amdgcn_inverse_ballot_w64); the eight key loads of a lane are one address and eight immediate offsets.  The first form of these passes did the
// same per row on the vector unit — a 64-bit row number, its clamp, the word's address, a 64-bit shift, per row — and ran at 46 (hist) / 215 (partition) vector
// instructions per 64 rows: at 2.5-4.3 cycles per wave-instruction (tools/ubench/valu_rate.hip) that, not HBM, was what both passes waited for.
// FULL = every row of the tile exists (base + 8192 <= nrows); the one partial tile of a table clamps its word and row numbers instead.
// A row takes part when it is selected, not missing and its image can be stored; the special rows go to aux in the partition pass: a missing key's
// smallest row to aux[1] here, the unstorable image's (all ones) to aux[0] where the keys are looked at.
template <int KIND, bool FULL>
__device__ __forceinline__ uint64_t tile_load(uint64_t (&key)[8], uint64_t (&in)[8], const uint64_t* __restrict__ sel, const void* __restrict__ col, int dtype,
                                              const uint64_t* __restrict__ missing, int64_t base, int64_t nrows, int wv, int lane) {
  const int64_t w0 = (base >> 6) + wv * 8;
  uint64_t mw[8];
  if (FULL) {
#pragma unroll
    for (int j = 0; j < 8; j++) in[j] = sel[w0 + j];
    if (missing) {
#pragma unroll
      for (int j = 0; j < 8; j++) mw[j] = missing[w0 + j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) mw[j] = 0;
    }
  } else {
    const int64_t wl = (nrows - 1) >> 6;                        // the last word that holds a row
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int64_t w = w0 + j, wc = w < wl ? w : wl;
      uint64_t sw = sel[wc];
      mw[j] = missing ? missing[wc] : 0ull;
      if (w > wl) sw = 0; else if (w == wl && (nrows & 63)) sw &= (1ull << (nrows & 63)) - 1ull;
      in[j] = sw;
    }
  }
  uint64_t selected_missing = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint64_t sw = wave_uniform(in[j]), m = wave_uniform(mw[j]);
    selected_missing |= sw & m;
    in[j] = sw & ~m;
  }
  // the keys of the rows that take part only: under a selective predicate most 128-byte lines of the column hold no such row and are never fetched (the words
  // are needed first — but the partition pass asks for a tile's keys a whole tile ahead)
  const uint32_t lo = (uint32_t)(wv * 512 + lane);              // the lane's first row of the tile
  const uint32_t last = FULL ? 8191u : (uint32_t)(nrows - 1 - base < 8191 ? nrows - 1 - base : 8191);
  if (KIND != kKindAny) {
    const uint64_t* kp = (const uint64_t*)col + base;           // (wave-uniform)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t o = lo + (uint32_t)(j * 64);
      key[j] = 0;
      if (__builtin_amdgcn_inverse_ballot_w64(in[j])) key[j] = __builtin_nontemporal_load(kp + (FULL || o < last ? o : last));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t o = lo + (uint32_t)(j * 64);
      key[j] = 0;
      if (__builtin_amdgcn_inverse_ballot_w64(in[j])) key[j] = rkey_fixed(col, dtype, base + (FULL || o < last ? o : last));
    }
  }
  return selected_missing;                                      // (wave-uniform: nonzero = the partition pass looks for the tile's first missing row, tile_first_missing)
}
// the rare side of the partition pass: the smallest selected row of the wave's 512 whose key is missing, to aux[1]
__device__ __forceinline__ void tile_first_missing(const uint64_t* __restrict__ sel, const uint64_t* __restrict__ missing, int64_t base, int64_t nrows, int wv, int lane, uint64_t* aux) {
  const int64_t w0 = (base >> 6) + wv * 8, wl = (nrows - 1) >> 6;
  for (int j = 0; j < 8; j++) {
    const int64_t w = w0 + j;
    if (w > wl) break;
    uint64_t m = wave_uniform(sel[w] & missing[w]);
    if (w == wl && (nrows & 63)) m &= (1ull << (nrows & 63)) - 1ull;
    if (m) { if (lane == 0) aux_min(&aux[1], (uint64_t)(w * 64 + __builtin_ctzll(m))); break; }
  }
}
// the lanes of the wave whose key can be stored, as a mask (all ones is the table's "empty"); Float64 keys become their isequal image first: one NaN — and all
// ones is a NaN
template <int KIND> __device__ __forceinline__ uint64_t keys_storable(uint64_t& k) {
  if (KIND == kKindF64) { if ((k << 1) > 0xFFE0000000000000ull) k = 0x7ff8000000000000ull; return ~0ull; }
  return __ballot(k != kREmpty);                                // (called with every lane active: the compare's result as it stands)
}

// ---- the SAMPLE: counts[p] = selected rows whose key falls into partition p among every `step`-th tile (query.cpp looks at the largest partition before anything
// is written: a skewed column gets the partition kernels that look for hot keys).  The partition pass needs no counts: it takes pages from a pool as it goes.
template <int KIND>
__global__ __launch_bounds__(kRBlock) void k_radix_hist(const uint64_t* __restrict__ sel, const void* __restrict__ col, int dtype, const uint64_t* __restrict__ missing,
                                                        int64_t nrows, int64_t rows_per_chunk, int kbits, uint32_t* __restrict__ counts, int step) {
  extern __shared__ uint32_t hist_sh[];
  const int P = 1 << kbits, c = (int)blockIdx.x;
  for (int p = threadIdx.x; p < P; p += kRBlock) hist_sh[p] = 0;
  __syncthreads();
  const int lane = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), sh = 32 - kbits;
  const int64_t r0 = (int64_t)c * rows_per_chunk, r1 = r0 + rows_per_chunk < nrows ? r0 + rows_per_chunk : nrows;
  for (int64_t base = r0 + (int64_t)(c % step) * kRTile; base < r1; base += (int64_t)kRTile * step) {
    uint64_t key[8], in[8];
    if (base + kRTile <= nrows) tile_load<KIND, true>(key, in, sel, col, dtype, missing, base, nrows, wv, lane);
    else tile_load<KIND, false>(key, in, sel, col, dtype, missing, base, nrows, wv, lane);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t ok = keys_storable<KIND>(key[j]);
      if (__builtin_amdgcn_inverse_ballot_w64(in[j] & ok)) atomicAdd(&hist_sh[rhash(key[j]) >> sh], 1u);
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += kRBlock) { const uint32_t h = hist_sh[p]; if (h) atomicAdd(&counts[p], h); }
}

// ---- pass 1: the records of chunk c, sorted by partition 8192 rows at a time, to their places
// Per tile: (1) every row's partition and its rank among the tile's rows of that partition (an LDS atomic that returns a value), the NEXT tile's loads issued;
// (2) thread p < P scans the tile's counts: where partition p's run starts in the sorted tile (lstart), and reserves the run behind the running position of its
// STREAM (partition p, the workgroup's share) with one global atomicAdd; (3) the records into LDS, sorted by partition; thread p turns the reserved position
// into places in the record pool; (4) slot s to its place — each partition's run a contiguous store.  Four barriers per tile.
// The pool (no counting pass: the first build read the column once more, 1.3 ms, only to size the partitions): a stream's records are numbered 0, 1, 2 … by the
// reservations, every 8192 of them are a PAGE, and pt[stream][k] is where the stream's k-th page lies in the pool.  The thread whose run holds a page's first
// record takes the page (an atomicAdd on the pool's counter) and publishes it; a thread whose run lies in a page somebody else's run began waits for that
// entry.  All the taking is done before any of the waiting (the thread that has to publish never waits for anybody first), so nothing can wait in a circle.
// A stream that needs more pages than the table has columns (a value that a large part of the column holds) raises the abort flag: the hash table answers.
// (Tried and dropped, profiles/r6_unique_radix.txt: ranks by ballots instead of LDS atomics that return a value — slower at 9-10 partition bits; whole 16-record
// units at 16-aligned positions with the remainders carried over in LDS, 4096-row tiles — every store a full line, and the pass took 11.7 ms instead of 7.3;
// 512-thread workgroups sorting 4096 rows, two per CU — shorter runs store slower than the overlap gains.)
constexpr int part_lds_words(int block) { return 6 * 1024 + 32 + 8 * block; }      // hist2, lstart (1024 each), place (1024 x 16 bytes), wave sums, srow — in 4-byte words; skey follows
// HASVAL (groupreduce by radix, below): a record carries the row's 8-byte VALUE too — 20 bytes {key image, row, value}; the value is fetched from its column where
// the record is written (the tile's values are read in row order behind the sort and laid down in LDS where the sorted keys were), and the rows whose key cannot be
// stored are counted and reduced
// here, in a workgroup-wide accumulator that is flushed to gspec {count, value} once per workgroup (GOP: 0 count only, 1 wrapping integer sum, 2 double sum, 3 min, 4 max
// of order images; vkind: how a value's image is made — k_unique.hip's order_image).
struct RadixVals { const void* col; int vdt; uint64_t* gspec; int gop; int vkind; };
// HOT KEYS.  A key that a large part of the rows hold would make one partition — one workgroup's work — of all those rows, every one an atomic on the same LDS
// word (and through the form this replaces, 3e8 global atomics on ONE address: 3.6 s per 1e9 rows).  A workgroup of the partition pass keeps kHotSlots keys in LDS
// with their own accumulators: a key that turns up three times among the 64 rows of a selection word is given a slot (if its slot is free), and from the next tile
// on a row whose key has a slot is counted and reduced THERE and never becomes a record.  When the workgroup ends its slots go to a list {key, value, rows, first
// row}; the table pass merges the entries of its partition into its table like records that stand for many rows.
constexpr int kHotSlots = 256;
__device__ __forceinline__ uint32_t hot_slot(uint64_t key) { return (((uint32_t)key ^ (uint32_t)(key >> 32) * 0x85EBCA77u) * 0x9E3779B1u) >> 24; }
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32;
}
// a value as the 64 bits the accumulators work on (= k_unique.hip's value_bits: integers widened, Float32 as the double it converts to)
__device__ __forceinline__ uint64_t rvalue_bits(const void* col, int dtype, int64_t row) {
  switch (dtype) {
    case DFDB_I8:  return (uint64_t)(int64_t)((const int8_t*)col)[row];
    case DFDB_I16: return (uint64_t)(int64_t)((const int16_t*)col)[row];
    case DFDB_I32: return (uint64_t)(int64_t)((const int32_t*)col)[row];
    case DFDB_U8: case DFDB_BOOL: return ((const uint8_t*)col)[row];
    case DFDB_U16: return ((const uint16_t*)col)[row];
    case DFDB_U32: return ((const uint32_t*)col)[row];
    case DFDB_F32: return (uint64_t)__double_as_longlong((double)((const float*)col)[row]);
    default: return ((const uint64_t*)col)[row];
  }
}
template <bool V8> __device__ __forceinline__ uint64_t rvalue_of(const void* col, int dtype, int64_t row) { return V8 ? ((const uint64_t*)col)[row] : rvalue_bits(col, dtype, row); }
__device__ __forceinline__ uint64_t order_image(uint64_t bits, int kind, bool is_min) {         // (= k_unique.hip's: unsigned compare; a NaN wins either reduction)
  if (kind == 1) return bits;
  if (kind == 0) return bits ^ (1ull << 63);
  const double d = __longlong_as_double((long long)bits);
  if (d != d) return is_min ? 0ull : ~0ull;
  return (bits >> 63) ? ~bits : (bits | (1ull << 63));
}
__device__ __forceinline__ void acc_value(uint64_t* slot, uint64_t v, int gop, int vkind) {      // (LDS or global: one atomic)
  if (gop == 1) atomicAdd((unsigned long long*)slot, (unsigned long long)v);
  else if (gop == 2) atomicAdd((double*)slot, __longlong_as_double((long long)v));
  else if (gop == 3) atomicMin((unsigned long long*)slot, (unsigned long long)order_image(v, vkind, true));
  else if (gop == 4) atomicMax((unsigned long long*)slot, (unsigned long long)order_image(v, vkind, false));
}
// V8: the value column is eight bytes wide (loaded as it is; the narrow types' switch — seventeen copies of it — lives in the !V8 kernels only)
// HOT: the hot keys' slots are compiled in (groupreduce always; unique only for a column its sample calls skewed: the code beside the sort costs the pass 1.1 of its 4.6 ms)
template <int KIND, int BLOCK, bool HASVAL, bool V8, bool HOT>
__global__ __launch_bounds__(BLOCK) void k_radix_partition(const uint64_t* __restrict__ sel, const void* __restrict__ col, int dtype, const uint64_t* __restrict__ missing,
                                                             int64_t nrows, int64_t rows_per_chunk, int kbits, RadixPool pool,
                                                             uint32_t* __restrict__ recs_out, uint64_t* aux, RadixVals vals, int xp) {
  __shared__ uint64_t spec_sh[4];                               // HASVAL: {rows, reduced value} of the rows whose key is the unstorable image, then of the rows whose key is missing: this workgroup's
  __shared__ uint64_t hc_key[HOT ? kHotSlots : 1], hc_val[HOT && HASVAL ? kHotSlots : 1];      // the hot keys' slots: key, (HASVAL) reduced value, ...
  __shared__ uint32_t hc_cnt[HOT ? kHotSlots : 1], hc_row[HOT ? kHotSlots : 1], hc_any;         // ... rows, smallest row; whether any slot is taken
  extern __shared__ uint64_t part_sh[];
  uint32_t* hist2 = (uint32_t*)part_sh;                         // [1024] this tile's records per partition
  uint32_t* lstart = hist2 + 1024;                              // [1024] their first slot in the sorted tile
  uint4* place = (uint4*)(lstart + 1024);                       // [1024] {d0, d1, slim}: sorted slot s of the partition goes to pool record s + (s < slim ? d0 : d1)
  uint32_t* wsum = (uint32_t*)(place + 1024);                   // [16]   scan scratch: one total per wave
  constexpr int TILE = 8 * BLOCK;                               // rows sorted at a time: 8192 (one workgroup per CU) or 4096 (two)
  uint32_t* srow = wsum + 32;                                   // [TILE] partition << 13 | the row's offset inside the tile
  uint64_t* skey = (uint64_t*)(srow + TILE);                    // [TILE]
  const int P = 1 << kbits, c = (int)blockIdx.x, sh = 32 - kbits;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)c * rows_per_chunk, r1 = r0 + rows_per_chunk < nrows ? r0 + rows_per_chunk : nrows;
  if (r0 >= r1) return;
  if (tid == 0) { hc_any = 0; if (HASVAL) { spec_sh[0] = spec_sh[2] = 0; spec_sh[1] = spec_sh[3] = vals.gop == 3 ? ~0ull : 0ull; } }
  if (HOT && tid < kHotSlots) { hc_key[tid] = kREmpty; hc_cnt[tid] = 0; hc_row[tid] = 0xFFFFFFFFu; if (HASVAL) hc_val[tid] = vals.gop == 3 ? ~0ull : 0ull; }
  const int fx = tid * kRShare + (c & (kRShare - 1));          // thread p < P: the stream (partition p, this workgroup's share)
  for (int p = tid; p < 1024; p += BLOCK) hist2[p] = 0;
  __syncthreads();
  uint64_t nkey[8], nin[8];                                     // the NEXT tile: loaded while this one is sorted and written
  uint32_t pk = 0xFFFFFFFFu, pb = 0;                            // thread p < P: the page of its stream its last run ended in, and where that page lies
  bool hot_on = false;                                          // some hot key has a slot (wave-uniform)
  uint64_t nmiss;                                               // (wave-uniform) selected rows of the next tile whose key is missing
  if (r0 + TILE <= nrows) nmiss = tile_load<KIND, true>(nkey, nin, sel, col, dtype, missing, r0, nrows, wv, lane);
  else nmiss = tile_load<KIND, false>(nkey, nin, sel, col, dtype, missing, r0, nrows, wv, lane);
#pragma unroll
  for (int j = 0; j < 8; j++) asm volatile("" : "+v"(nkey[j]));      // (arrived before the loop is entered — see step 4: no wait for them may sit at the loop's top)
  for (int64_t base = r0; base < r1; base += TILE) {
    uint64_t key[8]; uint32_t pr[8];                            // pr: partition << 13 | rank among the tile's records of that partition; ~0 = the row takes no part
    uint64_t unstorable = 0, take[8];                           // take: the rows that become records (wave-uniform masks)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      key[j] = nkey[j]; pr[j] = ~0u;
      const uint64_t ok = keys_storable<KIND>(key[j]);
      unstorable |= nin[j] & ~ok;
      take[j] = nin[j] & ok;
    }
    if (HOT) {                                                  // hot keys (see RadixVals)
      if (!hot_on) hot_on = __builtin_amdgcn_readfirstlane((int)hc_any) != 0;      // (as the tile begins: a slot taken during it counts from the next tile on)
      if (hot_on) {
        uint64_t ck[8];
#pragma unroll
        for (int j = 0; j < 8; j++) ck[j] = hc_key[hot_slot(key[j])];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint64_t hit = take[j] & __ballot(ck[j] == key[j]);
          if (!hit) continue;
          if ((hit >> lane) & 1ull) {
            const uint32_t sl = hot_slot(key[j]);
            const int64_t row = base + wv * 512 + j * 64 + lane;
            atomicAdd(&hc_cnt[sl], 1u);
            if (hc_row[sl] > (uint32_t)row) atomicMin(&hc_row[sl], (uint32_t)row);
            if (HASVAL && vals.col && vals.gop) acc_value(&hc_val[sl], rvalue_of<V8>(vals.col, vals.vdt, row), vals.gop, vals.vkind);
          }
          take[j] &= ~hit;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j += 4) {                          // a key that holds three of a word's 64 rows gets a slot (two of a wave's eight words are looked at)
        if (!take[j]) continue;
        const uint64_t fk = readlane64(key[j], __builtin_ctzll(take[j]));
        if (__builtin_popcountll(take[j] & __ballot(key[j] == fk)) < 3) continue;
        if (lane == 0) {
          const uint32_t sl = hot_slot(fk);
          if (hc_key[sl] == kREmpty) { atomicCAS((unsigned long long*)&hc_key[sl], (unsigned long long)kREmpty, (unsigned long long)fk); hc_any = 1; }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (__builtin_amdgcn_inverse_ballot_w64(take[j])) { const uint32_t p = rhash(key[j]) >> sh; pr[j] = p << 13 | atomicAdd(&hist2[p], 1u); }
    if (nmiss) {                                                // (rare: the two keys kept aside — a missing key's first row, aux[1] — and, HASVAL, its rows and their values; ...
      tile_first_missing(sel, missing, base, nrows, wv, lane, aux);
      if (HASVAL) {
        const int64_t w0 = (base >> 6) + wv * 8, wl = (nrows - 1) >> 6;
        for (int j = 0; j < 8 && w0 + j <= wl; j++) {
          uint64_t m = wave_uniform(sel[w0 + j] & missing[w0 + j]);
          if (w0 + j == wl && (nrows & 63)) m &= (1ull << (nrows & 63)) - 1ull;
          if (!m) continue;
          if (lane == 0) atomicAdd((unsigned long long*)&spec_sh[2], (unsigned long long)__builtin_popcountll(m));
          if (vals.col && ((m >> lane) & 1ull)) acc_value(&spec_sh[3], rvalue_of<V8>(vals.col, vals.vdt, (w0 + j) * 64 + lane), vals.gop, vals.vkind);
        }
      }
    }
    if (unstorable) {                                           // ... the unstorable image's, aux[0]: -1 in an Int64 column is that image; one lane of the wave reports)
      bool first = true;
      for (int j = 0; j < 8; j++) {
        const uint64_t bad = nin[j] & ~keys_storable<KIND>(key[j]);
        if (!bad) continue;
        if (first && lane == 0) aux_min(&aux[0], (uint64_t)(base + wv * 512 + j * 64 + __builtin_ctzll(bad)));
        first = false;
        if (!HASVAL) break;
        if (lane == 0) atomicAdd((unsigned long long*)&spec_sh[0], (unsigned long long)__builtin_popcountll(bad));
        if (vals.col && ((bad >> lane) & 1ull)) acc_value(&spec_sh[1], rvalue_of<V8>(vals.col, vals.vdt, base + wv * 512 + j * 64 + lane), vals.gop, vals.vkind);
      }
    }
    const int64_t nb = base + TILE;
    if (nb < r1) {
      if (nb + TILE <= nrows) nmiss = tile_load<KIND, true>(nkey, nin, sel, col, dtype, missing, nb, nrows, wv, lane);
      else nmiss = tile_load<KIND, false>(nkey, nin, sel, col, dtype, missing, nb, nrows, wv, lane);
    }
    __syncthreads();
    if (xp & 8) continue;                                       // (bit 3, timing only: loads and ranks only)
    // 2. exclusive scan of the tile's counts (thread p owns partition p; its count is cleared for the next tile as it is read)
    uint32_t h = 0;
    if (tid < P) { h = hist2[tid]; hist2[tid] = 0; }
    uint32_t incl = h;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { const uint32_t t = wsum[w]; total += t; if (w < wv) before += t; }
    if (total == 0) continue;                                   // (no row of this tile becomes a record — a selective predicate over clustered rows: most tiles; a flag behind
                                                                //  the FIRST barrier saved one more barrier there and cost the full selection 0.4 of its 4.8 ms)
    const uint32_t ex = before + incl - h;
    uint32_t got = 0;
    if (tid < P) { lstart[tid] = ex; if (h) got = atomicAdd(&pool.front[fx], h); } // the tile's run of partition `tid`: reserved behind whatever the XCD's other workgroups reserved last
    __syncthreads();
    if (xp & 4) continue;                                       // (bit 2, timing only: ranks and scan only)
    // 3. the tile's records into LDS, sorted by partition (the eight reads of lstart first, unconditionally: a branch per row made each wait for its own)
    uint32_t ls[8];
#pragma unroll
    for (int j = 0; j < 8; j++) ls[j] = lstart[(pr[j] >> 13) & 1023u];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (pr[j] != ~0u) {
        const uint32_t slot = ls[j] + (pr[j] & 8191u);
        skey[slot] = key[j];
        srow[slot] = (pr[j] & ~8191u) | (uint32_t)(wv * 512 + j * 64 + lane);
      }
    // HASVAL: the rows' values, asked for now in row order (coalesced: the registers of the keys are free) — they go through LDS behind the keys (step 4)
    uint64_t val[8];
    if (HASVAL) {
      const uint32_t lo = (uint32_t)(wv * 512 + lane);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        val[j] = 0;
        if (vals.col && !(xp & 256) && pr[j] != ~0u)                                           // (bit 8, timing only: no values)
          val[j] = rvalue_of<V8>(vals.col, vals.vdt, base + lo + (uint32_t)(j * 64));
      }
    }
    // the reserved positions [got, got + h) of the stream as places in the pool (the atomic's answer is waited for here, behind the sort)
    if (tid < P && h) {
      const uint32_t k0 = got >> 13, k1 = (got + h - 1u) >> 13;                 // the stream's pages the run lies in (k1 = k0 or k0 + 1)
      uint32_t* e0 = pool.pt + (size_t)fx * pool.maxv + k0;
      uint32_t b0 = pool.dump_page, b1 = pool.dump_page;
      const bool over = k1 >= pool.maxv;
      if (over) __atomic_store_n(&aux[3], 1ull, __ATOMIC_RELAXED);             // aux[kAuxAbort]: this column needs the hash-table form (the records go to the spare page)
      // first the taking ...  (publishing and reading the page table are read-modify-write atomics — exchanged, OR-ed with zero —: they are performed where the
      // other XCDs' are; an acquiring LOAD per thread and tile invalidated the CU's caches 512 times a tile and the pass took 19 ms)
      if (!over && k1 != k0) { b1 = atomicAdd(pool.next_page, 1u); atomicExch(e0 + 1, b1); }
      const bool mine = (got & 8191u) == 0u;
      if (!over && mine) { b0 = atomicAdd(pool.next_page, 1u); atomicExch(e0, b0); }
      // ... then the waiting (the page of the thread's last run is remembered: a stream's page takes 512 runs, a sixteenth of them this workgroup's)
      if (!over && !mine) {
        if (k0 == pk) b0 = pb;
        else {
          uint32_t spins = 0;                                                   // (bounded: a page that never comes — it cannot — would end in the abort flag, not in a hang)
          while ((b0 = atomicOr(e0, 0u)) == 0xFFFFFFFFu) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { __atomic_store_n(&aux[3], 1ull, __ATOMIC_RELAXED); b0 = pool.dump_page; break; }
          }
        }
      }
      if (k1 == k0) b1 = b0;
      pk = k1; pb = b1;
      // sorted slot s (ex <= s < ex + h) is the stream's record got + (s - ex): in page k0 while that is below (k0 + 1) * 8192
      uint4 pl;
      pl.x = (b0 << 13) + (got & 8191u) - ex;                                   // s + d0: its place in page k0
      pl.y = (b1 << 13) + got - ex - (k1 << 13);                                // s + d1: in page k1
      pl.z = ex + ((k0 + 1u) << 13) - got;                                      // slim: the first slot that lies in page k1
      pl.w = 0;
      place[tid] = pl;
    }
    __syncthreads();
    if (xp & 2) continue;                                       // (bit 1, timing only: nothing after the sort)
    // 4. out: all the LDS reads, then the stores
    uint64_t ok[8]; uint32_t ow[8]; uint4 od[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { ok[k] = skey[k * BLOCK + tid]; ow[k] = srow[k * BLOCK + tid]; }
#pragma unroll
    for (int k = 0; k < 8; k++) od[k] = place[(ow[k] >> 13) & 1023u];            // (a slot past `total` holds an older tile's record: read, not written)
    // the next tile's keys are waited for HERE, before the first store is issued: loads and stores share one in-order counter (vmcnt), and a wait for the loads at
    // the top of the next step would also be a wait for the sixteen stores issued after them — a tile's store latency, every tile
    uint64_t ov[8];                                             // HASVAL: the records' values: the rows' values laid down in row order where the sorted keys were
    if (HASVAL) {                                               // (LDS has no room for a third array beside keys and rows; a gathered read from the column per record: 2.6 ms)
      __syncthreads();                                          // every thread has its sorted keys and rows in registers
#pragma unroll
      for (int j = 0; j < 8; j++) skey[wv * 512 + j * 64 + lane] = val[j];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; k++) ov[k] = skey[ow[k] & 8191u];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) asm volatile("" : "+v"(nkey[j]));
    const uint32_t base32 = (uint32_t)base;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t s = (uint32_t)(k * BLOCK + tid);
      if (s < total) {
        const uint32_t dst = s + (s < od[k].z ? od[k].x : od[k].y);
        if (xp & 1) { if (ok[k] == 12345ull) recs_out[dst] = 1; continue; }              // (DFDB_RADIX_XP bit 0, timing only: no stores)
        // one 12-byte record {key image, row}: a partition's run of a tile is ONE piece of 192 bytes, not 128 + 64 in two arrays (the pass waits for its
        // stores, and what they cost goes by the number of pieces: tools/ubench/scatter_runs.hip; as nontemporal stores: 7.4 ms instead of 6.15)
        if (HASVAL) {
          Rec20 r; r.lo = (uint32_t)ok[k]; r.hi = (uint32_t)(ok[k] >> 32); r.row = base32 | (ow[k] & 8191u); r.vlo = (uint32_t)ov[k]; r.vhi = (uint32_t)(ov[k] >> 32);
          *(Rec20*)(recs_out + (size_t)dst * 5) = r;
          continue;
        }
        Rec12 r; r.lo = (uint32_t)ok[k]; r.hi = (uint32_t)(ok[k] >> 32); r.row = base32 | (ow[k] & 8191u);
        *(Rec12*)(recs_out + (size_t)dst * 3) = r;
      }
    }
    // (no barrier here: the next tile writes hist2 — cleared above — before its first barrier, and nothing this step still reads before its third)
  }
  if (HOT || HASVAL) __syncthreads();
  if (HOT && tid < kHotSlots && hc_key[tid] != kREmpty && hc_cnt[tid]) {     // the hot keys' slots -> the launch's list
    const uint32_t i = atomicAdd(pool.hot_n, 1u);
    if (i < pool.hot_cap) { pool.hot[3 * (size_t)i] = hc_key[tid]; pool.hot[3 * (size_t)i + 1] = HASVAL ? hc_val[tid] : 0ull; pool.hot[3 * (size_t)i + 2] = (uint64_t)hc_cnt[tid] << 32 | hc_row[tid]; }
    else __atomic_store_n(&aux[3], 1ull, __ATOMIC_RELAXED);                // (cannot happen: the list holds every workgroup's every slot)
  }
  if (HASVAL) {                                                 // the workgroup's rows of the unstorable key / the missing key -> the launch's
    if (tid < 2 && spec_sh[2 * tid]) {
      atomicAdd((unsigned long long*)&vals.gspec[2 * tid], (unsigned long long)spec_sh[2 * tid]);
      const uint64_t v = spec_sh[2 * tid + 1];
      if (vals.gop == 1) atomicAdd((unsigned long long*)&vals.gspec[2 * tid + 1], (unsigned long long)v);
      else if (vals.gop == 2) atomicAdd((double*)&vals.gspec[2 * tid + 1], __longlong_as_double((long long)v));
      else if (vals.gop == 3) atomicMin((unsigned long long*)&vals.gspec[2 * tid + 1], (unsigned long long)v);
      else if (vals.gop == 4) atomicMax((unsigned long long*)&vals.gspec[2 * tid + 1], (unsigned long long)v);
    }
  }
}

// ---- pass 2: one workgroup per partition (grid-strided): first occurrences out of a table in LDS
// Linear probing from an EVEN slot; a record first looks at the two slots its probing starts with (one 16-byte read of keys, one 8-byte read of rows): nearly
// every record of a partition is a key the table already holds, and at 24 % load all but a few per cent of the keys sit in one of those two.  What is left — a
// key's first record, the displaced keys' records — is claimed later, 64 at a time (the wave's list).  (FOUR slots per look — 48 bytes of LDS per record instead
// of 24 — left 0.5 % of the records for the list and took 5.7 ms instead of 3.0: the pass is bound by LDS reads at random addresses.)
constexpr int kRQueue = 128;                                  // records a wave's list holds (it is emptied by 64 whenever it holds 64: at most 63 + 64)
// where a key's probing starts: an EVEN slot, from the top bits of the hash's first product (the partition is the top bits of the MIXED
// hash): four vector instructions
__device__ __forceinline__ uint32_t table_home(uint64_t key) { return ((((uint32_t)key ^ (uint32_t)(key >> 32) * 0x85EBCA77u) * 0x9E3779B1u) >> 19) & ~1u; }
// one record per lane through the partition's table: linear probing from the key's home slot; the smallest row stays
__device__ __forceinline__ void table_claim(uint64_t* tkey, uint32_t* trow, uint64_t key, uint32_t row, uint32_t* claims, uint32_t* abort_flag) {
  uint32_t h = table_home(key);
  for (uint32_t probes = 0;; probes++) {
    uint64_t old = tkey[h];
    if (old == kREmpty) {
      old = atomicCAS((unsigned long long*)&tkey[h], (unsigned long long)kREmpty, (unsigned long long)key);
      if (old == kREmpty) atomicAdd(claims, 1u);
    }
    if (old == kREmpty || old == key) { if (trow[h] > row) atomicMin(&trow[h], row); break; }
    h = (h + 1) & (kRSlots - 1);
    if (probes >= (uint32_t)kRSlots) { *abort_flag = 1; break; }
  }
}
__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// four records of a thread: FULL = all four exist; otherwise a record past the partition's end reads the last one again and is made the empty image afterwards
template <bool FULL>
__device__ __forceinline__ void recs_load(uint64_t (&kk)[4], uint32_t (&rw)[4], const uint32_t* __restrict__ rp, uint32_t n, int tid) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t i = (uint32_t)(j * kRBlock + tid);
    const Rec12 r = *(const Rec12*)(rp + (size_t)(FULL || i < n ? i : n - 1u) * 3);
    kk[j] = (uint64_t)r.hi << 32 | r.lo; rw[j] = r.row;
  }
}
__global__ __launch_bounds__(kRBlock) void k_radix_unique(const uint32_t* __restrict__ recs, RadixPool pool,
                                                          int P, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, uint64_t* aux, int kbits, int xp) {
  extern __shared__ uint64_t tab_sh[];
  uint64_t* tkey = tab_sh;                                    // [kRSlots]
  uint32_t* trow = (uint32_t*)(tkey + kRSlots);               // [kRSlots]
  __shared__ uint32_t claims_sh, abort_sh;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint64_t* qk = (uint64_t*)(trow + kRSlots) + wv * kRQueue;                     // the wave's list of records still to be claimed: [kRQueue] keys ...
  uint32_t* qr = (uint32_t*)((uint64_t*)(trow + kRSlots) + (kRBlock / 64) * kRQueue) + wv * kRQueue;    // ... and their rows
  for (int p = (int)blockIdx.x; p < P; p += (int)gridDim.x) {
    uint32_t qn = 0;                                          // (wave-uniform)
    for (int i = tid; i < kRSlots; i += kRBlock) { tkey[i] = kREmpty; trow[i] = 0xFFFFFFFFu; }
    if (tid == 0) { claims_sh = 0; abort_sh = __atomic_load_n(&aux[3], __ATOMIC_RELAXED) != 0; }
    __syncthreads();
    if (abort_sh) return;                                     // (the partition pass gave up on a stream: the hash table answers)
    // the partition's records: its kRShare streams one after the other, a stream page by page, a page in two blocks of 4096 records (all wave-uniform).
    // What says where a block lies comes through VECTOR loads — the streams' record counts in lanes 0 .. 7 of one register, 64 entries of the current stream's
    // page table in another, read with v_readlane: a scalar load per block shares its counter with the LDS reads (returns out of order: every wait is a wait
    // for everything) and cost 0.4 of the pass's 3.5 ms
    const uint32_t vfront = pool.front[p * kRShare + (lane & (kRShare - 1))];
    uint32_t wnext = pool.pt[(size_t)(p * kRShare) * pool.maxv + lane], wcur = 0, wbase = 0;      // (the table has 64 entries of slack behind its last row)
    int sx = -1; uint32_t soff = 0, sn = 0;                   // the stream being read, where its next block starts, its records
    auto next_block = [&](const uint32_t*& bp, uint32_t& bc) {
      while (sx < kRShare && soff >= sn) {
        sx++; soff = 0; sn = 0;
        if (sx < kRShare) {
          sn = rl32(vfront, (uint32_t)sx);
          if ((uint64_t)sn > (uint64_t)pool.maxv * kRPage) sn = 0;               // (the partition pass raised the flag for this stream)
          wcur = wnext; wbase = 0;
          if (sx + 1 < kRShare) wnext = pool.pt[(size_t)(p * kRShare + sx + 1) * pool.maxv + lane];
        }
      }
      if (sx >= kRShare) { bc = 0; return; }
      const uint32_t k = soff >> 13;
      if (k - wbase >= 64u) { wbase = k & ~63u; wcur = pool.pt[(size_t)(p * kRShare + sx) * pool.maxv + wbase + lane]; }      // (a stream of more than 64 pages: twice its even share)
      const uint32_t pg = rl32(wcur, k - wbase);
      bp = recs + ((size_t)pg * kRPage + (soff & (uint32_t)(kRPage - 1))) * 3;
      bc = sn - soff < 4u * kRBlock ? sn - soff : 4u * kRBlock;
      soff += 4u * kRBlock;
    };
    const uint32_t* cp = recs; uint32_t cc = 0;
    next_block(cp, cc);
    uint64_t nk[4]; uint32_t nr[4];                           // the NEXT four records: loaded while these four go through the table
    if (cc) { if (cc == 4u * kRBlock) recs_load<true>(nk, nr, cp, cc, tid); else recs_load<false>(nk, nr, cp, cc, tid); }
    while (cc) {
      uint64_t kk[4]; uint32_t rw[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { kk[j] = nk[j]; rw[j] = nr[j]; }
      if (cc != 4u * kRBlock) {
#pragma unroll
        for (int j = 0; j < 4; j++) if ((uint32_t)(j * kRBlock + tid) >= cc) { kk[j] = kREmpty; rw[j] = 0xFFFFFFFFu; }       // (no record holds the empty image)
      }
      next_block(cp, cc);
      if (cc) { if (cc == 4u * kRBlock) recs_load<true>(nk, nr, cp, cc, tid); else recs_load<false>(nk, nr, cp, cc, tid); }
      if (xp & 32) { if ((kk[0] ^ kk[1] ^ kk[2] ^ kk[3]) == 12345ull && (rw[0] ^ rw[1] ^ rw[2] ^ rw[3]) == 77u) abort_sh = 1; continue; }   // (DFDB_RADIX_XP bit 5, timing only: the loads alone)
      uint32_t hb[4]; ulonglong2 tt[4]; uint2 qq[4];
#pragma unroll
      for (int j = 0; j < 4; j++) hb[j] = table_home(kk[j]);
#pragma unroll
      for (int j = 0; j < 4; j++) { tt[j] = *(const ulonglong2*)&tkey[hb[j]]; qq[j] = *(const uint2*)&trow[hb[j]]; }
      // a record that is not in its two slots — its key's first record, a displaced key's every record — does not go through the claiming loop here, where it
      // would hold its 63 neighbours up for a handful of dependent LDS round trips (1.4 of the pass's 3.6 ms when it did): it is put on the WAVE's own list in
      // LDS, and the wave claims 64 at a time, one per lane, whenever the list holds that many
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool e0 = tt[j].x == kk[j], e1 = tt[j].y == kk[j];
        bool pending = false;
        if (e0 | e1) { if ((e0 ? qq[j].x : qq[j].y) > rw[j]) atomicMin(&trow[hb[j] + (e0 ? 0u : 1u)], rw[j]); }
        else pending = kk[j] != kREmpty;                      // (an absent record compares equal to an empty slot — its row, all ones, changes nothing — or is skipped here)
        if (xp & 64) pending = false;                         // (DFDB_RADIX_XP bit 6, timing only: nothing is claimed)
        const uint64_t pm = __ballot(pending);
        if (pm) {
          if (pending) { const uint32_t e = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u)); qk[e] = kk[j]; qr[e] = rw[j]; }
          qn += (uint32_t)__builtin_popcountll(pm);
          if (qn >= 64u) { qn -= 64u; table_claim(tkey, trow, qk[qn + lane], qr[qn + lane], &claims_sh, &abort_sh); }
        }
      }
      // (the claims are read without a barrier: a late view only delays the stop — the flag is what the host looks at)
      if (claims_sh > (kRSlots * 7) / 8) { abort_sh = 1; break; }
    }
    if (qn && !abort_sh) { if (lane < (int)qn) table_claim(tkey, trow, qk[lane], qr[lane], &claims_sh, &abort_sh); }      // what is left on the wave's list
    // the hot keys the partition pass kept to itself (one entry per workgroup and key, the smallest of its rows): those of this partition
    { const uint32_t nhot = *pool.hot_n;
      for (uint32_t i = (uint32_t)tid; i < nhot && !abort_sh; i += kRBlock) {
        const uint64_t key = pool.hot[3 * (size_t)i];
        if ((int)(rhash(key) >> (32 - kbits)) == p) table_claim(tkey, trow, key, (uint32_t)pool.hot[3 * (size_t)i + 2], &claims_sh, &abort_sh);
      } }
    __syncthreads();
    if (abort_sh) { if (tid == 0) __atomic_store_n(&aux[3], 1ull, __ATOMIC_RELAXED); return; }      // aux[kAuxAbort]: this column needs the hash-table form
    for (int i = tid; i < kRSlots; i += kRBlock) {
      if (tkey[i] == kREmpty) continue;
      const uint64_t r = trow[i];
      atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63));
      atomicAdd(&tile_counts[r >> 10], 1u);
    }
    if (p == 0 && tid < 2) {                                  // the two keys kept aside: the unstorable image, missing
      const uint64_t r = aux[tid];
      if (r != kREmpty) { atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63)); atomicAdd(&tile_counts[r >> 10], 1u); }
    }
    __syncthreads();
  }
}

// ---- groupreduce by radix: the LDS-table pass with accumulators -------------------------------------------------------------------------------------------
// dfdb_query_groupreduce over more groups than a workgroup's LDS accumulators hold (9216) sent every selected row's value to its group through a global atomic:
// 85-90 ms per 1e9 rows whether the groups were 5e4 or 1e6.  Here the partition pass writes {key, row, value} records and one workgroup per partition reduces them
// through a table in LDS that holds, per key, its smallest row, the number of its rows and their reduced value; every occupied slot leaves one result {first row,
// rows, value} — a key lives in exactly one partition, so nothing is merged —, the first rows are marked in the bitmap as unique marks them (`mark`), and once the
// bitmap's prefix exists k_radix_group_finish sends every result to the place its first row's rank names: groups in order of first appearance.
constexpr int kGSlots = 4096;                                   // slots of a partition's table (24 bytes each + the waves' lists: 136 KB of LDS)
__device__ __forceinline__ uint32_t gtable_home(uint64_t key) { return ((((uint32_t)key ^ (uint32_t)(key >> 32) * 0x85EBCA77u) * 0x9E3779B1u) >> 20) & ~1u; }
__device__ __forceinline__ void gtable_add(uint32_t* trow, uint32_t* tcnt, uint64_t* tval, uint32_t slot, uint32_t seen_row, uint32_t row, uint64_t v, int gop, int vkind) {
  if (seen_row > row) atomicMin(&trow[slot], row);
  atomicAdd(&tcnt[slot], 1u);
  if (gop) acc_value(&tval[slot], v, gop, vkind);
}
__device__ __forceinline__ void gtable_claim(uint64_t* tkey, uint32_t* trow, uint32_t* tcnt, uint64_t* tval, uint64_t key, uint32_t row, uint64_t v, int gop, int vkind,
                                             uint32_t* claims, uint32_t* abort_flag) {
  uint32_t h = gtable_home(key);
  for (uint32_t probes = 0;; probes++) {
    uint64_t old = tkey[h];
    if (old == kREmpty) {
      old = atomicCAS((unsigned long long*)&tkey[h], (unsigned long long)kREmpty, (unsigned long long)key);
      if (old == kREmpty) atomicAdd(claims, 1u);
    }
    if (old == kREmpty || old == key) { gtable_add(trow, tcnt, tval, h, trow[h], row, v, gop, vkind); break; }
    h = (h + 1) & (kGSlots - 1);
    if (probes >= (uint32_t)kGSlots) { *abort_flag = 1; break; }
  }
}
// an entry of the hot keys' list into the partition's table: it stands for `cnt` rows whose values are already reduced to `v` (min / max: an order image)
__device__ __forceinline__ void gtable_merge(uint64_t* tkey, uint32_t* trow, uint32_t* tcnt, uint64_t* tval, uint64_t key, uint32_t row, uint32_t cnt, uint64_t v, int gop,
                                             uint32_t* claims, uint32_t* abort_flag) {
  uint32_t h = gtable_home(key);
  for (uint32_t probes = 0;; probes++) {
    uint64_t old = tkey[h];
    if (old == kREmpty) {
      old = atomicCAS((unsigned long long*)&tkey[h], (unsigned long long)kREmpty, (unsigned long long)key);
      if (old == kREmpty) atomicAdd(claims, 1u);
    }
    if (old == kREmpty || old == key) {
      atomicMin(&trow[h], row);
      atomicAdd(&tcnt[h], cnt);
      if (gop == 1) atomicAdd((unsigned long long*)&tval[h], (unsigned long long)v);
      else if (gop == 2) atomicAdd((double*)&tval[h], __longlong_as_double((long long)v));
      else if (gop == 3) atomicMin((unsigned long long*)&tval[h], (unsigned long long)v);
      else if (gop == 4) atomicMax((unsigned long long*)&tval[h], (unsigned long long)v);
      break;
    }
    h = (h + 1) & (kGSlots - 1);
    if (probes >= (uint32_t)kGSlots) { *abort_flag = 1; break; }
  }
}
template <bool FULL>
__device__ __forceinline__ void recs20_load(uint64_t (&kk)[4], uint32_t (&rw)[4], uint64_t (&vv)[4], const uint32_t* __restrict__ rp, uint32_t n, int tid) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t i = (uint32_t)(j * kRBlock + tid);
    const Rec20 r = *(const Rec20*)(rp + (size_t)(FULL || i < n ? i : n - 1u) * 5);
    kk[j] = (uint64_t)r.hi << 32 | r.lo; rw[j] = r.row; vv[j] = (uint64_t)r.vhi << 32 | r.vlo;
  }
}
__global__ __launch_bounds__(kRBlock) void k_radix_group(const uint32_t* __restrict__ recs, RadixPool pool, int P, int mark, uint64_t* __restrict__ bitmap,
                                                         uint32_t* __restrict__ tile_counts, uint64_t* aux, uint4* __restrict__ results, uint32_t* __restrict__ nres,
                                                         int gop, int vkind, int kbits) {
  extern __shared__ uint64_t tab_sh[];
  uint64_t* tkey = tab_sh;                                    // [kGSlots]
  uint64_t* tval = tkey + kGSlots;                            // [kGSlots] the key's reduced value (min / max: of order images)
  uint32_t* trow = (uint32_t*)(tval + kGSlots);               // [kGSlots] its smallest row
  uint32_t* tcnt = trow + kGSlots;                            // [kGSlots] its rows
  __shared__ uint32_t claims_sh, abort_sh;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint64_t* qbase = (uint64_t*)(tcnt + kGSlots);
  uint64_t* qk = qbase + wv * kRQueue;                                                   // the wave's list of records still to be claimed: keys ...
  uint64_t* qv = qbase + (kRBlock / 64) * kRQueue + wv * kRQueue;                        // ... values ...
  uint32_t* qr = (uint32_t*)(qbase + 2 * (kRBlock / 64) * kRQueue) + wv * kRQueue;       // ... rows
  const uint64_t vinit = gop == 3 ? ~0ull : 0ull;
  for (int p = (int)blockIdx.x; p < P; p += (int)gridDim.x) {
    uint32_t qn = 0;
    for (int i = tid; i < kGSlots; i += kRBlock) { tkey[i] = kREmpty; trow[i] = 0xFFFFFFFFu; tcnt[i] = 0; tval[i] = vinit; }
    if (tid == 0) { claims_sh = 0; abort_sh = __atomic_load_n(&aux[3], __ATOMIC_RELAXED) != 0; }
    __syncthreads();
    if (abort_sh) return;
    const uint32_t vfront = pool.front[p * kRShare + (lane & (kRShare - 1))];
    uint32_t wnext = pool.pt[(size_t)(p * kRShare) * pool.maxv + lane], wcur = 0, wbase = 0;
    int sx = -1; uint32_t soff = 0, sn = 0;
    auto next_block = [&](const uint32_t*& bp, uint32_t& bc) {
      while (sx < kRShare && soff >= sn) {
        sx++; soff = 0; sn = 0;
        if (sx < kRShare) {
          sn = rl32(vfront, (uint32_t)sx);
          if ((uint64_t)sn > (uint64_t)pool.maxv * kRPage) sn = 0;
          wcur = wnext; wbase = 0;
          if (sx + 1 < kRShare) wnext = pool.pt[(size_t)(p * kRShare + sx + 1) * pool.maxv + lane];
        }
      }
      if (sx >= kRShare) { bc = 0; return; }
      const uint32_t k = soff >> 13;
      if (k - wbase >= 64u) { wbase = k & ~63u; wcur = pool.pt[(size_t)(p * kRShare + sx) * pool.maxv + wbase + lane]; }
      const uint32_t pg = rl32(wcur, k - wbase);
      bp = recs + ((size_t)pg * kRPage + (soff & (uint32_t)(kRPage - 1))) * 5;
      bc = sn - soff < 4u * kRBlock ? sn - soff : 4u * kRBlock;
      soff += 4u * kRBlock;
    };
    const uint32_t* cp = recs; uint32_t cc = 0;
    next_block(cp, cc);
    uint64_t nk[4], nv[4]; uint32_t nr[4];
    if (cc) { if (cc == 4u * kRBlock) recs20_load<true>(nk, nr, nv, cp, cc, tid); else recs20_load<false>(nk, nr, nv, cp, cc, tid); }
    while (cc) {
      uint64_t kk[4], vv[4]; uint32_t rw[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { kk[j] = nk[j]; rw[j] = nr[j]; vv[j] = nv[j]; }
      if (cc != 4u * kRBlock) {
#pragma unroll
        for (int j = 0; j < 4; j++) if ((uint32_t)(j * kRBlock + tid) >= cc) { kk[j] = kREmpty; rw[j] = 0xFFFFFFFFu; }
      }
      next_block(cp, cc);
      if (cc) { if (cc == 4u * kRBlock) recs20_load<true>(nk, nr, nv, cp, cc, tid); else recs20_load<false>(nk, nr, nv, cp, cc, tid); }
      uint32_t hb[4]; ulonglong2 tt[4]; uint2 qq[4];
#pragma unroll
      for (int j = 0; j < 4; j++) hb[j] = gtable_home(kk[j]);
#pragma unroll
      for (int j = 0; j < 4; j++) { tt[j] = *(const ulonglong2*)&tkey[hb[j]]; qq[j] = *(const uint2*)&trow[hb[j]]; }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool real = kk[j] != kREmpty;                   // (a slot past the block's end: no record)
        const bool e0 = tt[j].x == kk[j], e1 = tt[j].y == kk[j];
        const bool hit = real && (e0 | e1), pending = real && !hit;
        if (hit) gtable_add(trow, tcnt, tval, hb[j] + (e0 ? 0u : 1u), e0 ? qq[j].x : qq[j].y, rw[j], vv[j], gop, vkind);
        const uint64_t pm = __ballot(pending);
        if (pm) {
          if (pending) { const uint32_t e = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u)); qk[e] = kk[j]; qv[e] = vv[j]; qr[e] = rw[j]; }
          qn += (uint32_t)__builtin_popcountll(pm);
          if (qn >= 64u) { qn -= 64u; gtable_claim(tkey, trow, tcnt, tval, qk[qn + lane], qr[qn + lane], qv[qn + lane], gop, vkind, &claims_sh, &abort_sh); }
        }
      }
      if (claims_sh > (kGSlots * 7) / 8) { abort_sh = 1; break; }
    }
    if (qn && !abort_sh) { if (lane < (int)qn) gtable_claim(tkey, trow, tcnt, tval, qk[lane], qr[lane], qv[lane], gop, vkind, &claims_sh, &abort_sh); }
    // the hot keys the partition pass reduced on its own (one entry per workgroup and key): those of this partition
    const uint64_t* hot = pool.hot;
    const uint32_t nhot = *pool.hot_n;
    for (uint32_t i = (uint32_t)tid; i < nhot && !abort_sh; i += kRBlock) {
      const uint64_t key = hot[3 * (size_t)i];
      if ((int)(rhash(key) >> (32 - kbits)) != p) continue;
      const uint64_t cr = hot[3 * (size_t)i + 2];
      gtable_merge(tkey, trow, tcnt, tval, key, (uint32_t)cr, (uint32_t)(cr >> 32), hot[3 * (size_t)i + 1], gop, &claims_sh, &abort_sh);
    }
    __syncthreads();
    if (abort_sh) { if (tid == 0) __atomic_store_n(&aux[3], 1ull, __ATOMIC_RELAXED); return; }
    for (int i = tid; i < kGSlots; i += kRBlock) {
      if (tkey[i] == kREmpty) continue;
      const uint64_t r = trow[i], v = tval[i];
      results[atomicAdd(nres, 1u)] = make_uint4((uint32_t)r, tcnt[i], (uint32_t)v, (uint32_t)(v >> 32));
      if (mark) { atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63)); atomicAdd(&tile_counts[r >> 10], 1u); }
    }
    if (mark && p == 0 && tid < 2) {                          // the two keys kept aside: the unstorable image, missing
      const uint64_t r = aux[tid];
      if (r != kREmpty) { atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63)); atomicAdd(&tile_counts[r >> 10], 1u); }
    }
    __syncthreads();
  }
}
// results -> the groups' places: a result's group is the rank of its first row among the first rows (the bitmap's set bits, `prefix` = set bits before every
// 1024-row tile); the unstorable key's group takes the partition pass's own accumulator
__global__ void k_radix_group_finish(const uint4* __restrict__ results, const uint32_t* __restrict__ nres, const uint64_t* __restrict__ ubits, const uint64_t* __restrict__ uprefix,
                                     const uint64_t* aux, const uint64_t* gspec, uint64_t* __restrict__ cnt, uint64_t* __restrict__ val) {
  auto rank_of = [&](uint64_t row) -> uint64_t {
    const uint64_t tile = row >> 10, w = (row & 1023) >> 6;
    uint64_t r = uprefix[tile];
    for (uint64_t k = 0; k < w; k++) r += (uint64_t)__popcll(ubits[tile * 16 + k]);
    return r + (uint64_t)__popcll(ubits[tile * 16 + w] & ((1ull << (row & 63)) - 1ull));
  };
  const uint32_t n = *nres;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint4 r = results[i];
    const uint64_t g = rank_of(r.x);
    cnt[g] = r.y; val[g] = (uint64_t)r.w << 32 | r.z;
  }
  if (blockIdx.x == 0 && threadIdx.x < 2 && aux[threadIdx.x] != kREmpty) { const uint64_t g = rank_of(aux[threadIdx.x]); cnt[g] = gspec[2 * threadIdx.x]; val[g] = gspec[2 * threadIdx.x + 1]; }
}
}  // namespace

int64_t radix_rows_per_chunk(int64_t nrows, int chunks) {
  const int64_t per = (nrows + chunks - 1) / chunks;
  return (per + kRTile - 1) / kRTile * kRTile;              // whole tiles of the partition pass (and whole bitmap words)
}
static int radix_xp() { static const int v = [] { const char* e = getenv("DFDB_RADIX_XP"); return e ? atoi(e) : 0; }(); return v; }   // timing experiments only: results are WRONG with any bit set
static size_t radix_partition_lds_bytes(int block) { return (size_t)part_lds_words(block) * 4 + (size_t)block * 8 * 8; }
static int radix_kind(int dtype) { return dtype == DFDB_F64 ? kKindF64 : (dtype == DFDB_I64 || dtype == DFDB_U64 ? kKindRaw8 : kKindAny); }
int radix_share() { return kRShare; }
// the record pool for `cnt` selected rows in 2^kbits partitions: every stream (partition, share) ends in a page that is not full, one page is nobody's
// (where a stream that gave up puts its records); a stream may take `maxv` pages: 16 times its even share — the sample already turned skewed columns away
int64_t radix_pool_pages(int64_t cnt, int kbits) { return (cnt + kRPage - 1) / kRPage + ((int64_t)kRShare << kbits) + 1; }
int64_t radix_pool_record_bytes(int64_t cnt, int kbits, bool with_values) { return radix_pool_pages(cnt, kbits) * kRPage * (with_values ? 20 : 12); }
int radix_group_slots() { return kGSlots; }
int radix_hot_slots() { return kHotSlots; }
uint32_t radix_pool_maxv(int64_t cnt, int kbits) { const int64_t even = ((cnt + kRPage - 1) / kRPage + ((int64_t)kRShare << kbits) - 1) / ((int64_t)kRShare << kbits); return (uint32_t)(16 * even + 16); }

bool launch_radix_sample(hipStream_t s, const uint64_t* sel, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int kbits, int chunks, int step,
                         uint32_t* counts) {
  if (kbits < 6 || kbits > 10 || nrows < 1 || step < 1) return false;
  const int64_t rpc = radix_rows_per_chunk(nrows, chunks);
  const size_t lds = ((size_t)1 << kbits) * 4;
  switch (radix_kind(dtype)) {
    case kKindRaw8: hipLaunchKernelGGL(k_radix_hist<kKindRaw8>, dim3(chunks), dim3(kRBlock), lds, s, sel, col, dtype, missing, nrows, rpc, kbits, counts, step); break;
    case kKindF64: hipLaunchKernelGGL(k_radix_hist<kKindF64>, dim3(chunks), dim3(kRBlock), lds, s, sel, col, dtype, missing, nrows, rpc, kbits, counts, step); break;
    default: hipLaunchKernelGGL(k_radix_hist<kKindAny>, dim3(chunks), dim3(kRBlock), lds, s, sel, col, dtype, missing, nrows, rpc, kbits, counts, step); break;
  }
  return true;
}
template <int KIND, bool HASVAL, bool V8, bool HOT>
static bool radix_partition_go(hipStream_t s, const uint64_t* sel, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int kbits, int chunks,
                               const RadixPool& pool, uint32_t* recs_out, uint64_t* aux, const RadixVals& vals) {
  static const bool ok = hipFuncSetAttribute((const void*)k_radix_partition<KIND, kRBlock, HASVAL, V8, HOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)radix_partition_lds_bytes(kRBlock)) == hipSuccess;      // (+ the static arrays: under 160 KB)
  if (!ok) { (void)hipGetLastError(); return false; }
  hipLaunchKernelGGL((k_radix_partition<KIND, kRBlock, HASVAL, V8, HOT>), dim3(chunks), dim3(kRBlock), radix_partition_lds_bytes(kRBlock), s, sel, col, dtype, missing, nrows,
                     radix_rows_per_chunk(nrows, chunks), kbits, pool, recs_out, aux, vals, radix_xp());
  return true;
}
// (512-thread workgroups sorting 4096 rows, two per CU, instead of one of 1024 sorting 8192: 5.36-5.43 ms against 5.32-5.33 — the pass waits for its stores either way)
// group = nullptr: 12-byte records for unique; otherwise 20-byte records {key, row, value} for groupreduce (group->valcol may be null: count only)
bool launch_radix_partition(hipStream_t s, const uint64_t* sel, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int kbits, int chunks,
                            const RadixPool& pool, uint32_t* recs_out, uint64_t* aux, const RadixGroup* group, bool hot) {
  if (kbits < 6 || kbits > 10 || nrows < 1 || chunks % kRShare) return false;
  RadixVals v{};
  if (group) { v.col = group->valcol; v.vdt = group->valdt; v.gspec = group->gspec; v.gop = group->gop; v.vkind = group->vkind; }
  const bool v8 = !group || !group->valcol || group->valdt == DFDB_I64 || group->valdt == DFDB_U64 || group->valdt == DFDB_F64;
#define DFDB_RP(K) (group ? (!v8 ? radix_partition_go<K, true, false, true>(s, sel, col, dtype, missing, nrows, kbits, chunks, pool, recs_out, aux, v) \
                                 : hot ? radix_partition_go<K, true, true, true>(s, sel, col, dtype, missing, nrows, kbits, chunks, pool, recs_out, aux, v) \
                                       : radix_partition_go<K, true, true, false>(s, sel, col, dtype, missing, nrows, kbits, chunks, pool, recs_out, aux, v)) \
                          : (hot ? radix_partition_go<K, false, true, true>(s, sel, col, dtype, missing, nrows, kbits, chunks, pool, recs_out, aux, v) \
                                 : radix_partition_go<K, false, true, false>(s, sel, col, dtype, missing, nrows, kbits, chunks, pool, recs_out, aux, v)))
  switch (radix_kind(dtype)) {
    case kKindRaw8: return DFDB_RP(kKindRaw8);
    case kKindF64: return DFDB_RP(kKindF64);
    default: return DFDB_RP(kKindAny);
  }
#undef DFDB_RP
}
bool launch_radix_group(hipStream_t s, const uint32_t* recs, const RadixPool& pool, int kbits, bool mark, uint64_t* bitmap, uint32_t* tile_counts, uint64_t* aux,
                        const RadixGroup& group, int cus) {
  const size_t lds = (size_t)kGSlots * 24 + (size_t)(kRBlock / 64) * kRQueue * 20;
  static bool ok = [] { return hipFuncSetAttribute((const void*)k_radix_group, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess; }();
  if (!ok) { (void)hipGetLastError(); return false; }
  const int P = 1 << kbits;
  hipLaunchKernelGGL(k_radix_group, dim3(P < cus ? P : cus), dim3(kRBlock), lds, s, recs, pool, P, mark ? 1 : 0, bitmap, tile_counts, aux, (uint4*)group.results, group.nres,
                     group.gop, group.vkind, kbits);
  return true;
}
void launch_radix_group_finish(hipStream_t s, const RadixGroup& group, const uint64_t* ubits, const uint64_t* uprefix, const uint64_t* aux, uint64_t* cnt, uint64_t* val) {
  hipLaunchKernelGGL(k_radix_group_finish, dim3(1024), dim3(256), 0, s, (const uint4*)group.results, group.nres, ubits, uprefix, aux, group.gspec, cnt, val);
}
bool launch_radix_unique(hipStream_t s, const uint32_t* recs, const RadixPool& pool, int kbits, uint64_t* bitmap, uint32_t* tile_counts, uint64_t* aux, int cus) {
  const size_t lds = (size_t)kRSlots * 12 + (size_t)(kRBlock / 64) * kRQueue * 12;
  static bool ok = [] { return hipFuncSetAttribute((const void*)k_radix_unique, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess; }();
  if (!ok) { (void)hipGetLastError(); return false; }
  const int P = 1 << kbits;
  const int grid = P < cus ? P : cus;                       // one 96-KB table per CU at a time
  hipLaunchKernelGGL(k_radix_unique, dim3(grid), dim3(kRBlock), lds, s, recs, pool, P, bitmap, tile_counts, aux, kbits, radix_xp());
  return true;
}

}  // namespace dfdb
