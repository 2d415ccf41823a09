// k_unique.hip — unique(col) over the selected rows as a SELECTION: the rows that hold the first occurrence of their value.
//
// Replaces Base.unique driven by Base.iterate(::DFColumn) (src/tables/column.jl:102-126; docs/src/index.md:171-182,
// 479-486: "unique(t.brand[t.brand .!= ""])", 7-11 MRows/s in the reference).  Julia's unique keeps the FIRST occurrence
// of every value in iteration order and compares with isequal (NaN == NaN, 0.0 != -0.0, missing == missing).  Here:
//   insert  every selected row puts (key, row) into an open-addressing table of 16-byte {key, smallest row} entries in HBM:
//           a 64-bit atomicCAS claims the slot of a key, a 64-bit atomicMin keeps the smallest row that holds it — both behind
//           plain loads, so a row whose key is in place with a smaller row recorded touches one cache line and no atomic;
//   mark    the bitmap of first occurrences (+ per-tile counts): SCATTERED out of the table when the distinct values are few
//           beside the rows (one bit per occupied slot), else every selected row looks its key up and keeps its bit iff it IS
//           that smallest row; the ordinary count / gather / materialize machinery then returns the distinct values in order
//           of first appearance.  No sort.
// The table is sized by the DISTINCT values, which nobody knows beforehand (round 4; it used to be sized by the selected rows:
// 34 GB of table for 1e9 rows of 1e6 values, every probe a miss in every cache — 86 ms, this form 20): the host feeds the
// rows in three chunks (1 M rows, 16 M, the rest), reads the number of claimed slots after each, estimates the distinct count
// of the whole selection from it (query.cpp: unique_capacity_wanted) and MIGRATES the entries to a larger table when needed; a
// probe sequence of kMaxProbe slots raises an abort flag, the host grows the table and repeats that chunk (inserts are idempotent).
// Fixed-width values are their own keys.  A String's key is a salted 64-bit hash of its bytes; the slot remembers where one
// holder's bytes start, a verify pass compares every selected row with that representative and reports a true hash collision
// (two different strings, one key), in which case the host repeats with another salt: the result is exact, not probabilistic.
// Integer keys of a small range take the dense form at the end of this file instead (no hashing: a presence bit per value in LDS).
#include "device_utils.hpp"
#include <algorithm>
#include <atomic>
#include <type_traits>
#include "kernels.hpp"
#include "../../include/dfdb_ir.h"

namespace dfdb {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int64_t kTile = 1024;
constexpr uint64_t kEmpty = 0xFFFFFFFFFFFFFFFFull;    // never stored as a key: a value with this image uses aux[0]
constexpr int kMaxProbe = 256;                        // a longer probe sequence means the table is too full: abort, the host grows it
// aux (8 x u64 beside the table): 0 = smallest row whose key is kEmpty, 1 = smallest missing row, 2 = claimed slots, 3 = abort flag, 4 = (int) hash collision seen
constexpr int kAuxClaims = 2, kAuxAbort = 3;
constexpr uint64_t kNoSlot = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ uint64_t slot_of(uint64_t key, uint64_t mask) { return splitmix64(key) & mask; }

// 64-bit image of row `row` of a fixed-width column under isequal: integers by value, floats by bits with one NaN
__device__ __forceinline__ uint64_t key_fixed(const void* col, int dtype, int64_t row) {
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

// -> the key's slot, bit 63 set when the key was already there (clear: this call claimed the slot), kNoSlot when the probe sequence got too long
// (h: where the probe sequence starts — slot_of(key) for values that are their own keys, the low bits of a String's key, which is a hash already)
__device__ __forceinline__ uint64_t table_insert(UniqueEntry* ent, uint64_t mask, uint64_t key, uint64_t row, uint64_t* aux, uint64_t h) {
  for (int probes = 0; probes < kMaxProbe; probes++) {
    // plain loads first (key and row of a slot share a line): with few distinct values nearly every row finds its key in place and a smaller row
    // recorded, and skips both atomics (5e8 rows of 10 distinct strings: 0.60 s with every row hammering the same 10 addresses, 0.023 s so).  A
    // stale line (the XCDs' L2s are not coherent) shows an emptier slot or a larger row than memory holds: an atomic too many, never one too few
    uint64_t old = __atomic_load_n(&ent[h].key, __ATOMIC_RELAXED);
    const uint64_t seen = __atomic_load_n(&ent[h].row, __ATOMIC_RELAXED);
    if (old == kEmpty) old = atomicCAS((unsigned long long*)&ent[h].key, (unsigned long long)kEmpty, (unsigned long long)key);
    if (old == kEmpty || old == key) {
      if (seen > row) atomicMin((unsigned long long*)&ent[h].row, (unsigned long long)row);
      return old == kEmpty ? h : (h | (1ull << 63));
    }
    h = (h + 1) & mask;
  }
  __atomic_store_n(&aux[kAuxAbort], 1ull, __ATOMIC_RELAXED);
  return kNoSlot;
}
__device__ __forceinline__ uint64_t table_find(const UniqueEntry* ent, uint64_t mask, uint64_t key, uint64_t h) {
  while (ent[h].key != key) h = (h + 1) & mask;     // present by construction (the insert pass put it there)
  return h;
}
// the same for a key that need not be there (groupreduce over an optimistically filled table): kNoSlot when its probe sequence reaches an empty slot
__device__ __forceinline__ uint64_t table_find_maybe(const UniqueEntry* ent, uint64_t mask, uint64_t key, uint64_t h) {
  for (int probes = 0; probes <= kMaxProbe; probes++) {
    const uint64_t k = ent[h].key;
    if (k == key) return h;
    if (k == kEmpty) return kNoSlot;
    h = (h + 1) & mask;
  }
  return kNoSlot;
}
// the group number (entry.row after k_group_ids) of a key that need not be there, the entry read ONCE as 16 bytes (key and row together: finding the slot and then
// loading its row was a second dependent round trip per row of groupreduce's accumulate pass); kEmpty when the key has no slot
__device__ __forceinline__ uint64_t table_row_maybe(const UniqueEntry* ent, uint64_t mask, uint64_t key, uint64_t h) {
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  for (int probes = 0; probes <= kMaxProbe; probes++) {
    const u64x2 e = *(const u64x2*)&ent[h];
    if (e.x == key) return e.y;
    if (e.x == kEmpty) return kEmpty;
    h = (h + 1) & mask;
  }
  return kEmpty;
}
// a workgroup's claimed slots -> aux[kAuxClaims] (one atomic per workgroup)
__device__ __forceinline__ void add_claims(uint32_t mine, uint32_t* sh, uint64_t* aux) {
  if (mine) atomicAdd(sh, mine);
  __syncthreads();
  if (threadIdx.x == 0 && *sh) atomicAdd((unsigned long long*)&aux[kAuxClaims], (unsigned long long)*sh);
}

// ---------------------------------------------------------------- fixed-width columns
__global__ __launch_bounds__(kBlock) void k_unique_insert(const uint64_t* __restrict__ bitmap, const void* __restrict__ col, int dtype,
                                                          const uint64_t* __restrict__ missing, int64_t row0, int64_t row1, UniqueEntry* ent,
                                                          uint64_t mask, uint64_t* aux) {
  __shared__ uint32_t claims_sh;
  if (threadIdx.x == 0) claims_sh = 0;
  __syncthreads();
  uint32_t mine = 0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t row = row0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; row < row1; row += stride) {
    if (!((bitmap[row >> 6] >> (row & 63)) & 1ull)) continue;
    if (missing && ((missing[row >> 6] >> (row & 63)) & 1ull)) { if (__atomic_load_n(&aux[1], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&aux[1], (unsigned long long)row); continue; }
    const uint64_t key = key_fixed(col, dtype, row);
    if (key == kEmpty) { if (__atomic_load_n(&aux[0], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&aux[0], (unsigned long long)row); continue; }
    const uint64_t r = table_insert(ent, mask, key, (uint64_t)row, aux, slot_of(key, mask));
    if (r == kNoSlot) break;                          // the table is too full: the host grows it and repeats the chunk
    mine += (uint32_t)!(r >> 63);
  }
  add_claims(mine, &claims_sh, aux);
}

// the entries of a table that became too small, into its successor (strings: with their representatives)
__global__ __launch_bounds__(kBlock) void k_unique_migrate(const UniqueEntry* __restrict__ from, const uint64_t* __restrict__ from_off, const uint32_t* __restrict__ from_len,
                                                           uint64_t from_cap, UniqueEntry* ent, uint64_t* rep_off, uint32_t* rep_len, uint64_t mask, uint64_t* aux) {
  __shared__ uint32_t claims_sh;
  if (threadIdx.x == 0) claims_sh = 0;
  __syncthreads();
  uint32_t mine = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < from_cap; i += stride) {
    const uint64_t key = from[i].key;
    if (key == kEmpty) continue;
    const uint64_t r = table_insert(ent, mask, key, from[i].row, aux, from_off ? (key & mask) : slot_of(key, mask));
    if (r == kNoSlot) break;
    mine += (uint32_t)!(r >> 63);
    if (from_off && !(r >> 63)) { rep_off[r] = from_off[i]; rep_len[r] = from_len[i]; }
  }
  add_claims(mine, &claims_sh, aux);
}

__global__ __launch_bounds__(kBlock) void k_unique_mark(uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, const void* __restrict__ col,
                                                        int dtype, const uint64_t* __restrict__ missing, int64_t nrows, int64_t ntiles,
                                                        const UniqueEntry* __restrict__ ent, uint64_t mask, const uint64_t* __restrict__ aux) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    uint64_t myword = 0; uint32_t cnt = 0;
    const uint64_t mine = lane < 16 ? bitmap[tile * 16 + lane] : 0ull;
    if (__ballot(mine != 0) == 0) { if (lane == 0) tile_counts[tile] = 0; continue; }
    for (int j = 0; j < 16; j++) {
      const uint64_t w = __shfl(mine, j, 64);
      bool first = false;
      if (w) {
        const int64_t row = tile * kTile + j * 64 + lane;
        if (row < nrows && ((w >> lane) & 1ull)) {
          if (missing && ((missing[row >> 6] >> (row & 63)) & 1ull)) first = aux[1] == (uint64_t)row;
          else {
            const uint64_t key = key_fixed(col, dtype, row);
            first = key == kEmpty ? aux[0] == (uint64_t)row : ent[table_find(ent, mask, key, slot_of(key, mask))].row == (uint64_t)row;
          }
        }
      }
      const uint64_t m = __ballot(first);
      if (lane == j) myword = m;
      cnt += (uint32_t)__popcll(m);
    }
    if (lane < 16) bitmap[tile * 16 + lane] = myword;
    if (lane == 0) tile_counts[tile] = cnt;
  }
}

// the first occurrences straight out of the table (few distinct values beside the rows): one bit + one tile count per occupied slot, into a
// bitmap and tile counts the host has cleared
__global__ __launch_bounds__(kBlock) void k_unique_scatter(const UniqueEntry* __restrict__ ent, uint64_t cap, const uint64_t* __restrict__ aux,
                                                           uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < cap + 2; i += stride) {
    uint64_t r;
    if (i < cap) { if (ent[i].key == kEmpty) continue; r = ent[i].row; }
    else { r = aux[i - cap]; if (r == kEmpty) continue; }
    atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63));
    atomicAdd(&tile_counts[r >> 10], 1u);
  }
}

// ---------------------------------------------------------------- String columns (FlatStringsVector: sizes + arena, offsets per 1024-row tile)
// A String's 64-bit key.  `head` / `head2` = its bytes 0..7 / 8..15 as loaded (bytes past its end are masked here; head2 only where len > 8).  Two 32-bit lanes of
// state, one 32-bit multiply each per 8 bytes and two to finish (64-bit multiplies are four quarter-rate instructions each on this chip: three rounds of splitmix64 per
// row were ~500 of the ~1000 cycles a wave spent per 64 rows).  Every step is a bijection of the state for a given chunk, so two different strings of one length
// <= 8 never share a key; any others that do are found by the compare against the slot's representative, and the salt changes.
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }
__device__ __forceinline__ void hash_chunk(uint32_t& a, uint32_t& b, uint64_t v) {
  a = (a ^ (uint32_t)v) * 0x85EBCA6Bu;
  b = (b ^ (uint32_t)(v >> 32)) * 0xC2B2AE35u;
  a = rotl32(a, 15) + b;
  b = rotl32(b, 13) ^ a;
}
__device__ __forceinline__ uint64_t low_bytes(uint64_t v, int n) { return n >= 8 ? v : (n > 0 ? v & (~0ull >> (64 - 8 * n)) : 0ull); }
__device__ __forceinline__ uint64_t load8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }    // (the arena ends in 16 bytes of padding: an 8-byte probe never faults)
__device__ __forceinline__ uint64_t hash_bytes(const uint8_t* p, int32_t len, uint64_t salt, uint64_t head, uint64_t head2) {
  uint32_t a = (uint32_t)salt ^ __umul24((uint32_t)len & 0xFFFFFFu, 0x9E3779u), b = (uint32_t)(salt >> 32) + ((uint32_t)len >> 24);
  int32_t k = 0;
  if (len >= 8) {
    hash_chunk(a, b, head); k = 8;
    if (len >= 16) { hash_chunk(a, b, head2); for (k = 16; k + 8 <= len; k += 8) hash_chunk(a, b, load8(p + k)); }
  }
  if (k < len) hash_chunk(a, b, low_bytes(k == 0 ? head : (k == 8 ? head2 : load8(p + k)), len - k));
  a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13;
  b ^= a; b *= 0xC2B2AE35u; b ^= b >> 16;
  a ^= b;
  return ((uint64_t)a << 32) | b;
}

// What a wave has already met: strings of up to 16 bytes (length + bytes fit one entry and identify the string EXACTLY) in a cache in LDS, one per wave.
// A wave walks its tiles, the words of a tile and the lanes of a word in increasing row order, so whatever it does for a string the first time it meets it — insert
// its key with that row (smaller than any row it will meet later), compare it with its slot's representative, find its group — holds for every later meeting:
// those rows touch neither the hash nor the table.  An entry has ONE writer per instruction: lanes that want a slot write their lane number beside it first, the one
// that reads its own number back writes the entry (two lanes storing different entries to one slot in the same instruction could tear it).
constexpr int kMetSlots = 256;
struct MetEntry { uint64_t head, head2; uint32_t len, val; uint64_t pad; };
template <int SLOTS>
struct MetCacheT {
  MetEntry* e; uint32_t* claim;
  __device__ __forceinline__ void init(MetEntry* entries, uint32_t* claims, int lane) {
    e = entries; claim = claims;
    for (int i = lane; i < SLOTS; i += 64) e[i].len = 0xFFFFFFFFu;
  }
  // two places per string (the 32-bit mix gives both): direct-mapped, two of a column's FREQUENT strings that share a slot evict each other on every row — measured on the
  // ten brands with 128 slots: the accumulate pass 3x slower.  A string goes to its first place if that is empty or already its own, to its second otherwise
  static constexpr int kLog = SLOTS == 256 ? 8 : SLOTS == 128 ? 7 : 6;
  static __device__ __forceinline__ uint32_t mix(uint64_t head, uint64_t head2, uint32_t len) {
    const uint32_t x = (uint32_t)head ^ rotl32((uint32_t)(head >> 32), 7) ^ rotl32((uint32_t)head2, 13) ^ rotl32((uint32_t)(head2 >> 32), 19) ^ (len << 27);
    return x * 0x9E3779B1u;
  }
  static __device__ __forceinline__ uint32_t slot1(uint32_t m) { return m >> (32 - kLog); }
  static __device__ __forceinline__ uint32_t slot2(uint32_t m) { const uint32_t a = m >> (32 - kLog), b = (m >> (32 - 2 * kLog)) & (uint32_t)(SLOTS - 1); return b == a ? (b ^ 1u) : b; }
  static __device__ __forceinline__ bool is(const MetEntry& x, uint64_t head, uint64_t head2, uint32_t len) { return x.len == len && x.head == head && x.head2 == head2; }
  // (called by every lane of the wave; `want`: this lane's answer matters — the second place is read by all lanes when one of those missed the first)
  __device__ __forceinline__ bool find(uint64_t head, uint64_t head2, uint32_t len, uint32_t& val, bool want = true) const {
    const uint32_t m = mix(head, head2, len);
    const MetEntry x = e[slot1(m)];
    bool hit = is(x, head, head2, len);
    val = x.val;
    if (__ballot(want && !hit) != 0) { const MetEntry y = e[slot2(m)]; const bool hit2 = is(y, head, head2, len); val = hit ? val : y.val; hit = hit || hit2; }
    // the value is read HERE, with the tag: left to the compiler the load sinks into the caller's hit branch, which may run after the miss branch of other
    // lanes of the same instruction has overwritten the slot (found by the forms test: rows counted for a neighbouring group)
    asm volatile("" : "+v"(val));
    return hit;
  }
  __device__ __forceinline__ void put(uint64_t head, uint64_t head2, uint32_t len, uint32_t val, int lane) {
    const uint32_t m = mix(head, head2, len);
    const MetEntry x = e[slot1(m)];
    const uint32_t s = (x.len == 0xFFFFFFFFu || is(x, head, head2, len)) ? slot1(m) : slot2(m);
    claim[s] = (uint32_t)lane;
    asm volatile("" ::: "memory");                           // the read-back must be a read (of LDS, not of the register just stored): it decides which lane goes on
    if (claim[s] == (uint32_t)lane) { MetEntry y; y.head = head; y.head2 = head2; y.len = len; y.val = val; y.pad = 0; e[s] = y; }
  }
};
constexpr int32_t kMetMaxLen = 16;

// do the two strings hold the same bytes?  (8 at a time; both live in the padded arena)
__device__ __forceinline__ bool same_bytes(const uint8_t* a, int32_t la, const uint8_t* b, uint32_t lb) {
  if ((uint32_t)la != lb) return false;
  int32_t k = 0;
  for (; k + 8 <= la; k += 8) if (load8(a + k) != load8(b + k)) return false;
  if (k < la && ((load8(a + k) ^ load8(b + k)) & (~0ull >> (64 - 8 * (la - k))))) return false;
  return true;
}

// ---- groupreduce's accumulators (the comment at `groupreduce` below says what they are for)
constexpr int kGroupLds = 1024;                      // groups that fit the per-workgroup accumulators
// the value of row `row` as the accumulator sees it: 0 = int64 (wrapping sum / signed order), 1 = uint64, 2 = double
__device__ __forceinline__ uint64_t value_bits(const void* col, int dtype, int64_t row, int& kind) {
  switch (dtype) {
    case DFDB_I8:  kind = 0; return (uint64_t)(int64_t)((const int8_t*)col)[row];
    case DFDB_I16: kind = 0; return (uint64_t)(int64_t)((const int16_t*)col)[row];
    case DFDB_I32: kind = 0; return (uint64_t)(int64_t)((const int32_t*)col)[row];
    case DFDB_I64: kind = 0; return ((const uint64_t*)col)[row];
    case DFDB_U8: case DFDB_BOOL: kind = 1; return ((const uint8_t*)col)[row];
    case DFDB_U16: kind = 1; return ((const uint16_t*)col)[row];
    case DFDB_U32: kind = 1; return ((const uint32_t*)col)[row];
    case DFDB_U64: kind = 1; return ((const uint64_t*)col)[row];
    case DFDB_F32: { kind = 2; const double d = (double)((const float*)col)[row]; return (uint64_t)__double_as_longlong(d); }
    default:       { kind = 2; return ((const uint64_t*)col)[row]; }
  }
}
// order-preserving 64-bit image for min / max (unsigned compare); a NaN maps to the end that wins the reduction
__device__ __forceinline__ uint64_t order_image(uint64_t bits, int kind, int op) {
  if (kind == 1) return bits;
  if (kind == 0) return bits ^ (1ull << 63);
  const double d = __longlong_as_double((long long)bits);
  if (d != d) return op == DFDB_AGG_MIN ? 0ull : ~0ull;
  return (bits >> 63) ? ~bits : (bits | (1ull << 63));
}
__device__ __forceinline__ void group_add(uint64_t* cnt, uint64_t* val, uint64_t gid, uint64_t bits, int kind, int op, bool has_val) {
  atomicAdd((unsigned long long*)&cnt[gid], 1ull);
  if (!has_val) return;
  if (op == DFDB_AGG_SUM) { if (kind == 2) atomicAdd((double*)&val[gid], __longlong_as_double((long long)bits)); else atomicAdd((unsigned long long*)&val[gid], (unsigned long long)bits); }
  else if (op == DFDB_AGG_MIN) atomicMin((unsigned long long*)&val[gid], (unsigned long long)order_image(bits, kind, op));
  else if (op == DFDB_AGG_MAX) atomicMax((unsigned long long*)&val[gid], (unsigned long long)order_image(bits, kind, op));
}
// the same with the operator fixed at compile time (OPK: 0 count only, 1 wrapping integer sum, 2 double sum, 3 min, 4 max): inside the unrolled row loops of the
// String pass the runtime form was a tree of scalar compares and branches per row (a 10 000-line kernel; SGPRs spilled to lanes)
template <int OPK>
__device__ __forceinline__ void group_add_t(uint64_t* cnt, uint64_t* val, uint64_t gid, uint64_t bits, int kind) {
  atomicAdd((unsigned long long*)&cnt[gid], 1ull);
  if (OPK == 1) atomicAdd((unsigned long long*)&val[gid], (unsigned long long)bits);
  if (OPK == 2) atomicAdd((double*)&val[gid], __longlong_as_double((long long)bits));
  if (OPK == 3) atomicMin((unsigned long long*)&val[gid], (unsigned long long)order_image(bits, kind, DFDB_AGG_MIN));
  if (OPK == 4) atomicMax((unsigned long long*)&val[gid], (unsigned long long)order_image(bits, kind, DFDB_AGG_MAX));
}
static int opk_of(int op, bool has_val, int kind) {
  if (!has_val || op == DFDB_AGG_COUNT) return 0;
  if (op == DFDB_AGG_SUM) return kind == 2 ? 2 : 1;
  if (op == DFDB_AGG_MIN) return 3;
  if (op == DFDB_AGG_MAX) return 4;
  return 0;
}
// merge a workgroup's LDS accumulators into the global ones (val_kind: 2 = double sums)
__device__ __forceinline__ void group_flush(const uint64_t* lcnt, const uint64_t* lval, uint64_t* cnt, uint64_t* val, int ngroups, int op, int val_kind, bool has_val, int nthreads = kBlock) {
  for (int g = threadIdx.x; g < ngroups; g += nthreads) {
    const uint64_t c = lcnt[g];
    if (!c) continue;
    atomicAdd((unsigned long long*)&cnt[g], (unsigned long long)c);
    if (!has_val) continue;
    if (op == DFDB_AGG_SUM) { if (val_kind == 2) atomicAdd((double*)&val[g], __longlong_as_double((long long)lval[g])); else atomicAdd((unsigned long long*)&val[g], (unsigned long long)lval[g]); }
    else if (op == DFDB_AGG_MIN) atomicMin((unsigned long long*)&val[g], (unsigned long long)lval[g]);
    else if (op == DFDB_AGG_MAX) atomicMax((unsigned long long*)&val[g], (unsigned long long)lval[g]);
  }
}

// ---- one pass over the selected rows of a String column, a wave per 1024-row tile, in two steps (round 4).
// FAST: the sixteen sizes of a lane in flight together, then per half tile the byte offsets (a wave prefix sum of the sizes per 64 rows) and the first 8 (16) bytes
//   of every selected string, again in flight together — a row whose string the wave has met before (MetCache) is DONE here: nothing to do (insert, verify) or its
//   group is known (accumulate).  The other rows leave a bit in the word's miss mask.
// SLOW: only the words with misses — sizes again, the word's own prefix sum from its recorded start, hash, table, representative, and the cache learns the string.
// Few distinct values: after a wave's first tiles everything is FAST and the pass runs at what the loads allow; many: both steps run, the second as the whole pass did.
// KIND 0: insert {key, row} for the tiles [tile0, tile1); 1: compare every selected row with its slot's representative; 2: groupreduce's accumulate pass (+ that compare)
struct StrPassArgs {
  const uint64_t* sel; const int32_t* sizes; const int64_t* tile_off; const uint8_t* bytes; int64_t nrows, tile0, tile1;
  UniqueEntry* ent; uint64_t* rep_off; uint32_t* rep_len; uint64_t mask; uint64_t* aux; uint64_t salt;
  const void* valcol; int valdt, op; uint64_t* cnt; uint64_t* val; int ngroups; uint64_t val_init;       // KIND 2
  int maybe;                                                   // KIND 1: the table may not hold every string (an optimistic unique): an unknown one raises aux[kAuxAbort]
};
constexpr int kHotGroups = 256;                      // LDS slots for hot groups where every row's value goes through a global atomic (k_group_acc<0>, k_str_pass<2, 0>)
// NGL: groups the workgroup's LDS accumulators hold (0: global atomics — and kHotGroups slots for hot groups, as in k_group_acc<0>); OPK: group_add_t; V8: the value column is 8 bytes wide (loaded as is; the narrow types'
// switch, sixteen copies of it in the unrolled loops, lives in the !V8 kernels only)
template <int KIND, int NGL, int OPK, bool V8>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void k_str_pass(const StrPassArgs A) {   // (four waves per SIMD: 128 VGPRs)
  constexpr bool LDS = NGL > 0;
  __shared__ uint32_t claims_sh;
  constexpr bool HOT = KIND == 2 && !LDS;
  __shared__ uint64_t lcnt[KIND == 2 ? (LDS ? NGL : (int)kHotGroups) : 1], lval[KIND == 2 ? (LDS ? NGL : (int)kHotGroups) : 1], hg_gid[HOT ? (int)kHotGroups : 1];
  __shared__ uint32_t hg_any;
  constexpr int kSlots = kMetSlots;
  static_assert(kSlots == 256 || kSlots == 128 || kSlots == 64, "MetCacheT::slot shifts");
  __shared__ MetEntry met_e[kWavesPerBlock][kSlots];
  __shared__ uint32_t met_c[kWavesPerBlock][kSlots];
  const int lane = lane_id();
  constexpr bool has_val = KIND == 2 && OPK != 0;
  if (KIND == 0 && threadIdx.x == 0) claims_sh = 0;
  if (KIND == 2 && LDS) for (int g = threadIdx.x; g < A.ngroups; g += kBlock) { lcnt[g] = 0; lval[g] = A.val_init; }
  if (HOT) { for (int g = threadIdx.x; g < kHotGroups; g += kBlock) { lcnt[g] = 0; lval[g] = A.val_init; hg_gid[g] = kEmpty; } if (threadIdx.x == 0) hg_any = 0; }
  if (KIND == 0 || KIND == 2) __syncthreads();
  bool hot_on = false;
  // a row of a hot group into the group's LDS slot (true), or not (k_group_acc<0> describes the scheme; volatile: no barrier in these loops)
  auto hot_add = [&](uint64_t gid, uint64_t bits, int kind) -> bool {
    if (!HOT || !hot_on || ((volatile uint64_t*)hg_gid)[gid & (kHotGroups - 1)] != gid) return false;
    group_add_t<OPK>(lcnt, lval, gid & (kHotGroups - 1), bits, kind);
    return true;
  };
  MetCacheT<kSlots> met;
  met.init(met_e[threadIdx.x >> 6], met_c[threadIdx.x >> 6], lane);
  uint32_t claimed = 0;
  int* collision = (int*)(A.aux + 4);
  const uint64_t gid_missing = KIND == 2 ? A.aux[1] : 0ull;
  // (properties of the value column, looked at ONCE: inside the unrolled row loops the dtype switch was a chain of scalar compares per row)
  int vkind = 0; if (KIND == 2 && has_val) (void)value_bits(A.valcol, A.valdt, 0, vkind);
  constexpr bool val8 = KIND == 2 && has_val && V8;
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = A.tile0 + wave; tile < A.tile1; tile += nwaves) {
    if (KIND == 0 && __atomic_load_n(&A.aux[kAuxAbort], __ATOMIC_RELAXED)) break;      // (wave-uniform) too full: the host grows the table and repeats the chunk
    const uint64_t mine = lane < 16 ? A.sel[tile * 16 + lane] : 0ull;
    if (__ballot(mine != 0) == 0) continue;
    if (HOT && !hot_on) hot_on = __builtin_amdgcn_readfirstlane((int)*(volatile uint32_t*)&hg_any) != 0;
    const int64_t base = tile * kTile;
    const bool whole = base + kTile <= A.nrows;
    const uint8_t* tb = A.bytes + A.tile_off[tile];
    int32_t sz[16];
    if (whole) {
#pragma unroll
      for (int j = 0; j < 16; j++) sz[j] = __builtin_nontemporal_load(A.sizes + base + j * 64 + lane);
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) { const int64_t row = base + j * 64 + lane; sz[j] = row < A.nrows ? A.sizes[row] : 0; }
    }
    uint64_t missw = 0; uint32_t wordoff = 0, run = 0;        // lane j: the miss mask / the byte offset (inside the tile) of word j
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t rel[8]; uint64_t head[8], head2[8], bits[KIND == 2 ? 8 : 1]; bool on[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t c = sz[h * 8 + j] > 0 ? (uint32_t)sz[h * 8 + j] : 0u;
        const uint32_t incl = wave_incl_scan(c);
        rel[j] = run + incl - c;
        if (lane == h * 8 + j) wordoff = run;
        run += (uint32_t)__shfl(incl, 63, 64);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint64_t w = __shfl(mine, h * 8 + j, 64);
        on[j] = ((w >> lane) & 1ull) && (whole || base + (h * 8 + j) * 64 + lane < A.nrows);
        const int32_t s0 = sz[h * 8 + j];
        head[j] = (on[j] && s0 > 0) ? load8(tb + rel[j]) : 0ull;
        head2[j] = (on[j] && s0 > 8) ? load8(tb + rel[j] + 8) : 0ull;
        if (KIND == 2) {
          const int64_t row = base + (h * 8 + j) * 64 + lane;
          if (val8) bits[j] = on[j] ? ((const uint64_t*)A.valcol)[row] : 0ull;
          else { int kind = 0; bits[j] = (on[j] && has_val) ? value_bits(A.valcol, A.valdt, row, kind) : 0ull; }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int32_t s0 = sz[h * 8 + j];
        const int64_t row = base + (h * 8 + j) * 64 + lane;
        bool miss = false;
        if (KIND == 0) {                                        // the word's smallest selected missing row
          const uint64_t mm = __ballot(on[j] && s0 < 0);
          if (mm && lane == __builtin_ctzll(mm) && __atomic_load_n(&A.aux[1], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&A.aux[1], (unsigned long long)row);
        }
        // (flat: every lane probes the cache — a lane that is off or holds a missing value probes with zeros and ignores the answer —, ONE predicated region does
        // the accumulation; nested `if`s cost an exec-mask save / branch / restore each, 80 scalar instructions per 64 rows before)
        const bool str = on[j] && s0 >= 0;
        const int32_t sl = str ? s0 : 0;
        uint32_t g32 = 0;
        const bool hit = met.find(low_bytes(head[j], sl), low_bytes(head2[j], sl - 8), (uint32_t)sl, g32, str && s0 <= kMetMaxLen);
        const bool known = str && s0 <= kMetMaxLen && hit;
        miss = str && !known;
        const bool acc = KIND == 2 && on[j] && !miss;
        const uint64_t gidv = s0 >= 0 ? (uint64_t)g32 : gid_missing;
        if (acc) {
          if (LDS) group_add_t<OPK>(lcnt, lval, gidv, bits[j], vkind); else if (!hot_add(gidv, bits[j], vkind)) group_add_t<OPK>(A.cnt, A.val, gidv, bits[j], vkind);
        }
        if (HOT && j == 0) {                                    // a group that holds three of these 64 rows gets a slot (one word in eight is looked at)
          const uint64_t am = __ballot(acc);
          if (am) {
            const int fl = __builtin_ctzll(am);
            const uint64_t fg = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gidv, fl) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gidv >> 32), fl) << 32;
            if (__builtin_popcountll(am & __ballot(acc && gidv == fg)) >= 3 && lane == 0) {
              const uint64_t sl2 = fg & (kHotGroups - 1);
              if (((volatile uint64_t*)hg_gid)[sl2] == kEmpty) { atomicCAS((unsigned long long*)&hg_gid[sl2], (unsigned long long)kEmpty, (unsigned long long)fg); *(volatile uint32_t*)&hg_any = 1; }
            }
          }
        }
        const uint64_t m = __ballot(miss);
        if (lane == h * 8 + j) missw = m;
      }
    }
    if (__ballot(missw != 0) == 0) continue;
#pragma unroll 1
    for (int jj = 0; jj < 16; jj++) {
      const uint64_t m = __shfl(missw, jj, 64);
      if (m == 0) continue;                                     // (wave-uniform)
      const int64_t row = base + jj * 64 + lane;
      const int32_t s0 = row < A.nrows ? A.sizes[row] : 0;
      const uint32_t c = s0 > 0 ? (uint32_t)s0 : 0u;
      const uint32_t incl = wave_incl_scan(c);
      const int64_t off = A.tile_off[tile] + (int64_t)((uint32_t)__shfl(wordoff, jj, 64) + incl - c);
      if (!((m >> lane) & 1ull)) continue;
      const uint8_t* p = A.bytes + off;
      const uint64_t h1 = s0 > 0 ? load8(p) : 0ull, h2 = s0 > 8 ? load8(p + 8) : 0ull;
      const uint64_t c1 = low_bytes(h1, s0), c2 = low_bytes(h2, s0 - 8);
      // the cache once more: the FAST step looked at the whole tile before any of its words came here, so on a wave's first tile every row of every word was a
      // miss — but the words before this one have taught the cache since (a column of ten brands: 1024 hashes, table probes and hot-line atomics per wave became ~64)
      if (s0 <= kMetMaxLen) {
        uint32_t g32 = 0;
        if (met.find(c1, c2, (uint32_t)s0, g32)) {
          if (KIND == 2) {
            int kind = 0; const uint64_t bits = has_val ? value_bits(A.valcol, A.valdt, row, kind) : 0ull;
            if (LDS) group_add_t<OPK>(lcnt, lval, (uint64_t)g32, bits, kind); else if (!hot_add((uint64_t)g32, bits, kind)) group_add_t<OPK>(A.cnt, A.val, (uint64_t)g32, bits, kind);
          }
          continue;
        }
      }
      uint64_t key = hash_bytes(p, s0, A.salt, h1, h2);
      if (key == kEmpty) key = 0x1234567ull;                    // (any fixed remap: equal strings still get equal keys)
      if (KIND == 0) {
        const uint64_t r = table_insert(A.ent, A.mask, key, (uint64_t)row, A.aux, key & A.mask);
        if (r != kNoSlot && !(r >> 63)) { A.rep_off[r] = (uint64_t)off; A.rep_len[r] = (uint32_t)s0; claimed++; }   // I claimed the slot: my bytes represent it
        if (r != kNoSlot && s0 <= kMetMaxLen) met.put(c1, c2, (uint32_t)s0, 1u, lane);
      } else {
        const uint64_t hs = (KIND == 2 || A.maybe) ? table_find_maybe(A.ent, A.mask, key, key & A.mask) : table_find(A.ent, A.mask, key, key & A.mask);
        if (hs == kNoSlot) { __atomic_store_n(&A.aux[kAuxAbort], 1ull, __ATOMIC_RELAXED); continue; }   // a string the (optimistically filled) table does not hold: the host runs everything again
        const bool same = !A.rep_off || same_bytes(p, s0, A.bytes + A.rep_off[hs], A.rep_len[hs]);
        if (!same) atomicOr(collision, 1);                      // two different strings, one key: the host repeats with another salt
        if (KIND == 2) {
          const uint64_t gid = A.ent[hs].row;
          if (same && s0 <= kMetMaxLen && gid < 0xFFFFFFFFull) met.put(c1, c2, (uint32_t)s0, (uint32_t)gid, lane);
          int kind = 0; const uint64_t bits = has_val ? value_bits(A.valcol, A.valdt, row, kind) : 0ull;
          if (LDS) group_add_t<OPK>(lcnt, lval, gid, bits, kind); else if (!hot_add(gid, bits, kind)) group_add_t<OPK>(A.cnt, A.val, gid, bits, kind);
        } else if (same && s0 <= kMetMaxLen) met.put(c1, c2, (uint32_t)s0, 1u, lane);
      }
    }
  }
  if (KIND == 0) add_claims(claimed, &claims_sh, A.aux);
  if (KIND == 2 && LDS) {
    __syncthreads();
    int k2 = 0; if (has_val) (void)value_bits(A.valcol, A.valdt, 0, k2);
    group_flush(lcnt, lval, A.cnt, A.val, A.ngroups, A.op, k2, has_val);
  }
  if (HOT) {                                                   // the hot groups' slots -> their groups
    __syncthreads();
    int k2 = 0; if (has_val) (void)value_bits(A.valcol, A.valdt, 0, k2);
    for (int sl = threadIdx.x; sl < kHotGroups; sl += kBlock) {
      const uint64_t c = lcnt[sl], g = hg_gid[sl];
      if (!c || g == kEmpty) continue;
      atomicAdd((unsigned long long*)&A.cnt[g], (unsigned long long)c);
      if (!has_val) continue;
      if (A.op == DFDB_AGG_SUM) { if (k2 == 2) atomicAdd((double*)&A.val[g], __longlong_as_double((long long)lval[sl])); else atomicAdd((unsigned long long*)&A.val[g], (unsigned long long)lval[sl]); }
      else if (A.op == DFDB_AGG_MIN) atomicMin((unsigned long long*)&A.val[g], (unsigned long long)lval[sl]);
      else if (A.op == DFDB_AGG_MAX) atomicMax((unsigned long long*)&A.val[g], (unsigned long long)lval[sl]);
    }
  }
}

// mark the first occurrences row by row (many distinct values: the table's rows are compared with every selected row's own)
__global__ __launch_bounds__(kBlock) void k_unique_str_mark(uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, const int32_t* __restrict__ sizes,
                                                            const int64_t* __restrict__ tile_off, const uint8_t* __restrict__ bytes, int64_t nrows, int64_t ntiles,
                                                            const UniqueEntry* __restrict__ ent, uint64_t mask, const uint64_t* __restrict__ aux, uint64_t salt) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const uint64_t mine = lane < 16 ? bitmap[tile * 16 + lane] : 0ull;
    if (__ballot(mine != 0) == 0) { if (lane == 0) tile_counts[tile] = 0; continue; }
    uint64_t myword = 0; uint32_t cnt = 0;
    int64_t run = tile_off[tile];
    for (int j = 0; j < 16; j++) {
      const int64_t row = tile * kTile + j * 64 + lane;
      const int32_t sz = row < nrows ? sizes[row] : 0;
      const uint32_t c = sz > 0 ? (uint32_t)sz : 0u;
      const uint32_t incl = wave_incl_scan(c);
      const int64_t off = run + (int64_t)(incl - c);
      run += (int64_t)__shfl(incl, 63, 64);
      const uint64_t w = __shfl(mine, j, 64);
      bool first = false;
      if (row < nrows && ((w >> lane) & 1ull)) {
        if (sz < 0) first = aux[1] == (uint64_t)row;
        else {
          const uint8_t* p = bytes + off;
          uint64_t key = hash_bytes(p, sz, salt, sz > 0 ? load8(p) : 0ull, sz > 8 ? load8(p + 8) : 0ull);
          if (key == kEmpty) key = 0x1234567ull;
          first = ent[table_find(ent, mask, key, key & mask)].row == (uint64_t)row;
        }
      }
      const uint64_t m = __ballot(first);
      if (lane == j) myword = m;
      cnt += (uint32_t)__popcll(m);
    }
    if (lane < 16) bitmap[tile * 16 + lane] = myword;
    if (lane == 0) tile_counts[tile] = cnt;
  }
}

static int grid_rows(int64_t n) { int64_t b = (n + kBlock - 1) / kBlock; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }
static int grid_tiles(int64_t nt) { int64_t b = (nt + kWavesPerBlock - 1) / kWavesPerBlock; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }
// the String passes: a wave's met cache starts cold, so every wave's FIRST tile goes through the slow step whole — with one workgroup per four tiles (grid_tiles)
// that was 10 % of the rows of a 5e8-row column (a tile or fourteen per wave, and all of a 16 M-row chunk).  At least 32 tiles per wave where the column has
// them, never fewer workgroups than fill the chip once
static int grid_str_pass(int64_t nt) {
  int64_t b = nt / (32 * kWavesPerBlock);
  const int64_t fill = 1280;                                          // ~ resident workgroups of these kernels (5 per CU x 256)
  if (b < fill) b = std::min<int64_t>(fill, (nt + kWavesPerBlock - 1) / kWavesPerBlock);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

void launch_unique_insert(hipStream_t s, const uint64_t* bitmap, const void* col, int dtype, const uint64_t* missing, int64_t row0, int64_t row1,
                          UniqueEntry* ent, uint64_t mask, uint64_t* aux) {
  if (row1 <= row0) return;
  hipLaunchKernelGGL(k_unique_insert, dim3(grid_rows(row1 - row0)), dim3(kBlock), 0, s, bitmap, col, dtype, missing, row0, row1, ent, mask, aux);
}
void launch_unique_mark(hipStream_t s, uint64_t* bitmap, uint32_t* tile_counts, const void* col, int dtype, const uint64_t* missing, int64_t nrows,
                        const UniqueEntry* ent, uint64_t mask, const uint64_t* aux) {
  if (nrows <= 0) return;
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  hipLaunchKernelGGL(k_unique_mark, dim3(grid_tiles(ntiles)), dim3(kBlock), 0, s, bitmap, tile_counts, col, dtype, missing, nrows, ntiles, ent, mask, aux);
}
void launch_unique_migrate(hipStream_t s, const UniqueEntry* from, const uint64_t* from_off, const uint32_t* from_len, uint64_t from_cap, UniqueEntry* ent,
                           uint64_t* rep_off, uint32_t* rep_len, uint64_t mask, uint64_t* aux) {
  hipLaunchKernelGGL(k_unique_migrate, dim3(grid_rows((int64_t)from_cap)), dim3(kBlock), 0, s, from, from_off, from_len, from_cap, ent, rep_off, rep_len, mask, aux);
}
void launch_unique_scatter(hipStream_t s, const UniqueEntry* ent, uint64_t cap, const uint64_t* aux, uint64_t* bitmap, uint32_t* tile_counts) {
  hipLaunchKernelGGL(k_unique_scatter, dim3(grid_rows((int64_t)cap + 2)), dim3(kBlock), 0, s, ent, cap, aux, bitmap, tile_counts);
}
void launch_unique_str(hipStream_t s, int pass, uint64_t* bitmap, uint32_t* tile_counts, const int32_t* sizes, const int64_t* tile_off,
                       const uint8_t* bytes, int64_t nrows, int64_t tile0, int64_t tile1, UniqueEntry* ent, uint64_t* rep_off, uint32_t* rep_len, uint64_t mask,
                       uint64_t* aux, uint64_t salt) {
  if (tile1 <= tile0) return;
  const dim3 b(kBlock);
  if (pass == 2) { hipLaunchKernelGGL(k_unique_str_mark, dim3(grid_tiles(tile1 - tile0)), b, 0, s, bitmap, tile_counts, sizes, tile_off, bytes, nrows, tile1, ent, mask, aux, salt); return; }
  const dim3 g(grid_str_pass(tile1 - tile0));
  StrPassArgs A{};
  A.sel = bitmap; A.sizes = sizes; A.tile_off = tile_off; A.bytes = bytes; A.nrows = nrows; A.tile0 = tile0; A.tile1 = tile1;
  A.ent = ent; A.rep_off = rep_off; A.rep_len = rep_len; A.mask = mask; A.aux = aux; A.salt = salt;
  A.maybe = pass == 3 ? 1 : 0;
  if (pass == 0) hipLaunchKernelGGL((k_str_pass<0, 0, 0, false>), g, b, 0, s, A);
  else hipLaunchKernelGGL((k_str_pass<1, 0, 0, false>), g, b, 0, s, A);
}

// ---------------------------------------------------------------- groupreduce (src/tables/aggregate.jl:1-36)
// The reference's groupreduce numbers the groups in order of first appearance of the key (group_map[elem] = length(group_map) + 1) and stops
// there (it is unfinished: it prints the map).  Completed to that intent on top of unique's table: after the unique passes the table maps a
// key to the row of its first occurrence and the bitmap holds exactly those rows, so a group's number is the RANK of its first row among them
// (k_group_ids turns the table's row slots into group numbers, once per group); k_group_accumulate then sends every selected row's value to its
// group's accumulator — privatised in LDS per workgroup when there are few groups (ten brands over 5e8 rows would otherwise be 5e8 atomics on ten
// addresses), global atomics otherwise.  Accumulators are 64-bit: counts, wrapping integer sums (Julia's), double sums, and min / max through an
// order-preserving image (a NaN wins both, like Julia's minimum / maximum).

__device__ __forceinline__ uint64_t rank_of_row(const uint64_t* __restrict__ ubits, const uint64_t* __restrict__ uprefix, uint64_t row) {
  const uint64_t tile = row >> 10, w = (row & 1023) >> 6;
  uint64_t r = uprefix[tile];
  for (uint64_t k = 0; k < w; k++) r += (uint64_t)__popcll(ubits[tile * 16 + k]);
  return r + (uint64_t)__popcll(ubits[tile * 16 + w] & ((1ull << (row & 63)) - 1ull));
}
__global__ __launch_bounds__(kBlock) void k_group_ids(UniqueEntry* __restrict__ ent, uint64_t cap, uint64_t* __restrict__ special,
                                                      const uint64_t* __restrict__ ubits, const uint64_t* __restrict__ uprefix) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < cap && ent[i].key != kEmpty) ent[i].row = rank_of_row(ubits, uprefix, ent[i].row);
  if (i < 2 && special[i] != kEmpty) special[i] = rank_of_row(ubits, uprefix, special[i]);
}

// ---- groupreduce's accumulate pass for keys that are not flat Strings (those: k_str_pass).  SRC 0: the key's group number out of the hash table, 1: out of the
// dictionary codes' rank table, 2: out of the dense form's table.  NG = how many groups the workgroup's own accumulators hold: 1024 (16 KB of LDS, 256 threads, several
// workgroups per CU), 9216 (144 KB, one 1024-thread workgroup per CU: round 4 — 1e9 rows in 5 000 groups were 2e9 global atomics, 85 ms), 0 = global atomics
constexpr int kGroupLdsBig = 9216;
struct AccArgs {
  const uint64_t* sel; const void* keycol; int keydt; const uint64_t* missing; const void* valcol; int valdt, op; int64_t nrows;
  const UniqueEntry* ent; uint64_t mask; const uint64_t* special;       // SRC 0 (special: aux — the groups of the unstorable key and of missing)
  const uint16_t* codes; const uint32_t* rank_of_code;                  // SRC 1
  uint64_t lo; const uint64_t* gids;                                    // SRC 2 (special[1]: the group of missing)
  uint64_t* cnt; uint64_t* val; int ngroups; uint64_t val_init;
  uint64_t* unknown_flag;                                               // k_group_acc_dense_lds: raised by a selected row whose key has no group (an optimistic, head-only table)
};
// OPK: group_add_t's operator; W8: the value column AND (SRC 0 / 2) the key column are 8-byte integers or doubles, loaded as they are — the dtype switches of value_bits
// and key_fixed, copied four times by the unrolled trip, stay in the !W8 kernels
// NG = 0 (more groups than any LDS holds: every row's value through a global atomic) keeps kHotGroups slots in LDS for HOT groups (round 6): a group that a
// large part of the rows belong to was 3e8 atomics on one address — 3.6 s per 1e9 rows.  A group that holds three of the 64 rows a wave looks at is given a slot
// (if its slot is free); from the next trip on a row of a group with a slot is added THERE, and the slots are flushed with one global atomic each when the
// workgroup ends.  (groupreduce by radix — k_radix.hip — does the same in its partition pass; this is for what it does not take: String keys, > 2.4 M groups.)
template <int NG, int SRC, int OPK, bool W8>
__global__ __launch_bounds__(NG > kGroupLds ? 1024 : kBlock) void k_group_acc(const AccArgs A) {
  __shared__ uint64_t lcnt[NG ? NG : (int)kHotGroups], lval[NG ? NG : (int)kHotGroups];      // (NG = 0: the hot groups' slots)
  __shared__ uint64_t hg_gid[NG ? 1 : (int)kHotGroups];
  __shared__ uint32_t hg_any;
  if (!NG) { for (int g = threadIdx.x; g < kHotGroups; g += kBlock) { lcnt[g] = 0; lval[g] = A.val_init; hg_gid[g] = kEmpty; } if (threadIdx.x == 0) hg_any = 0; __syncthreads(); }
  bool hot_on = false;
  const int nthreads = NG > kGroupLds ? 1024 : kBlock;
  const bool has_val = OPK < 0 ? (A.valcol != nullptr && A.op != DFDB_AGG_COUNT) : OPK != 0;      // (OPK -1: the operator at run time — the !W8 kernels)
  if (NG) { for (int g = threadIdx.x; g < A.ngroups; g += nthreads) { lcnt[g] = 0; lval[g] = A.val_init; } __syncthreads(); }
  // four rows per thread and trip, stage by stage (selection bits, keys, group numbers, values, adds): a row's loads depend on each other, the four rows' do not —
  // one row at a time the pass ran at the latency of three dependent loads per trip (5e8 rows by dictionary codes: 5 ms)
  constexpr int U = 4;          // (eight: the same for integer keys, 1.6x slower on dictionary codes — measured)
  const int64_t stride = (int64_t)gridDim.x * nthreads;
  int vkind = 0; if (has_val) (void)value_bits(A.valcol, A.valdt, 0, vkind);      // (the value kind is a property of the column)
  bool unknown = false;
  for (int64_t row0 = (int64_t)blockIdx.x * nthreads + threadIdx.x; row0 < A.nrows; row0 += U * stride) {
    bool on[U], miss[U]; uint64_t key[U], gid[U], bits[U], w[U], mw[U];
    (void)unknown;
    // every load of a trip is issued before any is looked at (round 5): the selection word, the key and the value of a row do not wait for each other — a key loaded only
    // where its selection bit is set is a second dependent round trip per trip (rows that are off read a key nobody uses)
#pragma unroll
    for (int k = 0; k < U; k++) {
      const int64_t row = row0 + k * stride;
      const bool inb = row < A.nrows;
      w[k] = inb ? A.sel[row >> 6] : 0ull;
      mw[k] = (SRC != 1 && inb && A.missing) ? A.missing[row >> 6] : 0ull;
      if (SRC == 1) key[k] = inb ? (uint64_t)A.codes[row] : 0ull;
      else if (W8 && A.keydt != DFDB_F64) key[k] = inb ? ((const uint64_t*)A.keycol)[row] : 0ull;
      else key[k] = inb ? key_fixed(A.keycol, A.keydt, row) : 0ull;
      if (W8) bits[k] = (inb && has_val) ? ((const uint64_t*)A.valcol)[row] : 0ull;
      else { int kind = 0; bits[k] = (inb && has_val) ? value_bits(A.valcol, A.valdt, row, kind) : 0ull; }
    }
#pragma unroll
    for (int k = 0; k < U; k++) { const int64_t row = row0 + k * stride; on[k] = (w[k] >> (row & 63)) & 1ull; miss[k] = on[k] && ((mw[k] >> (row & 63)) & 1ull); }
#pragma unroll
    for (int k = 0; k < U; k++) {
      gid[k] = 0;
      if (!on[k]) continue;
      if (SRC == 1) gid[k] = A.rank_of_code[key[k]];
      else if (miss[k]) gid[k] = A.special[1];
      else if (SRC == 2) gid[k] = A.gids[key[k] - A.lo];
      else if (key[k] == kEmpty) gid[k] = A.special[0];
      else gid[k] = table_row_maybe(A.ent, A.mask, key[k], slot_of(key[k], A.mask));      // (kEmpty: no slot — only an optimistically filled table can say that)
      // (no group: the key, `missing` or the unstorable key never turned up among the rows the table was made from — the host runs everything again)
      if (SRC == 0 && gid[k] >= (uint64_t)A.ngroups) { unknown = true; on[k] = false; }
    }
    if (!NG) {                                                 // hot groups: rows of a group that has a slot are added in LDS
      // (volatile: there is no barrier in this loop, and a plain read of what another wave sets was hoisted out of it — only the wave that gave a slot away ever used it)
      if (!hot_on) hot_on = __builtin_amdgcn_readfirstlane((int)*(volatile uint32_t*)&hg_any) != 0;
#pragma unroll
      for (int k = 0; k < U; k++) {
        if (hot_on && on[k] && ((volatile uint64_t*)hg_gid)[gid[k] & (kHotGroups - 1)] == gid[k]) {
          const uint64_t sl = gid[k] & (kHotGroups - 1);
          if (OPK < 0) group_add(lcnt, lval, sl, bits[k], vkind, A.op, has_val); else group_add_t<(OPK < 0 ? 0 : OPK)>(lcnt, lval, sl, bits[k], vkind);
          on[k] = false;
        }
        const uint64_t m = __ballot(on[k]);
        if (k == 0 && m) {                                       // (one of the trip's four rows is looked at)
          const int fl = __builtin_ctzll(m);
          const uint64_t fg = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gid[k], fl) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gid[k] >> 32), fl) << 32;
          if (__builtin_popcountll(m & __ballot(on[k] && gid[k] == fg)) >= 3 && (threadIdx.x & 63) == 0) {
            const uint64_t sl = fg & (kHotGroups - 1);
            if (((volatile uint64_t*)hg_gid)[sl] == kEmpty) { atomicCAS((unsigned long long*)&hg_gid[sl], (unsigned long long)kEmpty, (unsigned long long)fg); *(volatile uint32_t*)&hg_any = 1; }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (!on[k]) continue;
      if (OPK < 0) { if (NG) group_add(lcnt, lval, gid[k], bits[k], vkind, A.op, has_val); else group_add(A.cnt, A.val, gid[k], bits[k], vkind, A.op, has_val); }
      else if (NG) group_add_t<(OPK < 0 ? 0 : OPK)>(lcnt, lval, gid[k], bits[k], vkind); else group_add_t<(OPK < 0 ? 0 : OPK)>(A.cnt, A.val, gid[k], bits[k], vkind);
    }
  }
  if (SRC == 0 && unknown && A.unknown_flag) __atomic_store_n(A.unknown_flag, 1ull, __ATOMIC_RELAXED);
  if (NG) {
    __syncthreads();
    int k2 = 0; if (has_val) (void)value_bits(A.valcol, A.valdt, 0, k2);      // (the value kind is a property of the column)
    group_flush(lcnt, lval, A.cnt, A.val, A.ngroups, A.op, k2, has_val, nthreads);
  } else {                                                     // the hot groups' slots -> their groups (group_flush, a slot's group from hg_gid)
    __syncthreads();
    int k2 = 0; if (has_val) (void)value_bits(A.valcol, A.valdt, 0, k2);
    for (int sl = threadIdx.x; sl < kHotGroups; sl += nthreads) {
      const uint64_t c = lcnt[sl], g = hg_gid[sl];
      if (!c || g == kEmpty) continue;
      atomicAdd((unsigned long long*)&A.cnt[g], (unsigned long long)c);
      if (!has_val) continue;
      if (A.op == DFDB_AGG_SUM) { if (k2 == 2) atomicAdd((double*)&A.val[g], __longlong_as_double((long long)lval[sl])); else atomicAdd((unsigned long long*)&A.val[g], (unsigned long long)lval[sl]); }
      else if (A.op == DFDB_AGG_MIN) atomicMin((unsigned long long*)&A.val[g], (unsigned long long)lval[sl]);
      else if (A.op == DFDB_AGG_MAX) atomicMax((unsigned long long*)&A.val[g], (unsigned long long)lval[sl]);
    }
  }
}
template <int SRC>
static void launch_group_acc(hipStream_t s, const AccArgs& A) {
  if (A.nrows <= 0) return;
  const int64_t b256 = (A.nrows + kBlock - 1) / kBlock;
  int vkind = 0;
  switch (A.valdt) {                                           // (as value_bits sees the column)
    case DFDB_I8: case DFDB_I16: case DFDB_I32: case DFDB_I64: vkind = 0; break;
    case DFDB_U8: case DFDB_BOOL: case DFDB_U16: case DFDB_U32: case DFDB_U64: vkind = 1; break;
    default: vkind = 2; break;
  }
  const int opk = opk_of(A.op, A.valcol != nullptr, vkind);
  const bool v8 = opk == 0 || A.valdt == DFDB_I64 || A.valdt == DFDB_U64 || A.valdt == DFDB_F64;
  const bool k8 = SRC == 1 || A.keydt == DFDB_I64 || A.keydt == DFDB_U64 || A.keydt == DFDB_F64;
  auto go = [&](auto opk_c, auto w8_c) {
    constexpr int OPK = decltype(opk_c)::value; constexpr bool W8 = decltype(w8_c)::value;
    if (A.ngroups <= kGroupLds) hipLaunchKernelGGL((k_group_acc<kGroupLds, SRC, OPK, W8>), dim3((unsigned)std::min<int64_t>(2048, std::max<int64_t>(1, b256))), dim3(kBlock), 0, s, A);
    else if (A.ngroups <= kGroupLdsBig) hipLaunchKernelGGL((k_group_acc<kGroupLdsBig, SRC, OPK, W8>), dim3((unsigned)std::min<int64_t>(256, std::max<int64_t>(1, (A.nrows + 1023) / 1024))), dim3(1024), 0, s, A);
    else hipLaunchKernelGGL((k_group_acc<0, SRC, OPK, W8>), dim3((unsigned)std::min<int64_t>(8192, std::max<int64_t>(1, b256))), dim3(kBlock), 0, s, A);
  };
  auto by_w = [&](auto opk_c) { go(opk_c, std::true_type{}); };
  if (!(v8 && k8)) { go(std::integral_constant<int, -1>{}, std::false_type{}); return; }      // narrow keys or values: one kernel per (NG, SRC), operator at run time
  switch (opk) {
    case 0: by_w(std::integral_constant<int, 0>{}); break;
    case 1: by_w(std::integral_constant<int, 1>{}); break;
    case 2: by_w(std::integral_constant<int, 2>{}); break;
    case 3: by_w(std::integral_constant<int, 3>{}); break;
    default: by_w(std::integral_constant<int, 4>{}); break;
  }
}

void launch_group_ids(hipStream_t s, UniqueEntry* ent, uint64_t cap, uint64_t* special, const uint64_t* ubits, const uint64_t* uprefix) {
  hipLaunchKernelGGL(k_group_ids, dim3((unsigned)((cap + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, ent, cap, special, ubits, uprefix);
}
// ---- K9 fast path: the key is a String column with a dictionary (k_dict.hip) — its 16-bit codes ARE group labels.  What is left of `unique` is the
// first selected row of every code (the reference numbers groups by first appearance), and `groupreduce` accumulates by rank_of_code[code]: no hash
// table (the generic path sizes one by the number of selected rows: 30 GB for 5e8 rows of ten brands).
// A wave walks its tiles and the rows of a tile in increasing order, so the first time it meets a code is its smallest row for that code: a per-wave
// bit set in LDS keeps every later meeting away from the global atomicMin (ten brands over 5e8 rows: <= waves x 10 atomics, not 5e8).
__global__ __launch_bounds__(kBlock) void k_dict_first_rows(const uint64_t* __restrict__ sel, const uint16_t* __restrict__ codes, int64_t nrows, int64_t tile0, int64_t ntiles,
                                                            unsigned long long* __restrict__ first, int lut_words) {
  __shared__ uint32_t seen_sh[kBlock / 64][2048];
  uint32_t* seen = seen_sh[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  for (int k = lane; k < lut_words; k += 64) seen[k] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t tile = tile0 + wave; tile < ntiles; tile += nwaves) {
    const uint64_t mine = lane < 16 ? sel[tile * 16 + lane] : 0ull;
    if (__ballot(mine != 0) == 0) continue;
    // a lane takes 8 CONSECUTIVE rows per half tile (one 16-byte load: 1 KB per wave instruction; 2-byte loads per lane were 128 B per instruction, 1.9 ms per
    // 5e8 rows).  Rows are no longer met in increasing order inside a half tile, so every row of it is checked against what the wave had seen BEFORE it, and only
    // then do the unseen ones record their row and set their bit
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int64_t base = tile * 1024 + h * 512 + lane * 8;
      uint16_t c[8];
      if (tile * 1024 + 1024 <= nrows) { typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); const u32x4 v = __builtin_nontemporal_load((const u32x4*)(codes + base)); __builtin_memcpy(c, &v, 16); }
      else {
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = base + k < nrows ? codes[base + k] : (uint16_t)0;
      }
      const uint64_t w = __shfl(mine, h * 8 + (lane >> 3), 64);               // the selection word that holds this lane's 8 rows
      const uint32_t bits8 = (uint32_t)(w >> ((lane & 7) * 8)) & 0xFFu;
      uint32_t need = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t cc = c[k];
        if (((bits8 >> k) & 1u) && base + k < nrows && !((seen[cc >> 5] >> (cc & 31u)) & 1u)) need |= 1u << k;
      }
      if (__ballot(need != 0) == 0) continue;
      asm volatile("" ::: "memory");                                           // (every check above reads `seen` before any lane below writes it)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (!((need >> k) & 1u)) continue;
        const uint32_t cc = c[k];
        // (behind a plain load: a wave's first half tile sends all its 512 rows here, 8192 waves at once onto as few addresses as there are codes)
        if (__atomic_load_n(&first[cc], __ATOMIC_RELAXED) > (unsigned long long)(base + k)) atomicMin(&first[cc], (unsigned long long)(base + k));
        atomicOr(&seen[cc >> 5], 1u << (cc & 31u));
      }
    }
  }
}
// (the tiles [tile0, tile1): a first row only ever gets smaller, so the column can be walked in pieces)
void launch_dict_first_rows(hipStream_t s, const uint64_t* sel, const uint16_t* codes, int64_t nrows, uint64_t* first, int dict_n, int64_t tile0, int64_t tile1) {
  const int64_t ntiles = std::min<int64_t>((nrows + 1023) / 1024, tile1);
  if (ntiles <= tile0) return;
  // every wave pays one global atomicMin per code it meets, all of them on the same few words: 8192 waves x 10 brands were 0.4 ms of serialised atomics whatever the
  // rows.  A short range (the head that dict_unique walks first) goes to few waves with 32 tiles each — their later tiles are filtered by the wave's own seen-set
  int g = grid_tiles(ntiles - tile0) > 2048 ? 2048 : grid_tiles(ntiles - tile0);
  if (ntiles - tile0 <= 8192) g = (int)std::max<int64_t>(1, (ntiles - tile0 + 127) / 128);
  hipLaunchKernelGGL(k_dict_first_rows, dim3(g), dim3(kBlock), 0, s, sel, codes, nrows, tile0, ntiles, (unsigned long long*)first, (dict_n + 31) / 32);
}
// the bitmap that holds exactly `rows` (ascending, n <= 65 535): what unique leaves behind
__global__ void k_set_rows(const uint64_t* __restrict__ rows, int n, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t r = rows[i];
  atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63));
  atomicAdd(&tile_counts[r >> 10], 1u);
}
void launch_set_rows(hipStream_t s, const uint64_t* rows, int n, uint64_t* bitmap, uint32_t* tile_counts) {
  if (n > 0) hipLaunchKernelGGL(k_set_rows, dim3((n + 255) / 256), dim3(256), 0, s, rows, n, bitmap, tile_counts);
}
void launch_group_accumulate_codes(hipStream_t s, const uint64_t* sel, const uint16_t* codes, const uint32_t* rank_of_code, const void* valcol, int valdt, int op,
                                   int64_t nrows, uint64_t* cnt, uint64_t* val, int64_t ngroups, uint64_t val_init) {
  AccArgs A{};
  A.sel = sel; A.codes = codes; A.rank_of_code = rank_of_code; A.valcol = valcol; A.valdt = valdt; A.op = op; A.nrows = nrows; A.cnt = cnt; A.val = val; A.ngroups = (int)ngroups; A.val_init = val_init;
  launch_group_acc<1>(s, A);
}

// Keys that take the hash table (Float64, wide integers), a few thousand groups: the accumulate pass probed the GLOBAL table per row — a 64-bit mix, a 16-byte read
// out of L2, a dependent round trip — and ran at 2 TB/s.  The groups' keys (the key column at the groups' first rows: `gkeys`, group order) are put into a small
// open-addressing table IN LDS by every workgroup, beside its accumulators, and a row's group is a few ds_reads away (k_group_acc_dense_lds's shape otherwise).
// Dynamic LDS: [ngp] counts, [ngp] values, [slots] 8-byte keys, [slots] 2-byte group numbers.  8-byte keys, 8-byte values (or none), at most 9216 groups.
__device__ __forceinline__ uint32_t lds_slot_of(uint64_t key, uint32_t slots) {
  uint32_t h = ((uint32_t)key ^ (uint32_t)(key >> 32) * 0x85EBCA77u) * 0x9E3779B1u;
  h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13;
  return (uint32_t)(((uint64_t)h * slots) >> 32) & ~1u;      // (an EVEN slot: a probe of the accumulate pass reads two slots at a time; slots is a multiple of 4)
}
template <int OPK>
__global__ __launch_bounds__(1024) void k_group_acc_hash_lds(const AccArgs A, const void* __restrict__ gkeys, uint32_t slots, int ngp) {
  extern __shared__ uint64_t dyn_sh[];
  // (a workgroup meets fewer than 2^32 rows: its counts are 4 bytes, which leaves the key table more room — a lower load factor, shorter probe sequences)
  uint64_t* lval = dyn_sh; uint64_t* lkey = dyn_sh + ngp; uint32_t* lcnt = (uint32_t*)(lkey + slots); uint16_t* lgid = (uint16_t*)(lcnt + ngp);
  constexpr bool has_val = OPK != 0;
  for (int g = threadIdx.x; g < A.ngroups; g += 1024) { lcnt[g] = 0; lval[g] = A.val_init; }
  for (uint32_t i = threadIdx.x; i < slots; i += 1024) lkey[i] = kEmpty;
  __syncthreads();
  const uint64_t g_unstorable = A.special[0], g_missing = A.special[1];       // the groups of the key that cannot be stored / of `missing` (kEmpty: there is none)
  for (int g = threadIdx.x; g < A.ngroups; g += 1024) {
    if ((uint64_t)g == g_unstorable || (uint64_t)g == g_missing) continue;     // (their first rows hold no key of the table)
    const uint64_t key = key_fixed(gkeys, A.keydt, g);
    uint32_t h = lds_slot_of(key, slots);
    for (;;) {
      const uint64_t old = atomicCAS((unsigned long long*)&lkey[h], (unsigned long long)kEmpty, (unsigned long long)key);
      if (old == kEmpty || old == key) { lgid[h] = (uint16_t)g; break; }
      h = h + 1 == slots ? 0u : h + 1;
    }
  }
  __syncthreads();
  constexpr int U = 8;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  int vkind = 0; if (has_val) (void)value_bits(A.valcol, A.valdt, 0, vkind);
  const bool fkey = A.keydt == DFDB_F64;
  bool unknown = false;
  for (int64_t row0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; row0 < A.nrows; row0 += U * stride) {
    uint64_t w[U], mw[U], key[U], bits[U]; uint64_t gid[U];
#pragma unroll
    for (int k = 0; k < U; k++) {                              // every load of the trip before any is used
      const int64_t row = row0 + k * stride;
      const bool inb = row < A.nrows;
      w[k] = inb ? A.sel[row >> 6] : 0ull;
      mw[k] = (inb && A.missing) ? A.missing[row >> 6] : 0ull;
      key[k] = inb ? ((const uint64_t*)A.keycol)[row] : 0ull;
      bits[k] = (inb && has_val) ? ((const uint64_t*)A.valcol)[row] : 0ull;
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      const int64_t row = row0 + k * stride;
      const bool on = (w[k] >> (row & 63)) & 1ull, miss = (mw[k] >> (row & 63)) & 1ull;
      uint64_t kk = key[k];
      if (fkey) { const double d = __longlong_as_double((long long)kk); if (d != d) kk = 0x7ff8000000000000ull; }      // (one NaN: key_fixed)
      uint64_t g = kEmpty;
      if (on) {
        if (miss) g = g_missing;
        else if (kk == kEmpty) g = g_unstorable;
        else {
          uint32_t h = lds_slot_of(kk, slots);
          for (;;) {                                             // two slots per probe (one 16-byte read): half the trips of a loop the wave leaves with its slowest lane
            const ulonglong2 t = *(const ulonglong2*)&lkey[h];
            if (t.x == kk) { g = lgid[h]; break; }
            if (t.x == kEmpty) break;                          // not among the groups: only a table made from a prefix of the rows can say that
            if (t.y == kk) { g = lgid[h + 1]; break; }
            if (t.y == kEmpty) break;
            h = h + 2 >= slots ? 0u : h + 2;
          }
        }
        if (g >= (uint64_t)A.ngroups) { unknown = true; g = kEmpty; }
      }
      gid[k] = g;
    }
#pragma unroll
    for (int k = 0; k < U; k++) if (gid[k] != kEmpty) {
      atomicAdd(&lcnt[gid[k]], 1u);
      if (OPK == 1) atomicAdd((unsigned long long*)&lval[gid[k]], (unsigned long long)bits[k]);
      if (OPK == 2) atomicAdd((double*)&lval[gid[k]], __longlong_as_double((long long)bits[k]));
      if (OPK == 3) atomicMin((unsigned long long*)&lval[gid[k]], (unsigned long long)order_image(bits[k], vkind, DFDB_AGG_MIN));
      if (OPK == 4) atomicMax((unsigned long long*)&lval[gid[k]], (unsigned long long)order_image(bits[k], vkind, DFDB_AGG_MAX));
    }
  }
  if (unknown && A.unknown_flag) __atomic_store_n(A.unknown_flag, 1ull, __ATOMIC_RELAXED);
  __syncthreads();
  for (int g = threadIdx.x; g < A.ngroups; g += 1024) {         // (group_flush with 4-byte counts)
    const uint32_t c = lcnt[g];
    if (!c) continue;
    atomicAdd((unsigned long long*)&A.cnt[g], (unsigned long long)c);
    if (OPK == 1) atomicAdd((unsigned long long*)&A.val[g], (unsigned long long)lval[g]);
    if (OPK == 2) atomicAdd((double*)&A.val[g], __longlong_as_double((long long)lval[g]));
    if (OPK == 3) atomicMin((unsigned long long*)&A.val[g], (unsigned long long)lval[g]);
    if (OPK == 4) atomicMax((unsigned long long*)&A.val[g], (unsigned long long)lval[g]);
  }
}
template <int OPK>
static bool try_hash_lds(hipStream_t s, const AccArgs& A, const void* gkeys) {
  const int ngp = (A.ngroups + 1) & ~1;
  // LDS: [ngp] 8-byte values, [slots] 8-byte keys, [ngp] 4-byte counts, [slots] 2-byte group numbers
  const size_t budget = 156 * 1024, acc = (size_t)ngp * 12;
  if (acc + 64 * 10 > budget) return false;
  size_t slots = std::min<size_t>((size_t)A.ngroups * 3 + 64, (budget - acc) / 10);
  slots &= ~(size_t)3;
  if (slots < (size_t)A.ngroups + A.ngroups / 4 + 8) return false;             // (a load factor above 0.8: the probes get long)
  const size_t lds = acc + slots * 10;
  static std::atomic<bool> raised[64] = {};
  int dev = 0; (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || !raised[dev].load(std::memory_order_acquire))) {
    if (hipFuncSetAttribute((const void*)k_group_acc_hash_lds<OPK>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (dev >= 0 && dev < 64) raised[dev].store(true, std::memory_order_release);
  }
  const unsigned g = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, (A.nrows + 1023) / 1024));
  hipLaunchKernelGGL((k_group_acc_hash_lds<OPK>), dim3(g), dim3(1024), lds, s, A, gkeys, (uint32_t)slots, ngp);
  return true;
}
int launch_group_accumulate(hipStream_t s, const uint64_t* sel, const void* keycol, int keydt, const uint64_t* missing, const void* valcol, int valdt, int op,
                             int64_t nrows, const UniqueEntry* ent, uint64_t mask, const uint64_t* special, uint64_t* cnt, uint64_t* val,
                             int64_t ngroups, uint64_t val_init, uint64_t* unknown_flag, const void* gkeys) {
  AccArgs A{};
  A.unknown_flag = unknown_flag;                               // the table was filled from a prefix of the rows: a key without a slot raises this word
  A.sel = sel; A.keycol = keycol; A.keydt = keydt; A.missing = missing; A.valcol = valcol; A.valdt = valdt; A.op = op; A.nrows = nrows; A.ent = ent; A.mask = mask; A.special = special;
  A.cnt = cnt; A.val = val; A.ngroups = (int)ngroups; A.val_init = val_init;
  if (gkeys && nrows > 0 && ngroups > 0 && (keydt == DFDB_I64 || keydt == DFDB_U64 || keydt == DFDB_F64)) {      // (1: the form with the groups' keys in an LDS table)
    const int vkind = valdt == DFDB_F64 ? 2 : (valdt == DFDB_U64 ? 1 : 0);
    const int opk = opk_of(op, valcol != nullptr, vkind);
    const bool v8 = opk == 0 || valdt == DFDB_I64 || valdt == DFDB_U64 || valdt == DFDB_F64;
    bool done = false;
    if (v8) switch (opk) {
      case 0: done = try_hash_lds<0>(s, A, gkeys); break;
      case 1: done = try_hash_lds<1>(s, A, gkeys); break;
      case 2: done = try_hash_lds<2>(s, A, gkeys); break;
      case 3: done = try_hash_lds<3>(s, A, gkeys); break;
      default: done = try_hash_lds<4>(s, A, gkeys); break;
    }
    if (done) return 1;
  }
  launch_group_acc<0>(s, A);
  return 0;
}
// String keys: k_str_pass, KIND 2
void launch_group_accumulate_str(hipStream_t s, const uint64_t* sel, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const void* valcol, int valdt,
                                 int op, int64_t nrows, const UniqueEntry* ent, const uint64_t* rep_off, const uint32_t* rep_len, uint64_t mask, uint64_t* special, uint64_t salt,
                                 uint64_t* cnt, uint64_t* val, int64_t ngroups, uint64_t val_init) {
  if (nrows <= 0) return;
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  StrPassArgs A{};
  A.sel = sel; A.sizes = sizes; A.tile_off = tile_off; A.bytes = bytes; A.nrows = nrows; A.tile0 = 0; A.tile1 = ntiles;
  A.ent = const_cast<UniqueEntry*>(ent); A.rep_off = const_cast<uint64_t*>(rep_off); A.rep_len = const_cast<uint32_t*>(rep_len); A.mask = mask; A.aux = special; A.salt = salt;
  A.valcol = valcol; A.valdt = valdt; A.op = op; A.cnt = cnt; A.val = val; A.ngroups = (int)ngroups; A.val_init = val_init;
  // the LDS forms run ONE round of workgroups: as many as are resident at once (a workgroup's met caches warm up on its first tiles and its accumulators are
  // flushed once, and 2048 workgroups over 768 resident places left a third of the chip idle in the last round)
  auto one_round = [&](const void* fn) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 3; }
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int g = per_cu * cus;
    return (int)std::min<int64_t>(g, std::max<int64_t>(1, (ntiles + kWavesPerBlock - 1) / kWavesPerBlock));
  };
  int vkind = 0;
  switch (valdt) {                                             // (as value_bits sees the column)
    case DFDB_I8: case DFDB_I16: case DFDB_I32: case DFDB_I64: vkind = 0; break;
    case DFDB_U8: case DFDB_BOOL: case DFDB_U16: case DFDB_U32: case DFDB_U64: vkind = 1; break;
    default: vkind = 2; break;
  }
  const int opk = opk_of(op, valcol != nullptr, vkind);
  auto go = [&](auto ngl, auto opk_c) {
    constexpr int NGL = decltype(ngl)::value, OPK = decltype(opk_c)::value;
    if (OPK != 0 && (valdt == DFDB_I64 || valdt == DFDB_U64 || valdt == DFDB_F64)) {
      const int g = NGL ? one_round((const void*)k_str_pass<2, NGL, OPK, true>) : grid_str_pass(ntiles);
      hipLaunchKernelGGL((k_str_pass<2, NGL, OPK, true>), dim3(g), dim3(kBlock), 0, s, A);
    } else {
      const int g = NGL ? one_round((const void*)k_str_pass<2, NGL, OPK, false>) : grid_str_pass(ntiles);
      hipLaunchKernelGGL((k_str_pass<2, NGL, OPK, false>), dim3(g), dim3(kBlock), 0, s, A);
    }
  };
  auto by_op = [&](auto ngl) {
    switch (opk) {
      case 0: go(ngl, std::integral_constant<int, 0>{}); break;
      case 1: go(ngl, std::integral_constant<int, 1>{}); break;
      case 2: go(ngl, std::integral_constant<int, 2>{}); break;
      case 3: go(ngl, std::integral_constant<int, 3>{}); break;
      default: go(ngl, std::integral_constant<int, 4>{}); break;
    }
  };
  if (ngroups <= 64) by_op(std::integral_constant<int, 64>{});
  else if (ngroups <= kGroupLds) by_op(std::integral_constant<int, kGroupLds>{});
  else by_op(std::integral_constant<int, 0>{});
}
// accumulators -> results: min / max images back to values (in place)
__global__ void k_group_finish(uint64_t* val, int64_t ng, int kind, int op) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng || (op != DFDB_AGG_MIN && op != DFDB_AGG_MAX)) return;
  const uint64_t im = val[g];
  uint64_t bits;
  if (kind == 1) bits = im;
  else if (kind == 0) bits = im ^ (1ull << 63);
  else if (im == 0ull || im == ~0ull) bits = 0x7ff8000000000000ull;                      // the NaN images
  else bits = (im >> 63) ? (im & ~(1ull << 63)) : ~im;
  val[g] = bits;
}
void launch_group_finish(hipStream_t s, uint64_t* val, int64_t ng, int kind, int op) {
  if (ng <= 0) return;
  hipLaunchKernelGGL(k_group_finish, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, val, ng, kind, op);
}

// ---------------------------------------------------------------- integer keys of a small range: no hash table (round 4)
// max - min of the selected keys (one pass at memory speed, k_dense_minmax) fits kDenseRange: the key IS its slot.
//   presence  one 1024-thread workgroup per CU owns a bit per value in LDS (156 KB: 1 277 952 values); the column streams by once, every selected row
//             sets its bit (ds_or, no return), the workgroups OR their bits into one global set at the end -> the distinct values and their number D;
//   first     the row of every value's first occurrence: the rows again IN ORDER, a million or a few at a time (launches of 1 M, 1 M, 2 M, 4 M, 16 M, 64 M ... rows), a
//             guarded 64-bit atomicMin per row into first[value]; every launch starts by comparing the number of values whose first row is known with
//             D and returns at once when they are equal, which for keys that are spread over the column happens after the first launch or two (1e9 rows
//             of 1e6 uniform values: all met within 20 M rows) — in the worst case (a value that turns up late) the column is read a second time;
//   mark      one bit + one tile count per value, scattered into a cleared bitmap.
// unique over 1e9 Int64 rows of 1e6 values: 86 ms through the row-sized hash table of round 3, 2.9 ms so.
constexpr int kDenseBlock = 1024;
constexpr int kDenseWords = 39936;                                   // u32 words of LDS per workgroup (156 KB of the CU's 160)
constexpr int64_t kDenseRange = (int64_t)kDenseWords * 32;
constexpr int kAuxMissing = 1, kAuxOutside = 5, kAuxDistinct = 6, kAuxFound = 7, kAuxMin = 8, kAuxMax = 9, kAuxFoundBefore = 10;
constexpr int kAuxSpanLo = 11, kAuxSpanHi = 12;            // smallest / largest value index with its presence bit set (k_dense_count): the span the keys really cover

int64_t unique_dense_max_range() { return kDenseRange; }
bool unique_dense_dtype(int dtype) {
  switch (dtype) { case DFDB_I8: case DFDB_I16: case DFDB_I32: case DFDB_I64: case DFDB_U8: case DFDB_U16: case DFDB_U32: case DFDB_U64: case DFDB_BOOL: return true; default: return false; }
}
template <typename T> __device__ __forceinline__ uint64_t widen_key(T x) { return (T)-1 < (T)0 ? (uint64_t)(int64_t)x : (uint64_t)x; }

// f(key image, row) for every selected, non-missing row of the tiles [tile0, tile1) this wave owns, sixteen loads in flight per lane;
// g(row) for the smallest selected missing row of a 64-row word (lanes 0-15, one word each)
template <typename T, typename F, typename G>
__device__ __forceinline__ void walk_selected(const uint64_t* __restrict__ bitmap, const T* __restrict__ col, const uint64_t* __restrict__ missing, int64_t nrows,
                                              int64_t tile0, int64_t tile1, int64_t wave, int64_t nwaves, F&& f, G&& g) {
  const int lane = lane_id();
  for (int64_t tile = tile0 + wave; tile < tile1; tile += nwaves) {
    const uint64_t sel = lane < 16 ? bitmap[tile * 16 + lane] : 0ull;
    const uint64_t ms = (missing && lane < 16) ? missing[tile * 16 + lane] : 0ull;
    const int64_t base = tile * kTile;
    if (sel & ms) { const int64_t row = base + lane * 64 + __builtin_ctzll(sel & ms); if (row < nrows) g(row); }
    const uint64_t live = sel & ~ms;
    if (__ballot(live != 0) == 0) continue;
    T v[16];
    if (base + kTile <= nrows) {
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(col + base + j * 64 + lane);
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) { const int64_t row = base + j * 64 + lane; v[j] = row < nrows ? col[row] : (T)0; }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t w = __shfl(live, j, 64);
      const int64_t row = base + j * 64 + lane;
      if (((w >> lane) & 1ull) && row < nrows) f(widen_key(v[j]), row);
    }
  }
}
__device__ __forceinline__ void note_missing(uint64_t* aux, int64_t row) {
  if (__atomic_load_n(&aux[kAuxMissing], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&aux[kAuxMissing], (unsigned long long)row);
}
__device__ __forceinline__ uint64_t wave_min64(uint64_t v) { for (int d = 32; d; d >>= 1) { const uint64_t o = __shfl_xor(v, d, 64); v = o < v ? o : v; } return v; }
__device__ __forceinline__ uint64_t wave_max64(uint64_t v) { for (int d = 32; d; d >>= 1) { const uint64_t o = __shfl_xor(v, d, 64); v = o > v ? o : v; } return v; }

// smallest and largest selected key as order-preserving images (signed: sign bit flipped) -> aux[kAuxMin], aux[kAuxMax]
// (every tile_step-th tile: a sample)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dense_minmax(const uint64_t* __restrict__ bitmap, const T* __restrict__ col, const uint64_t* __restrict__ missing,
                                                         int64_t nrows, int64_t ntiles, int64_t tile_step, uint64_t* aux) {
  const uint64_t flip = (T)-1 < (T)0 ? (1ull << 63) : 0ull;
  uint64_t lo = ~0ull, hi = 0ull;
  walk_selected<T>(bitmap, col, missing, nrows, 0, ntiles, ((int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * tile_step, (int64_t)gridDim.x * kWavesPerBlock * tile_step,
                   [&](uint64_t key, int64_t) { const uint64_t im = key ^ flip; lo = im < lo ? im : lo; hi = im > hi ? im : hi; }, [&](int64_t) {});
  lo = wave_min64(lo); hi = wave_max64(hi);
  if (lane_id() == 0 && lo <= hi) { atomicMin((unsigned long long*)&aux[kAuxMin], (unsigned long long)lo); atomicMax((unsigned long long*)&aux[kAuxMax], (unsigned long long)hi); }
}

template <typename T>
__global__ __launch_bounds__(kDenseBlock) void k_dense_presence(const uint64_t* __restrict__ bitmap, const T* __restrict__ col, const uint64_t* __restrict__ missing,
                                                                int64_t nrows, int64_t ntiles, uint64_t lo, uint32_t range, uint32_t* __restrict__ present, uint64_t* aux) {
  __shared__ uint32_t bits[kDenseWords];
  const int words = (int)((range + 31u) >> 5);
  for (int i = threadIdx.x; i < words; i += kDenseBlock) bits[i] = 0;
  __syncthreads();
  bool outside = false;
  walk_selected<T>(bitmap, col, missing, nrows, 0, ntiles, (int64_t)blockIdx.x * (kDenseBlock / 64) + (threadIdx.x >> 6), (int64_t)gridDim.x * (kDenseBlock / 64),
                   // (the bit is looked at before it is set: a column of seven values sent 64 lanes' atomicOr to ONE word per instruction — 2.35 ms per 1e9 rows
                   // against 1.24 for 300 or 5000 values; a plain read of one address is a broadcast)
                   [&](uint64_t key, int64_t) {
                     const uint64_t k = key - lo;
                     if (k < range) { const uint32_t b = 1u << (k & 31); if (!(__atomic_load_n(&bits[k >> 5], __ATOMIC_RELAXED) & b)) atomicOr(&bits[k >> 5], b); }
                     else outside = true;
                   },
                   [&](int64_t row) { note_missing(aux, row); });
  if (outside) __atomic_store_n(&aux[kAuxOutside], 1ull, __ATOMIC_RELAXED);
  __syncthreads();
  for (int i = threadIdx.x; i < words; i += kDenseBlock) { const uint32_t b = bits[i]; if (b) atomicOr(&present[i], b); }
}
__global__ __launch_bounds__(kBlock) void k_dense_count(const uint32_t* __restrict__ present, int words, uint64_t* aux) {
  uint32_t n = 0, lo = 0xFFFFFFFFu, hi = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < words; i += gridDim.x * kBlock) {
    const uint32_t b = present[i];
    if (b) { n += (uint32_t)__popc(b); lo = min(lo, (uint32_t)i * 32u + (uint32_t)__builtin_ctz(b)); hi = max(hi, (uint32_t)i * 32u + 31u - (uint32_t)__builtin_clz(b)); }
  }
  for (int d = 32; d; d >>= 1) { n += __shfl_xor(n, d, 64); lo = min(lo, (uint32_t)__shfl_xor(lo, d, 64)); hi = max(hi, (uint32_t)__shfl_xor(hi, d, 64)); }
  if (lane_id() == 0 && n) {
    atomicAdd((unsigned long long*)&aux[kAuxDistinct], (unsigned long long)n);
    atomicMin((unsigned long long*)&aux[kAuxSpanLo], (unsigned long long)lo); atomicMax((unsigned long long*)&aux[kAuxSpanHi], (unsigned long long)hi);
  }
}

// (three round trips per tile — the keys, their table entries, the atomics — each sixteen deep: with a returning atomic inside the loop over the sixteen words
// every one of them waited for the one before, ~30 us per tile, and the launches that find the first rows are a few tiles per wave)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dense_first(const uint64_t* __restrict__ bitmap, const T* __restrict__ col, const uint64_t* __restrict__ missing,
                                                        int64_t nrows, int64_t tile0, int64_t tile1, uint64_t lo, uint32_t range, uint64_t distinct, uint64_t* first, uint64_t* aux) {
  // every value's first row lies in an EARLIER launch?  The test reads kAuxFoundBefore — kAuxFound as the launches before this one left it (k_dense_snap copies it
  // between two launches of the row-ordered series) — never the live counter: workgroups of THIS launch add to that as they finish, a workgroup that starts
  // late could see found == distinct and skip its tiles although they hold an earlier row of a value a sibling has just found in a later tile (first[value]
  // would then not be the smallest row, and unique's order with it).  `distinct` comes from the host, which read it after the presence pass.
  if (aux[kAuxFoundBefore] >= distinct) return;
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  uint32_t fresh = 0;
  for (int64_t tile = tile0 + wave; tile < tile1; tile += nwaves) {
    const uint64_t sel = lane < 16 ? bitmap[tile * 16 + lane] : 0ull;
    const uint64_t ms = (missing && lane < 16) ? missing[tile * 16 + lane] : 0ull;
    const uint64_t live = sel & ~ms;
    if (__ballot(live != 0) == 0) continue;
    const int64_t base = tile * kTile;
    T v[16];
    if (base + kTile <= nrows) {
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(col + base + j * 64 + lane);
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) { const int64_t row = base + j * 64 + lane; v[j] = row < nrows ? col[row] : (T)0; }
    }
    uint64_t seen[16], old[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t w = __shfl(live, j, 64);
      const uint64_t k = widen_key(v[j]) - lo;
      const bool on = ((w >> lane) & 1ull) && base + j * 64 + lane < nrows && k < range;
      seen[j] = on ? __atomic_load_n(&first[k], __ATOMIC_RELAXED) : 0ull;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t row = (uint64_t)(base + j * 64 + lane);
      old[j] = seen[j] > row ? atomicMin((unsigned long long*)&first[widen_key(v[j]) - lo], (unsigned long long)row) : 0ull;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) fresh += (uint32_t)(old[j] == kEmpty);
  }
  for (int d = 32; d; d >>= 1) fresh += __shfl_xor(fresh, d, 64);
  if (lane == 0 && fresh) atomicAdd((unsigned long long*)&aux[kAuxFound], (unsigned long long)fresh);
}

__global__ void k_dense_snap(uint64_t* aux) { aux[kAuxFoundBefore] = aux[kAuxFound]; }

__global__ __launch_bounds__(kBlock) void k_dense_scatter(const uint64_t* __restrict__ first, uint32_t range, const uint64_t* __restrict__ aux,
                                                          uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i > range) return;
  const uint64_t r = i < range ? first[i] : aux[kAuxMissing];
  if (r == kEmpty) return;
  atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63));
  atomicAdd(&tile_counts[r >> 10], 1u);
}
__global__ __launch_bounds__(kBlock) void k_dense_group_ids(uint64_t* __restrict__ first, uint32_t range, uint64_t* __restrict__ aux,
                                                            const uint64_t* __restrict__ ubits, const uint64_t* __restrict__ uprefix) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < range) { if (first[i] != kEmpty) first[i] = rank_of_row(ubits, uprefix, first[i]); }
  else if (i == range && aux[kAuxMissing] != kEmpty) aux[kAuxMissing] = rank_of_row(ubits, uprefix, aux[kAuxMissing]);
}

#define DENSE_BY_DTYPE(CALL)                                                       \
  switch (dtype) {                                                                 \
    case DFDB_I8:  { using T = int8_t;   CALL; } break;                            \
    case DFDB_I16: { using T = int16_t;  CALL; } break;                            \
    case DFDB_I32: { using T = int32_t;  CALL; } break;                            \
    case DFDB_I64: { using T = int64_t;  CALL; } break;                            \
    case DFDB_U8: case DFDB_BOOL: { using T = uint8_t; CALL; } break;              \
    case DFDB_U16: { using T = uint16_t; CALL; } break;                            \
    case DFDB_U32: { using T = uint32_t; CALL; } break;                            \
    default:       { using T = uint64_t; CALL; } break;                            \
  }
void launch_dense_minmax(hipStream_t s, const uint64_t* bitmap, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int64_t tile_step, uint64_t* aux) {
  if (nrows <= 0) return;
  const int64_t ntiles = (nrows + kTile - 1) / kTile, visited = (ntiles + tile_step - 1) / tile_step;
  const dim3 g(grid_tiles(visited) > 2048 ? 2048 : grid_tiles(visited)), b(kBlock);
  DENSE_BY_DTYPE(hipLaunchKernelGGL((k_dense_minmax<T>), g, b, 0, s, bitmap, (const T*)col, missing, nrows, ntiles, tile_step, aux))
}
void launch_dense_presence(hipStream_t s, const uint64_t* bitmap, const void* col, int dtype, const uint64_t* missing, int64_t nrows, uint64_t lo, uint32_t range,
                           uint32_t* present, uint64_t* aux) {
  if (nrows <= 0) return;
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  int64_t nb = (ntiles + kDenseBlock / 64 - 1) / (kDenseBlock / 64); if (nb > 256) nb = 256;      // one workgroup per CU: its LDS is the whole CU's
  const dim3 g((unsigned)nb), b(kDenseBlock);
  DENSE_BY_DTYPE(hipLaunchKernelGGL((k_dense_presence<T>), g, b, 0, s, bitmap, (const T*)col, missing, nrows, ntiles, lo, range, present, aux))
  const int words = (int)((range + 31u) >> 5);
  hipLaunchKernelGGL(k_dense_count, dim3((words + kBlock - 1) / kBlock > 64 ? 64 : (words + kBlock - 1) / kBlock), dim3(kBlock), 0, s, present, words, aux);
}
void launch_dense_first(hipStream_t s, const uint64_t* bitmap, const void* col, int dtype, const uint64_t* missing, int64_t row0, int64_t row1, uint64_t lo,
                        uint32_t range, uint64_t distinct, uint64_t* first, uint64_t* aux) {
  if (row1 <= row0) return;
  const int64_t tile0 = row0 / kTile, tile1 = (row1 + kTile - 1) / kTile;
  const dim3 g(grid_tiles(tile1 - tile0) > 2048 ? 2048 : grid_tiles(tile1 - tile0)), b(kBlock);
  hipLaunchKernelGGL(k_dense_snap, dim3(1), dim3(1), 0, s, aux);      // what the launches before this one have found: the only count this launch may stop on
  DENSE_BY_DTYPE(hipLaunchKernelGGL((k_dense_first<T>), g, b, 0, s, bitmap, (const T*)col, missing, row1, tile0, tile1, lo, range, distinct, first, aux))
}
void launch_dense_scatter(hipStream_t s, const uint64_t* first, uint32_t range, const uint64_t* aux, uint64_t* bitmap, uint32_t* tile_counts) {
  hipLaunchKernelGGL(k_dense_scatter, dim3((range + 1 + kBlock) / kBlock), dim3(kBlock), 0, s, first, range, aux, bitmap, tile_counts);
}
void launch_dense_group_ids(hipStream_t s, uint64_t* first, uint32_t range, uint64_t* aux, const uint64_t* ubits, const uint64_t* uprefix) {
  hipLaunchKernelGGL(k_dense_group_ids, dim3((range + 1 + kBlock) / kBlock), dim3(kBlock), 0, s, first, range, aux, ubits, uprefix);
}
// The dense form with its group-number table IN LDS (round 5): per row the pass looked its key's group up in `gids` — a random 8-byte load per lane, 64 different lines
// per wave instruction (64 cycles of the CU's vector cache each, and a second dependent round trip per trip of rows): 1e9 rows by 5000 keys ran at 3.2 TB/s.  When the
// table (4 bytes per value of the keys' range) fits beside the accumulators it is copied into LDS once per workgroup and the lookup is a ds_read.
// Dynamic LDS: [ngroups] counts, [ngroups] values, [range + 1] group numbers (the last: missing).  8-byte integer keys and 8-byte values (or none).
template <int OPK>
__global__ __launch_bounds__(1024) void k_group_acc_dense_lds(const AccArgs A, uint32_t range, int ngp) {
  extern __shared__ uint64_t dyn_sh[];
  uint64_t* lcnt = dyn_sh; uint64_t* lval = dyn_sh + ngp; uint32_t* tbl = (uint32_t*)(dyn_sh + 2 * ngp);
  constexpr bool has_val = OPK != 0;
  for (int g = threadIdx.x; g < A.ngroups; g += 1024) { lcnt[g] = 0; lval[g] = A.val_init; }
  for (uint32_t i = threadIdx.x; i < range; i += 1024) tbl[i] = (uint32_t)A.gids[i];
  if (threadIdx.x == 0) tbl[range] = (uint32_t)A.special[1];
  __syncthreads();
  // every load of a trip is issued before any of them is looked at: the selection word, the key and the value of a row do not wait for each other (a key loaded only
  // where the selection bit is set costs a second dependent round trip per trip: the counters showed waves waiting 89 % of their cycles with ~37 lines in flight per CU)
  constexpr int U = 8;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  int vkind = 0; if (has_val) (void)value_bits(A.valcol, A.valdt, 0, vkind);
  bool unknown = false;
  for (int64_t row0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; row0 < A.nrows; row0 += U * stride) {
    uint64_t w[U], mw[U], key[U], bits[U]; uint32_t gid[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      const int64_t row = row0 + k * stride;
      const bool inb = row < A.nrows;
      w[k] = inb ? A.sel[row >> 6] : 0ull;
      mw[k] = (inb && A.missing) ? A.missing[row >> 6] : 0ull;
      key[k] = inb ? ((const uint64_t*)A.keycol)[row] : 0ull;
      bits[k] = (inb && has_val) ? ((const uint64_t*)A.valcol)[row] : 0ull;
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      const int64_t row = row0 + k * stride;
      const bool on = (w[k] >> (row & 63)) & 1ull, miss = (mw[k] >> (row & 63)) & 1ull;
      const uint64_t idx = key[k] - A.lo;
      const bool inr = idx < (uint64_t)range;
      gid[k] = tbl[(on && !miss && inr) ? (uint32_t)idx : range];      // (missing, and rows that are off: the table's last entry)
      // a selected row without a group: its key lies outside the span laid out, or it (or `missing`) never turned up among the rows the table was made from
      const bool bad = on && (gid[k] == 0xFFFFFFFFu || (!miss && !inr));
      unknown |= bad;
      w[k] = (on && !bad) ? 1ull : 0ull;
    }
#pragma unroll
    for (int k = 0; k < U; k++) if (w[k]) group_add_t<OPK>(lcnt, lval, (uint64_t)gid[k], bits[k], vkind);
  }
  if (unknown && A.unknown_flag) __atomic_store_n(A.unknown_flag, 1ull, __ATOMIC_RELAXED);
  __syncthreads();
  group_flush(lcnt, lval, A.cnt, A.val, A.ngroups, A.op, vkind, has_val, 1024);
}
template <int OPK>
static bool try_dense_lds(hipStream_t s, const AccArgs& A, uint32_t range) {
  const int ngp = (A.ngroups + 1) & ~1;
  const size_t lds = (size_t)ngp * 16 + ((size_t)range + 2) * 4;
  if (lds > 156 * 1024) return false;
  // (the attribute belongs to the function ON A DEVICE: a process that drives several GPUs raises it on each)
  static std::atomic<bool> raised[64] = {};
  int dev = 0; (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || !raised[dev].load(std::memory_order_acquire))) {
    if (hipFuncSetAttribute((const void*)k_group_acc_dense_lds<OPK>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (dev >= 0 && dev < 64) raised[dev].store(true, std::memory_order_release);
  }
  const unsigned g = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, (A.nrows + 1023) / 1024));
  hipLaunchKernelGGL((k_group_acc_dense_lds<OPK>), dim3(g), dim3(1024), lds, s, A, range, ngp);
  return true;
}
int launch_group_accumulate_dense(hipStream_t s, const uint64_t* sel, const void* keycol, int keydt, const uint64_t* missing, const void* valcol, int valdt, int op,
                                   int64_t nrows, uint64_t lo, uint32_t range, uint64_t span_lo, uint64_t span_hi, const uint64_t* gids, const uint64_t* aux, uint64_t* cnt, uint64_t* val,
                                   int64_t ngroups, uint64_t val_init, uint64_t* unknown_flag) {
  // unknown_flag != nullptr: the table was made from the head of the column only — the pass must be the LDS form, which reports a key without a group; returns -1,
  // nothing launched, when that form cannot take the job (the caller then makes the table from every row)
  AccArgs A{};
  A.unknown_flag = unknown_flag;
  A.sel = sel; A.keycol = keycol; A.keydt = keydt; A.missing = missing; A.valcol = valcol; A.valdt = valdt; A.op = op; A.nrows = nrows; A.lo = lo; A.gids = gids; A.special = aux;
  A.cnt = cnt; A.val = val; A.ngroups = (int)ngroups; A.val_init = val_init;
  if (nrows > 0 && (ngroups > kGroupLds || unknown_flag) && (keydt == DFDB_I64 || keydt == DFDB_U64) && span_lo <= span_hi && span_hi < range) {   // (few groups: the 256-thread kernels, several workgroups per CU)
    // the table is laid out for the widest span the form can hold; the keys that are there cover [span_lo, span_hi] of it (k_dense_count) and only that goes to LDS
    AccArgs B = A;
    B.lo = lo + span_lo; B.gids = gids + span_lo;
    const uint32_t brange = (uint32_t)(span_hi - span_lo + 1);
    int vkind = (valdt == DFDB_F64) ? 2 : (valdt == DFDB_U64 ? 1 : 0);
    const int opk = opk_of(op, valcol != nullptr, vkind);
    const bool v8 = opk == 0 || valdt == DFDB_I64 || valdt == DFDB_U64 || valdt == DFDB_F64;
    bool done = false;
    if (v8) switch (opk) {
      case 0: done = try_dense_lds<0>(s, B, brange); break;
      case 1: done = try_dense_lds<1>(s, B, brange); break;
      case 2: done = try_dense_lds<2>(s, B, brange); break;
      case 3: done = try_dense_lds<3>(s, B, brange); break;
      default: done = try_dense_lds<4>(s, B, brange); break;
    }
    if (done) return 1;                                        // (1: the form with the group table in LDS)
  }
  if (unknown_flag) return -1;
  launch_group_acc<2>(s, A);
  return 0;
}

}  // namespace dfdb
