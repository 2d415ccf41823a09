// k_unique.hip — unique(col) over the selected rows as a SELECTION: the rows that hold the first occurrence of their value.
//
// Replaces Base.unique driven by Base.iterate(::DFColumn) (src/tables/column.jl:102-126; docs/src/index.md:171-182,
// 479-486: "unique(t.brand[t.brand .!= ""])", 7-11 MRows/s in the reference).  Julia's unique keeps the FIRST occurrence
// of every value in iteration order and compares with isequal (NaN == NaN, 0.0 != -0.0, missing == missing).  Here:
//   pass 1  every selected row inserts (key, row) into an open-addressing table in HBM: 64-bit atomicCAS claims the slot of
//           a key, 64-bit atomicMin keeps the smallest row that holds it;
//   pass 2  every selected row looks its key up and keeps its bit iff it IS that smallest row -> the bitmap of first
//           occurrences (+ per-tile counts), and the ordinary count / gather / materialize machinery returns the distinct
//           values in order of first appearance.  No sort.
// Fixed-width values are their own keys.  A String's key is a salted 64-bit hash of its bytes; the slot remembers where one
// holder's bytes start, pass 1b compares every selected row with that representative and reports a true hash collision
// (two different strings, one key), in which case the host repeats with another salt: the result is exact, not probabilistic.
#include "device_utils.hpp"
#include "kernels.hpp"
#include "../../include/dfdb_ir.h"

namespace dfdb {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int64_t kTile = 1024;
constexpr uint64_t kEmpty = 0xFFFFFFFFFFFFFFFFull;    // never stored as a key: a value with this image uses special[0]

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

__device__ __forceinline__ uint64_t table_insert(uint64_t* keys, uint64_t* rows, uint64_t mask, uint64_t key, uint64_t row) {
  uint64_t h = slot_of(key, mask);
  for (;;) {
    // plain loads first: with few distinct values nearly every row finds its key in place and a smaller row recorded, and
    // skips both atomics (5e8 rows of 10 distinct strings: 0.60 s with every row hammering the same 10 addresses, 0.023 s so)
    uint64_t old = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
    if (old == kEmpty) old = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)kEmpty, (unsigned long long)key);
    if (old == kEmpty || old == key) {
      if (__atomic_load_n(&rows[h], __ATOMIC_RELAXED) > row) atomicMin((unsigned long long*)&rows[h], (unsigned long long)row);
      return old == kEmpty ? h : (h | (1ull << 63));
    }
    h = (h + 1) & mask;
  }
}
__device__ __forceinline__ uint64_t table_find(const uint64_t* keys, uint64_t mask, uint64_t key) {
  uint64_t h = slot_of(key, mask);
  while (keys[h] != key) h = (h + 1) & mask;     // present by construction (pass 1 inserted it)
  return h;
}

// ---------------------------------------------------------------- fixed-width columns
__global__ __launch_bounds__(kBlock) void k_unique_insert(const uint64_t* __restrict__ bitmap, const void* __restrict__ col, int dtype,
                                                          const uint64_t* __restrict__ missing, int64_t nrows, uint64_t* keys, uint64_t* rows,
                                                          uint64_t mask, uint64_t* special) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < nrows; row += stride) {
    if (!((bitmap[row >> 6] >> (row & 63)) & 1ull)) continue;
    if (missing && ((missing[row >> 6] >> (row & 63)) & 1ull)) { if (__atomic_load_n(&special[1], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&special[1], (unsigned long long)row); continue; }
    const uint64_t key = key_fixed(col, dtype, row);
    if (key == kEmpty) { if (__atomic_load_n(&special[0], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&special[0], (unsigned long long)row); continue; }
    table_insert(keys, rows, mask, key, (uint64_t)row);
  }
}

__global__ __launch_bounds__(kBlock) void k_unique_mark(uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, const void* __restrict__ col,
                                                        int dtype, const uint64_t* __restrict__ missing, int64_t nrows, int64_t ntiles,
                                                        const uint64_t* __restrict__ keys, const uint64_t* __restrict__ rows, uint64_t mask,
                                                        const uint64_t* __restrict__ special) {
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
          if (missing && ((missing[row >> 6] >> (row & 63)) & 1ull)) first = special[1] == (uint64_t)row;
          else {
            const uint64_t key = key_fixed(col, dtype, row);
            first = key == kEmpty ? special[0] == (uint64_t)row : rows[table_find(keys, mask, key)] == (uint64_t)row;
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

// ---------------------------------------------------------------- String columns (FlatStringsVector: sizes + arena, offsets per 1024-row tile)
__device__ __forceinline__ uint64_t hash_bytes(const uint8_t* p, int32_t len, uint64_t salt) {
  uint64_t h = splitmix64(salt ^ (uint64_t)(uint32_t)len);
  int32_t k = 0;
  for (; k + 8 <= len; k += 8) { uint64_t v; __builtin_memcpy(&v, p + k, 8); h = splitmix64(h ^ v); }
  uint64_t tail = 0;
  for (int b = 0; k < len; k++, b += 8) tail |= (uint64_t)p[k] << b;
  return splitmix64(h ^ tail);
}

// MODE 0: insert (hash, row) and remember one holder's bytes; 1: verify every selected row against its slot's representative;
// 2: mark first occurrences.  One wave per 1024-row tile: the rows' byte offsets are a wave prefix sum of the sizes.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_unique_str(uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, const int32_t* __restrict__ sizes,
                                                       const int64_t* __restrict__ tile_off, const uint8_t* __restrict__ bytes, int64_t nrows, int64_t ntiles,
                                                       uint64_t* keys, uint64_t* rows, uint64_t* rep_off, uint32_t* rep_len, uint64_t mask,
                                                       uint64_t* special, uint64_t salt, int* __restrict__ collision) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const uint64_t mine = lane < 16 ? bitmap[tile * 16 + lane] : 0ull;
    if (__ballot(mine != 0) == 0) { if (MODE == 2 && lane == 0) tile_counts[tile] = 0; continue; }
    int64_t run = tile_off[tile];
    uint64_t myword = 0; uint32_t cnt = 0;
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
        if (sz < 0) {                                                     // missing
          if (MODE == 0) { if (__atomic_load_n(&special[1], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&special[1], (unsigned long long)row); }
          else if (MODE == 2) first = special[1] == (uint64_t)row;
        } else {
          uint64_t key = hash_bytes(bytes + off, sz, salt);
          if (key == kEmpty) key = 0x1234567ull;                         // (any fixed remap: equal strings still get equal keys)
          if (MODE == 0) {
            const uint64_t r = table_insert(keys, rows, mask, key, (uint64_t)row);
            if (!(r >> 63)) { rep_off[r] = (uint64_t)off; rep_len[r] = (uint32_t)sz; }   // I claimed the slot: my bytes represent it
          } else {
            const uint64_t h = table_find(keys, mask, key);
            if (MODE == 1) {
              bool same = rep_len[h] == (uint32_t)sz;
              const uint8_t* a = bytes + off; const uint8_t* b = bytes + rep_off[h];
              for (int32_t k = 0; same && k < sz; k++) same = a[k] == b[k];
              if (!same) atomicOr(collision, 1);
            } else first = rows[h] == (uint64_t)row;
          }
        }
      }
      if (MODE == 2) { const uint64_t m = __ballot(first); if (lane == j) myword = m; cnt += (uint32_t)__popcll(m); }
    }
    if (MODE == 2) {
      if (lane < 16) bitmap[tile * 16 + lane] = myword;
      if (lane == 0) tile_counts[tile] = cnt;
    }
  }
}

static int grid_rows(int64_t n) { int64_t b = (n + kBlock - 1) / kBlock; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }
static int grid_tiles(int64_t nt) { int64_t b = (nt + kWavesPerBlock - 1) / kWavesPerBlock; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }

void launch_unique_fixed(hipStream_t s, int pass, uint64_t* bitmap, uint32_t* tile_counts, const void* col, int dtype, const uint64_t* missing,
                         int64_t nrows, uint64_t* keys, uint64_t* rows, uint64_t mask, uint64_t* special) {
  if (nrows <= 0) return;
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (pass == 0) hipLaunchKernelGGL(k_unique_insert, dim3(grid_rows(nrows)), dim3(kBlock), 0, s, bitmap, col, dtype, missing, nrows, keys, rows, mask, special);
  else hipLaunchKernelGGL(k_unique_mark, dim3(grid_tiles(ntiles)), dim3(kBlock), 0, s, bitmap, tile_counts, col, dtype, missing, nrows, ntiles, keys, rows, mask, special);
}

void launch_unique_str(hipStream_t s, int pass, uint64_t* bitmap, uint32_t* tile_counts, const int32_t* sizes, const int64_t* tile_off,
                       const uint8_t* bytes, int64_t nrows, uint64_t* keys, uint64_t* rows, uint64_t* rep_off, uint32_t* rep_len, uint64_t mask,
                       uint64_t* special, uint64_t salt, int* collision) {
  if (nrows <= 0) return;
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  const dim3 g(grid_tiles(ntiles)), b(kBlock);
  if (pass == 0) hipLaunchKernelGGL((k_unique_str<0>), g, b, 0, s, bitmap, tile_counts, sizes, tile_off, bytes, nrows, ntiles, keys, rows, rep_off, rep_len, mask, special, salt, collision);
  else if (pass == 1) hipLaunchKernelGGL((k_unique_str<1>), g, b, 0, s, bitmap, tile_counts, sizes, tile_off, bytes, nrows, ntiles, keys, rows, rep_off, rep_len, mask, special, salt, collision);
  else hipLaunchKernelGGL((k_unique_str<2>), g, b, 0, s, bitmap, tile_counts, sizes, tile_off, bytes, nrows, ntiles, keys, rows, rep_off, rep_len, mask, special, salt, collision);
}

// ---------------------------------------------------------------- groupreduce (src/tables/aggregate.jl:1-36)
// The reference's groupreduce numbers the groups in order of first appearance of the key (group_map[elem] = length(group_map) + 1) and stops
// there (it is unfinished: it prints the map).  Completed to that intent on top of unique's table: after the unique passes the table maps a
// key to the row of its first occurrence and the bitmap holds exactly those rows, so a group's number is the RANK of its first row among them
// (k_group_ids turns the table's row slots into group numbers, once per group); k_group_accumulate then sends every selected row's value to its
// group's accumulator — privatised in LDS per workgroup when there are few groups (ten brands over 5e8 rows would otherwise be 5e8 atomics on ten
// addresses), global atomics otherwise.  Accumulators are 64-bit: counts, wrapping integer sums (Julia's), double sums, and min / max through an
// order-preserving image (a NaN wins both, like Julia's minimum / maximum).
constexpr int kGroupLds = 1024;                      // groups that fit the per-workgroup accumulators

__device__ __forceinline__ uint64_t rank_of_row(const uint64_t* __restrict__ ubits, const uint64_t* __restrict__ uprefix, uint64_t row) {
  const uint64_t tile = row >> 10, w = (row & 1023) >> 6;
  uint64_t r = uprefix[tile];
  for (uint64_t k = 0; k < w; k++) r += (uint64_t)__popcll(ubits[tile * 16 + k]);
  return r + (uint64_t)__popcll(ubits[tile * 16 + w] & ((1ull << (row & 63)) - 1ull));
}
__global__ __launch_bounds__(kBlock) void k_group_ids(const uint64_t* __restrict__ keys, uint64_t* __restrict__ rows, uint64_t cap, uint64_t* __restrict__ special,
                                                      const uint64_t* __restrict__ ubits, const uint64_t* __restrict__ uprefix) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < cap && keys[i] != kEmpty) rows[i] = rank_of_row(ubits, uprefix, rows[i]);
  if (i < 2 && special[i] != kEmpty) special[i] = rank_of_row(ubits, uprefix, special[i]);
}

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
// merge a workgroup's LDS accumulators into the global ones (val_kind: 2 = double sums)
__device__ __forceinline__ void group_flush(const uint64_t* lcnt, const uint64_t* lval, uint64_t* cnt, uint64_t* val, int ngroups, int op, int val_kind, bool has_val) {
  for (int g = threadIdx.x; g < ngroups; g += kBlock) {
    const uint64_t c = lcnt[g];
    if (!c) continue;
    atomicAdd((unsigned long long*)&cnt[g], (unsigned long long)c);
    if (!has_val) continue;
    if (op == DFDB_AGG_SUM) { if (val_kind == 2) atomicAdd((double*)&val[g], __longlong_as_double((long long)lval[g])); else atomicAdd((unsigned long long*)&val[g], (unsigned long long)lval[g]); }
    else if (op == DFDB_AGG_MIN) atomicMin((unsigned long long*)&val[g], (unsigned long long)lval[g]);
    else if (op == DFDB_AGG_MAX) atomicMax((unsigned long long*)&val[g], (unsigned long long)lval[g]);
  }
}

template <bool LDS>
__global__ __launch_bounds__(kBlock) void k_group_accumulate(const uint64_t* __restrict__ sel, const void* __restrict__ keycol, int keydt, const uint64_t* __restrict__ missing,
                                                             const void* __restrict__ valcol, int valdt, int op, int64_t nrows,
                                                             const uint64_t* __restrict__ keys, const uint64_t* __restrict__ gids, uint64_t mask,
                                                             const uint64_t* __restrict__ special, uint64_t* cnt, uint64_t* val, int ngroups, uint64_t val_init) {
  __shared__ uint64_t lcnt[LDS ? kGroupLds : 1], lval[LDS ? kGroupLds : 1];
  const bool has_val = valcol != nullptr && op != DFDB_AGG_COUNT;
  int val_kind = 0;
  if (LDS) { for (int g = threadIdx.x; g < ngroups; g += kBlock) { lcnt[g] = 0; lval[g] = val_init; } __syncthreads(); }
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < nrows; row += stride) {
    if (!((sel[row >> 6] >> (row & 63)) & 1ull)) continue;
    uint64_t gid;
    if (missing && ((missing[row >> 6] >> (row & 63)) & 1ull)) gid = special[1];
    else { const uint64_t key = key_fixed(keycol, keydt, row); gid = key == kEmpty ? special[0] : gids[table_find(keys, mask, key)]; }
    int kind = 0; const uint64_t bits = has_val ? value_bits(valcol, valdt, row, kind) : 0ull;
    val_kind = kind;
    if (LDS) group_add(lcnt, lval, gid, bits, kind, op, has_val); else group_add(cnt, val, gid, bits, kind, op, has_val);
  }
  if (LDS) {
    __syncthreads();
    int k2 = 0; if (has_val) (void)value_bits(valcol, valdt, 0, k2);      // (the value kind is a property of the column)
    (void)val_kind;
    group_flush(lcnt, lval, cnt, val, ngroups, op, k2, has_val);
  }
}

// String keys: one wave per 1024-row tile (byte offsets are a wave prefix sum of the sizes, as in k_unique_str)
template <bool LDS>
__global__ __launch_bounds__(kBlock) void k_group_accumulate_str(const uint64_t* __restrict__ sel, const int32_t* __restrict__ sizes, const int64_t* __restrict__ tile_off,
                                                                 const uint8_t* __restrict__ bytes, const void* __restrict__ valcol, int valdt, int op, int64_t nrows, int64_t ntiles,
                                                                 const uint64_t* __restrict__ keys, const uint64_t* __restrict__ gids, uint64_t mask,
                                                                 const uint64_t* __restrict__ special, uint64_t salt, uint64_t* cnt, uint64_t* val, int ngroups, uint64_t val_init) {
  __shared__ uint64_t lcnt[LDS ? kGroupLds : 1], lval[LDS ? kGroupLds : 1];
  const bool has_val = valcol != nullptr && op != DFDB_AGG_COUNT;
  if (LDS) { for (int g = threadIdx.x; g < ngroups; g += kBlock) { lcnt[g] = 0; lval[g] = val_init; } __syncthreads(); }
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const uint64_t mine = lane < 16 ? sel[tile * 16 + lane] : 0ull;
    if (__ballot(mine != 0) == 0) continue;
    int64_t run = tile_off[tile];
    for (int j = 0; j < 16; j++) {
      const int64_t row = tile * kTile + j * 64 + lane;
      const int32_t sz = row < nrows ? sizes[row] : 0;
      const uint32_t c = sz > 0 ? (uint32_t)sz : 0u;
      const uint32_t incl = wave_incl_scan(c);
      const int64_t off = run + (int64_t)(incl - c);
      run += (int64_t)__shfl(incl, 63, 64);
      const uint64_t w = __shfl(mine, j, 64);
      if (row < nrows && ((w >> lane) & 1ull)) {
        uint64_t gid;
        if (sz < 0) gid = special[1];
        else { uint64_t key = hash_bytes(bytes + off, sz, salt); if (key == kEmpty) key = 0x1234567ull; gid = gids[table_find(keys, mask, key)]; }
        int kind = 0; const uint64_t bits = has_val ? value_bits(valcol, valdt, row, kind) : 0ull;
        if (LDS) group_add(lcnt, lval, gid, bits, kind, op, has_val); else group_add(cnt, val, gid, bits, kind, op, has_val);
      }
    }
  }
  if (LDS) {
    __syncthreads();
    int k2 = 0; if (has_val) (void)value_bits(valcol, valdt, 0, k2);
    group_flush(lcnt, lval, cnt, val, ngroups, op, k2, has_val);
  }
}

void launch_group_ids(hipStream_t s, const uint64_t* keys, uint64_t* rows, uint64_t cap, uint64_t* special, const uint64_t* ubits, const uint64_t* uprefix) {
  hipLaunchKernelGGL(k_group_ids, dim3((unsigned)((cap + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, keys, rows, cap, special, ubits, uprefix);
}
// ---- K9 fast path: the key is a String column with a dictionary (k_dict.hip) — its 16-bit codes ARE group labels.  What is left of `unique` is the
// first selected row of every code (the reference numbers groups by first appearance), and `groupreduce` accumulates by rank_of_code[code]: no hash
// table (the generic path sizes one by the number of selected rows: 30 GB for 5e8 rows of ten brands).
// A wave walks its tiles and the rows of a tile in increasing order, so the first time it meets a code is its smallest row for that code: a per-wave
// bit set in LDS keeps every later meeting away from the global atomicMin (ten brands over 5e8 rows: <= waves x 10 atomics, not 5e8).
__global__ __launch_bounds__(kBlock) void k_dict_first_rows(const uint64_t* __restrict__ sel, const uint16_t* __restrict__ codes, int64_t nrows, int64_t ntiles,
                                                            unsigned long long* __restrict__ first, int lut_words) {
  __shared__ uint32_t seen_sh[kBlock / 64][2048];
  uint32_t* seen = seen_sh[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  for (int k = lane; k < lut_words; k += 64) seen[k] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    for (int j = 0; j < 16; j++) {
      const uint64_t w = sel[tile * 16 + j];
      if (w == 0) continue;                                                  // (wave-uniform)
      const int64_t row = tile * 1024 + j * 64 + lane;
      if (!((w >> lane) & 1ull) || row >= nrows) continue;
      const uint32_t c = codes[row];
      if ((seen[c >> 5] >> (c & 31u)) & 1u) continue;
      atomicMin(&first[c], (unsigned long long)row);
      atomicOr(&seen[c >> 5], 1u << (c & 31u));
    }
  }
}
void launch_dict_first_rows(hipStream_t s, const uint64_t* sel, const uint16_t* codes, int64_t nrows, uint64_t* first, int dict_n) {
  const int64_t ntiles = (nrows + 1023) / 1024;
  if (ntiles == 0) return;
  hipLaunchKernelGGL(k_dict_first_rows, dim3(grid_tiles(ntiles) > 2048 ? 2048 : grid_tiles(ntiles)), dim3(kBlock), 0, s, sel, codes, nrows, ntiles, (unsigned long long*)first, (dict_n + 31) / 32);
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
template <bool LDS>
__global__ __launch_bounds__(kBlock) void k_group_accumulate_codes(const uint64_t* __restrict__ sel, const uint16_t* __restrict__ codes, const uint32_t* __restrict__ rank_of_code,
                                                                   const void* __restrict__ valcol, int valdt, int op, int64_t nrows, uint64_t* cnt, uint64_t* val,
                                                                   int ngroups, uint64_t val_init) {
  __shared__ uint64_t lcnt[LDS ? kGroupLds : 1], lval[LDS ? kGroupLds : 1];
  const bool has_val = valcol != nullptr && op != DFDB_AGG_COUNT;
  if (LDS) { for (int g = threadIdx.x; g < ngroups; g += kBlock) { lcnt[g] = 0; lval[g] = val_init; } __syncthreads(); }
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < nrows; row += stride) {
    if (!((sel[row >> 6] >> (row & 63)) & 1ull)) continue;
    const uint64_t gid = rank_of_code[codes[row]];
    int kind = 0; const uint64_t bits = has_val ? value_bits(valcol, valdt, row, kind) : 0ull;
    if (LDS) group_add(lcnt, lval, gid, bits, kind, op, has_val); else group_add(cnt, val, gid, bits, kind, op, has_val);
  }
  if (LDS) {
    __syncthreads();
    int k2 = 0; if (has_val) (void)value_bits(valcol, valdt, 0, k2);
    group_flush(lcnt, lval, cnt, val, ngroups, op, k2, has_val);
  }
}
void launch_group_accumulate_codes(hipStream_t s, const uint64_t* sel, const uint16_t* codes, const uint32_t* rank_of_code, const void* valcol, int valdt, int op,
                                   int64_t nrows, uint64_t* cnt, uint64_t* val, int64_t ngroups, uint64_t val_init) {
  if (nrows <= 0) return;
  if (ngroups <= kGroupLds) hipLaunchKernelGGL((k_group_accumulate_codes<true>), dim3(grid_rows(nrows) > 2048 ? 2048 : grid_rows(nrows)), dim3(kBlock), 0, s, sel, codes, rank_of_code, valcol, valdt, op, nrows, cnt, val, (int)ngroups, val_init);
  else hipLaunchKernelGGL((k_group_accumulate_codes<false>), dim3(grid_rows(nrows)), dim3(kBlock), 0, s, sel, codes, rank_of_code, valcol, valdt, op, nrows, cnt, val, (int)ngroups, val_init);
}

int group_lds_limit() { return kGroupLds; }
void launch_group_accumulate(hipStream_t s, const uint64_t* sel, const void* keycol, int keydt, const uint64_t* missing, const void* valcol, int valdt, int op,
                             int64_t nrows, const uint64_t* keys, const uint64_t* gids, uint64_t mask, const uint64_t* special, uint64_t* cnt, uint64_t* val,
                             int64_t ngroups, uint64_t val_init) {
  if (nrows <= 0) return;
  if (ngroups <= kGroupLds) hipLaunchKernelGGL((k_group_accumulate<true>), dim3(grid_rows(nrows) > 2048 ? 2048 : grid_rows(nrows)), dim3(kBlock), 0, s, sel, keycol, keydt, missing, valcol, valdt, op, nrows, keys, gids, mask, special, cnt, val, (int)ngroups, val_init);
  else hipLaunchKernelGGL((k_group_accumulate<false>), dim3(grid_rows(nrows)), dim3(kBlock), 0, s, sel, keycol, keydt, missing, valcol, valdt, op, nrows, keys, gids, mask, special, cnt, val, (int)ngroups, val_init);
}
void launch_group_accumulate_str(hipStream_t s, const uint64_t* sel, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const void* valcol, int valdt,
                                 int op, int64_t nrows, const uint64_t* keys, const uint64_t* gids, uint64_t mask, const uint64_t* special, uint64_t salt,
                                 uint64_t* cnt, uint64_t* val, int64_t ngroups, uint64_t val_init) {
  if (nrows <= 0) return;
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  const int g = grid_tiles(ntiles) > 2048 ? 2048 : grid_tiles(ntiles);
  if (ngroups <= kGroupLds) hipLaunchKernelGGL((k_group_accumulate_str<true>), dim3(g), dim3(kBlock), 0, s, sel, sizes, tile_off, bytes, valcol, valdt, op, nrows, ntiles, keys, gids, mask, special, salt, cnt, val, (int)ngroups, val_init);
  else hipLaunchKernelGGL((k_group_accumulate_str<false>), dim3(grid_tiles(ntiles)), dim3(kBlock), 0, s, sel, sizes, tile_off, bytes, valcol, valdt, op, nrows, ntiles, keys, gids, mask, special, salt, cnt, val, (int)ngroups, val_init);
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

}  // namespace dfdb
