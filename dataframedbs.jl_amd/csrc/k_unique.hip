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

}  // namespace dfdb
