// k_radix.hip — unique(col) by RADIX PARTITION (round 6): the general (hash-table) form of k_unique.hip for many distinct values.
//
// Replaces Base.unique driven by Base.iterate(::DFColumn) (src/tables/column.jl:102-126; docs/src/index.md:479-487), like k_unique.hip.  The open-addressing
// table of {key, smallest row} in HBM costs one random 128-byte line per selected row once it outgrows the L2s (1e6 distinct Int64 / Float64 values: 32 MB of
// table, 1e9 probes, 19.8 ms per 1e9 rows = 0.05 of the HBM roofline on the 8 bytes per row it needs).  Here every byte moves in STREAMS:
//   hist       one pass over the key column: how many selected rows of each of C contiguous chunks fall into each of P = 2^k partitions (the top k bits of
//              splitmix64(key image)); an exclusive scan of the P x C counts (partition-major) is every chunk's write position in every partition
//   partition  the same pass again: a workgroup sorts 8192 rows at a time by partition in LDS and writes each partition's run — {key image 8 B} and
//              {row 4 B} in two arrays — at its running position: no global atomic, every store a contiguous run
//   unique     one workgroup per partition: its keys go through a table that lives in LDS (8192 slots: 64-bit compare-and-swap claims a slot, a 32-bit
//              atomic minimum keeps the smallest row), and the occupied slots leave one bit each in the bitmap of first occurrences (+ per-tile counts)
// 8 + (8 + 12) + 12 bytes per selected row, all sequential.  A partition that holds more distinct keys than its table takes raises a flag and the host
// runs the hash-table form instead (query.cpp: unique_hashed); keys are isequal images (one NaN, -0.0 apart from 0.0), a missing key and the one image that
// cannot be stored (all ones) are kept aside in aux[1] / aux[0] exactly as k_unique_insert keeps them.
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
// A row takes part when it is selected, not missing and its image can be stored (the special rows go to aux, once, in the hist pass).  In two steps, so that
// a thread can have several rows' loads in flight before it looks at any of them: `radix_load` issues the (unconditional) loads
// of a row — its selection word, its missing word, its value —, `radix_take` decides.  (One row at a time — bitmap word, then the value, then the atomics — kept
// ~8 KB in flight per CU: the hist pass read its 8 GB at 2.8 TB/s.)
struct RadixRow { uint64_t selw, missw, key; };
// (no control flow and no use of a loaded value in here: a branch around a row's loads, or the NaN select on its value, made the compiler wait for that row before it
// issued the next one's — the hist pass read its 8 GB at 2.2 TB/s with or without its atomics.  A row past the end loads the last row again and is masked in radix_take.)
__device__ __forceinline__ RadixRow radix_load(const uint64_t* __restrict__ sel, const void* __restrict__ col, int dtype, const uint64_t* __restrict__ missing,
                                               int64_t row, int64_t nrows) {
  const int64_t rc = row < nrows ? row : nrows - 1;            // (nrows >= 1: the launchers never run over an empty table)
  RadixRow r;
  r.selw = sel[rc >> 6];
  r.missw = missing ? missing[rc >> 6] : 0ull;
  if (dtype == DFDB_I64 || dtype == DFDB_U64 || dtype == DFDB_F64) r.key = __builtin_nontemporal_load((const uint64_t*)col + rc);   // (wave-uniform: 8-byte keys, raw)
  else r.key = rkey_fixed(col, dtype, rc);
  return r;
}
template <bool SPECIALS>
__device__ __forceinline__ bool radix_take(RadixRow& r, int dtype, int64_t row, int64_t nrows, uint64_t* aux) {
  if (dtype == DFDB_F64 && (r.key & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) r.key = 0x7ff8000000000000ull;      // isequal: one NaN
  if (row >= nrows) return false;
  if (!((r.selw >> (row & 63)) & 1ull)) return false;
  if ((r.missw >> (row & 63)) & 1ull) {
    if (SPECIALS && __atomic_load_n(&aux[1], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&aux[1], (unsigned long long)row);
    return false;
  }
  if (r.key == kREmpty) {
    if (SPECIALS && __atomic_load_n(&aux[0], __ATOMIC_RELAXED) > (uint64_t)row) atomicMin((unsigned long long*)&aux[0], (unsigned long long)row);
    return false;
  }
  return true;
}

// ---- pass 1: counts_T[p * C + c] = selected rows of chunk c whose key falls into partition p
__global__ __launch_bounds__(kRBlock) void k_radix_hist(const uint64_t* __restrict__ sel, const void* __restrict__ col, int dtype, const uint64_t* __restrict__ missing,
                                                        int64_t nrows, int64_t rows_per_chunk, int kbits, uint32_t* __restrict__ counts_T, uint64_t* aux, int xp) {
  extern __shared__ uint32_t hist_sh[];
  const int P = 1 << kbits, C = (int)gridDim.x, c = (int)blockIdx.x;
  for (int p = threadIdx.x; p < P; p += kRBlock) hist_sh[p] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)c * rows_per_chunk, r1 = r0 + rows_per_chunk < nrows ? r0 + rows_per_chunk : nrows;
  for (int64_t base = r0; base < r1; base += 8 * kRBlock) {                // eight rows per thread, their loads issued together
    RadixRow rr[8];
#pragma unroll
    for (int j = 0; j < 8; j++) rr[j] = radix_load(sel, col, dtype, missing, base + j * kRBlock + threadIdx.x, r1);
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (radix_take<true>(rr[j], dtype, base + j * kRBlock + threadIdx.x, r1, aux)) {
        if (xp & 1) { if (rhash(rr[j].key) == 12345u) hist_sh[0] = 1; }              // (DFDB_RADIX_XP bit 0, timing only: the pass without its LDS atomics)
        else atomicAdd(&hist_sh[rhash(rr[j].key) >> (32 - kbits)], 1u);
      }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += kRBlock) counts_T[(size_t)p * C + c] = hist_sh[p];
}

// ---- pass 2: the records of chunk c, sorted by partition 8192 rows at a time, to their places
// (Tried and dropped, profiles/r6_unique_radix.txt: ranks by ballots instead of LDS atomics that return a value — slower at 9-10 partition bits; whole 16-record
// units at 16-aligned positions with the remainders carried over in LDS, 4096-row tiles — every store a full line, and the pass took 11.7 ms instead of 7.3.)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_radix_partition(const uint64_t* __restrict__ sel, const void* __restrict__ col, int dtype, const uint64_t* __restrict__ missing,
                                                             int64_t nrows, int64_t rows_per_chunk, int kbits, const uint64_t* __restrict__ offsets_T,
                                                             uint64_t* __restrict__ keys_out, uint32_t* __restrict__ rows_out, int xp) {
  extern __shared__ uint64_t part_sh[];
  constexpr int TILE = 8 * BLOCK;                            // rows sorted at a time (8 per thread)
  const int P = 1 << kbits, C = (int)gridDim.x, c = (int)blockIdx.x;
  uint64_t* skey = part_sh;                                   // [TILE]
  uint64_t* cursor = skey + TILE;                           // [P]   where the chunk's next record of partition p goes
  uint32_t* srow = (uint32_t*)(cursor + P);                   // [TILE]  partition << 13 | row's offset inside the tile (the partition is not hashed again on the way out)
  uint32_t* hist2 = srow + TILE;                            // [P]   this tile's records per partition
  uint32_t* lstart = hist2 + P;                               // [P]   their first slot in skey / srow
  uint32_t* wsum = lstart + P;                                // [16]  scan scratch: one total per wave
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int p = tid; p < P; p += BLOCK) { cursor[p] = offsets_T[(size_t)p * C + c]; hist2[p] = 0; }
  __syncthreads();
  const int64_t r0 = (int64_t)c * rows_per_chunk, r1 = r0 + rows_per_chunk < nrows ? r0 + rows_per_chunk : nrows;
  RadixRow nx[8];                                             // the NEXT tile's rows: loaded while this tile is sorted and written
#pragma unroll
  for (int j = 0; j < 8; j++) nx[j] = radix_load(sel, col, dtype, missing, r0 + j * BLOCK + tid, r1);
  for (int64_t base = r0; base < r1; base += TILE) {
    // 1. keys, partitions, rank inside the tile's partition (an LDS atomic that returns a value: 0.8 ms of the pass per 1e9 rows)
    uint64_t key[8]; uint32_t pr[8];                          // pr: partition << 13 | rank  (rank < 8192), ~0 = no record
    bool tk[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      tk[j] = radix_take<false>(nx[j], dtype, base + j * BLOCK + tid, r1, nullptr);
      key[j] = nx[j].key;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      pr[j] = ~0u;
      if (tk[j]) {
        const uint32_t p = rhash(key[j]) >> (32 - kbits);
        pr[j] = p << 13 | ((xp & 2) ? (uint32_t)(j * BLOCK + tid) / (uint32_t)P : atomicAdd(&hist2[p], 1u));     // (DFDB_RADIX_XP bit 1, timing only: no rank atomics — and with them no stores)
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) nx[j] = radix_load(sel, col, dtype, missing, base + TILE + j * BLOCK + tid, r1);
    __syncthreads();
    // 2. exclusive scan of hist2 over the partitions (P <= 2048: at most two per thread)
    uint32_t h0 = tid < P ? hist2[tid] : 0u, h1 = tid + BLOCK < P ? hist2[tid + BLOCK] : 0u;
    uint32_t mine = h0 + h1, incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wv; w++) before += wsum[w];
    // (thread t scans partitions t and t + 1024 as one element: their slots are adjacent, t's first)
    const uint32_t ex = before + incl - mine;
    if (tid < P) lstart[tid] = ex;
    if (tid + BLOCK < P) lstart[tid + BLOCK] = ex + h0;
    __syncthreads();
    // 3. the tile's records into LDS, sorted by partition
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (pr[j] == ~0u) continue;
      const uint32_t p = pr[j] >> 13, slot = lstart[p] + (pr[j] & 8191u);
      skey[slot] = key[j];
      srow[slot] = p << 13 | (uint32_t)(j * BLOCK + tid);
    }
    __syncthreads();
    // 4. out: slot s's place is its partition's cursor + its rank in the tile's run
    uint32_t total = 0;
    for (int w = 0; w < BLOCK / 64; w++) total += wsum[w];
    for (uint32_t s = (uint32_t)tid; s < total; s += BLOCK) {
      const uint64_t k = skey[s];
      const uint32_t pw = srow[s], p = pw >> 13;
      const uint64_t dst = cursor[p] + (s - lstart[p]);
      if (xp & 4) { if (k == 12345ull) keys_out[dst] = k; continue; }                 // (DFDB_RADIX_XP bit 2, timing only: no stores)
      keys_out[dst] = k;
      if (!(xp & 16)) rows_out[dst] = (uint32_t)base + (pw & 8191u);                   // (DFDB_RADIX_XP bit 4, timing only: keys without their rows)
    }
    __syncthreads();
    // 5. the cursors move on
    for (int p = tid; p < P; p += BLOCK) { cursor[p] += hist2[p]; hist2[p] = 0; }
    __syncthreads();
  }
}

// ---- pass 3: one workgroup per partition (grid-strided): first occurrences out of a table in LDS
__global__ __launch_bounds__(kRBlock) void k_radix_unique(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ rows, const uint64_t* __restrict__ offsets_T,
                                                          int P, int C, uint64_t total, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, uint64_t* aux) {
  extern __shared__ uint64_t tab_sh[];
  uint64_t* tkey = tab_sh;                                    // [kRSlots]
  uint32_t* trow = (uint32_t*)(tkey + kRSlots);               // [kRSlots]
  __shared__ uint32_t claims_sh, abort_sh;
  for (int p = (int)blockIdx.x; p < P; p += (int)gridDim.x) {
    for (int i = threadIdx.x; i < kRSlots; i += kRBlock) { tkey[i] = kREmpty; trow[i] = 0xFFFFFFFFu; }
    if (threadIdx.x == 0) { claims_sh = 0; abort_sh = 0; }
    __syncthreads();
    const uint64_t a = offsets_T[(size_t)p * C], b = p + 1 < P ? offsets_T[(size_t)(p + 1) * C] : total;
    for (uint64_t i0 = a; i0 < b; i0 += 4 * kRBlock) {
      uint64_t kk[4]; uint32_t rw[4], hh[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { const uint64_t i = i0 + (uint64_t)j * kRBlock + threadIdx.x; kk[j] = i < b ? keys[i] : kREmpty; rw[j] = i < b ? rows[i] : 0u; }
#pragma unroll
      for (int j = 0; j < 4; j++) hh[j] = rhash(kk[j]) & (uint32_t)(kRSlots - 1);                        // (the partition is the hash's TOP bits)
      // every lane walks ITS four records at its own pace: one probe per trip, the next record as soon as this one is placed — a wave waits for the lane
      // with the most probes over four records, not for the slowest lane of every record (linear probing at 25-50 % load has a long tail)
      int j = 0; uint32_t probes = 0;
      uint64_t key = kk[0]; uint32_t row = rw[0], h = hh[0];
      while (j < 4) {
        bool placed = key == kREmpty;                                            // (past the partition's end: no record holds this image)
        if (!placed) {
          uint64_t old = tkey[h];
          if (old == kREmpty) {
            old = atomicCAS((unsigned long long*)&tkey[h], (unsigned long long)kREmpty, (unsigned long long)key);
            if (old == kREmpty) atomicAdd(&claims_sh, 1u);
          }
          if (old == kREmpty || old == key) { if (trow[h] > row) atomicMin(&trow[h], row); placed = true; }
          else { h = (h + 1) & (kRSlots - 1); if (++probes >= (uint32_t)kRSlots) { abort_sh = 1; placed = true; } }
        }
        if (placed) {
          j++; probes = 0;
          key = j == 1 ? kk[1] : (j == 2 ? kk[2] : kk[3]); row = j == 1 ? rw[1] : (j == 2 ? rw[2] : rw[3]); h = j == 1 ? hh[1] : (j == 2 ? hh[2] : hh[3]);
        }
      }
      // (the claims are read without a barrier: a late view only delays the stop — the flag is what the host looks at)
      if (claims_sh > (kRSlots * 7) / 8) { abort_sh = 1; break; }
    }
    __syncthreads();
    if (abort_sh) { if (threadIdx.x == 0) __atomic_store_n(&aux[3], 1ull, __ATOMIC_RELAXED); return; }      // aux[kAuxAbort]: this column needs the hash-table form
    for (int i = threadIdx.x; i < kRSlots; i += kRBlock) {
      if (tkey[i] == kREmpty) continue;
      const uint64_t r = trow[i];
      atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63));
      atomicAdd(&tile_counts[r >> 10], 1u);
    }
    if (p == 0 && threadIdx.x < 2) {                          // the two keys kept aside: the unstorable image, missing
      const uint64_t r = aux[threadIdx.x];
      if (r != kREmpty) { atomicOr((unsigned long long*)&bitmap[r >> 6], 1ull << (r & 63)); atomicAdd(&tile_counts[r >> 10], 1u); }
    }
    __syncthreads();
  }
}
}  // namespace

int64_t radix_rows_per_chunk(int64_t nrows, int chunks) {
  const int64_t per = (nrows + chunks - 1) / chunks;
  return (per + kRTile - 1) / kRTile * kRTile;              // whole tiles of the partition pass (and whole bitmap words)
}
static int radix_xp() { static const int v = [] { const char* e = getenv("DFDB_RADIX_XP"); return e ? atoi(e) : 0; }(); return v; }   // timing experiments only: results are WRONG with any bit set
size_t radix_partition_lds_bytes(int kbits) { const size_t P = (size_t)1 << kbits; return (size_t)kRTile * 12 + P * 16 + 64 + 64; }

bool launch_radix_hist(hipStream_t s, const uint64_t* sel, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int kbits, int chunks,
                       uint32_t* counts_T, uint64_t* aux) {
  if (kbits < 6 || kbits > 11) return false;
  hipLaunchKernelGGL(k_radix_hist, dim3(chunks), dim3(kRBlock), ((size_t)1 << kbits) * 4, s, sel, col, dtype, missing, nrows, radix_rows_per_chunk(nrows, chunks), kbits, counts_T, aux, radix_xp());
  return true;
}
bool launch_radix_partition(hipStream_t s, const uint64_t* sel, const void* col, int dtype, const uint64_t* missing, int64_t nrows, int kbits, int chunks,
                            const uint64_t* offsets_T, uint64_t* keys_out, uint32_t* rows_out) {
  // DFDB_RADIX_BLOCK (an A/B switch, read once): 1024-thread workgroups sorting 8192 rows at a time, one per CU (default), or 512-thread ones sorting 4096, two per CU
  static const int block = [] { const char* e = getenv("DFDB_RADIX_BLOCK"); return e && atoi(e) == 512 ? 512 : 1024; }();
  const size_t P = (size_t)1 << kbits;
  const size_t lds = (size_t)(8 * block) * 12 + P * 16 + 64 + 64;
  static bool ok = [] { return hipFuncSetAttribute((const void*)k_radix_partition<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess &&
                               hipFuncSetAttribute((const void*)k_radix_partition<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess; }();
  if (!ok || lds > 156 * 1024 || P > (size_t)2 * block) { (void)hipGetLastError(); return false; }
  const int64_t rpc = radix_rows_per_chunk(nrows, chunks);
  if (block == 512) hipLaunchKernelGGL(k_radix_partition<512>, dim3(chunks), dim3(512), lds, s, sel, col, dtype, missing, nrows, rpc, kbits, offsets_T, keys_out, rows_out, radix_xp());
  else hipLaunchKernelGGL(k_radix_partition<1024>, dim3(chunks), dim3(1024), lds, s, sel, col, dtype, missing, nrows, rpc, kbits, offsets_T, keys_out, rows_out, radix_xp());
  return true;
}
bool launch_radix_unique(hipStream_t s, const uint64_t* keys, const uint32_t* rows, const uint64_t* offsets_T, int kbits, int chunks, uint64_t total,
                         uint64_t* bitmap, uint32_t* tile_counts, uint64_t* aux, int cus) {
  const size_t lds = (size_t)kRSlots * 12;
  static bool ok = [] { return hipFuncSetAttribute((const void*)k_radix_unique, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess; }();
  if (!ok) { (void)hipGetLastError(); return false; }
  const int P = 1 << kbits;
  const int grid = P < cus ? P : cus;                       // one 96-KB table per CU at a time
  hipLaunchKernelGGL(k_radix_unique, dim3(grid), dim3(kRBlock), lds, s, keys, rows, offsets_T, P, chunks, total, bitmap, tile_counts, aux);
  return true;
}

}  // namespace dfdb
