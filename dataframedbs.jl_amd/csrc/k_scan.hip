// k_scan.hip — K1: predicate scan kernels for gfx950 (MI355X).
//
// Replaces, per 65 536-row block of the reference: _extract_for_eval! + materialize! + mask write-back
// (src/tables/broadcast.jl:96-133, src/tables/selection.jl:133-157) and the LogicalIndex count
// (selection.jl:166).  HBM-bound: every row of every referenced column is read exactly once with fully
// coalesced wave loads (lane l reads row base+l: one 512-B request per wave instruction for 8-byte
// types), the predicate result is a wavefront ballot (= one 64-bit word of the selection bitmap, LSB =
// lowest row), and a wave emits one 128-B line of bitmap + one tile count per 1024 rows.
//   algorithmic bytes / row: sum of referenced column widths + 1/8 (bitmap) + 4/1024 (count)
#include <type_traits>
#include "device_utils.hpp"
#include "kernels.hpp"
#include "../../include/dfdb_ir.h"

namespace dfdb {

constexpr int kBlock = 256;           // 4 waves
constexpr int kWavesPerBlock = 4;
constexpr int64_t kTile = 1024;       // rows per wave step
constexpr int kWordsPerTile = 16;

static inline int grid_for_tiles(int64_t ntiles) {
  // memory-bound streaming: 256 CUs x 8 blocks, grid-stride over tiles (cdna guide G11)
  int64_t blocks = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <int OP, typename T>
__device__ __forceinline__ bool cmp_op(T x, T c) {
  if constexpr (OP == CMP_EQ) return x == c;
  else if constexpr (OP == CMP_NE) return x != c;
  else if constexpr (OP == CMP_LT) return x < c;
  else if constexpr (OP == CMP_LE) return x <= c;
  else if constexpr (OP == CMP_GT) return x > c;
  else return x >= c;
}

// sum of popcounts held by lanes 0..15
__device__ __forceinline__ uint32_t tile_popcount(uint64_t myword, int lane) {
  uint32_t c = lane < kWordsPerTile ? (uint32_t)__popcll(myword) : 0u;
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  return c;  // valid in lanes 0..15 (each 16-lane group reduced separately; group 0 holds the tile)
}

// ------------------------------------------------------------------------------------------------
// single column  x OP c
// ------------------------------------------------------------------------------------------------
// lanes below this one that are set in m
__device__ __forceinline__ uint32_t rank_in(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// Capture: the selected values of a 64-row word are a run of ~6 at 10 % selectivity; stored straight to the tile's slot they are sixteen sub-line
// stores per tile.  They go to a 2-KB per-wave LDS buffer instead (one ds_write_b64 per word: nothing waits for it) and leave as contiguous
// NONTEMPORAL stores when the tile ends (or the buffer fills: selectivities above 25 %).  Measured on 1e9 rows at 10 % (tools/r3_scan2.py, one box):
// k_scan_cmp with capture 1.90 ms -> 1.55 with nontemporal stores alone (the plain scan is 1.25: written through L2 the captured values cost four
// times their share of the bytes); the two-term scan 3.62-3.75 -> 3.14-3.19 with the values packed in registers by ds_permute (a round trip per
// word) -> below with the LDS buffer.
#ifndef DFDB_CAP_STORE
#define DFDB_CAP_STORE 1
#endif
constexpr uint32_t kCapBuf = 256;      // values per wave
__device__ __forceinline__ void cap_store(uint64_t* p, uint64_t v) {
#if DFDB_CAP_STORE == 1
  __builtin_nontemporal_store(v, p);
#elif DFDB_CAP_STORE == 2
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
  *p = v;
#endif
}
struct CapQueue {
  uint64_t* lds;                        // this wave's kCapBuf slots
  uint32_t run = 0, out = 0;            // values in the buffer, values already stored (both wave-uniform)
  __device__ __forceinline__ explicit CapQueue(uint64_t* l) : lds(l) {}
};
__device__ __forceinline__ void cap_flush(CapQueue& cq, uint64_t* __restrict__ stage, int lane) {
  wave_lds_fence();
  for (uint32_t i = (uint32_t)lane; i < cq.run; i += 64u) cap_store(stage + cq.out + i, cq.lds[i]);
  wave_lds_fence();
  cq.out += cq.run; cq.run = 0;
}
__device__ __forceinline__ void cap_push(CapQueue& cq, uint64_t* __restrict__ stage, uint64_t m, uint64_t bits, int lane) {
  const uint32_t cnt = (uint32_t)__popcll(m);                       // wave-uniform
  if (cq.run + cnt > kCapBuf) cap_flush(cq, stage, lane);
  if ((m >> lane) & 1ull) cq.lds[cq.run + rank_in(m)] = bits;
  cq.run += cnt;
}
__device__ __forceinline__ void cap_finish(CapQueue& cq, uint64_t* __restrict__ stage, int lane) { if (cq.run) cap_flush(cq, stage, lane); }
template <typename T> __device__ __forceinline__ uint64_t bits_of(T v) { static_assert(sizeof(T) <= 8, ""); uint64_t b = 0; __builtin_memcpy(&b, &v, sizeof(T)); return b; }

// CAP: the values of the selected rows are also written, compacted per GROUP of four 1024-row tiles (what a wave takes per trip), to cap[group*4096 + rank in group]
// (round 5; per tile before: four 0.8-KB runs at 8-KB strides where there is now one 3.2-KB run — the scan's capture stores are what it pays for keeping the values,
// 0.45 ms per 0.8 GB mixed into its read stream, not the instructions around them):
// a projection of the predicate column itself is then a contiguous copy per group (k_compact_captured) instead of a gather
// that re-reads ~81 % of the column's 128-B lines at 10 % selectivity (late materialization: the scan already holds the values)
template <typename T, int OP, bool AND_EXISTING, bool NT, bool CAP>
__global__ __launch_bounds__(kBlock) void k_scan_cmp(const T* __restrict__ col, T c, uint64_t* __restrict__ bitmap,
                                                     uint32_t* __restrict__ tile_counts, int64_t nrows, int64_t ntiles, T* __restrict__ cap, int wt_store) {
  // CAP: the tile's selected values go straight to the tile's slot, a contiguous run per 64-row word (merged into full lines in L2).  Staging them in
  // LDS for full 512-byte stores (32 KB per workgroup: 5 workgroups per CU instead of 8) measured 3-4 % slower.
  uint64_t* stage = nullptr;
  __shared__ uint64_t cap_lds[CAP ? kWavesPerBlock * kCapBuf : 1];
  uint64_t* const cap_mine = cap_lds + (CAP ? (threadIdx.x >> 6) * kCapBuf : 0);
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  // A wave takes FOUR consecutive tiles per trip: lane 16k + j ends up with word j of tile k, so the bitmap leaves as ONE 512-byte
  // store (the buffer is padded to whole 4096-row groups) and the loads of the next tile overlap the ballots of this one: same-process
  // A/B against one tile per trip 1.342 -> 1.308, 1.375 -> 1.343 ms per 1e9 rows (-2.3 %).
  const int64_t ngroups = (ntiles + 3) / 4;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    const int64_t t0 = g * 4;
    uint64_t myword = 0;
    uint64_t existing = ~0ull;
    uint64_t live = ~0ull;
    if (AND_EXISTING) {   // late materialization: a tile no earlier stage left a survivor in is never read
      existing = bitmap[g * 64 + lane];
      live = __ballot(existing != 0);
      if (live == 0) { if ((lane & 15) == 0 && t0 + (lane >> 4) < ntiles) tile_counts[t0 + (lane >> 4)] = 0; continue; }
    }
    uint32_t gout = 0;                                                         // CAP: values this GROUP of four tiles has kept so far (its tiles' runs lie back to back)
    for (int k = 0; k < 4; k++) {
      const int64_t tile = t0 + k;
      if (tile >= ntiles) break;                                               // (wave-uniform)
      if (AND_EXISTING && ((live >> (16 * k)) & 0xffffull) == 0) continue;
      const int64_t base = tile * kTile;
      const T* p = col + base + lane;
      const int l0 = 16 * k;
      if (base + kTile <= nrows) {
        T v[kWordsPerTile];
#pragma unroll
        for (int j = 0; j < kWordsPerTile; j++) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];   // 16 independent coalesced loads in flight; NT: streamed once
        CapQueue cq(cap_mine);
        if (CAP) stage = (uint64_t*)cap + t0 * kTile + gout;
#pragma unroll
        for (int j = 0; j < kWordsPerTile; j++) {
          uint64_t m = __ballot(cmp_op<OP, T>(v[j], c));
          if (lane == l0 + j) myword = m;
          if (CAP) cap_push(cq, stage, m, bits_of(v[j]), lane);
        }
        if (CAP) { cap_finish(cq, stage, lane); gout += cq.out; }
      } else {
        CapQueue cq(cap_mine);
        if (CAP) stage = (uint64_t*)cap + t0 * kTile + gout;
#pragma unroll
        for (int j = 0; j < kWordsPerTile; j++) {
          const int64_t row = base + j * 64 + lane;
          bool r = false; T x = T(0);
          if (row < nrows) { x = p[j * 64]; r = cmp_op<OP, T>(x, c); }
          uint64_t m = __ballot(r);
          if (lane == l0 + j) myword = m;
          if (CAP) cap_push(cq, stage, m, bits_of(x), lane);
        }
        if (CAP) { cap_finish(cq, stage, lane); gout += cq.out; }
      }
    }
    if (AND_EXISTING) myword &= existing;
    uint32_t cnt = (uint32_t)__popcll(myword);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);          // every 16-lane group adds up its own tile
    // wt_store: write-through (system-scope) store — the bitmap is 1.5 % of the kernel's traffic but its write-backs out of L2,
    // interleaved with the read stream, cost 7-19 % of the pure read time; pushed straight through they cost 2-3 % less
    if (wt_store) __hip_atomic_store(&bitmap[g * 64 + lane], myword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else bitmap[g * 64 + lane] = myword;
    if ((lane & 15) == 0 && t0 + (lane >> 4) < ntiles) tile_counts[t0 + (lane >> 4)] = cnt;
  }
}


// ------------------------------------------------------------------------------------------------
// single column x OP c for 1-, 2- and 4-byte types (round 4)
// ------------------------------------------------------------------------------------------------
// k_scan_cmp's lane loads ONE element per instruction whatever its width: an Int32 column moves in 256-byte wave requests and a Bool / Int8 column in
// 64-byte ones, and the sixteen loads + sixteen ballots per 1024 rows that make an Int64 scan bandwidth-bound cap an Int32 scan at 0.4 and an Int8
// scan at 0.1 of the HBM peak.  Here a lane loads 16 BYTES = E = 16 / sizeof(T) consecutive rows (every wave request is 1024 bytes: 256 / 512 / 1024
// rows), compares them in registers into an E-bit piece of the mask, and the 64 / E lanes that share a 64-row word OR their pieces together with
// log2(64 / E) xor-shuffles.  A load therefore yields E finished words, held by the lane groups in row order; one ds_bpermute moves word j of the tile
// to lane l0 + j, where k_scan_cmp's ballots put it, and everything after that (four tiles per trip, one 512-byte bitmap store, tile counts, the
// late-materialization tile skip) is k_scan_cmp's.
template <typename T, int OP, bool AND_EXISTING>
__global__ __launch_bounds__(kBlock) void k_scan_cmp_narrow(const T* __restrict__ col, T c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts,
                                                            int64_t nrows, int64_t ntiles, int wt_store) {
  static_assert(sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4, "narrow types only");
  constexpr int E = 16 / (int)sizeof(T);        // rows per lane per load = words per load
  constexpr int G = 64 / E;                     // lanes per word
  constexpr int LOADS = kWordsPerTile / E;      // loads per tile: 4 / 2 / 1
  constexpr int64_t kSpan = 64 * E;             // rows per wave load
  // EIGHT 16-byte loads per lane are issued before the first is looked at (8 KB per wave in flight, what k_scan_cmp's sixteen 8-byte loads keep in
  // flight): 2 tiles of a 4-byte column, 4 tiles (one group) of a 2-byte column, 8 tiles (two groups = two bitmap lines) of a 1-byte column
  constexpr int GPT = sizeof(T) == 1 ? 2 : 1;   // four-tile groups per trip
  constexpr int BATCH = 8;                      // loads per batch
  constexpr int NB = GPT * 4 * LOADS / BATCH;   // batches per trip: 2 / 1 / 1
  constexpr int TPB = BATCH / LOADS;            // tiles per batch: 2 / 4 / 8
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ngroups = (ntiles + 3) / 4;
  const int64_t ntrips = (ngroups + GPT - 1) / GPT;
  const int j16 = lane & 15;                    // the word of its tile this lane will hold
  const int src_lane = G * (j16 % E);           // where that word sits after the group reduction of load j16 / E
  const int my_load = j16 / E;
  for (int64_t trip = wave; trip < ntrips; trip += nwaves) {
    const int64_t g0 = trip * GPT;
    uint64_t myword[GPT], existing[GPT], live[GPT];
#pragma unroll
    for (int u = 0; u < GPT; u++) {
      myword[u] = 0; existing[u] = ~0ull; live[u] = ~0ull;
      if (AND_EXISTING) {
        existing[u] = g0 + u < ngroups ? bitmap[(g0 + u) * 64 + lane] : 0ull;
        live[u] = __ballot(existing[u] != 0);
      }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
      // tiles [tb, tb + TPB) of the trip, all of one batch; tile x of the trip belongs to group x / 4, slot x % 4
      const int tb = b * TPB;
      u32x4 raw[BATCH];
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        const int x = tb + i / LOADS;                                          // tile of the trip
        const int64_t tile = g0 * 4 + x;
        const int64_t r0 = tile * kTile + (i % LOADS) * kSpan + (int64_t)E * lane;
        const bool dead = AND_EXISTING && ((live[x / 4] >> (16 * (x % 4))) & 0xffffull) == 0;   // (wave-uniform: no earlier stage left a survivor in this tile)
        const u32x4* p = (const u32x4*)(col + r0);
        if (dead || r0 >= nrows) raw[i] = u32x4{0u, 0u, 0u, 0u};              // (a partly valid vector ends < 16 bytes past the column: inside its 256-byte pad)
        else raw[i] = __builtin_nontemporal_load(p);
      }
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        const int x = tb + i / LOADS;
        const int64_t tile = g0 * 4 + x;
        const int64_t r0 = tile * kTile + (i % LOADS) * kSpan + (int64_t)E * lane;
        T v[E];
        __builtin_memcpy(v, &raw[i], 16);
        uint32_t piece = 0;
#pragma unroll
        for (int e = 0; e < E; e++) piece |= (cmp_op<OP, T>(v[e], c) ? 1u : 0u) << e;
        const int64_t left = nrows - r0;                                       // rows of this lane's vector that exist
        piece = left >= E ? piece : (left <= 0 ? 0u : piece & ((1u << left) - 1u));
        uint64_t w = (uint64_t)piece << (E * (lane % G));
#pragma unroll
        for (int d = G / 2; d >= 1; d >>= 1) w |= __shfl_xor(w, d, 64);       // every lane of a G-lane group now holds the group's 64-row word
        const uint64_t t = __shfl(w, src_lane, 64);
        if (my_load == i % LOADS && (lane >> 4) == x % 4) myword[x / 4] = t;
      }
    }
#pragma unroll
    for (int u = 0; u < GPT; u++) {
      const int64_t g = g0 + u;
      if (g >= ngroups) break;
      const int64_t t0 = g * 4;
      uint64_t mw = myword[u];
      if (AND_EXISTING) mw &= existing[u];
      uint32_t cnt = (uint32_t)__popcll(mw);
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
      if (wt_store) __hip_atomic_store(&bitmap[g * 64 + lane], mw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else bitmap[g * 64 + lane] = mw;
      if ((lane & 15) == 0 && t0 + (lane >> 4) < ntiles) tile_counts[t0 + (lane >> 4)] = cnt;
    }
  }
}

template <typename T, int OP>
static void launch_cmp_t(hipStream_t s, const void* col, uint64_t cbits, uint64_t* bitmap, uint32_t* tc, int64_t nrows, bool and_existing, bool nt, void* cap, int wt_store) {
  const T c = from_bits<T>(cbits);
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  const int grid = grid_for_tiles((ntiles + 3) / 4);
  const int narrow = (wt_store >> 1) & 3;        // ctx option "scan_narrow" rides in bits 1-2
  wt_store &= 1;
  if constexpr (sizeof(T) == 8) {
    if (cap && !and_existing) { hipLaunchKernelGGL((k_scan_cmp<T, OP, false, true, true>), dim3(grid), dim3(kBlock), 0, s, (const T*)col, c, bitmap, tc, nrows, ntiles, (T*)cap, wt_store); return; }
  }
  if constexpr (sizeof(T) < 8) {
    // sixteen bytes per lane.  Measured at 1e9 rows (tools/diag_types.py, profiles/r4_types.txt): 1-byte columns 0.72-0.73 of the HBM peak against 0.51-0.52
    // one element per lane; 2- and 4-byte columns 0.71-0.80 against 0.69-0.81 — no consistent difference, so by default only 1-byte columns take it
    // (bits 1-2 of wt_store = ctx option "scan_narrow": 1 = 1-byte columns (default), 2 = every narrow column, 0 = never)
    if ((narrow == 2 || (narrow == 1 && sizeof(T) == 1)) && ((uintptr_t)col & 15u) == 0) {
      const int ngrid = grid_for_tiles(sizeof(T) == 1 ? (ntiles + 7) / 8 : (ntiles + 3) / 4);
      if (and_existing) hipLaunchKernelGGL((k_scan_cmp_narrow<T, OP, true>), dim3(ngrid), dim3(kBlock), 0, s, (const T*)col, c, bitmap, tc, nrows, ntiles, wt_store & 1);
      else hipLaunchKernelGGL((k_scan_cmp_narrow<T, OP, false>), dim3(ngrid), dim3(kBlock), 0, s, (const T*)col, c, bitmap, tc, nrows, ntiles, wt_store & 1);
      return;
    }
  }
  (void)narrow;
  if (and_existing) hipLaunchKernelGGL((k_scan_cmp<T, OP, true, true, false>), dim3(grid), dim3(kBlock), 0, s, (const T*)col, c, bitmap, tc, nrows, ntiles, (T*)nullptr, wt_store);
  else if (nt) hipLaunchKernelGGL((k_scan_cmp<T, OP, false, true, false>), dim3(grid), dim3(kBlock), 0, s, (const T*)col, c, bitmap, tc, nrows, ntiles, (T*)nullptr, wt_store);
  else hipLaunchKernelGGL((k_scan_cmp<T, OP, false, false, false>), dim3(grid), dim3(kBlock), 0, s, (const T*)col, c, bitmap, tc, nrows, ntiles, (T*)nullptr, wt_store);
}
template <typename T>
static void launch_cmp_op(hipStream_t s, const void* col, int op, uint64_t cbits, uint64_t* bitmap, uint32_t* tc, int64_t nrows, bool ae, bool nt, void* cap, int wt) {
  switch (op) {
    case CMP_EQ: launch_cmp_t<T, CMP_EQ>(s, col, cbits, bitmap, tc, nrows, ae, nt, cap, wt); break;
    case CMP_NE: launch_cmp_t<T, CMP_NE>(s, col, cbits, bitmap, tc, nrows, ae, nt, cap, wt); break;
    case CMP_LT: launch_cmp_t<T, CMP_LT>(s, col, cbits, bitmap, tc, nrows, ae, nt, cap, wt); break;
    case CMP_LE: launch_cmp_t<T, CMP_LE>(s, col, cbits, bitmap, tc, nrows, ae, nt, cap, wt); break;
    case CMP_GT: launch_cmp_t<T, CMP_GT>(s, col, cbits, bitmap, tc, nrows, ae, nt, cap, wt); break;
    default:     launch_cmp_t<T, CMP_GE>(s, col, cbits, bitmap, tc, nrows, ae, nt, cap, wt); break;
  }
}

void launch_scan_cmp(hipStream_t s, const void* col, int32_t dtype, int op, uint64_t cbits, uint64_t* bitmap, uint32_t* tile_counts,
                     int64_t nrows, bool and_existing, bool nt, void* cap, int wt_store) {
  switch (dtype) {
    case DFDB_I8:  launch_cmp_op<int8_t>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    case DFDB_I16: launch_cmp_op<int16_t>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    case DFDB_I32: launch_cmp_op<int32_t>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    case DFDB_I64: launch_cmp_op<int64_t>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    case DFDB_U8: case DFDB_BOOL: launch_cmp_op<uint8_t>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    case DFDB_U16: launch_cmp_op<uint16_t>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    case DFDB_U32: launch_cmp_op<uint32_t>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    case DFDB_U64: launch_cmp_op<uint64_t>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    case DFDB_F32: launch_cmp_op<float>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
    default:       launch_cmp_op<double>(s, col, op, cbits, bitmap, tile_counts, nrows, and_existing, nt, cap, wt_store); break;
  }
}

// ------------------------------------------------------------------------------------------------
// read probe: what THIS box's HBM gives K1's load shape with nothing written beside it (dfdb_table_read_probe)
// ------------------------------------------------------------------------------------------------
// k_scan_cmp's trip — four consecutive tiles per wave, sixteen nontemporal 512-byte wave loads in flight per tile, the same grid — with the ballots, the bitmap
// and the tile counts taken out: every loaded value is folded into one register and a store happens only if the fold hits a value the host picks so that it
// cannot (the loads stay, nothing leaves).  bench.py prints the rate beside K1's: the distance between the two is the price of the 1/64 of bitmap written
// into the read stream, and the ceiling itself is what tells a slow allocation / a slow box from a slower kernel.
__global__ __launch_bounds__(kBlock) void k_read_probe(const uint64_t* __restrict__ col, int64_t nrows, int64_t ntiles, uint64_t never, uint64_t* __restrict__ sink) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ngroups = (ntiles + 3) / 4;
  uint64_t acc = 0;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    for (int k = 0; k < 4; k++) {
      const int64_t tile = g * 4 + k;
      if (tile >= ntiles) break;
      const int64_t base = tile * kTile;
      const uint64_t* p = col + base + lane;
      if (base + kTile <= nrows) {
        uint64_t v[kWordsPerTile];
#pragma unroll
        for (int j = 0; j < kWordsPerTile; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
        for (int j = 0; j < kWordsPerTile; j++) acc += v[j] ^ (uint64_t)j;
      } else {
        for (int j = 0; j < kWordsPerTile; j++) if (base + j * 64 + lane < nrows) acc += p[j * 64];
      }
    }
  }
  if (acc == never) sink[0] = acc;
}
void launch_read_probe(hipStream_t s, const void* col, int64_t nrows, uint64_t* sink) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  const int grid = grid_for_tiles((ntiles + 3) / 4);
  hipLaunchKernelGGL(k_read_probe, dim3(grid), dim3(kBlock), 0, s, (const uint64_t*)col, nrows, ntiles, 0x9e3779b97f4a7c15ull, sink);
}

// ------------------------------------------------------------------------------------------------
// conjunction / disjunction of simple terms over several columns (config 3: (a > c1) & (x < c2))
// ------------------------------------------------------------------------------------------------
// op -> which of {lt, eq, gt, unordered} satisfy it
__device__ __forceinline__ uint32_t op_sel(int op) {
  switch (op) {
    case CMP_EQ: return 2u; case CMP_NE: return 1u | 4u | 8u; case CMP_LT: return 1u;
    case CMP_LE: return 1u | 2u; case CMP_GT: return 4u; default: return 4u | 2u;
  }
}
template <typename T>
__device__ __forceinline__ bool cmp_sel(T x, T c, uint32_t sel) {
  const bool lt = x < c, eq = x == c, gt = x > c;
  return ((sel & 1u) && lt) || ((sel & 2u) && eq) || ((sel & 4u) && gt) || ((sel & 8u) && !(lt || eq || gt));
}
// the 16 bitmap words of one tile for one term; word j lands in lane l0 + j (l0 = 16 x the tile's place in its group of four)
// sel2 != 0: an interval term, both comparisons on the value just loaded (wave-uniform branch outside the unrolled loops)
template <typename T, bool NT = true>
__device__ __forceinline__ uint64_t term_word(const void* colv, uint64_t cbits, uint32_t sel, int64_t base, int64_t nrows, int lane, int l0 = 0,
                                              uint32_t sel2 = 0, uint64_t cbits2 = 0) {
  const T* p = (const T*)colv + base + lane;
  const T c = from_bits<T>(cbits), c2 = from_bits<T>(cbits2);
  uint64_t myword = 0;
  if (base + kTile <= nrows) {
    T v[kWordsPerTile];
#pragma unroll
    for (int j = 0; j < kWordsPerTile; j++) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
    if (sel2) {
#pragma unroll
      for (int j = 0; j < kWordsPerTile; j++) { uint64_t m = __ballot(cmp_sel<T>(v[j], c, sel) && cmp_sel<T>(v[j], c2, sel2)); if (lane == l0 + j) myword = m; }
    } else {
#pragma unroll
      for (int j = 0; j < kWordsPerTile; j++) { uint64_t m = __ballot(cmp_sel<T>(v[j], c, sel)); if (lane == l0 + j) myword = m; }
    }
  } else {
#pragma unroll
    for (int j = 0; j < kWordsPerTile; j++) {
      bool r = false;
      if (base + j * 64 + lane < nrows) { const T x = p[j * 64]; r = cmp_sel<T>(x, c, sel) && (sel2 == 0 || cmp_sel<T>(x, c2, sel2)); }
      uint64_t m = __ballot(r); if (lane == l0 + j) myword = m;
    }
  }
  return myword;
}

// ScanTerm.pre = 1: the compared value is rem(x, m) in Int64 (Julia `%`: the sign of the dividend).  |x| / |m| by the branch-free multiply-shift form
// of division by an invariant (magic and shift from the host: expr.cpp), one 64 x 64 -> high 64 multiply per row.
template <typename T>
__device__ __forceinline__ uint64_t term_word_rem(const ScanTerm& tm, uint32_t sel, uint32_t sel2, int64_t base, int64_t nrows, int lane, int l0) {
  const T* p = (const T*)tm.col + base + lane;
  const int64_t c = (int64_t)tm.cbits, c2 = (int64_t)tm.cbits2;
  const uint64_t magic = tm.pre_magic, d = tm.pre_d;
  const int sh = tm.pre_shift;
  auto rem64 = [&](int64_t x) -> int64_t {
    const uint64_t ux = x < 0 ? 0ull - (uint64_t)x : (uint64_t)x;
    const uint64_t q0 = __umul64hi(magic, ux);
    const uint64_t q = (((ux - q0) >> 1) + q0) >> sh;
    const uint64_t r = ux - q * d;
    return x < 0 ? -(int64_t)r : (int64_t)r;
  };
  uint64_t myword = 0;
  if (base + kTile <= nrows) {
    T v[kWordsPerTile];
#pragma unroll
    for (int j = 0; j < kWordsPerTile; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
    for (int j = 0; j < kWordsPerTile; j++) {
      const int64_t r = rem64((int64_t)v[j]);
      const uint64_t m = __ballot(cmp_sel<int64_t>(r, c, sel) && (sel2 == 0 || cmp_sel<int64_t>(r, c2, sel2)));
      if (lane == l0 + j) myword = m;
    }
  } else {
#pragma unroll
    for (int j = 0; j < kWordsPerTile; j++) {
      bool ok = false;
      if (base + j * 64 + lane < nrows) { const int64_t r = rem64((int64_t)p[j * 64]); ok = cmp_sel<int64_t>(r, c, sel) && (sel2 == 0 || cmp_sel<int64_t>(r, c2, sel2)); }
      const uint64_t m = __ballot(ok); if (lane == l0 + j) myword = m;
    }
  }
  return myword;
}

// ScanTerm.pre = 2 / 3: the compared value is x * k + d — in wrapping Int64 (FLT = false), or in Float64 with a rounding after the multiplication
// and another after the addition, as Julia's two operations round (FLT = true; __dmul_rn / __dadd_rn are never contracted into an fma)
template <typename T, int FLT>      // 0: Int64, 1: Float64 multiply + add, 2: Float64 division
__device__ __forceinline__ uint64_t term_word_affine(const ScanTerm& tm, uint32_t sel, uint32_t sel2, int64_t base, int64_t nrows, int lane, int l0) {
  const T* p = (const T*)tm.col + base + lane;
  using V = typename std::conditional<FLT != 0, double, int64_t>::type;
  const V c = from_bits<V>(tm.cbits), c2 = from_bits<V>(tm.cbits2);
  const V k = from_bits<V>(tm.pre_magic), d = from_bits<V>(tm.pre_d);
  auto f = [&](T x) -> V {
    if constexpr (FLT == 2) return __ddiv_rn((double)x, k);
    else if constexpr (FLT == 1) {
      // two roundings, like Julia's two operations: hipcc contracts a * b + c into an fma by default (and __dmul_rn / __dadd_rn are plain * and +
      // to it), which rounds once — one ulp off in 2 % of the rows of a 1e19-sized column (found by the fuzz soak).  The empty asm makes the
      // product an opaque value the add cannot fuse with.
      double prod = (double)x * k;
      asm volatile("" : "+v"(prod));
      return prod + d;
    }
    else return (int64_t)((uint64_t)(int64_t)x * (uint64_t)k + (uint64_t)d);
  };
  uint64_t myword = 0;
  if (base + kTile <= nrows) {
    T v[kWordsPerTile];
#pragma unroll
    for (int j = 0; j < kWordsPerTile; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
    for (int j = 0; j < kWordsPerTile; j++) {
      const V y = f(v[j]);
      const uint64_t m = __ballot(cmp_sel<V>(y, c, sel) && (sel2 == 0 || cmp_sel<V>(y, c2, sel2)));
      if (lane == l0 + j) myword = m;
    }
  } else {
#pragma unroll
    for (int j = 0; j < kWordsPerTile; j++) {
      bool ok = false;
      if (base + j * 64 + lane < nrows) { const V y = f(p[j * 64]); ok = cmp_sel<V>(y, c, sel) && (sel2 == 0 || cmp_sel<V>(y, c2, sel2)); }
      const uint64_t m = __ballot(ok); if (lane == l0 + j) myword = m;
    }
  }
  return myword;
}

// EXTRA = 2 / 3 / 4: sum / min / max of the finally selected values (Julia: Int sums wrap, min / max of Float64 propagate NaN)
template <typename T, int EXTRA> __device__ __forceinline__ T agg_identity() {
  if (EXTRA == 3) return std::is_same<T, double>::value ? (T)__builtin_inf() : (std::is_same<T, int64_t>::value ? (T)INT64_MAX : (T)~0ull);
  if (EXTRA == 4) return std::is_same<T, double>::value ? (T)-__builtin_inf() : (std::is_same<T, int64_t>::value ? (T)INT64_MIN : (T)0);
  return (T)0;
}
template <typename T, int EXTRA> __device__ __forceinline__ T agg_combine(T a, T b) {
  if (EXTRA == 2) return a + b;
  if (std::is_same<T, double>::value) {
    if (a != a) return a;
    if (b != b) return b;
    if (a == b) {                       // the two zeros: -0.0 is the minimum, 0.0 the maximum, whichever came first (Base.min / Base.max; k_compact.hip red_combine_f)
      const unsigned long long x = __double_as_longlong((double)a), y = __double_as_longlong((double)b);
      return (T)__longlong_as_double(EXTRA == 3 ? (x | y) : (x & y));
    }
  }
  if (EXTRA == 3) return b < a ? b : a;
  return b > a ? b : a;
}
template <typename T, int EXTRA> __device__ __forceinline__ T wave_agg(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const T t = __shfl_xor(v, d, 64); v = agg_combine<T, EXTRA>(v, t); }
  return v;
}
template <int EXTRA> __device__ __forceinline__ uint64_t agg_identity_bits(int dtype) {
  if (dtype == DFDB_F64) { const double d = agg_identity<double, EXTRA>(); uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }
  if (dtype == DFDB_I64) return (uint64_t)agg_identity<int64_t, EXTRA>();
  return agg_identity<uint64_t, EXTRA>();
}
// The LAST term of an AND of terms can do more than compare: when it is evaluated the mask of everything before it (the other
// terms, the earlier stages) is known, so the tile's final mask falls out word by word while the term's values are still in
// registers.  EXTRA = 1 (capture, see k_scan_cmp CAP): the values of the finally selected rows go to an LDS staging tile in rank
// order.  EXTRA = 2 (sum): they are added up per lane.  Nothing is kept across terms (the first capture version held the 16
// loaded values in 32 VGPRs until the mask was complete: +0.75 ms on the two-term scan of 1e9 rows).
template <typename T, int EXTRA>
__device__ __forceinline__ uint64_t term_word_last(const void* colv, uint64_t cbits, uint32_t sel, int64_t base, int64_t nrows, int lane, uint64_t before,
                                                   uint64_t* stage, uint32_t& run, T& lsum, int l0, uint32_t sel2 = 0, uint64_t cbits2 = 0, uint64_t* cap_lds = nullptr) {
  const T* p = (const T*)colv + base + lane;
  const T c = from_bits<T>(cbits), c2 = from_bits<T>(cbits2);
  uint64_t myword = 0;
  CapQueue cq(cap_lds);
  const bool full = base + kTile <= nrows;
  T v[kWordsPerTile];
#pragma unroll
  for (int j = 0; j < kWordsPerTile; j++) {
    if (full) v[j] = __builtin_nontemporal_load(p + j * 64);               // 16 loads in flight, like term_word
    else v[j] = base + j * 64 + lane < nrows ? p[j * 64] : T(0);
  }
#pragma unroll
  for (int j = 0; j < kWordsPerTile; j++) {
    const bool inb = full || base + j * 64 + lane < nrows;
    const uint64_t bj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)before, l0 + j) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(before >> 32), l0 + j) << 32;   // word j of the tile's mask so far
    const uint64_t m = __ballot(inb && cmp_sel<T>(v[j], c, sel) && (sel2 == 0 || cmp_sel<T>(v[j], c2, sel2))) & bj;
    if (lane == l0 + j) myword = m;
    const bool mine = (m >> lane) & 1ull;
    if (EXTRA == 1) cap_push(cq, stage, m, bits_of(v[j]), lane);
    if (EXTRA >= 2) { if (mine) lsum = agg_combine<T, EXTRA>(lsum, v[j]); }
  }
  if (EXTRA == 1) { cap_finish(cq, stage, lane); run = cq.out; }
  return myword;
}

// A term whose column is captured although it is NOT the last one evaluated (EXTRA = 5: two projected predicate columns): its mask word like
// term_word, and the tile's sixteen values per lane parked in LDS (stash[j * 64 + lane]) until the term after it has produced the final mask.
// LDS as explicit spill space: held in registers across the last term the 32 VGPRs cost the two-term scan +0.75 ms per 1e9 rows (round 1).
template <typename T>
__device__ __forceinline__ uint64_t term_word_stash(const void* colv, uint64_t cbits, uint32_t sel, int64_t base, int64_t nrows, int lane, int l0,
                                                    uint32_t sel2, uint64_t cbits2, uint64_t* stash) {
  const T* p = (const T*)colv + base + lane;
  const T c = from_bits<T>(cbits), c2 = from_bits<T>(cbits2);
  uint64_t myword = 0;
  const bool full = base + kTile <= nrows;
  T v[kWordsPerTile];
#pragma unroll
  for (int j = 0; j < kWordsPerTile; j++) {
    if (full) v[j] = __builtin_nontemporal_load(p + j * 64);
    else v[j] = base + j * 64 + lane < nrows ? p[j * 64] : T(0);
  }
#pragma unroll
  for (int j = 0; j < kWordsPerTile; j++) {
    const bool inb = full || base + j * 64 + lane < nrows;
    const uint64_t m = __ballot(inb && cmp_sel<T>(v[j], c, sel) && (sel2 == 0 || cmp_sel<T>(v[j], c2, sel2)));
    if (lane == l0 + j) myword = m;
    stash[j * 64 + lane] = bits_of(v[j]);
  }
  return myword;
}

template <typename T> __device__ __forceinline__ T wave_sum_t(T v);
template <> __device__ __forceinline__ double wave_sum_t<double>(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
template <> __device__ __forceinline__ uint64_t wave_sum_t<uint64_t>(uint64_t v) { return wave_sum64(v); }

// EXTRA: 0 plain; 1 capture the values of the LAST term's 8-byte column at the finally selected rows -> extra_out[tile*1024 + rank];
// 2 / 3 / 4 sum / min / max of them -> one partial per tile in extra_out (Float64 column: doubles; Int64 / UInt64 column: wrapping
// 64-bit sums, as Julia's; an empty tile holds the identity)
// 5 capture TWO columns: the last term's like EXTRA = 1 -> extra_out, and the term's before it -> extra_out2 (its tile parked in LDS while the last
// term is evaluated: term_word_stash).  materialize(t[(a > c1) & (x < c2) & ..., [:a, :x]]) then gathers neither column (never
// gather a column the scan already held; projection.jl:128-154 reads them per block for the same reason).
template <bool AND_EXISTING, int EXTRA>
__global__ __launch_bounds__(kBlock) void k_scan_terms(ScanTerms terms, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts,
                                                       int64_t nrows, int64_t ntiles, uint64_t* __restrict__ extra_out, uint64_t* __restrict__ extra_out2) {
  constexpr bool CAPQ = EXTRA == 1 || EXTRA == 5;      // a per-wave capture queue
  constexpr bool AGG = EXTRA >= 2 && EXTRA <= 4;
  constexpr int LASTX = EXTRA == 5 ? 1 : EXTRA;         // what the LAST term does with its values
  uint64_t* stage = nullptr;      // capture: the tile's slot in extra_out (see k_scan_cmp CAP)
  __shared__ uint64_t cap_lds[CAPQ ? kWavesPerBlock * kCapBuf : 1];
  __shared__ uint64_t stash_lds[EXTRA == 5 ? kWavesPerBlock * kTile : 1];
  uint64_t* const cap_mine = cap_lds + (CAPQ ? (threadIdx.x >> 6) * kCapBuf : 0);
  uint64_t* const stash = stash_lds + (EXTRA == 5 ? (threadIdx.x >> 6) * kTile : 0);
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int nplain = EXTRA == 5 ? terms.n - 2 : (EXTRA ? terms.n - 1 : terms.n);
  // four consecutive tiles per trip, like k_scan_cmp: lane 16k + j holds word j of tile k, one 512-byte bitmap store per group
  const int64_t ngroups = (ntiles + 3) / 4;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    const int64_t t0 = g * 4;
    const int nk = ntiles - t0 < 4 ? (int)(ntiles - t0) : 4;
    uint64_t existing = ~0ull, live = ~0ull;
    if (AND_EXISTING) {
      existing = bitmap[g * 64 + lane];
      live = __ballot(existing != 0);
      if (live == 0) {
        if ((lane & 15) == 0 && (lane >> 4) < nk) { tile_counts[t0 + (lane >> 4)] = 0; if (AGG) extra_out[t0 + (lane >> 4)] = agg_identity_bits<LASTX>(terms.t[terms.n - 1].dtype); }
        continue;
      }
    }
    uint64_t acc = terms.combine_or ? 0ull : ~0ull;
    for (int t = 0; t < nplain; t++) {
      const ScanTerm& tm = terms.t[t];
      const uint32_t sel = op_sel(tm.op), sel2 = tm.op2 >= 0 ? op_sel(tm.op2) : 0u;
      const uint64_t cb2 = tm.cbits2;
      uint64_t w = 0;
      for (int k = 0; k < nk; k++) {
        if (AND_EXISTING && ((live >> (16 * k)) & 0xffffull) == 0) continue;   // late materialization, tile by tile
        const int64_t base = (t0 + k) * kTile;
        const int l0 = 16 * k;
        if (tm.pre == 1) {     // rem(col, m) OP c: signed integer columns only (the host checks)
          switch (tm.dtype) {
            case DFDB_I8:  w |= term_word_rem<int8_t>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I16: w |= term_word_rem<int16_t>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I32: w |= term_word_rem<int32_t>(tm, sel, sel2, base, nrows, lane, l0); break;
            default:       w |= term_word_rem<int64_t>(tm, sel, sel2, base, nrows, lane, l0); break;
          }
          continue;
        }
        if (tm.pre == 2) {     // (col * k + d) OP c in wrapping Int64
          switch (tm.dtype) {
            case DFDB_I8:  w |= term_word_affine<int8_t, 0>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I16: w |= term_word_affine<int16_t, 0>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I32: w |= term_word_affine<int32_t, 0>(tm, sel, sel2, base, nrows, lane, l0); break;
            default:       w |= term_word_affine<int64_t, 0>(tm, sel, sel2, base, nrows, lane, l0); break;
          }
          continue;
        }
        if (tm.pre == 4) {     // Float64(col) / k
          switch (tm.dtype) {
            case DFDB_I8:  w |= term_word_affine<int8_t, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I16: w |= term_word_affine<int16_t, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I32: w |= term_word_affine<int32_t, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I64: w |= term_word_affine<int64_t, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_U8:  w |= term_word_affine<uint8_t, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_U16: w |= term_word_affine<uint16_t, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_U32: w |= term_word_affine<uint32_t, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_U64: w |= term_word_affine<uint64_t, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
            default:       w |= term_word_affine<double, 2>(tm, sel, sel2, base, nrows, lane, l0); break;
          }
          continue;
        }
        if (tm.pre == 3) {     // the same in Float64, any numeric column
          switch (tm.dtype) {
            case DFDB_I8:  w |= term_word_affine<int8_t, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I16: w |= term_word_affine<int16_t, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I32: w |= term_word_affine<int32_t, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_I64: w |= term_word_affine<int64_t, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_U8:  w |= term_word_affine<uint8_t, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_U16: w |= term_word_affine<uint16_t, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_U32: w |= term_word_affine<uint32_t, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_U64: w |= term_word_affine<uint64_t, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            case DFDB_F32: w |= term_word_affine<float, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
            default:       w |= term_word_affine<double, 1>(tm, sel, sel2, base, nrows, lane, l0); break;
          }
          continue;
        }
        switch (tm.dtype) {   // wave-uniform
          case DFDB_I8:  w |= term_word<int8_t>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          case DFDB_I16: w |= term_word<int16_t>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          case DFDB_I32: w |= term_word<int32_t>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          case DFDB_I64: w |= term_word<int64_t>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          case DFDB_U8: case DFDB_BOOL: w |= term_word<uint8_t>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          case DFDB_U16: w |= term_word<uint16_t>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          case DFDB_U32: w |= term_word<uint32_t>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          case DFDB_U64: w |= term_word<uint64_t>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          case DFDB_F32: w |= term_word<float>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
          default:       w |= term_word<double>(tm.col, tm.cbits, sel, base, nrows, lane, l0, sel2, cb2); break;
        }
      }
      acc = terms.combine_or ? (acc | w) : (acc & w);
    }
    // rows past nrows never set: every term's tail ballot is false (AND) — for OR also false
    if (AND_EXISTING) acc &= existing;
    if (EXTRA) {     // the last term (AND only, 8-byte dtypes only: the host checks), evaluated against the mask so far
      const ScanTerm& tm = terms.t[terms.n - 1];
      const uint32_t sel = op_sel(tm.op), sel2 = tm.op2 >= 0 ? op_sel(tm.op2) : 0u;
      const uint64_t cb2 = tm.cbits2;
      const uint64_t before = acc;
      uint64_t fin = 0;
      uint32_t gout = 0, gout2 = 0;          // captured values of this group so far (first / second capture buffer): a group's four runs lie back to back
      for (int k = 0; k < nk; k++) {
        const int64_t tile = t0 + k, base = tile * kTile;
        const int l0 = 16 * k;
        if (AND_EXISTING && ((live >> l0) & 0xffffull) == 0) { if (AGG && lane == 0) extra_out[tile] = agg_identity_bits<LASTX>(tm.dtype); continue; }
        uint32_t run = 0;
        uint64_t bef = before;
        if (EXTRA == 5) {           // the term before the last: its word of the mask, its values parked in LDS
          const ScanTerm& ta = terms.t[terms.n - 2];
          const uint32_t sa = op_sel(ta.op), sa2 = ta.op2 >= 0 ? op_sel(ta.op2) : 0u;
          wave_lds_fence();         // (the previous tile's reads of the stash are done)
          uint64_t wa;
          if (ta.dtype == DFDB_F64) wa = term_word_stash<double>(ta.col, ta.cbits, sa, base, nrows, lane, l0, sa2, ta.cbits2, stash);
          else if (ta.dtype == DFDB_I64) wa = term_word_stash<int64_t>(ta.col, ta.cbits, sa, base, nrows, lane, l0, sa2, ta.cbits2, stash);
          else wa = term_word_stash<uint64_t>(ta.col, ta.cbits, sa, base, nrows, lane, l0, sa2, ta.cbits2, stash);
          const bool mine16 = (lane >> 4) == k;            // the lanes that hold this tile's words
          bef = mine16 ? (before & wa) : before;
        }
        if (CAPQ) stage = extra_out + t0 * kTile + gout;   // the group's slot, behind what its earlier tiles kept
        uint64_t ft;
        if (tm.dtype == DFDB_F64) {
          double ls = agg_identity<double, LASTX>();
          ft = term_word_last<double, LASTX>(tm.col, tm.cbits, sel, base, nrows, lane, bef, stage, run, ls, l0, sel2, cb2, cap_mine);
          if (AGG) { ls = wave_agg<double, LASTX>(ls); if (lane == 0) { uint64_t b; __builtin_memcpy(&b, &ls, 8); extra_out[tile] = b; } }
        } else if (tm.dtype == DFDB_I64) {
          int64_t ls = agg_identity<int64_t, LASTX>();
          ft = term_word_last<int64_t, LASTX>(tm.col, tm.cbits, sel, base, nrows, lane, bef, stage, run, ls, l0, sel2, cb2, cap_mine);
          if (AGG) { ls = wave_agg<int64_t, LASTX>(ls); if (lane == 0) extra_out[tile] = (uint64_t)ls; }
        } else {
          uint64_t ls = agg_identity<uint64_t, LASTX>();
          ft = term_word_last<uint64_t, LASTX>(tm.col, tm.cbits, sel, base, nrows, lane, bef, stage, run, ls, l0, sel2, cb2, cap_mine);
          if (AGG) { ls = wave_agg<uint64_t, LASTX>(ls); if (lane == 0) extra_out[tile] = ls; }
        }
        fin |= ft;
        if (CAPQ) gout += run;
        if (EXTRA == 5) {           // the parked values of the rows that made it, in rank order, to the second capture buffer
          wave_lds_fence();
          CapQueue cq(cap_mine);
          uint64_t* const stage2 = extra_out2 + t0 * kTile + gout2;
#pragma unroll
          for (int j = 0; j < kWordsPerTile; j++) {
            const uint64_t mj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ft, l0 + j) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ft >> 32), l0 + j) << 32;
            cap_push(cq, stage2, mj, stash[j * 64 + lane], lane);
          }
          cap_finish(cq, stage2, lane);
          gout2 += cq.out;
        }
      }
      acc = fin;
    }
    uint32_t cnt = (uint32_t)__popcll(acc);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);          // every 16-lane group adds up its own tile
    __hip_atomic_store(&bitmap[g * 64 + lane], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // write-through, see k_scan_cmp
    if ((lane & 15) == 0 && (lane >> 4) < nk) tile_counts[t0 + (lane >> 4)] = cnt;
  }
}

// The two-column conjunction `(a OP c1) & (b OP c2)` over 8-byte columns (BASELINE configs 3 and 5; docs/src/index.md:503-517) with both
// column types known at compile time and the two read streams software-pipelined: while the sixteen values of a tile of `a` are compared, the
// loads of the same tile of `b` and then of the NEXT tile of `a` are already in flight (k_scan_terms takes one term after another through a
// dtype switch: a wave waits for sixteen loads, compares, and only then asks for the next sixteen).  EXTRA as in k_scan_terms, on `b`.
template <typename TA, typename TB, int EXTRA>
__global__ __launch_bounds__(kBlock) void k_scan_pair(ScanTerms terms, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts,
                                                      int64_t nrows, int64_t ntiles, uint64_t* __restrict__ extra_out) {
  __shared__ uint64_t cap_lds[EXTRA == 1 ? kWavesPerBlock * kCapBuf : 1];
  uint64_t* const cap_mine = cap_lds + (EXTRA == 1 ? (threadIdx.x >> 6) * kCapBuf : 0);
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const ScanTerm& ta = terms.t[0];
  const ScanTerm& tb = terms.t[1];
  const uint32_t sela = op_sel(ta.op), sela2 = ta.op2 >= 0 ? op_sel(ta.op2) : 0u;
  const uint32_t selb = op_sel(tb.op), selb2 = tb.op2 >= 0 ? op_sel(tb.op2) : 0u;
  const TA ca = from_bits<TA>(ta.cbits), ca2 = from_bits<TA>(ta.cbits2);
  const TB cb = from_bits<TB>(tb.cbits), cb2 = from_bits<TB>(tb.cbits2);
  const TA* __restrict__ cola = (const TA*)ta.col;
  const TB* __restrict__ colb = (const TB*)tb.col;
  const int64_t ngroups = (ntiles + 3) / 4;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    const int64_t t0 = g * 4;
    const int nk = ntiles - t0 < 4 ? (int)(ntiles - t0) : 4;
    uint64_t acc = 0;
    if ((t0 + 4) * kTile <= nrows) {
      // ---- four full tiles: the pipelined form
      const TA* pa = cola + t0 * kTile + lane;
      const TB* pb = colb + t0 * kTile + lane;
      TA va[kWordsPerTile]; TB vb[kWordsPerTile];
      uint32_t gout = 0;                     // captured values of this group so far
#pragma unroll
      for (int j = 0; j < kWordsPerTile; j++) va[j] = __builtin_nontemporal_load(pa + j * 64);
#pragma unroll
      for (int j = 0; j < kWordsPerTile; j++) vb[j] = __builtin_nontemporal_load(pb + j * 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int l0 = 16 * k;
        uint64_t wa = 0;
        if (sela2) {
#pragma unroll
          for (int j = 0; j < kWordsPerTile; j++) { const uint64_t m = __ballot(cmp_sel<TA>(va[j], ca, sela) && cmp_sel<TA>(va[j], ca2, sela2)); if (lane == l0 + j) wa = m; }
        } else {
#pragma unroll
          for (int j = 0; j < kWordsPerTile; j++) { const uint64_t m = __ballot(cmp_sel<TA>(va[j], ca, sela)); if (lane == l0 + j) wa = m; }
        }
        if (k < 3) {
#pragma unroll
          for (int j = 0; j < kWordsPerTile; j++) va[j] = __builtin_nontemporal_load(pa + (k + 1) * kTile + j * 64);
        }
        CapQueue cq(cap_mine);
        TB ls = agg_identity<TB, EXTRA>();
        uint64_t* stage = EXTRA == 1 ? extra_out + t0 * kTile + gout : nullptr;
#pragma unroll
        for (int j = 0; j < kWordsPerTile; j++) {
          uint64_t m = __ballot(cmp_sel<TB>(vb[j], cb, selb) && (selb2 == 0 || cmp_sel<TB>(vb[j], cb2, selb2)));
          if (EXTRA) {
            const uint64_t aj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wa, l0 + j) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wa >> 32), l0 + j) << 32;     // word j of a's mask
            m &= aj;
            const bool mine = (m >> lane) & 1ull;
            if (EXTRA == 1) cap_push(cq, stage, m, bits_of(vb[j]), lane);
            if (EXTRA >= 2) { if (mine) ls = agg_combine<TB, EXTRA>(ls, vb[j]); }
            if (lane == l0 + j) acc = m;
          } else if (lane == l0 + j) acc = m & wa;
        }
        if (k < 3) {
#pragma unroll
          for (int j = 0; j < kWordsPerTile; j++) vb[j] = __builtin_nontemporal_load(pb + (k + 1) * kTile + j * 64);
        }
        if (EXTRA == 1) { cap_finish(cq, stage, lane); gout += cq.out; }
        if (EXTRA >= 2) { ls = wave_agg<TB, EXTRA>(ls); if (lane == 0) { uint64_t b; __builtin_memcpy(&b, &ls, 8); extra_out[t0 + k] = b; } }
      }
    } else {
      // ---- the column's last group: tile by tile, bounds checked
      uint64_t wa = 0;
      for (int k = 0; k < nk; k++) wa |= term_word<TA>(ta.col, ta.cbits, sela, (t0 + k) * kTile, nrows, lane, 16 * k, sela2, ta.cbits2);
      if (EXTRA) {
        uint32_t gout = 0;
        for (int k = 0; k < nk; k++) {
          uint32_t run = 0;
          TB ls = agg_identity<TB, EXTRA>();
          acc |= term_word_last<TB, EXTRA>(tb.col, tb.cbits, selb, (t0 + k) * kTile, nrows, lane, wa, EXTRA == 1 ? extra_out + t0 * kTile + gout : nullptr, run, ls, 16 * k, selb2, tb.cbits2, cap_mine);
          gout += run;
          if (EXTRA >= 2) { ls = wave_agg<TB, EXTRA>(ls); if (lane == 0) { uint64_t b; __builtin_memcpy(&b, &ls, 8); extra_out[t0 + k] = b; } }
        }
      } else {
        uint64_t wb = 0;
        for (int k = 0; k < nk; k++) wb |= term_word<TB>(tb.col, tb.cbits, selb, (t0 + k) * kTile, nrows, lane, 16 * k, selb2, tb.cbits2);
        acc = wa & wb;
      }
    }
    uint32_t cnt = (uint32_t)__popcll(acc);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    __hip_atomic_store(&bitmap[g * 64 + lane], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // write-through, see k_scan_cmp
    if ((lane & 15) == 0 && (lane >> 4) < nk) tile_counts[t0 + (lane >> 4)] = cnt;
  }
}
template <typename TA, typename TB>
static void launch_pair_e(hipStream_t s, const ScanTerms& terms, uint64_t* bitmap, uint32_t* tc, int64_t nrows, int64_t ntiles, int extra, uint64_t* eo, dim3 g, dim3 b) {
  switch (extra) {
    case 1:  hipLaunchKernelGGL((k_scan_pair<TA, TB, 1>), g, b, 0, s, terms, bitmap, tc, nrows, ntiles, eo); break;
    case 2:  hipLaunchKernelGGL((k_scan_pair<TA, TB, 2>), g, b, 0, s, terms, bitmap, tc, nrows, ntiles, eo); break;
    case 3:  hipLaunchKernelGGL((k_scan_pair<TA, TB, 3>), g, b, 0, s, terms, bitmap, tc, nrows, ntiles, eo); break;
    case 4:  hipLaunchKernelGGL((k_scan_pair<TA, TB, 4>), g, b, 0, s, terms, bitmap, tc, nrows, ntiles, eo); break;
    default: hipLaunchKernelGGL((k_scan_pair<TA, TB, 0>), g, b, 0, s, terms, bitmap, tc, nrows, ntiles, eo); break;
  }
}
bool scan_pair_applies(const ScanTerms& terms, bool and_existing) {
  if (and_existing || terms.n != 2 || terms.combine_or || terms.t[0].pre || terms.t[1].pre) return false;
  const int da = terms.t[0].dtype, db = terms.t[1].dtype;
  return (da == DFDB_I64 || da == DFDB_F64) && (db == DFDB_I64 || db == DFDB_F64);
}
// true when the pair kernel took the launch: an AND of exactly two plain comparisons (or intervals) on Int64 / Float64 columns over a fresh mask
static bool launch_scan_pair(hipStream_t s, const ScanTerms& terms, uint64_t* bitmap, uint32_t* tc, int64_t nrows, int64_t ntiles, bool and_existing, int extra, uint64_t* eo) {
  if (!scan_pair_applies(terms, and_existing)) return false;
  const int da = terms.t[0].dtype, db = terms.t[1].dtype;
  const dim3 g(grid_for_tiles((ntiles + 3) / 4)), b(kBlock);
  if (da == DFDB_I64 && db == DFDB_I64) launch_pair_e<int64_t, int64_t>(s, terms, bitmap, tc, nrows, ntiles, extra, eo, g, b);
  else if (da == DFDB_I64) launch_pair_e<int64_t, double>(s, terms, bitmap, tc, nrows, ntiles, extra, eo, g, b);
  else if (db == DFDB_I64) launch_pair_e<double, int64_t>(s, terms, bitmap, tc, nrows, ntiles, extra, eo, g, b);
  else launch_pair_e<double, double>(s, terms, bitmap, tc, nrows, ntiles, extra, eo, g, b);
  return true;
}

void launch_scan_terms(hipStream_t s, const ScanTerms& terms, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows, bool and_existing,
                       int extra, void* extra_out, int pair, void* extra_out2) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  const dim3 g(grid_for_tiles((ntiles + 3) / 4)), b(kBlock);
  uint64_t* eo = (uint64_t*)extra_out; uint64_t* eo2 = (uint64_t*)extra_out2;
  if (pair && extra != 5 && launch_scan_pair(s, terms, bitmap, tile_counts, nrows, ntiles, and_existing, extra, eo)) return;
  if (extra == 1 && !and_existing) hipLaunchKernelGGL((k_scan_terms<false, 1>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 1) hipLaunchKernelGGL((k_scan_terms<true, 1>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 5 && !and_existing) hipLaunchKernelGGL((k_scan_terms<false, 5>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 5) hipLaunchKernelGGL((k_scan_terms<true, 5>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 2 && !and_existing) hipLaunchKernelGGL((k_scan_terms<false, 2>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 2) hipLaunchKernelGGL((k_scan_terms<true, 2>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 3 && !and_existing) hipLaunchKernelGGL((k_scan_terms<false, 3>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 3) hipLaunchKernelGGL((k_scan_terms<true, 3>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 4 && !and_existing) hipLaunchKernelGGL((k_scan_terms<false, 4>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (extra == 4) hipLaunchKernelGGL((k_scan_terms<true, 4>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, eo, eo2);
  else if (and_existing) hipLaunchKernelGGL((k_scan_terms<true, 0>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, (uint64_t*)nullptr, (uint64_t*)nullptr);
  else hipLaunchKernelGGL((k_scan_terms<false, 0>), g, b, 0, s, terms, bitmap, tile_counts, nrows, ntiles, (uint64_t*)nullptr, (uint64_t*)nullptr);
}

// ------------------------------------------------------------------------------------------------
// all-ones mask (fill!(range_buffer, 1): selection.jl:163) for an empty SelectionQueue
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ones_word(int64_t word_row0, int64_t nrows) {
  if (word_row0 + 64 <= nrows) return ~0ull;
  if (word_row0 >= nrows) return 0ull;
  return (1ull << (nrows - word_row0)) - 1ull;
}
__global__ __launch_bounds__(kBlock) void k_fill_ones(uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, int64_t nrows, int64_t ntiles) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    uint64_t w = lane < kWordsPerTile ? ones_word(tile * kTile + lane * 64, nrows) : 0ull;
    const uint32_t cnt = tile_popcount(w, lane);
    if (lane < kWordsPerTile) bitmap[tile * kWordsPerTile + lane] = w;
    if (lane == 0) tile_counts[tile] = cnt;
  }
}
void launch_fill_ones(hipStream_t s, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  hipLaunchKernelGGL(k_fill_ones, dim3(grid_for_tiles(ntiles)), dim3(kBlock), 0, s, bitmap, tile_counts, nrows, ntiles);
}

// ismissing(col) / !ismissing(col) over a Union{T,Missing} column: the column's missing bitmap IS the answer (docs/src/index.md:326-328
// count missing values this way); the interpreter spent 1.5 ms per 1e9 rows re-deriving it row by row
__global__ __launch_bounds__(kBlock) void k_missing_mask(const uint64_t* __restrict__ missing, int negate, int and_existing, uint64_t* __restrict__ bitmap,
                                                         uint32_t* __restrict__ tile_counts, int64_t nrows, int64_t ntiles) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    uint64_t w = 0;
    if (lane < kWordsPerTile) {
      const uint64_t valid = ones_word(tile * kTile + lane * 64, nrows);
      w = missing[tile * kWordsPerTile + lane];
      w = (negate ? ~w : w) & valid;
      if (and_existing) w &= bitmap[tile * kWordsPerTile + lane];
    }
    const uint32_t cnt = tile_popcount(w, lane);
    if (lane < kWordsPerTile) bitmap[tile * kWordsPerTile + lane] = w;
    if (lane == 0) tile_counts[tile] = cnt;
  }
}
void launch_missing_mask(hipStream_t s, const uint64_t* missing, bool negate, bool and_existing, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  hipLaunchKernelGGL(k_missing_mask, dim3(grid_for_tiles(ntiles)), dim3(kBlock), 0, s, missing, negate ? 1 : 0, and_existing ? 1 : 0, bitmap, tile_counts, nrows, ntiles);
}

// ------------------------------------------------------------------------------------------------
// range stage (selection.jl:94-111): survivor number offset+i is kept iff it is in the range.
// The cross-block `offset` of RangeToProcess is the global exclusive prefix of the mask popcounts.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool range_contains(const RangeSpec& r, int64_t rank) {
  if (r.kind == 0) {
    if (rank < r.first || rank > r.last) return false;
    return r.step == 1 || ((rank - r.first) % r.step) == 0;
  }
  int64_t lo = 0, hi = r.nsorted;
  while (lo < hi) { int64_t m = (lo + hi) >> 1; if (r.sorted[m] < rank) lo = m + 1; else hi = m; }
  return lo < r.nsorted && r.sorted[lo] == rank;
}

__global__ __launch_bounds__(kBlock) void k_range_stage(RangeSpec r, uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ prefix,
                                                        uint32_t* __restrict__ tile_counts, int64_t nrows, int64_t tile_first, int64_t tile_end,
                                                        int64_t rank_base, int implicit_ones) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = tile_first + wave; tile < tile_end; tile += nwaves) {
    uint64_t w = 0;
    if (lane < kWordsPerTile) w = implicit_ones ? ones_word(tile * kTile + lane * 64, nrows) : bitmap[tile * kWordsPerTile + lane];
    const uint32_t c = (uint32_t)__popcll(w);
    const uint32_t incl = wave_incl_scan(c);
    const int64_t tile_rank0 = rank_base + (implicit_ones ? tile * kTile : (int64_t)prefix[tile]);
    int64_t rank = tile_rank0 + (int64_t)(incl - c) + 1;   // rank of this word's first survivor
    uint64_t nw = 0;
    if (c) {
      const int64_t rlast = rank + c - 1;
      if (rlast < r.first || rank > r.last) nw = 0;
      else if (r.kind == 0 && r.step == 1 && rank >= r.first && rlast <= r.last) nw = w;
      else if (r.kind == 0) {
        // a:s:b — one modulo per word, then a running remainder (a 64-bit modulo per survivor made t[1:10:end, :] 6 ms per 1e9 rows)
        const int64_t d = rank - r.first;
        int64_t rem = r.step == 1 ? 0 : (d >= 0 ? d % r.step : (r.step - ((-d) % r.step)) % r.step);
        if (w == ~0ull) {   // a full word (a leading a:s:b over the table itself): the hits are every s-th bit from the first one
          const int64_t bmax = r.last - rank < 63 ? r.last - rank : 63;
          int64_t b = rem == 0 ? 0 : r.step - rem;
          if (rank + b < r.first) b += (r.first - rank - b + r.step - 1) / r.step * r.step;   // (only near the range's start)
          for (; b <= bmax; b += r.step) nw |= 1ull << b;
        } else {
          uint64_t ww = w;
          while (ww) {
            const uint64_t bit = ww & (0 - ww);
            if (rem == 0 && rank >= r.first && rank <= r.last) nw |= bit;
            ww ^= bit; rank++;
            if (r.step != 1) { rem++; if (rem == r.step) rem = 0; }
          }
        }
      } else {
        // sorted unique index list: one binary search for the word's first survivor, then the cursor only moves forward
        int64_t lo = 0, hi = r.nsorted;
        while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (r.sorted[m] < rank) lo = m + 1; else hi = m; }
        uint64_t ww = w;
        while (ww && lo < r.nsorted) {
          const uint64_t bit = ww & (0 - ww);
          if (r.sorted[lo] < rank) lo++;
          if (lo < r.nsorted && r.sorted[lo] == rank) nw |= bit;
          ww ^= bit; rank++;
        }
      }
    }
    const uint32_t cnt = tile_popcount(nw, lane);
    if (lane < kWordsPerTile) bitmap[tile * kWordsPerTile + lane] = nw;
    if (lane == 0) tile_counts[tile] = cnt;
  }
}
void launch_range_stage(hipStream_t s, const RangeSpec& r, uint64_t* bitmap, const uint64_t* prefix, uint32_t* tile_counts, int64_t nrows,
                        int64_t rank_base, bool implicit_ones) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  int64_t t0 = 0, t1 = ntiles;
  if (implicit_ones) {
    // a LEADING range numbers the table's own rows: only the tiles between its first and last element can hold a survivor
    // (skip_if_can / is_finished, selection.jl:177-196); the rest of the mask is cleared with two memsets — head(t) on 1e9 rows
    // was 1.3 ms of writing zeros tile by tile
    const int64_t lo = r.first - rank_base - 1, hi = r.last - rank_base - 1;      // local 0-based rows
    t0 = lo <= 0 ? 0 : (lo / kTile < ntiles ? lo / kTile : ntiles);
    t1 = hi < 0 ? t0 : (hi / kTile + 1 < ntiles ? hi / kTile + 1 : ntiles);
    if (t1 < t0) t1 = t0;
    if (t0 > 0) { (void)hipMemsetAsync(bitmap, 0, (size_t)t0 * kWordsPerTile * 8, s); (void)hipMemsetAsync(tile_counts, 0, (size_t)t0 * 4, s); }
    if (t1 < ntiles) {
      (void)hipMemsetAsync(bitmap + t1 * kWordsPerTile, 0, (size_t)(ntiles - t1) * kWordsPerTile * 8, s);
      (void)hipMemsetAsync(tile_counts + t1, 0, (size_t)(ntiles - t1) * 4, s);
    }
    if (t1 == t0) return;
  }
  hipLaunchKernelGGL(k_range_stage, dim3(grid_for_tiles(t1 - t0)), dim3(kBlock), 0, s, r, bitmap, prefix, tile_counts, nrows, t0, t1, rank_base,
                     implicit_ones ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of the per-tile counts: u32[ntiles] -> u64[ntiles+1]
// ------------------------------------------------------------------------------------------------
constexpr int kScChunk = 4096;       // counts per workgroup
constexpr int kScPerThread = kScChunk / kBlock;  // 16

__device__ __forceinline__ uint64_t block_sum64(uint64_t v, uint64_t* sh /*4*/) {
  v = wave_sum64(v);
  if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  uint64_t t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(kBlock) void k_sc_reduce(const uint32_t* __restrict__ counts, uint64_t* __restrict__ chunk_sums, int64_t ntiles) {
  __shared__ uint64_t sh[4];
  const int64_t base = (int64_t)blockIdx.x * kScChunk;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScPerThread; k++) {
    const int64_t i = base + k * kBlock + threadIdx.x;
    if (i < ntiles) s += counts[i];
  }
  s = block_sum64(s, sh);
  if (threadIdx.x == 0) chunk_sums[blockIdx.x] = s;
}

// single workgroup: in-place exclusive scan of chunk_sums[0..nchunks), total -> chunk_sums[nchunks]
__global__ __launch_bounds__(kBlock) void k_sc_scan_chunks(uint64_t* __restrict__ chunk_sums, int64_t nchunks) {
  __shared__ uint64_t wave_tot[4];
  __shared__ uint64_t carry_sh;
  if (threadIdx.x == 0) carry_sh = 0;
  __syncthreads();
  for (int64_t base = 0; base < nchunks; base += kBlock) {
    const int64_t i = base + threadIdx.x;
    const uint64_t v = i < nchunks ? chunk_sums[i] : 0;
    const uint64_t incl = wave_incl_scan64(v);
    if (lane_id() == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wbase += wave_tot[w];
    const uint64_t carry = carry_sh;
    if (i < nchunks) chunk_sums[i] = carry + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry_sh = carry + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_sums[nchunks] = carry_sh;
}

// carry_in / carry_out: the scan of one PIECE of a column continues the previous piece's (pipelined select_indices)
__global__ __launch_bounds__(kBlock) void k_sc_down(const uint32_t* __restrict__ counts, const uint64_t* __restrict__ chunk_sums,
                                                    uint64_t* __restrict__ prefix, int64_t ntiles, int64_t nchunks,
                                                    const uint64_t* __restrict__ carry_in, uint64_t* __restrict__ carry_out) {
  __shared__ uint64_t wave_tot[4];
  const int64_t base = (int64_t)blockIdx.x * kScChunk + (int64_t)threadIdx.x * kScPerThread;
  uint32_t v[kScPerThread];
  uint64_t tsum = 0;
#pragma unroll
  for (int k = 0; k < kScPerThread; k++) { const int64_t i = base + k; v[k] = i < ntiles ? counts[i] : 0u; tsum += v[k]; }
  const uint64_t incl = wave_incl_scan64(tsum);
  if (lane_id() == 63) wave_tot[threadIdx.x >> 6] = incl;
  __syncthreads();
  const uint64_t carry = carry_in ? *carry_in : 0ull;
  uint64_t run = carry + chunk_sums[blockIdx.x] + incl - tsum;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wave_tot[w];
#pragma unroll
  for (int k = 0; k < kScPerThread; k++) { const int64_t i = base + k; if (i < ntiles) prefix[i] = run; run += v[k]; }
  if (blockIdx.x == 0 && threadIdx.x == 0) { prefix[ntiles] = carry + chunk_sums[nchunks]; if (carry_out) *carry_out = carry + chunk_sums[nchunks]; }
}

// up to 4096 tiles (4 M rows): the three passes in ONE workgroup — a small table's query is a handful of 5-10 us launches, two fewer matter
__global__ __launch_bounds__(kBlock) void k_sc_small(const uint32_t* __restrict__ counts, uint64_t* __restrict__ prefix, int64_t ntiles,
                                                     const uint64_t* __restrict__ carry_in, uint64_t* __restrict__ carry_out) {
  __shared__ uint64_t wave_tot[4];
  const int64_t base = (int64_t)threadIdx.x * kScPerThread;
  uint32_t v[kScPerThread];
  uint64_t tsum = 0;
#pragma unroll
  for (int k = 0; k < kScPerThread; k++) { const int64_t i = base + k; v[k] = i < ntiles ? counts[i] : 0u; tsum += v[k]; }
  const uint64_t incl = wave_incl_scan64(tsum);
  if (lane_id() == 63) wave_tot[threadIdx.x >> 6] = incl;
  __syncthreads();
  const uint64_t carry = carry_in ? *carry_in : 0ull;
  uint64_t run = carry + incl - tsum;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wave_tot[w];
#pragma unroll
  for (int k = 0; k < kScPerThread; k++) { const int64_t i = base + k; if (i < ntiles) prefix[i] = run; run += v[k]; }
  if (threadIdx.x == kBlock - 1) { prefix[ntiles] = run; if (carry_out) *carry_out = run; }
}

size_t scan_counts_scratch_bytes(int64_t ntiles) { return (size_t)((ntiles + kScChunk - 1) / kScChunk + 2) * 8; }

void launch_scan_counts(hipStream_t s, const uint32_t* counts, uint64_t* prefix, int64_t ntiles, uint64_t* scratch, const uint64_t* carry_in,
                        uint64_t* carry_out) {
  if (ntiles <= 0) { (void)hipMemsetAsync(prefix, 0, 8, s); return; }
  if (ntiles <= kScChunk) { hipLaunchKernelGGL(k_sc_small, dim3(1), dim3(kBlock), 0, s, counts, prefix, ntiles, carry_in, carry_out); return; }
  const int64_t nchunks = (ntiles + kScChunk - 1) / kScChunk;
  hipLaunchKernelGGL(k_sc_reduce, dim3((unsigned)nchunks), dim3(kBlock), 0, s, counts, scratch, ntiles);
  hipLaunchKernelGGL(k_sc_scan_chunks, dim3(1), dim3(kBlock), 0, s, scratch, nchunks);
  hipLaunchKernelGGL(k_sc_down, dim3((unsigned)nchunks), dim3(kBlock), 0, s, counts, scratch, prefix, ntiles, nchunks, carry_in, carry_out);
}

}  // namespace dfdb
