// k_fused.hip — single-pass predicate scan + row-index compaction (K1 + count scan + K2 in one launch).
//
// For the headline query `selection(x -> x OP c)` -> ascending row indices, the three-kernel pipeline reads
// the bitmap back, pays two extra launches, and serialises index stores behind column loads.  Here one
// workgroup (16 waves) claims a 262 144-row chunk (= four blocks of the reference's DEFAULT_BLOCK_SIZE) with a ticket,
//   1. scans it exactly like K1 (coalesced nontemporal loads, ballot = bitmap word), keeping the chunk's
//      bitmap (8 KB) and 64 tile counts in LDS while also writing them to HBM for later gathers,
//   2. publishes the chunk's survivor count and obtains its global output offset by DECOUPLED LOOK-BACK
//      over the chunk descriptors (Merrill-Garland): the cross-block `offset` of RangeToProcess
//      (src/tables/selection.jl:68-75,107) computed without a separate scan pass,
//   3. expands the LDS bitmap into row numbers (per-lane word expansion into an LDS staging buffer, then
//      coalesced stores) at that offset.
// Ordering/visibility: a descriptor is ONE naturally aligned 8-byte word {status:2 | value:62} written and
// read with relaxed agent-scope atomics (sc1), so value and flag can never be seen torn or stale
// (MI355X_MICROARCH.md "R2 granule needs no ordering"); tickets are claimed in increasing order by running
// workgroups, so every predecessor of a chunk is already running -> forward progress without assuming
// dispatch order or co-residency.  Spins are bounded; an overrun raises a flag the host turns into an error.
//   algorithmic bytes / row: 8 + 1/8 + 12/1024 + 8 sigma
#include "device_utils.hpp"
#include "kernels.hpp"

namespace dfdb {

constexpr int kWaves = 16;                      // 1024-thread workgroups: few chunks in flight -> short look-backs
constexpr int kBlock = 64 * kWaves;
constexpr int kTilesPerWave = 16;
constexpr int kChunkTiles = kTilesPerWave * kWaves;   // 256 x 1024 rows = 262 144 rows (4 reference blocks) per chunk
constexpr int kTPL = kChunkTiles / 64;          // tile counts per lane in the chunk-level scan
constexpr uint64_t kStAgg = 1ull << 62, kStPrefix = 2ull << 62, kValMask = (1ull << 62) - 1;

template <int OP, typename T>
__device__ __forceinline__ bool fcmp(T x, T c) {
  if constexpr (OP == CMP_EQ) return x == c;
  else if constexpr (OP == CMP_NE) return x != c;
  else if constexpr (OP == CMP_LT) return x < c;
  else if constexpr (OP == CMP_LE) return x <= c;
  else if constexpr (OP == CMP_GT) return x > c;
  else return x >= c;
}

template <typename T, int OP>
__global__ __launch_bounds__(kBlock, 8) void k_scan_compact(const T* __restrict__ col, T c, uint64_t* __restrict__ bitmap,
                                                         uint32_t* __restrict__ tile_counts, uint64_t* __restrict__ prefix,
                                                         int64_t* __restrict__ out, int64_t out_cap, int64_t nrows, int64_t ntiles,
                                                         int64_t nchunks, int64_t row_base, uint64_t* __restrict__ desc,
                                                         uint32_t* __restrict__ ticket /* [0] ticket, [1] overrun flag */, int diag) {
  __shared__ uint64_t words[kChunkTiles * 16];      // the chunk's bitmap
  __shared__ uint32_t tcount[kChunkTiles];
  __shared__ uint32_t tpre[kChunkTiles];
  __shared__ uint16_t pos_sh[kWaves][1024];
  __shared__ int64_t chunk_sh;
  __shared__ uint64_t base_sh;
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  uint16_t* pos = pos_sh[wib];
  for (;;) {
    if (tid == 0) chunk_sh = (int64_t)atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t chunk = chunk_sh;
    if (chunk >= nchunks) return;

    // ---- 1. scan this wave's 16 tiles
    for (int k = 0; k < kTilesPerWave; k++) {
      const int lt = wib * kTilesPerWave + k;
      const int64_t tile = chunk * kChunkTiles + lt;
      uint64_t myword = 0;
      if (tile < ntiles) {
        const int64_t base = tile * 1024;
        const T* p = col + base + lane;
        if (base + 1024 <= nrows) {
          T v[16];
#pragma unroll
          for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
          for (int j = 0; j < 16; j++) { const uint64_t m = __ballot(fcmp<OP, T>(v[j], c)); if (lane == j) myword = m; }
        } else {
#pragma unroll
          for (int j = 0; j < 16; j++) {
            bool r = false;
            if (base + j * 64 + lane < nrows) r = fcmp<OP, T>(p[j * 64], c);
            const uint64_t m = __ballot(r); if (lane == j) myword = m;
          }
        }
      }
      uint32_t cnt = lane < 16 ? (uint32_t)__popcll(myword) : 0u;
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
      if (lane < 16) words[lt * 16 + lane] = myword;
      if (lane == 0) tcount[lt] = cnt;
      if (tile < ntiles) {
        if (lane < 16) bitmap[tile * 16 + lane] = myword;
        if (lane == 0) tile_counts[tile] = cnt;
      }
    }
    __syncthreads();

    // ---- 2. chunk aggregate, publish, look back
    if (wib == 0) {
      uint32_t c4[kTPL], cnt = 0;
#pragma unroll
      for (int i = 0; i < kTPL; i++) { c4[i] = tcount[lane * kTPL + i]; cnt += c4[i]; }
      const uint32_t incl = wave_incl_scan(cnt);
      uint32_t run = incl - cnt;
#pragma unroll
      for (int i = 0; i < kTPL; i++) { tpre[lane * kTPL + i] = run; run += c4[i]; }
      const uint64_t agg = __shfl(incl, 63, 64);
      if (lane == 0) __hip_atomic_store(&desc[chunk], (chunk == 0 ? kStPrefix : kStAgg) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint64_t base = 0;
      if (chunk > 0 && !(diag & 2)) {
        int64_t look = chunk - 1;
        uint32_t spins = 0;
        for (;;) {
          const int64_t idx = look - lane;
          const uint64_t d = idx >= 0 ? __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStPrefix;
          const uint64_t st = d >> 62;
          const uint64_t mp = __ballot(st == 2), mx = __ballot(st == 0);
          const int fp = mp ? __builtin_ctzll(mp) : 64, fx = mx ? __builtin_ctzll(mx) : 64;
          if (fx < fp) {                                   // a nearer predecessor has not published yet
            if (++spins > (1u << 24)) { if (lane == 0) atomicOr(&ticket[1], 1u); break; }
            __builtin_amdgcn_s_sleep(2);
            continue;
          }
          base += wave_sum64(lane <= fp ? (d & kValMask) : 0ull);
          if (fp < 64) break;
          look -= 64;
        }
        if (lane == 0) __hip_atomic_store(&desc[chunk], kStPrefix | (base + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        base_sh = base;
        if (chunk == nchunks - 1) prefix[ntiles] = base + agg;   // grand total
      }
    }
    __syncthreads();
    const uint64_t base = base_sh;
    if (tid < kChunkTiles) { const int64_t tile = chunk * kChunkTiles + tid; if (tile < ntiles) prefix[tile] = base + tpre[tid]; }

    // ---- 3. compaction of the 16 tiles this wave scanned, out of LDS: lane l owns 16 bits (rows 16l..16l+15 of
    //         the tile), expands them at its wave-prefix into a 2-KB staging buffer, then coalesced stores
    for (int k = 0; k < kTilesPerWave; k++) {
      const int lt = wib * kTilesPerWave + k;
      if (tcount[lt] == 0 || (diag & 1)) continue;         // wave-uniform
      uint32_t w = (uint32_t)(words[lt * 16 + (lane >> 2)] >> ((lane & 3) * 16)) & 0xffffu;
      const uint32_t pc = (uint32_t)__popc(w);
      const uint32_t incl = wave_incl_scan(pc);
      const uint32_t total = __shfl(incl, 63, 64);
      uint32_t o = incl - pc;
      const uint32_t lbase = (uint32_t)lane << 4;
      while (w) { const int b = __builtin_ctz(w); w &= w - 1; pos[o++] = (uint16_t)(lbase + (uint32_t)b); }
      wave_lds_fence();
      const int64_t obase = (int64_t)(base + tpre[lt]);
      const int64_t row1 = row_base + (chunk * kChunkTiles + lt) * 1024 + 1;
      for (uint32_t i = lane; i < total; i += 64) { const int64_t oo = obase + i; if (oo < out_cap) out[oo] = row1 + pos[i]; }
      wave_lds_fence();
    }
    __syncthreads();   // words/tcount/tpre are reused by the next chunk
  }
}

int g_fused_diag = 0;   // diagnostics only: 1 = skip index stores, 2 = skip look-back (wrong results, timing only)
void set_fused_diag(int d) { g_fused_diag = d; }
size_t fused_scratch_bytes(int64_t nrows) { return (size_t)((nrows + 65535) / 65536 + 8) * 8 + 64; }   // >= one descriptor per chunk

template <typename T, int OP>
static void launch_fused_t(hipStream_t s, const void* col, uint64_t cbits, uint64_t* bitmap, uint32_t* tc, uint64_t* prefix, int64_t* out,
                           int64_t out_cap, int64_t nrows, int64_t row_base, void* scratch) {
  const int64_t ntiles = (nrows + 1023) / 1024, nchunks = (ntiles + kChunkTiles - 1) / kChunkTiles;
  uint32_t* ticket = (uint32_t*)scratch;
  uint64_t* desc = (uint64_t*)((char*)scratch + 64);
  (void)hipMemsetAsync(scratch, 0, fused_scratch_bytes(nrows), s);
  int64_t grid = nchunks < 512 ? nchunks : 512;   // 2 workgroups of 16 waves per CU (67 KB LDS each)
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_scan_compact<T, OP>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const T*)col, from_bits<T>(cbits), bitmap, tc, prefix, out,
                     out_cap, nrows, ntiles, nchunks, row_base, desc, ticket, g_fused_diag);
}
template <typename T>
static void launch_fused_op(hipStream_t s, const void* col, int op, uint64_t cbits, uint64_t* bitmap, uint32_t* tc, uint64_t* prefix, int64_t* out,
                            int64_t out_cap, int64_t nrows, int64_t row_base, void* scratch) {
  switch (op) {
    case CMP_EQ: launch_fused_t<T, CMP_EQ>(s, col, cbits, bitmap, tc, prefix, out, out_cap, nrows, row_base, scratch); break;
    case CMP_NE: launch_fused_t<T, CMP_NE>(s, col, cbits, bitmap, tc, prefix, out, out_cap, nrows, row_base, scratch); break;
    case CMP_LT: launch_fused_t<T, CMP_LT>(s, col, cbits, bitmap, tc, prefix, out, out_cap, nrows, row_base, scratch); break;
    case CMP_LE: launch_fused_t<T, CMP_LE>(s, col, cbits, bitmap, tc, prefix, out, out_cap, nrows, row_base, scratch); break;
    case CMP_GT: launch_fused_t<T, CMP_GT>(s, col, cbits, bitmap, tc, prefix, out, out_cap, nrows, row_base, scratch); break;
    default:     launch_fused_t<T, CMP_GE>(s, col, cbits, bitmap, tc, prefix, out, out_cap, nrows, row_base, scratch); break;
  }
}

bool fused_supported(int32_t dtype) { return dtype == DFDB_I64 || dtype == DFDB_U64 || dtype == DFDB_F64 || dtype == DFDB_I32 || dtype == DFDB_F32; }

void launch_scan_compact(hipStream_t s, const void* col, int32_t dtype, int op, uint64_t cbits, uint64_t* bitmap, uint32_t* tile_counts,
                         uint64_t* prefix, int64_t* out, int64_t out_cap, int64_t nrows, int64_t row_base, void* scratch) {
  if (nrows <= 0) { (void)hipMemsetAsync(prefix, 0, 8, s); return; }
  switch (dtype) {
    case DFDB_I64: launch_fused_op<int64_t>(s, col, op, cbits, bitmap, tile_counts, prefix, out, out_cap, nrows, row_base, scratch); break;
    case DFDB_U64: launch_fused_op<uint64_t>(s, col, op, cbits, bitmap, tile_counts, prefix, out, out_cap, nrows, row_base, scratch); break;
    case DFDB_F64: launch_fused_op<double>(s, col, op, cbits, bitmap, tile_counts, prefix, out, out_cap, nrows, row_base, scratch); break;
    case DFDB_I32: launch_fused_op<int32_t>(s, col, op, cbits, bitmap, tile_counts, prefix, out, out_cap, nrows, row_base, scratch); break;
    default:       launch_fused_op<float>(s, col, op, cbits, bitmap, tile_counts, prefix, out, out_cap, nrows, row_base, scratch); break;
  }
}

}  // namespace dfdb
