// k_strings.hip — K4/K5/K6: FlatStringsVector columns on gfx950.
//
// Device layout of a String column (reference: src/FlatStringsVectors.jl:5-9): int32 sizes[nrows]
// (-1 = missing, contributes 0 bytes), one byte arena for the whole resident column, and — instead of the
// reference's per-row Int64 offsets vector (rebuilt serially by unsafe_remake_offsets!, :61-70) — one
// u64 byte offset per 1024-row tile.  A row's offset is tile_off[tile] + the wave prefix-sum of the sizes
// before it, recomputed on the fly, so the scan reads 4 B (size) + len B per row and no 8-B offset.
//   K4 tile byte totals + scan              replaces unsafe_remake_offsets!
//   K5 s OP "const" -> bitmap                replaces getindex -> unsafe_string -> == per element (:83-85,116-121)
//   K6 gather sizes + bytes of selected rows replaces getindex(a, r) (:136-157)
#include "device_utils.hpp"
#include "kernels.hpp"
#include <cstring>

namespace dfdb {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int64_t kTile = 1024;

static inline int grid_for(int64_t nunits, int cap = 4096) {
  int64_t b = (nunits + kWavesPerBlock - 1) / kWavesPerBlock;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ uint32_t clamp_size(int32_t s) { return s > 0 ? (uint32_t)s : 0u; }

// ---------------------------------------------------------------- K4
__global__ __launch_bounds__(kBlock) void k_str_tile_bytes(const int32_t* __restrict__ sizes, uint32_t* __restrict__ tile_bytes, int64_t nrows,
                                                           int64_t ntiles) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { const int64_t i = tile * kTile + j * 64 + lane; if (i < nrows) s += clamp_size(sizes[i]); }
    s = wave_sum(s);
    if (lane == 0) tile_bytes[tile] = s;
  }
}
void launch_str_tile_bytes(hipStream_t s, const int32_t* sizes, uint32_t* tile_bytes, int64_t nrows) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  hipLaunchKernelGGL(k_str_tile_bytes, dim3(grid_for(ntiles)), dim3(kBlock), 0, s, sizes, tile_bytes, nrows, ntiles);
}

// ---------------------------------------------------------------- K5
struct Pattern { uint64_t w[8]; int32_t len; };   // patterns up to 64 bytes travel in the kernel arguments

__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  return *(const u64u*)p;
}
// compare `len` bytes at p with the pattern (8 bytes at a time; the arena is padded so the last probe is safe)
__device__ __forceinline__ bool bytes_equal(const uint8_t* p, const Pattern& pat, int len) {
  int k = 0, wi = 0;
  for (; k + 8 <= len; k += 8, wi++) if (load_u64_unaligned(p + k) != pat.w[wi]) return false;
  const int rem = len - k;
  if (rem > 0) {
    const uint64_t mask = ~0ull >> (64 - 8 * rem);
    if ((load_u64_unaligned(p + k) & mask) != (pat.w[wi] & mask)) return false;
  }
  return true;
}
__device__ __forceinline__ bool bytes_equal_long(const uint8_t* p, const uint8_t* pat, int len) {
  for (int k = 0; k < len; k++) if (p[k] != pat[k]) return false;
  return true;
}

template <bool AND_EXISTING>
__global__ __launch_bounds__(kBlock) void k_str_match(const int32_t* __restrict__ sizes, const int64_t* __restrict__ tile_off,
                                                      const uint8_t* __restrict__ bytes, Pattern pat, const uint8_t* __restrict__ pat_long,
                                                      int mode, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, int64_t nrows,
                                                      int64_t ntiles) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int plen = pat.len;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    int64_t run = tile_off[tile];
    const int64_t base = tile * kTile;
    int32_t sz[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const int64_t i = base + j * 64 + lane; sz[j] = i < nrows ? sizes[i] : 0; }
    uint64_t myword = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t c = clamp_size(sz[j]);
      const uint32_t incl = wave_incl_scan(c);
      const int64_t off = run + (int64_t)(incl - c);
      run += (int64_t)__shfl(incl, 63, 64);
      const bool inb = base + j * 64 + lane < nrows;
      bool r = false;
      const int len = (int)c;
      if (inb) {
        if (mode <= 1) {            // == / != : length first, then bytes
          bool eq = len == plen;
          if (eq && plen > 0) eq = plen <= 64 ? bytes_equal(bytes + off, pat, plen) : bytes_equal_long(bytes + off, pat_long, plen);
          r = mode == 0 ? eq : !eq;
        } else if (len >= plen) {   // startswith / endswith
          const uint8_t* p = bytes + off + (mode == 3 ? len - plen : 0);
          r = plen == 0 ? true : (plen <= 64 ? bytes_equal(p, pat, plen) : bytes_equal_long(p, pat_long, plen));
        }
      }
      const uint64_t m = __ballot(r);
      if (lane == j) myword = m;
    }
    if (AND_EXISTING) { if (lane < 16) myword &= bitmap[tile * 16 + lane]; }
    uint32_t cnt = lane < 16 ? (uint32_t)__popcll(myword) : 0u;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane < 16) bitmap[tile * 16 + lane] = myword;
    if (lane == 0) tile_counts[tile] = cnt;
  }
}

void launch_str_match(hipStream_t s, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const uint8_t* pat_host,
                      const uint8_t* pat_dev, int32_t patlen, int mode, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows,
                      bool and_existing) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  Pattern pat; memset(&pat, 0, sizeof pat); pat.len = patlen;
  if (patlen > 0 && patlen <= 64) memcpy(pat.w, pat_host, (size_t)patlen);   // short patterns ride in the kernel arguments
  const int grid = grid_for(ntiles, 2048);
  if (and_existing) hipLaunchKernelGGL((k_str_match<true>), dim3(grid), dim3(kBlock), 0, s, sizes, tile_off, bytes, pat, pat_dev, mode, bitmap, tile_counts, nrows, ntiles);
  else hipLaunchKernelGGL((k_str_match<false>), dim3(grid), dim3(kBlock), 0, s, sizes, tile_off, bytes, pat, pat_dev, mode, bitmap, tile_counts, nrows, ntiles);
}

// ---------------------------------------------------------------- K6
constexpr int64_t kCTile = 4096;

// selected sizes -> out_sizes at the tile's output offset, plus the selected byte total per 4096-row ctile
__global__ __launch_bounds__(kBlock) void k_str_gather_sizes(const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ prefix,
                                                             const int32_t* __restrict__ sizes, int32_t* __restrict__ out_sizes,
                                                             uint32_t* __restrict__ sel_tile_bytes, int64_t nctiles, int64_t out_cap) {
  __shared__ uint16_t pos_sh[kWavesPerBlock][kCTile];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  uint16_t* pos = pos_sh[wib];
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t ct = wave; ct < nctiles; ct += nwaves) {
    uint64_t w = bitmap[ct * 64 + lane];
    const uint32_t c = (uint32_t)__popcll(w);
    const uint32_t incl = wave_incl_scan(c);
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t o = incl - c;
    while (w) { const int b = __builtin_ctzll(w); w &= w - 1; pos[o++] = (uint16_t)((lane << 6) + b); }
    wave_lds_fence();
    const int64_t obase = (int64_t)prefix[ct * 4];
    const int32_t* ts = sizes + ct * kCTile;
    uint32_t bsum = 0;
    for (uint32_t k = lane; k < total; k += 64) {
      const int32_t sz = ts[pos[k]];
      bsum += clamp_size(sz);
      if (obase + k < out_cap) out_sizes[obase + k] = sz;
    }
    bsum = wave_sum(bsum);
    if (lane == 0) sel_tile_bytes[ct] = bsum;
    wave_lds_fence();
  }
}
void launch_str_gather_sizes(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const int32_t* sizes, int32_t* out_sizes,
                             uint32_t* sel_tile_bytes, int64_t nrows, int64_t out_cap) {
  const int64_t nct = (nrows + kCTile - 1) / kCTile;
  if (nct == 0) return;
  hipLaunchKernelGGL(k_str_gather_sizes, dim3(grid_for(nct)), dim3(kBlock), 0, s, bitmap, prefix, sizes, out_sizes, sel_tile_bytes, nct, out_cap);
}

// bytes: walk the ctile 64 rows at a time carrying the source offset (all rows) and the destination offset
// (selected rows); a selected lane copies its string
__global__ __launch_bounds__(kBlock) void k_str_gather_bytes(const uint64_t* __restrict__ bitmap, const int32_t* __restrict__ sizes,
                                                             const int64_t* __restrict__ tile_off, const uint8_t* __restrict__ bytes,
                                                             const uint64_t* __restrict__ out_tile_off, uint8_t* __restrict__ out_bytes,
                                                             int64_t nrows, int64_t nctiles, int64_t out_cap) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t ct = wave; ct < nctiles; ct += nwaves) {
    const uint64_t myw = bitmap[ct * 64 + lane];
    if (__ballot(myw != 0) == 0) continue;           // nothing selected in this ctile
    int64_t src = tile_off[ct * 4];
    int64_t dst = (int64_t)out_tile_off[ct];
    for (int j = 0; j < 64; j++) {
      if (ct * kCTile + j * 64 >= nrows) break;      // wave-uniform: past the last row
      const uint64_t w = __shfl(myw, j, 64);         // word j, broadcast
      const int64_t i = ct * kCTile + j * 64 + lane;
      if (w == 0) {
        // still advance the source offset past these 64 rows; use the next tile offset when we cross a
        // 1024-row boundary instead of summing sizes
        if ((j & 15) == 15) { src = tile_off[ct * 4 + (j >> 4) + 1]; continue; }
        const uint32_t c = i < nrows ? clamp_size(sizes[i]) : 0u;
        src += (int64_t)wave_sum(c);
        continue;
      }
      const uint32_t c = i < nrows ? clamp_size(sizes[i]) : 0u;
      const uint32_t incl = wave_incl_scan(c);
      const bool sel = (w >> lane) & 1ull;
      const uint32_t cs = sel ? c : 0u;
      const uint32_t incls = wave_incl_scan(cs);
      if (sel && cs) {
        const uint8_t* sp = bytes + src + (int64_t)(incl - c);
        const int64_t d0 = dst + (int64_t)(incls - cs);
        if (d0 + cs <= out_cap) { uint8_t* dp = out_bytes + d0; for (uint32_t k = 0; k < cs; k++) dp[k] = sp[k]; }
      }
      src += (int64_t)__shfl(incl, 63, 64);
      dst += (int64_t)__shfl(incls, 63, 64);
    }
  }
}
void launch_str_gather_bytes(hipStream_t s, const uint64_t* bitmap, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes,
                             const uint64_t* out_tile_off, uint8_t* out_bytes, int64_t nrows, int64_t out_bytes_cap) {
  const int64_t nct = (nrows + kCTile - 1) / kCTile;
  if (nct == 0) return;
  hipLaunchKernelGGL(k_str_gather_bytes, dim3(grid_for(nct)), dim3(kBlock), 0, s, bitmap, sizes, tile_off, bytes, out_tile_off, out_bytes, nrows,
                     nct, out_bytes_cap);
}

}  // namespace dfdb
