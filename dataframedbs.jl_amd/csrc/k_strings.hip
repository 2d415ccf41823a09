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
                                                           int64_t ntiles, uint32_t* __restrict__ max_tile_bytes) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  uint32_t most = 0;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { const int64_t i = tile * kTile + j * 64 + lane; if (i < nrows) s += clamp_size(sizes[i]); }
    s = wave_sum(s);
    if (lane == 0) tile_bytes[tile] = s;
    most = s > most ? s : most;
  }
  if (max_tile_bytes && lane == 0 && most) atomicMax(max_tile_bytes, most);      // (one per wave: K5 stages a tile's bytes in LDS when the largest tile fits)
}
void launch_str_tile_bytes(hipStream_t s, const int32_t* sizes, uint32_t* tile_bytes, int64_t nrows, uint32_t* max_tile_bytes) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  hipLaunchKernelGGL(k_str_tile_bytes, dim3(grid_for(ntiles)), dim3(kBlock), 0, s, sizes, tile_bytes, nrows, ntiles, max_tile_bytes);
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
    uint64_t existing = ~0ull;
    if (AND_EXISTING) {
      existing = lane < 16 ? bitmap[tile * 16 + lane] : 0ull;
      if (__ballot(existing != 0) == 0) { if (lane == 0) tile_counts[tile] = 0; continue; }
    }
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
      const bool inb = base + j * 64 + lane < nrows && sz[j] >= 0;          // (a missing row — size -1 — selects nothing under any of the four terms)
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
    if (AND_EXISTING) myword &= existing;
    uint32_t cnt = lane < 16 ? (uint32_t)__popcll(myword) : 0u;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane < 16) bitmap[tile * 16 + lane] = myword;
    if (lane == 0) tile_counts[tile] = cnt;
  }
}

// Patterns up to 64 bytes (brand / event-type equality; the first 8 bytes decide for almost every row): three phases per 8 steps so that
// the byte probes of 8 x 64 rows are all in flight together instead of one dependent round trip per 64 rows:
//   A  wave prefix-sums of the sizes -> per-row byte offsets       (ALU only)
//   B  one unaligned 8-byte probe per candidate row (size matches) (8 loads in flight per lane)
//   C  masked compare + ballot = bitmap word
// exact copy of one string (len bytes): unaligned 8-byte moves, then ONE 8-byte load (the arenas are padded) and <= 3 stores
__device__ __forceinline__ void copy_string(uint8_t* dp, const uint8_t* sp, uint32_t len) {
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  typedef uint32_t __attribute__((aligned(1), may_alias)) u32u;
  typedef uint16_t __attribute__((aligned(1), may_alias)) u16u;
  uint32_t b = 0;
  for (; b + 8 <= len; b += 8) *(u64u*)(dp + b) = *(const u64u*)(sp + b);
  const uint32_t rem = len - b;
  if (rem) {
    uint64_t v = *(const u64u*)(sp + b);
    uint8_t* d = dp + b;
    if (rem & 4u) { *(u32u*)d = (uint32_t)v; d += 4; v >>= 32; }
    if (rem & 2u) { *(u16u*)d = (uint16_t)v; d += 2; v >>= 16; }
    if (rem & 1u) *d = (uint8_t)v;
  }
}

// the low `len` (<= 8) bytes of v
__device__ __forceinline__ void store_small(uint8_t* d, uint64_t v, uint32_t len) {
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  typedef uint32_t __attribute__((aligned(1), may_alias)) u32u;
  typedef uint16_t __attribute__((aligned(1), may_alias)) u16u;
  if (len >= 8u) { *(u64u*)d = v; return; }
  if (len & 4u) { *(u32u*)d = (uint32_t)v; d += 4; v >>= 32; }
  if (len & 2u) { *(u16u*)d = (uint16_t)v; d += 2; v >>= 16; }
  if (len & 1u) *d = (uint8_t)v;
}

// round 4: the tile's bytes arrive through LDS.  The probes of the candidate rows used to be the only readers of the arena — 8-byte loads at byte-granular
// addresses, a few active lanes per instruction, every 128-byte line of the arena fetched anyway because a line holds ~20 strings — and the pass ran at 0.58-0.64
// of the HBM peak.  Now the wave streams the tile's byte range [tile_off[t], tile_off[t+1]) with aligned 16-byte loads that are in flight together with the sixteen
// size loads, parks it in 8 KB of LDS, and the probes read there.  A tile whose strings do not fit (mean length above ~7) keeps the direct probes.
constexpr uint32_t kStageBytes = 8192;
__device__ __forceinline__ uint64_t lds_probe(const uint32_t* st, uint32_t off) {     // 8 bytes at byte offset `off` of the staged range
  const uint32_t i = off >> 2, sh = off & 3u;
  const uint32_t w0 = st[i], w1 = st[i + 1], w2 = st[i + 2];
  return (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32);
}

// CAP (dfdb_query_hint_materialize, the string column itself is projected): the pass that decides the rows also keeps them — the
// size of every selected row at cap_sizes[tile * 1024 + rank], its bytes packed at the tile's own arena offset in cap_bytes, the
// tile's selected byte total in sel_tile_bytes — so K6 does not read the column again: the projection is a contiguous copy per tile.
// LONG: patterns of 9..64 bytes (a second probe rides along, the rest is compared where 16 bytes matched); !LONG keeps the one-probe kernel as it was
template <bool AND_EXISTING, int MODE, bool CAP, bool LONG, bool STAGE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(CAP ? 2 : 4))) void k_str_match_short(const int32_t* __restrict__ sizes, const int64_t* __restrict__ tile_off,
                                                            const uint8_t* __restrict__ bytes, uint64_t patw, uint64_t patw1, int plen,
                                                            const uint8_t* __restrict__ pat_dev, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, int64_t nrows,
                                                            int64_t ntiles, int32_t* __restrict__ cap_sizes, uint8_t* __restrict__ cap_bytes,
                                                            uint32_t* __restrict__ sel_tile_bytes) {
  // patw / patw1: bytes 0..7 / 8..15 of the pattern; pat_dev: the whole pattern in device memory (read only where 16 bytes matched)
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const uint64_t mask = plen >= 8 ? ~0ull : ((1ull << (8 * plen)) - 1ull);
  const uint64_t want = patw & mask;
  const uint64_t mask2 = plen >= 16 ? ~0ull : (plen > 8 ? ((1ull << (8 * (plen - 8))) - 1ull) : 0ull);
  const uint64_t want2 = patw1 & mask2;
  __shared__ __attribute__((aligned(16))) uint32_t stage_sh[STAGE ? kWavesPerBlock : 1][STAGE ? kStageBytes / 4 + 8 : 4];
  uint32_t* const stage = stage_sh[STAGE ? (threadIdx.x >> 6) : 0];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    uint64_t existing = ~0ull;
    if (AND_EXISTING) {
      existing = lane < 16 ? bitmap[tile * 16 + lane] : 0ull;
      if (__ballot(existing != 0) == 0) { if (lane == 0) tile_counts[tile] = 0; continue; }
    }
    const uint8_t* tb = bytes + tile_off[tile];
    const int64_t base = tile * kTile;
    int32_t sz[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const int64_t i = base + j * 64 + lane; sz[j] = i < nrows ? __builtin_nontemporal_load(sizes + i) : -2; }
    // the tile's byte range (from the 16-byte boundary below it to 31 bytes past its end: a probe may read 15 bytes behind its string, and every arena is
    // allocated with 64 bytes of slack), eight 1-KB pieces at most, all in flight with the size loads
    const int64_t o0 = tile_off[tile], a0 = o0 & ~15ll;
    const uint32_t lead = (uint32_t)(o0 - a0), need = (uint32_t)(tile_off[tile + 1] - a0) + 16u;
    if (STAGE) {                                                       // (the launcher has checked that the column's largest tile fits)
      // (a piece past the range's end is the range's last piece once more, loaded and stored by several lanes alike: no predication, no divergence)
      const uint32_t lastc = (need - 1u) & ~15u;
      u32x4 piece[8];
#pragma unroll
      for (int i = 0; i < 8; i++) { uint32_t c = (uint32_t)i * 1024u + (uint32_t)lane * 16u; c = c < lastc ? c : lastc; piece[i] = __builtin_nontemporal_load((const u32x4*)(bytes + a0 + c)); }
#pragma unroll
      for (int i = 0; i < 8; i++) { uint32_t c = (uint32_t)i * 1024u + (uint32_t)lane * 16u; c = c < lastc ? c : lastc; *(u32x4*)(stage + (c >> 2)) = piece[i]; }
      wave_lds_fence();
    }
    uint64_t myword = 0;
    uint32_t run = 0;
    uint32_t cap_n = 0, cap_b = 0;                                    // CAP: selected rows / bytes of this tile so far
    uint8_t* const cb = CAP ? cap_bytes + tile_off[tile] : nullptr;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t rel[8]; uint64_t v[8], v2[LONG ? 8 : 1]; bool cand[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {                                   // A
        const uint32_t c = clamp_size(sz[h * 8 + j]);
        const uint32_t incl = wave_incl_scan(c);
        rel[j] = run + incl - c;
        run += __shfl(incl, 63, 64);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {                                   // B
        const int32_t s0 = sz[h * 8 + j];
        const int len = s0 > 0 ? s0 : 0;
        cand[j] = s0 >= 0 && (MODE <= 1 ? len == plen : len >= plen);          // (-2: past the last row; -1: missing — a comparison with missing selects nothing: coalesce(term, false))
        v[j] = 0; if (LONG) v2[j] = 0;
        if (cand[j] && plen > 0) {
          const uint32_t po = rel[j] + (uint32_t)(MODE == 3 ? len - plen : 0);
          if (STAGE) {
            v[j] = lds_probe(stage, lead + po);
            if (LONG) v2[j] = lds_probe(stage, lead + po + 8u);
          } else {
            v[j] = load_u64_unaligned(tb + po);
            if (LONG) v2[j] = load_u64_unaligned(tb + po + 8);         // bytes 8..15 of the compared span ride along
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {                                   // C
        bool eq = cand[j] && ((v[j] & mask) == want);
        if (LONG) eq = eq && ((v2[j] & mask2) == want2);
        if (LONG && plen > 16 && eq) {                                        // still longer: the few rows whose first 16 bytes match compare the rest
          const int32_t s1 = sz[h * 8 + j];
          const uint8_t* p = tb + rel[j] + (MODE == 3 ? (s1 > 0 ? s1 : 0) - plen : 0);
          eq = bytes_equal_long(p + 16, pat_dev + 16, plen - 16);
        }
        const bool r = MODE == 1 ? (sz[h * 8 + j] >= 0 && !eq) : eq;
        const uint64_t m = __ballot(r);
        if (lane == h * 8 + j) myword = m;
        if (CAP && m != 0) {
          const int32_t s0 = sz[h * 8 + j];
          const uint32_t len = s0 > 0 ? (uint32_t)s0 : 0u;
          const uint32_t rank = cap_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          uint32_t bpos;
          if (MODE == 0) { bpos = rank * (uint32_t)plen; cap_b += (uint32_t)__popcll(m) * (uint32_t)plen; }   // every selected row is plen bytes long
          else {
            const uint32_t sl = r ? len : 0u;
            const uint32_t incl = wave_incl_scan(sl);
            bpos = cap_b + incl - sl;
            cap_b += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
          }
          if (r) {
            cap_sizes[base + rank] = s0;
            if (MODE == 0 && !LONG) store_small(cb + bpos, v[j], len);       // the probe already holds the whole string (<= 8 bytes)
            else if (len) copy_string(cb + bpos, tb + rel[j], len);
          }
          cap_n += (uint32_t)__popcll(m);
        }
      }
    }
    if (CAP && lane == 0) sel_tile_bytes[tile] = cap_b;
    if (AND_EXISTING) myword &= existing;
    uint32_t cnt = lane < 16 ? (uint32_t)__popcll(myword) : 0u;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane < 16) __hip_atomic_store(&bitmap[tile * 16 + lane], myword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // write-through: see k_scan_cmp
    if (lane == 0) tile_counts[tile] = cnt;
  }
}

template <int MODE, bool STAGE>
static void launch_short(hipStream_t s, int grid, bool ae, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const Pattern& pat,
                         const uint8_t* pat_dev, uint64_t* bitmap, uint32_t* tc, int64_t nrows, int64_t ntiles, const StrCapture* cap) {
  dim3 g(grid), b(kBlock);
  int32_t* cs = cap ? cap->sizes : nullptr; uint8_t* cby = cap ? cap->bytes : nullptr; uint32_t* ctb = cap ? cap->tile_bytes : nullptr;
  if (pat.len > 8) {
    // the two-probe form keeps fewer waves resident and has little in flight per wave: with more workgroups than fit at once the last round runs on a
    // part-empty chip (s == "microsoft" over 5e8 rows: 1.38 ms on 2048 workgroups, 1.18 on exactly the resident 1280; the one-probe form does not care)
    static int resident = 0;
    if (resident == 0) {
      int per_cu = 0, dev = 0; hipDeviceProp_t pr;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_str_match_short<false, MODE, false, true, STAGE>, kBlock, 0) == hipSuccess && per_cu > 0 &&
          hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) resident = per_cu * pr.multiProcessorCount;
      else { (void)hipGetLastError(); resident = -1; }
    }
    if (resident > 0 && (int)g.x > resident) g.x = (unsigned)resident;
    if (cap) hipLaunchKernelGGL((k_str_match_short<false, MODE, true, true, STAGE>), g, b, 0, s, sizes, tile_off, bytes, pat.w[0], pat.w[1], (int)pat.len, pat_dev, bitmap, tc, nrows, ntiles, cs, cby, ctb);
    else if (ae) hipLaunchKernelGGL((k_str_match_short<true, MODE, false, true, STAGE>), g, b, 0, s, sizes, tile_off, bytes, pat.w[0], pat.w[1], (int)pat.len, pat_dev, bitmap, tc, nrows, ntiles, cs, cby, ctb);
    else hipLaunchKernelGGL((k_str_match_short<false, MODE, false, true, STAGE>), g, b, 0, s, sizes, tile_off, bytes, pat.w[0], pat.w[1], (int)pat.len, pat_dev, bitmap, tc, nrows, ntiles, cs, cby, ctb);
  } else {
    if (cap) hipLaunchKernelGGL((k_str_match_short<false, MODE, true, false, STAGE>), g, b, 0, s, sizes, tile_off, bytes, pat.w[0], pat.w[1], (int)pat.len, pat_dev, bitmap, tc, nrows, ntiles, cs, cby, ctb);
    else if (ae) hipLaunchKernelGGL((k_str_match_short<true, MODE, false, false, STAGE>), g, b, 0, s, sizes, tile_off, bytes, pat.w[0], pat.w[1], (int)pat.len, pat_dev, bitmap, tc, nrows, ntiles, cs, cby, ctb);
    else hipLaunchKernelGGL((k_str_match_short<false, MODE, false, false, STAGE>), g, b, 0, s, sizes, tile_off, bytes, pat.w[0], pat.w[1], (int)pat.len, pat_dev, bitmap, tc, nrows, ntiles, cs, cby, ctb);
  }
}

void launch_str_match(hipStream_t s, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const uint8_t* pat_host,
                      const uint8_t* pat_dev, int32_t patlen, int mode, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows,
                      bool and_existing, const StrCapture* cap, uint32_t max_tile_bytes) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  if (ntiles == 0) return;
  Pattern pat; memset(&pat, 0, sizeof pat); pat.len = patlen;
  if (patlen > 0 && patlen <= 64) memcpy(pat.w, pat_host, (size_t)patlen);   // short patterns ride in the kernel arguments
  const int grid = grid_for(ntiles, 2048);
  if (patlen <= 64) {
    // every tile of the column fits the 8 KB a wave stages (+ 15 bytes below the tile's first, + 16 behind its last): the bytes come through LDS
    const bool stage = patlen > 0 && max_tile_bytes > 0 && max_tile_bytes + 48u <= kStageBytes;
#define DFDB_SHORT(M) do { if (stage) launch_short<M, true>(s, grid, and_existing, sizes, tile_off, bytes, pat, pat_dev, bitmap, tile_counts, nrows, ntiles, cap); \
                           else launch_short<M, false>(s, grid, and_existing, sizes, tile_off, bytes, pat, pat_dev, bitmap, tile_counts, nrows, ntiles, cap); } while (0)
    switch (mode) { case 0: DFDB_SHORT(0); break; case 1: DFDB_SHORT(1); break; case 2: DFDB_SHORT(2); break; default: DFDB_SHORT(3); break; }
#undef DFDB_SHORT
    return;
  }
  if (and_existing) hipLaunchKernelGGL((k_str_match<true>), dim3(grid), dim3(kBlock), 0, s, sizes, tile_off, bytes, pat, pat_dev, mode, bitmap, tile_counts, nrows, ntiles);
  else hipLaunchKernelGGL((k_str_match<false>), dim3(grid), dim3(kBlock), 0, s, sizes, tile_off, bytes, pat, pat_dev, mode, bitmap, tile_counts, nrows, ntiles);
}

// ---------------------------------------------------------------- K6
// Both passes work on 1024-row tiles, one wave per tile.  Lane l owns a 16-bit slice of the tile's bitmap
// (rows 16l..16l+15) and expands it into a 2-KB LDS list of selected in-tile positions (wave prefix-sum of
// popcounts), so everything after that is proportional to the SELECTED rows.

// tiles a wave looks at together (their emptiness decided by one load): as many as leave ~8192 waves busy — a dense selection of a small table keeps a wave per tile
static int gather_group(int64_t nt) { int g = 1; while (g < 64 && nt / (g * 2) >= 8192) g *= 2; return g; }
// stage the selected positions of `tile`; returns how many
__device__ __forceinline__ uint32_t stage_tile_positions(const uint64_t* __restrict__ bitmap, int64_t tile, uint16_t* pos, int lane) {
  uint32_t w = (uint32_t)(bitmap[tile * 16 + (lane >> 2)] >> ((lane & 3) * 16)) & 0xffffu;
  const uint32_t pc = (uint32_t)__popc(w);
  const uint32_t incl = wave_incl_scan(pc);
  const uint32_t total = __shfl(incl, 63, 64);
  uint32_t o = incl - pc;
  const uint32_t lbase = (uint32_t)lane << 4;
  while (w) { const int b = __builtin_ctz(w); w &= w - 1; pos[o++] = (uint16_t)(lbase + (uint32_t)b); }
  wave_lds_fence();
  return total;
}

// pass 1: selected sizes -> out_sizes at the tile's row offset, plus the selected byte total of the tile
__global__ __launch_bounds__(kBlock) void k_str_gather_sizes(const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ prefix,
                                                             const int32_t* __restrict__ sizes, int32_t* __restrict__ out_sizes,
                                                             uint32_t* __restrict__ sel_tile_bytes, int64_t ntiles, int64_t out_cap, int group) {
  __shared__ uint16_t pos_sh[kWavesPerBlock][1024];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  uint16_t* pos = pos_sh[wib];
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  // `group` (<= 64) tiles at a time: their selected-row counts come out of the prefix sums with one coalesced load, the empty ones get their zero and are never looked at (a
  // sparse selection — ten first rows in 5e8 — used to pay one dependent bitmap load per tile: 60 us per pass)
  for (int64_t c0 = wave * group; c0 < ntiles; c0 += nwaves * group) {
   const int64_t tl = c0 + lane;
   const bool mine = lane < group && tl < ntiles;
   const bool some = mine && prefix[tl + 1] > prefix[tl];
   if (mine && !some) sel_tile_bytes[tl] = 0;
   for (uint64_t todo = __ballot(some); todo; todo &= todo - 1) {
    const int64_t tile = c0 + __builtin_ctzll(todo);
    const uint32_t total = stage_tile_positions(bitmap, tile, pos, lane);
    const int64_t obase = (int64_t)prefix[tile];
    const int32_t* ts = sizes + tile * kTile;
    uint32_t bsum = 0;
    for (uint32_t k = lane; k < total; k += 64) {
      const int32_t sz = ts[pos[k]];
      bsum += clamp_size(sz);
      if (obase + k < out_cap) out_sizes[obase + k] = sz;
    }
    bsum = wave_sum(bsum);
    if (lane == 0) sel_tile_bytes[tile] = bsum;
    wave_lds_fence();
   }
  }
}
void launch_str_gather_sizes(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const int32_t* sizes, int32_t* out_sizes,
                             uint32_t* sel_tile_bytes, int64_t nrows, int64_t out_cap) {
  const int64_t nt = (nrows + kTile - 1) / kTile;
  if (nt == 0) return;
  const int group = gather_group(nt);
  hipLaunchKernelGGL(k_str_gather_sizes, dim3(grid_for((nt + group - 1) / group)), dim3(kBlock), 0, s, bitmap, prefix, sizes, out_sizes, sel_tile_bytes, nt, out_cap, group);
}

// pass 2: bytes.  The tile's per-row source offsets (exclusive prefix of the sizes of ALL rows) go to LDS once
// (16 coalesced size loads + 16 in-register scans); then 64 selected rows at a time: destination offsets by a
// wave prefix-sum of the selected sizes, and every lane copies its own string.
__global__ __launch_bounds__(kBlock) void k_str_gather_bytes(const uint64_t* __restrict__ bitmap, const int32_t* __restrict__ sizes,
                                                             const int64_t* __restrict__ tile_off, const uint8_t* __restrict__ bytes,
                                                             const uint64_t* __restrict__ out_tile_off, uint8_t* __restrict__ out_bytes,
                                                             int64_t nrows, int64_t ntiles, int64_t out_cap, int group) {
  __shared__ uint16_t pos_sh[kWavesPerBlock][1024];
  __shared__ uint32_t pre_sh[kWavesPerBlock][1024];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  uint16_t* pos = pos_sh[wib];
  uint32_t* pre = pre_sh[wib];
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t c0 = wave * group; c0 < ntiles; c0 += nwaves * group) {      // (`group` tiles at a time, as in pass 1: a tile without selected BYTES has nothing to copy)
   const int64_t tl = c0 + lane;
   const bool some = lane < group && tl < ntiles && out_tile_off[tl + 1] > out_tile_off[tl];
   for (uint64_t todo = __ballot(some); todo; todo &= todo - 1) {
    const int64_t tile = c0 + __builtin_ctzll(todo);
    const uint32_t total = stage_tile_positions(bitmap, tile, pos, lane);
    if (total == 0) { wave_lds_fence(); continue; }        // wave-uniform: late materialization of the arena
    const int64_t base = tile * kTile;
    int32_t sz[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const int64_t i = base + j * 64 + lane; sz[j] = i < nrows ? sizes[i] : 0; }
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t c = clamp_size(sz[j]);
      const uint32_t incl = wave_incl_scan(c);
      pre[j * 64 + lane] = run + incl - c;
      run += __shfl(incl, 63, 64);
    }
    wave_lds_fence();
    const uint8_t* sb = bytes + tile_off[tile];
    int64_t drun = (int64_t)out_tile_off[tile];
    const int32_t* ts = sizes + base;
    for (uint32_t k0 = 0; k0 < total; k0 += 64) {
      const uint32_t k = k0 + lane;
      const bool valid = k < total;
      const uint32_t p = valid ? pos[k] : 0u;
      const uint32_t cs = valid ? clamp_size(ts[p]) : 0u;
      const uint32_t incl = wave_incl_scan(cs);
      if (cs) {
        const uint8_t* sp = sb + pre[p];
        const int64_t d0 = drun + (int64_t)(incl - cs);
        if (d0 + cs <= out_cap) {
          copy_string(out_bytes + d0, sp, cs);
        }
      }
      drun += (int64_t)__shfl(incl, 63, 64);
    }
    wave_lds_fence();
   }
  }
}
void launch_str_gather_bytes(hipStream_t s, const uint64_t* bitmap, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes,
                             const uint64_t* out_tile_off, uint8_t* out_bytes, int64_t nrows, int64_t out_bytes_cap) {
  const int64_t nt = (nrows + kTile - 1) / kTile;
  if (nt == 0) return;
  const int group = gather_group(nt);
  hipLaunchKernelGGL(k_str_gather_bytes, dim3(grid_for((nt + group - 1) / group)), dim3(kBlock), 0, s, bitmap, sizes, tile_off, bytes, out_tile_off, out_bytes, nrows,
                     nt, out_bytes_cap, group);
}

// projection of a String column whose selected rows K5 kept (CAP): per 1024-row tile a contiguous copy of its sizes and bytes
__global__ __launch_bounds__(kBlock) void k_str_compact_captured(const int32_t* __restrict__ cap_sizes, const uint8_t* __restrict__ cap_bytes,
                                                                 const uint64_t* __restrict__ prefix, const int64_t* __restrict__ tile_off,
                                                                 const uint64_t* __restrict__ out_tile_off, int32_t* __restrict__ out_sizes,
                                                                 uint8_t* __restrict__ out_bytes, int64_t ntiles, int64_t out_rows, int64_t out_bytes_cap) {
  // A tile keeps ~100 rows at 10 % selectivity: two dependent round trips (offsets, then the copy) for ~0.8 KB.  One tile per wave per trip was
  // latency-bound (0.256 ms per 5e8 rows = 3.1 TB/s); each 16-lane quarter of the wave takes its own tile, four tiles in flight per trip.
  const int lane = lane_id();
  const int sub = lane >> 4, sl = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  for (int64_t t0 = wave * 4; t0 < ntiles; t0 += nwaves * 4) {
    const int64_t tile = t0 + sub;
    if (tile >= ntiles) continue;
    const int64_t r0 = (int64_t)prefix[tile], r1 = (int64_t)prefix[tile + 1];
    const int64_t b0 = (int64_t)out_tile_off[tile], b1 = (int64_t)out_tile_off[tile + 1];
    const int32_t* ss = cap_sizes + tile * kTile;
    const uint8_t* sb = cap_bytes + tile_off[tile];
    for (int64_t k = sl; k < r1 - r0; k += 16) if (r0 + k < out_rows) out_sizes[r0 + k] = ss[k];
    const int64_t nb = b1 <= out_bytes_cap ? b1 - b0 : 0;
    const int64_t n8 = nb & ~7ll;
    for (int64_t k = (int64_t)sl * 8; k < n8; k += 128) *(u64u*)(out_bytes + b0 + k) = *(const u64u*)(sb + k);
    for (int64_t k = n8 + sl; k < nb; k += 16) out_bytes[b0 + k] = sb[k];
  }
}
void launch_str_compact_captured(hipStream_t s, const StrCapture& cap, const uint64_t* prefix, const int64_t* tile_off, const uint64_t* out_tile_off,
                                 int32_t* out_sizes, uint8_t* out_bytes, int64_t nrows, int64_t out_rows, int64_t out_bytes_cap) {
  const int64_t nt = (nrows + kTile - 1) / kTile;
  if (nt == 0) return;
  hipLaunchKernelGGL(k_str_compact_captured, dim3(grid_for((nt + 3) / 4)), dim3(kBlock), 0, s, cap.sizes, cap.bytes, prefix, tile_off, out_tile_off, out_sizes, out_bytes, nt,
                     out_rows, out_bytes_cap);
}

}  // namespace dfdb
