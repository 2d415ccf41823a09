// k_dict.hip — K9: dictionary codes for low-cardinality String columns (gfx950).
//
// The reference keeps every String column as a FlatStringsVector (sizes + bytes) and lists dictionary encoding under "Future plans"
// (docs/src/index.md); SURVEY.md §8f-4 carries it as a next row.  Here it is a SECOND, resident form of the column beside the flat one: one
// 16-bit code per row and the distinct strings once.  A string predicate (== / != / startswith / endswith) over such a column is evaluated
// on the few dictionary entries by the host and becomes a membership test of the codes (2 B per row instead of 4 + L); the projection of the
// column reads the selected rows' codes and copies the strings out of the dictionary.  Results are those of the flat kernels (K5 / K6) by
// construction: the codes only say WHICH string a row holds.
//
//   k_dict_encode   sizes + bytes -> codes, by lookup in an open-addressing table of the dictionary built so far (rows whose string is not in it
//                   report themselves: the host adds them and runs the pass again; the build gives up past max_entries)
//   k_dict_scan     codes -> selection bitmap + tile counts by a bit-table lookup (the K1 of dictionary columns)
//   k_dict_expand_* the projection: compacted codes (K3) -> sizes, then bytes out of the dictionary
#include "device_utils.hpp"
#include "kernels.hpp"

namespace dfdb {

namespace {
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int64_t kTile = 1024;

inline int grid_for(int64_t nunits, int cap = 4096) {
  int64_t b = (nunits + kWavesPerBlock - 1) / kWavesPerBlock;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}
__device__ __forceinline__ uint32_t clamp_size(int32_t s) { return s > 0 ? (uint32_t)s : 0u; }
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  return *(const u64u*)p;
}
__device__ __forceinline__ bool bytes_equal(const uint8_t* a, const uint8_t* b, uint32_t n) {
  for (uint32_t k = 0; k < n; k++) if (a[k] != b[k]) return false;
  return true;
}
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
}  // namespace

__host__ __device__ inline uint64_t dict_hash(uint64_t key8, uint32_t len) { return splitmix64(key8 ^ ((uint64_t)len * 0x9E3779B97F4A7C15ull)); }
uint64_t dict_hash_host(uint64_t key8, uint32_t len) { return dict_hash(key8, len); }

// one wave per 1024-row tile, rows j * 64 + lane like K5: sizes -> byte offsets by wave prefix sums, one unaligned 8-byte probe per row, table lookup
__global__ __launch_bounds__(kBlock) void k_dict_encode(const int32_t* __restrict__ sizes, const int64_t* __restrict__ tile_off, const uint8_t* __restrict__ bytes,
                                                        const DictSlot* __restrict__ slots, uint32_t slot_mask, const uint8_t* __restrict__ dict_bytes,
                                                        uint16_t* __restrict__ codes, int64_t nrows, int64_t ntiles, DictMiss miss) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const uint8_t* tb = bytes + tile_off[tile];
    const int64_t base = tile * kTile;
    uint32_t run = 0;
    for (int j = 0; j < 16; j++) {
      const int64_t i = base + j * 64 + lane;
      const int32_t s0 = i < nrows ? sizes[i] : 0;
      const uint32_t len = clamp_size(s0);
      const uint32_t incl = wave_incl_scan(len);
      const uint8_t* p = tb + run + incl - len;
      run += __shfl(incl, 63, 64);
      if (i >= nrows) continue;
      uint32_t code = 0xffffu;
      if (s0 >= 0) {                                                          // (a negative size is `missing`: nullable String columns are not encoded)
        const uint64_t v = len ? (load_u64_unaligned(p) & (len >= 8 ? ~0ull : ((1ull << (8 * len)) - 1ull))) : 0ull;
        uint32_t at = (uint32_t)dict_hash(v, len) & slot_mask;
        for (uint32_t probe = 0; probe <= slot_mask; probe++, at = (at + 1) & slot_mask) {
          const DictSlot sl = slots[at];
          if (sl.len == 0xffffffffu) break;                                   // empty slot: not in the dictionary (yet)
          if (sl.key8 == v && sl.len == len && (len <= 8 || bytes_equal(p + 8, dict_bytes + sl.off + 8, len - 8))) { code = sl.code; break; }
        }
      }
      codes[i] = (uint16_t)code;
      if (code == 0xffffu) {
        // report: the string itself goes to the staging arena if there is room, the host adds it to the dictionary
        // (a column with millions of distinct values must not queue millions of atomics on one address: once the records are full a row only looks)
        if (*(volatile unsigned long long*)miss.count >= (unsigned long long)miss.max_records) continue;
        const unsigned long long k = atomicAdd(miss.count, 1ull);
        if (k < (unsigned long long)miss.max_records && len <= (uint32_t)miss.max_len) {
          const unsigned long long o = atomicAdd(miss.bytes_used, (unsigned long long)len);
          if (o + len <= (unsigned long long)miss.bytes_cap) {
            miss.rec_off[k] = (uint32_t)o; miss.rec_len[k] = (int32_t)len;
            for (uint32_t b = 0; b < len; b++) miss.arena[o + b] = p[b];
          } else miss.rec_len[k] = -1;
        } else if (k < (unsigned long long)miss.max_records) miss.rec_len[k] = -2;   // longer than the dictionary takes
      }
    }
  }
}
void launch_dict_encode(hipStream_t s, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const DictSlot* slots, uint32_t nslots,
                        const uint8_t* dict_bytes, uint16_t* codes, int64_t nrows, const DictMiss& miss) {
  const int64_t nt = (nrows + kTile - 1) / kTile;
  if (nt == 0) return;
  hipLaunchKernelGGL(k_dict_encode, dim3(grid_for(nt, 2048)), dim3(kBlock), 0, s, sizes, tile_off, bytes, slots, nslots - 1, dict_bytes, codes, nrows, nt, miss);
}

// codes -> bitmap + tile counts: lane l of a wave takes 8 consecutive rows (one 16-byte load), looks each code up in the bit table (LDS) and writes
// one BYTE of the bitmap; a wave covers 512 rows per load, two loads per tile.  AND_EXISTING: the bytes are AND-ed into the mask of the stages before.
template <bool AND_EXISTING>
__global__ __launch_bounds__(kBlock) void k_dict_scan(const uint16_t* __restrict__ codes, const uint32_t* __restrict__ lut, int32_t lut_words,
                                                      uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts, int64_t nrows, int64_t ntiles) {
  __shared__ uint32_t lut_sh[2048];                                            // 65 536 codes at most
  for (int k = threadIdx.x; k < lut_words; k += kBlock) lut_sh[k] = lut[k];
  __syncthreads();
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  uint8_t* bm8 = (uint8_t*)bitmap;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * kTile;
    uint32_t cnt = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int64_t r0 = base + h * 512 + (int64_t)lane * 8;
      uint32_t w[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};    // (0xffff is never a member)
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      if (r0 + 8 <= nrows) { const u32x4 q = __builtin_nontemporal_load((const u32x4*)(codes + r0)); w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w; }
      else for (int k = 0; k < 8; k++) if (r0 + k < nrows) { const uint32_t c = codes[r0 + k]; w[k >> 1] = (w[k >> 1] & ~(0xffffu << (16 * (k & 1)))) | (c << (16 * (k & 1))); }
      uint32_t byte = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t c = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
        const uint32_t in = (c >> 5) < (uint32_t)lut_words ? (lut_sh[c >> 5] >> (c & 31u)) & 1u : 0u;
        byte |= in << k;
      }
      const int64_t bi = tile * 128 + h * 64 + lane;
      if (AND_EXISTING) byte &= bm8[bi];
      bm8[bi] = (uint8_t)byte;
      cnt += (uint32_t)__popc(byte);
    }
    cnt = wave_sum(cnt);
    if (lane == 0) tile_counts[tile] = cnt;
  }
}
void launch_dict_scan(hipStream_t s, const uint16_t* codes, const uint32_t* lut, int32_t lut_words, uint64_t* bitmap, uint32_t* tile_counts, int64_t nrows,
                      bool and_existing) {
  const int64_t nt = (nrows + kTile - 1) / kTile;
  if (nt == 0) return;
  if (and_existing) hipLaunchKernelGGL(k_dict_scan<true>, dim3(grid_for(nt, 2048)), dim3(kBlock), 0, s, codes, lut, lut_words, bitmap, tile_counts, nrows, nt);
  else hipLaunchKernelGGL(k_dict_scan<false>, dim3(grid_for(nt, 2048)), dim3(kBlock), 0, s, codes, lut, lut_words, bitmap, tile_counts, nrows, nt);
}

// Projection of a dictionary column.  K3 first compacts the selected rows' CODES (a 2-byte gather); what is left is flat work over the selected rows in
// table order, 1024 per wave: coalesced code loads, dictionary lookups that hit in L2, coalesced size stores; then, once the tiles' byte totals are
// scanned, every row copies its dictionary entry to its place (destination offsets by wave prefix sums).  (Gathering sizes and bytes tile by tile
// straight off the bitmap, as K6 does for flat columns, is a chain of dependent round trips per tile: 0.30 + 0.38 ms per 5e8 rows at 10 %.)
__global__ __launch_bounds__(kBlock) void k_dict_expand_sizes(const uint16_t* __restrict__ codes, int64_t n, const int32_t* __restrict__ dict_len,
                                                              int32_t* __restrict__ out_sizes, uint32_t* __restrict__ out_tile_bytes, int64_t ntiles) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * kTile;
    uint32_t c[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const int64_t i = base + j * 64 + lane; c[j] = i < n ? codes[i] : 0xffffffffu; }
    uint32_t bsum = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int64_t i = base + j * 64 + lane;
      if (c[j] != 0xffffffffu) { const int32_t sz = dict_len[c[j]]; out_sizes[i] = sz; bsum += clamp_size(sz); }
    }
    bsum = wave_sum(bsum);
    if (lane == 0) out_tile_bytes[tile] = bsum;
  }
}
__global__ __launch_bounds__(kBlock) void k_dict_expand_bytes(const uint16_t* __restrict__ codes, int64_t n, const int32_t* __restrict__ dict_len,
                                                              const uint32_t* __restrict__ dict_off, const uint8_t* __restrict__ dict_bytes,
                                                              const uint64_t* __restrict__ out_tile_off, uint8_t* __restrict__ out_bytes, int64_t ntiles, int64_t out_cap) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * kTile;
    uint32_t c[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const int64_t i = base + j * 64 + lane; c[j] = i < n ? codes[i] : 0xffffffffu; }
    uint32_t len[16], off[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { len[j] = c[j] != 0xffffffffu ? clamp_size(dict_len[c[j]]) : 0u; off[j] = c[j] != 0xffffffffu ? dict_off[c[j]] : 0u; }
    int64_t drun = (int64_t)out_tile_off[tile];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t incl = wave_incl_scan(len[j]);
      if (len[j]) {
        const int64_t d0 = drun + (int64_t)(incl - len[j]);
        if (d0 + len[j] <= out_cap) copy_string(out_bytes + d0, dict_bytes + off[j], len[j]);
      }
      drun += (int64_t)__shfl(incl, 63, 64);
    }
  }
}
void launch_dict_expand_sizes(hipStream_t s, const uint16_t* codes, int64_t n, const int32_t* dict_len, int32_t* out_sizes, uint32_t* out_tile_bytes) {
  const int64_t nt = (n + kTile - 1) / kTile;
  if (nt == 0) return;
  hipLaunchKernelGGL(k_dict_expand_sizes, dim3(grid_for(nt)), dim3(kBlock), 0, s, codes, n, dict_len, out_sizes, out_tile_bytes, nt);
}
void launch_dict_expand_bytes(hipStream_t s, const uint16_t* codes, int64_t n, const int32_t* dict_len, const uint32_t* dict_off, const uint8_t* dict_bytes,
                              const uint64_t* out_tile_off, uint8_t* out_bytes, int64_t out_bytes_cap) {
  const int64_t nt = (n + kTile - 1) / kTile;
  if (nt == 0) return;
  hipLaunchKernelGGL(k_dict_expand_bytes, dim3(grid_for(nt)), dim3(kBlock), 0, s, codes, n, dict_len, dict_off, dict_bytes, out_tile_off, out_bytes, nt, out_bytes_cap);
}

// the projection of a String column that a conjunct `col == "const"` pins to one value: the size n times, the bytes n times (8 bytes per store:
// byte b of the output is pattern byte b mod plen)
__global__ __launch_bounds__(kBlock) void k_fill_const_strings(int32_t* __restrict__ out_sizes, uint8_t* __restrict__ out_bytes, int64_t n,
                                                               const uint8_t* __restrict__ pat, int32_t plen) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  for (int64_t i = tid; i < n; i += stride) out_sizes[i] = plen;
  if (plen <= 0) return;
  const int64_t total = n * (int64_t)plen;
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  for (int64_t b = tid * 8; b < total; b += stride * 8) {
    uint64_t v = 0;
    uint32_t at = (uint32_t)(b % plen);
#pragma unroll
    for (int k = 0; k < 8; k++) { v |= (uint64_t)pat[at] << (8 * k); at = at + 1 == (uint32_t)plen ? 0u : at + 1; }
    if (b + 8 <= total) *(u64u*)(out_bytes + b) = v;
    else for (int k = 0; b + k < total; k++) out_bytes[b + k] = (uint8_t)(v >> (8 * k));
  }
}
void launch_fill_const_strings(hipStream_t s, int32_t* out_sizes, uint8_t* out_bytes, int64_t n, const uint8_t* pat_dev, int32_t plen) {
  if (n <= 0) return;
  int64_t blocks = (n + kBlock - 1) / kBlock; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_fill_const_strings, dim3((unsigned)blocks), dim3(kBlock), 0, s, out_sizes, out_bytes, n, pat_dev, plen);
}

}  // namespace dfdb
