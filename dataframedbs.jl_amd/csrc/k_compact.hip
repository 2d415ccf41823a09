// k_compact.hip — K2/K3: selection bitmap -> row indices / gathered projection columns (gfx950).
//
// Replaces Base.LogicalIndex iteration + the gather loops of the reference
// (src/tables/selection.jl:166, src/tables/broadcast.jl:106-110, src/tables/projection.jl:128-133) and the
// per-block append! of materialize (src/tables/materialization.jl:33-37).
//
// One wave owns a 4096-row compaction tile = 64 bitmap words, one word per lane.  Each lane expands the
// set bits of its word into a per-wave LDS staging buffer of 16-bit in-tile positions at its exclusive
// prefix (wave prefix-sum of popcounts), so the expensive part is O(max popcount in the wave) instead of
// O(64) ballots; the wave then streams the staged positions out as fully coalesced stores at the
// tile's global offset (exclusive scan of the per-1024-row counts produced by K1).  Output order is
// table order (stable), as the reference guarantees.  K2 itself ships in the "wide" form since round 3
// (k_compact_indices_wide: two adjacent ctiles per wave, 16-byte nontemporal stores, one pair per wave);
// the gathers (K3) keep the one-ctile form.
//   algorithmic bytes / row: 1/8 (bitmap) + sigma * 8 (index)  |  gather: 1/8 + sigma * 2 * width
#include "device_utils.hpp"
#include "kernels.hpp"
#include "../../include/dfdb_ir.h"

namespace dfdb {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int64_t kCTile = 4096;

static inline int grid_for_ctiles(int64_t n) {
  int64_t blocks = (n + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// expand this lane's word into pos[excl ...]; returns the tile's selected count
__device__ __forceinline__ uint32_t stage_positions(uint64_t w, uint16_t* pos, int lane) {
  const uint32_t c = (uint32_t)__popcll(w);
  const uint32_t incl = wave_incl_scan(c);
  const uint32_t total = __shfl(incl, 63, 64);
  uint32_t o = incl - c;
  const uint32_t lbase = (uint32_t)lane << 6;
  while (w) {
    const int b = __builtin_ctzll(w);
    w &= w - 1;
    pos[o++] = (uint16_t)(lbase + (uint32_t)b);
  }
  wave_lds_fence();
  return total;
}

// STORE: 0 plain, 1 nontemporal (default), 2 write-through.  K2 in isolation is fastest with plain stores (0.188 / 0.207 / 0.25 ms per 1e9 rows at 10 %),
// but the last few hundred MB of plain-stored indices are still leaving L2 / MALL when the next scan starts: the K1 that follows runs 4-7 % slower and
// varies from process to process (1.35-1.39 ms against a steady 1.285 after nontemporal stores; profiles/r2_k2_store_policy.txt).  The step is what counts.
// (With a host synchronisation between steps the drain happens in the gap and plain stores are 0.01-0.02 ms ahead; K3's and the captured copies' output
// stores measured the same either way and stay plain.)
template <int STORE>
__global__ __launch_bounds__(kBlock) void k_compact_indices(const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ prefix,
                                                            int64_t* __restrict__ out, int64_t nctiles, int64_t row_base, int64_t out_cap) {
  __shared__ uint16_t pos_sh[kWavesPerBlock][kCTile];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  uint16_t* pos = pos_sh[wib];
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  // software-pipelined: the next ctile's bitmap word and output offset are in flight while this one is expanded
  // (the loop is otherwise a chain of dependent HBM round trips: measured 0.24 ms -> latency-, not bandwidth-bound)
  int64_t ct = wave;
  uint64_t w_next = ct < nctiles ? bitmap[ct * 64 + lane] : 0ull;
  uint64_t ob_next = ct < nctiles ? prefix[ct * 4] : 0ull;
  for (; ct < nctiles; ct += nwaves) {
    const uint64_t w = w_next;
    const int64_t obase = (int64_t)ob_next;
    const int64_t nx = ct + nwaves;
    if (nx < nctiles) { w_next = bitmap[nx * 64 + lane]; ob_next = prefix[nx * 4]; }
    const uint32_t total = stage_positions(w, pos, lane);
    const int64_t row1 = row_base + ct * kCTile + 1;   // 1-based table row of in-tile position 0
    for (uint32_t k = lane; k < total; k += 64) {
      const int64_t o = obase + k;
      if (o < out_cap) {
        if (STORE == 1) __builtin_nontemporal_store(row1 + pos[k], out + o);
        else if (STORE == 2) __hip_atomic_store(out + o, row1 + pos[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else out[o] = row1 + pos[k];
      }
    }
    wave_lds_fence();
  }
}

// K2 "wide" (round 3): a wave takes TWO adjacent ctiles per trip — their survivors are contiguous in the output, so the pair is one run of
// t0 + t1 indices — and every lane stores TWO consecutive indices with one 16-byte store (global_store_dwordx4: 1 KB per wave instruction
// instead of 512 B, half the store instructions; an odd first element goes out alone so that the pairs are 16-byte aligned).  One packed DPP
// scan ranks both words of a lane (counts <= 4096 fit 16 bits each), the two expansion loops run interleaved, and the same 8 KB of LDS per
// wave holds the pair as long as it has at most 4096 survivors (sigma <= 0.5 on average); denser pairs fall back to one ctile after the other.
typedef long long dfdb_ll2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ void store_idx1(int64_t* p, int64_t a) { if (NT) __builtin_nontemporal_store(a, p); else *p = a; }
template <bool NT> __device__ __forceinline__ void store_idx2(int64_t* p, int64_t a, int64_t b) {
  dfdb_ll2 v; v.x = a; v.y = b;
  if (NT) __builtin_nontemporal_store(v, (dfdb_ll2*)p); else *(dfdb_ll2*)p = v;
}
// the wave's `total` staged positions -> out[obase ...] = row1 + pos[k]
template <bool NT>
__device__ __forceinline__ void stream_indices(const uint16_t* pos, uint32_t total, int64_t row1, int64_t* __restrict__ out, int64_t obase, int64_t out_cap, int lane) {
  if (obase + (int64_t)total > out_cap) total = out_cap > obase ? (uint32_t)(out_cap - obase) : 0u;
  if (total == 0) return;
  int64_t* o = out + obase;
  const uint32_t head = (uint32_t)(((uintptr_t)o >> 3) & 1u);            // 1: the run starts on the upper half of a 16-byte slot
  if (head && lane == 0) store_idx1<NT>(o, row1 + pos[0]);
  for (uint32_t k = head + 2u * (uint32_t)lane; k < total; k += 128u) {
    if (k + 1 < total) store_idx2<NT>(o + k, row1 + pos[k], row1 + pos[k + 1]);
    else store_idx1<NT>(o + k, row1 + pos[k]);
  }
}
// STAGE = staged positions per wave (2 bytes each): 4096 -> 8 KB per wave, 5 workgroups per CU; 2048 -> 4 KB, 10 workgroups (the 8-waves-per-SIMD cap).
// A pair with more survivors than STAGE goes out in rounds of (ctile, 16-lane group): at most 16 x 64 = 1024 survivors each, in output order.
template <bool NT, int STAGE>
__global__ __launch_bounds__(kBlock) void k_compact_indices_wide(const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ prefix,
                                                                 int64_t* __restrict__ out, int64_t nctiles, int64_t row_base, int64_t out_cap) {
  __shared__ uint16_t pos_sh[kWavesPerBlock][STAGE];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  uint16_t* pos = pos_sh[wib];
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t npairs = (nctiles + 1) >> 1;
  int64_t pt = wave;
  // the next pair's words and output offset are in flight while this pair is expanded (the bitmap is padded to whole ctiles, not to pairs)
  uint64_t w0_next = 0, w1_next = 0, ob_next = 0;
  if (pt < npairs) { w0_next = bitmap[pt * 128 + lane]; if (2 * pt + 1 < nctiles) w1_next = bitmap[pt * 128 + 64 + lane]; ob_next = prefix[pt * 8]; }
  for (; pt < npairs; pt += nwaves) {
    uint64_t w0 = w0_next, w1 = w1_next;
    const int64_t obase = (int64_t)ob_next;
    const int64_t nx = pt + nwaves;
    if (nx < npairs) { w0_next = bitmap[nx * 128 + lane]; w1_next = 2 * nx + 1 < nctiles ? bitmap[nx * 128 + 64 + lane] : 0ull; ob_next = prefix[nx * 8]; }
    const uint32_t c0 = (uint32_t)__popcll(w0), c1 = (uint32_t)__popcll(w1);
    const uint32_t incl = wave_incl_scan(c0 | (c1 << 16));
    const uint32_t tot = __shfl(incl, 63, 64);
    const uint32_t t0 = tot & 0xffffu, t1 = tot >> 16;
    const int64_t row1 = row_base + pt * (2 * kCTile) + 1;                // 1-based table row of position 0 of the pair
    const uint32_t lbase = (uint32_t)lane << 6;
    uint32_t o0 = (incl & 0xffffu) - c0, o1 = (incl >> 16) - c1;
    if (t0 + t1 <= (uint32_t)STAGE) {
      o1 += t0;
      while (w0 | w1) {                                                  // the two expansions interleaved: independent chains
        if (w0) { const int b = __builtin_ctzll(w0); w0 &= w0 - 1; pos[o0++] = (uint16_t)(lbase + (uint32_t)b); }
        if (w1) { const int b = __builtin_ctzll(w1); w1 &= w1 - 1; pos[o1++] = (uint16_t)((uint32_t)kCTile + lbase + (uint32_t)b); }
      }
      wave_lds_fence();
      stream_indices<NT>(pos, t0 + t1, row1, out, obase, out_cap, lane);
      wave_lds_fence();
    } else {
#pragma unroll 1
      for (int r = 0; r < 8; r++) {                                       // (ctile r / 4, lanes 16 * (r % 4) ... + 15): <= 1024 survivors, in output order
        const bool second = r >= 4;
        const int g = r & 3;
        uint64_t w = (lane >> 4) == g ? (second ? w1 : w0) : 0ull;
        const uint32_t gstart = __shfl(second ? o1 : o0, g * 16, 64);      // survivors of this ctile before the group
        const uint32_t gend = g == 3 ? (second ? t1 : t0) : __shfl(second ? o1 : o0, (g + 1) * 16, 64);
        uint32_t o = (second ? o1 : o0) - gstart;
        while (w) { const int b = __builtin_ctzll(w); w &= w - 1; pos[o++] = (uint16_t)(lbase + (uint32_t)b); }
        wave_lds_fence();
        stream_indices<NT>(pos, gend - gstart, row1 + (second ? kCTile : 0), out, obase + (second ? t0 : 0u) + gstart, out_cap, lane);
        wave_lds_fence();
      }
    }
  }
}

void launch_compact_indices(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, int64_t* out, int64_t nrows, int64_t row_base,
                            int64_t out_cap, int store, int grid_cap) {
  const int64_t nct = (nrows + kCTile - 1) / kCTile;
  if (nct == 0) return;
  if (store >= 3 && store <= 6) {        // 3 / 4: wide, nontemporal / plain 16-byte stores, 8 KB of LDS per wave; 5 / 6: the same with 4 KB per wave
    int64_t blocks = ((nct + 1) / 2 + kWavesPerBlock - 1) / kWavesPerBlock;
    // one pair per wave up to 65 536 workgroups (2e9 rows): measured in the bench step against a persistent grid of 4096 workgroups, 0.182 -> 0.173
    // and 0.207 -> 0.186 ms per 1e9 rows on two boxes (profiles/r3_k2_forms.txt) — a pure 0.8-GB fill shows the same preference for many short
    // workgroups over a grid-stride loop (tools/bench_fill: 6.0-6.3 TB/s at 16 384 workgroups, 4.2-5.2 at 1024-4096)
    const int64_t cap = grid_cap > 0 ? grid_cap : 65536;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const dim3 gr((unsigned)blocks), bl(kBlock);
    if (store == 3) hipLaunchKernelGGL((k_compact_indices_wide<true, 4096>), gr, bl, 0, s, bitmap, prefix, out, nct, row_base, out_cap);
    else if (store == 4) hipLaunchKernelGGL((k_compact_indices_wide<false, 4096>), gr, bl, 0, s, bitmap, prefix, out, nct, row_base, out_cap);
    else if (store == 5) hipLaunchKernelGGL((k_compact_indices_wide<true, 2048>), gr, bl, 0, s, bitmap, prefix, out, nct, row_base, out_cap);
    else hipLaunchKernelGGL((k_compact_indices_wide<false, 2048>), gr, bl, 0, s, bitmap, prefix, out, nct, row_base, out_cap);
    return;
  }
  if (store == 1) hipLaunchKernelGGL(k_compact_indices<1>, dim3(grid_for_ctiles(nct)), dim3(kBlock), 0, s, bitmap, prefix, out, nct, row_base, out_cap);
  else if (store == 2) hipLaunchKernelGGL(k_compact_indices<2>, dim3(grid_for_ctiles(nct)), dim3(kBlock), 0, s, bitmap, prefix, out, nct, row_base, out_cap);
  else hipLaunchKernelGGL(k_compact_indices<0>, dim3(grid_for_ctiles(nct)), dim3(kBlock), 0, s, bitmap, prefix, out, nct, row_base, out_cap);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather(const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ prefix,
                                                   const T* __restrict__ src, T* __restrict__ dst, int64_t nctiles, int64_t out_cap) {
  __shared__ uint16_t pos_sh[kWavesPerBlock][kCTile];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  uint16_t* pos = pos_sh[wib];
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  int64_t ct = wave;
  uint64_t w_next = ct < nctiles ? bitmap[ct * 64 + lane] : 0ull;
  uint64_t ob_next = ct < nctiles ? prefix[ct * 4] : 0ull;
  for (; ct < nctiles; ct += nwaves) {
    const uint64_t w = w_next;
    const int64_t obase = (int64_t)ob_next;
    const int64_t nx = ct + nwaves;
    if (nx < nctiles) { w_next = bitmap[nx * 64 + lane]; ob_next = prefix[nx * 4]; }
    const uint32_t total = stage_positions(w, pos, lane);
    const T* tsrc = src + ct * kCTile;
    // 4 gathers in flight per lane before the first store: the loop is a chain of dependent HBM round trips otherwise
    for (uint32_t k0 = 0; k0 < total; k0 += 256) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const uint32_t k = k0 + (uint32_t)u * 64 + lane; v[u] = k < total ? __builtin_nontemporal_load(tsrc + pos[k]) : T(0); }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = k0 + (uint32_t)u * 64 + lane;
        const int64_t o = obase + k;
        if (k < total && o < out_cap) dst[o] = v[u];
      }
    }
    wave_lds_fence();
  }
}

void launch_gather(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const void* src, void* dst, int width, int64_t nrows,
                   int64_t out_cap) {
  const int64_t nct = (nrows + kCTile - 1) / kCTile;
  if (nct == 0) return;
  const dim3 g(grid_for_ctiles(nct)), b(kBlock);
  switch (width) {
    case 1: hipLaunchKernelGGL((k_gather<uint8_t>), g, b, 0, s, bitmap, prefix, (const uint8_t*)src, (uint8_t*)dst, nct, out_cap); break;
    case 2: hipLaunchKernelGGL((k_gather<uint16_t>), g, b, 0, s, bitmap, prefix, (const uint16_t*)src, (uint16_t*)dst, nct, out_cap); break;
    case 4: hipLaunchKernelGGL((k_gather<uint32_t>), g, b, 0, s, bitmap, prefix, (const uint32_t*)src, (uint32_t*)dst, nct, out_cap); break;
    default: hipLaunchKernelGGL((k_gather<uint64_t>), g, b, 0, s, bitmap, prefix, (const uint64_t*)src, (uint64_t*)dst, nct, out_cap); break;
  }
}

// K3 with a column transform on the way (ScanTerm::pre: rem(col, m), col * k + d in wrapping Int64 or two-rounding Float64, col / k): the projection
// `k = a * 2 + 1` over a filtered view costs the gather of `a` (1.4 ms per 1e9 rows at 10 %), not an interpreter pass over every tile (3.6 ms).
// The arithmetic is the scan terms' (k_scan.hip term_word_rem / term_word_affine): same bits as the interpreter and as Julia.
template <typename T>
__device__ __forceinline__ uint64_t transform_value(T x, int pre, uint64_t magic, int shift, uint64_t dd) {
  if (pre == 1) {
    const int64_t xi = (int64_t)x;
    const uint64_t ux = xi < 0 ? 0ull - (uint64_t)xi : (uint64_t)xi;
    const uint64_t q0 = __umul64hi(magic, ux);
    const uint64_t q = (((ux - q0) >> 1) + q0) >> shift;
    const uint64_t r = ux - q * dd;
    return (uint64_t)(xi < 0 ? -(int64_t)r : (int64_t)r);
  }
  if (pre == 2) return (uint64_t)(int64_t)x * magic + dd;
  const double k = __longlong_as_double((long long)magic), d = __longlong_as_double((long long)dd);
  if (pre == 4) return (uint64_t)__double_as_longlong(__ddiv_rn((double)x, k));
  double prod = (double)x * k;
  asm volatile("" : "+v"(prod));                  // (no fma: see term_word_affine in k_scan.hip)
  return (uint64_t)__double_as_longlong(prod + d);
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_transform(const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ prefix,
                                                             const T* __restrict__ src, uint64_t* __restrict__ dst, int64_t nctiles, int64_t out_cap,
                                                             int pre, uint64_t magic, int shift, uint64_t dd) {
  __shared__ uint16_t pos_sh[kWavesPerBlock][kCTile];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  uint16_t* pos = pos_sh[wib];
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  int64_t ct = wave;
  uint64_t w_next = ct < nctiles ? bitmap[ct * 64 + lane] : 0ull;
  uint64_t ob_next = ct < nctiles ? prefix[ct * 4] : 0ull;
  for (; ct < nctiles; ct += nwaves) {
    const uint64_t w = w_next;
    const int64_t obase = (int64_t)ob_next;
    const int64_t nx = ct + nwaves;
    if (nx < nctiles) { w_next = bitmap[nx * 64 + lane]; ob_next = prefix[nx * 4]; }
    const uint32_t total = stage_positions(w, pos, lane);
    const T* tsrc = src + ct * kCTile;
    for (uint32_t k0 = 0; k0 < total; k0 += 256) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const uint32_t k = k0 + (uint32_t)u * 64 + lane; v[u] = k < total ? __builtin_nontemporal_load(tsrc + pos[k]) : T(0); }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = k0 + (uint32_t)u * 64 + lane;
        const int64_t o = obase + k;
        if (k < total && o < out_cap) dst[o] = transform_value<T>(v[u], pre, magic, shift, dd);
      }
    }
    wave_lds_fence();
  }
}
void launch_gather_transform(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const void* src, int32_t src_dtype, const ScanTerm& tf, void* dst,
                             int64_t nrows, int64_t out_cap) {
  const int64_t nct = (nrows + kCTile - 1) / kCTile;
  if (nct == 0) return;
  const dim3 g(grid_for_ctiles(nct)), b(kBlock);
#define DFDB_GT(T) hipLaunchKernelGGL((k_gather_transform<T>), g, b, 0, s, bitmap, prefix, (const T*)src, (uint64_t*)dst, nct, out_cap, (int)tf.pre, tf.pre_magic, (int)tf.pre_shift, tf.pre_d)
  switch (src_dtype) {
    case DFDB_I8: DFDB_GT(int8_t); break; case DFDB_I16: DFDB_GT(int16_t); break; case DFDB_I32: DFDB_GT(int32_t); break; case DFDB_I64: DFDB_GT(int64_t); break;
    case DFDB_U8: DFDB_GT(uint8_t); break; case DFDB_U16: DFDB_GT(uint16_t); break; case DFDB_U32: DFDB_GT(uint32_t); break; case DFDB_U64: DFDB_GT(uint64_t); break;
    case DFDB_F32: DFDB_GT(float); break; default: DFDB_GT(double); break;
  }
#undef DFDB_GT
}

// projection of a predicate column whose selected values the scan already wrote per tile (k_scan_cmp / k_scan_terms CAP):
// one wave per 4096-row ctile = one capture group: a contiguous copy of its pf[4] - pf[0] values
// XF: the captured values are those of an 8-byte column of type T and the output is a transform of them (transform_value: `x * 2` over a filtered view whose
// predicate already read x — the computed projection rides on the capture instead of on a second gather of the column)
template <bool XF, typename T>
__global__ __launch_bounds__(kBlock) void k_compact_captured(const uint64_t* __restrict__ cap, const uint64_t* __restrict__ prefix,
                                                             uint64_t* __restrict__ out, int64_t nctiles, int64_t ntiles, int64_t out_cap,
                                                             int pre, uint64_t magic, int shift, uint64_t dd) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t ct = wave; ct < nctiles; ct += nwaves) {
    const int64_t t0 = ct * 4;
    const uint64_t pf[2] = {prefix[t0], prefix[t0 + 4 < ntiles ? t0 + 4 : ntiles]};
    const uint32_t total = (uint32_t)(pf[1] - pf[0]);
    for (uint32_t k0 = 0; k0 < total; k0 += 256) {
      uint64_t v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = k0 + (uint32_t)u * 64 + lane;
        v[u] = k < total ? __builtin_nontemporal_load(cap + t0 * 1024 + (int64_t)k) : 0ull;      // (a group's four runs lie back to back: k_scan.hip CAP)
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = k0 + (uint32_t)u * 64 + lane;
        const int64_t o = (int64_t)pf[0] + k;
        if (k < total && o < out_cap) {
          if (XF) { T x; __builtin_memcpy(&x, &v[u], 8); out[o] = transform_value<T>(x, pre, magic, shift, dd); }
          else out[o] = v[u];
        }
      }
    }
  }
}
void launch_compact_captured(hipStream_t s, const uint64_t* cap, const uint64_t* prefix, uint64_t* out, int64_t nrows, int64_t out_cap) {
  const int64_t nct = (nrows + kCTile - 1) / kCTile, nt = (nrows + 1023) / 1024;
  if (nct == 0) return;
  hipLaunchKernelGGL((k_compact_captured<false, uint64_t>), dim3(grid_for_ctiles(nct)), dim3(kBlock), 0, s, cap, prefix, out, nct, nt, out_cap, 0, 0ull, 0, 0ull);
}
void launch_compact_captured_transform(hipStream_t s, const uint64_t* cap, const uint64_t* prefix, int32_t src_dtype, const ScanTerm& tf, uint64_t* out, int64_t nrows, int64_t out_cap) {
  const int64_t nct = (nrows + kCTile - 1) / kCTile, nt = (nrows + 1023) / 1024;
  if (nct == 0) return;
  const dim3 g(grid_for_ctiles(nct)), b(kBlock);
  if (src_dtype == DFDB_F64) hipLaunchKernelGGL((k_compact_captured<true, double>), g, b, 0, s, cap, prefix, out, nct, nt, out_cap, (int)tf.pre, tf.pre_magic, (int)tf.pre_shift, tf.pre_d);
  else if (src_dtype == DFDB_U64) hipLaunchKernelGGL((k_compact_captured<true, uint64_t>), g, b, 0, s, cap, prefix, out, nct, nt, out_cap, (int)tf.pre, tf.pre_magic, (int)tf.pre_shift, tf.pre_d);
  else hipLaunchKernelGGL((k_compact_captured<true, int64_t>), g, b, 0, s, cap, prefix, out, nct, nt, out_cap, (int)tf.pre, tf.pre_magic, (int)tf.pre_shift, tf.pre_d);
}

__global__ __launch_bounds__(kBlock) void k_gather_bits(const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ prefix,
                                                        const uint64_t* __restrict__ srcbits, uint8_t* __restrict__ dst, int64_t nctiles,
                                                        int64_t out_cap) {
  __shared__ uint16_t pos_sh[kWavesPerBlock][kCTile];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  uint16_t* pos = pos_sh[wib];
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t ct = wave; ct < nctiles; ct += nwaves) {
    const uint64_t w = bitmap[ct * 64 + lane];
    const uint32_t total = stage_positions(w, pos, lane);
    const int64_t obase = (int64_t)prefix[ct * 4];
    const uint64_t* tb = srcbits + ct * 64;
    for (uint32_t k = lane; k < total; k += 64) {
      const int64_t o = obase + k;
      const uint32_t p = pos[k];
      if (o < out_cap) dst[o] = (uint8_t)((tb[p >> 6] >> (p & 63)) & 1ull);
    }
    wave_lds_fence();
  }
}
void launch_gather_bits(hipStream_t s, const uint64_t* bitmap, const uint64_t* prefix, const uint64_t* srcbits, uint8_t* dst, int64_t nrows,
                        int64_t out_cap) {
  const int64_t nct = (nrows + kCTile - 1) / kCTile;
  if (nct == 0) return;
  hipLaunchKernelGGL(k_gather_bits, dim3(grid_for_ctiles(nct)), dim3(kBlock), 0, s, bitmap, prefix, srcbits, dst, nct, out_cap);
}

// ------------------------------------------------------------------------------------------------
// reductions over the selected rows (sum / min / max); integer results exact, Float64 sums pairwise
// per lane -> wave -> block -> final block (deterministic for a fixed grid)
// ------------------------------------------------------------------------------------------------
constexpr int kRedBlocks = 2048;

template <typename T> struct Acc;   // accumulator type
template <> struct Acc<int8_t> { using type = int64_t; };   template <> struct Acc<int16_t> { using type = int64_t; };
template <> struct Acc<int32_t> { using type = int64_t; };  template <> struct Acc<int64_t> { using type = int64_t; };
template <> struct Acc<uint8_t> { using type = uint64_t; }; template <> struct Acc<uint16_t> { using type = uint64_t; };
template <> struct Acc<uint32_t> { using type = uint64_t; }; template <> struct Acc<uint64_t> { using type = uint64_t; };
template <> struct Acc<float> { using type = double; };     template <> struct Acc<double> { using type = double; };

template <typename A> __device__ __forceinline__ A red_identity(int op);
template <> __device__ __forceinline__ int64_t red_identity<int64_t>(int op) { return op == DFDB_AGG_MIN ? INT64_MAX : (op == DFDB_AGG_MAX ? INT64_MIN : 0); }
template <> __device__ __forceinline__ uint64_t red_identity<uint64_t>(int op) { return op == DFDB_AGG_MIN ? ~0ull : 0ull; }
template <> __device__ __forceinline__ double red_identity<double>(int op) { return op == DFDB_AGG_MIN ? __builtin_inf() : (op == DFDB_AGG_MAX ? -__builtin_inf() : 0.0); }

template <typename A> __device__ __forceinline__ A red_combine(A a, A b, int op) {
  if (op == DFDB_AGG_SUM) return a + b;
  if (op == DFDB_AGG_MIN) return b < a ? b : a;
  return b > a ? b : a;
}
__device__ __forceinline__ double red_combine_f(double a, double b, int op) {   // Julia min/max propagate NaN, and -0.0 orders below 0.0 (Base.min / Base.max)
  if (op == DFDB_AGG_SUM) return a + b;
  if (a != a) return a;
  if (b != b) return b;
  // (a == b: equal values share their bits except the two zeros — OR keeps a sign bit either of them has, AND drops one either lacks.  `b < a ? b : a` kept
  // whichever zero came first, so minimum() of a column holding 0.0 and -0.0 depended on the grid: found by the block-streamed aggregates, round 6)
  if (a == b) { const unsigned long long x = __double_as_longlong(a), y = __double_as_longlong(b); return __longlong_as_double(op == DFDB_AGG_MIN ? (x | y) : (x & y)); }
  if (op == DFDB_AGG_MIN) return b < a ? b : a;
  return b > a ? b : a;
}
template <typename A> __device__ __forceinline__ A comb(A a, A b, int op) { return red_combine<A>(a, b, op); }
template <> __device__ __forceinline__ double comb<double>(double a, double b, int op) { return red_combine_f(a, b, op); }

template <typename A> __device__ __forceinline__ A wave_reduce(A v, int op) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { A t = __shfl_xor(v, d, 64); v = comb<A>(v, t, op); }
  return v;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_reduce_partial(const uint64_t* __restrict__ bitmap, const T* __restrict__ col, int op,
                                                           int64_t nwords, typename Acc<T>::type* __restrict__ partials,
                                                           uint64_t* __restrict__ pcounts) {
  using A = typename Acc<T>::type;
  __shared__ A sh[kWavesPerBlock];
  __shared__ uint64_t shc[kWavesPerBlock];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  A acc = red_identity<A>(op);
  uint64_t cnt = 0;
  const int64_t ntiles = (nwords + 15) / 16;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {   // 1024 rows per wave step, 16 column loads in flight
    const int64_t wi = tile * 16 + lane;
    const uint64_t myw = (lane < 16 && wi < nwords) ? bitmap[wi] : 0ull;
    if (__ballot(myw != 0) == 0) continue;                     // wave-uniform: dead tiles never touch the column
    const T* p = col + tile * 1024 + lane;
    T v[16];
    uint64_t w[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      w[j] = __shfl(myw, j, 64);                               // word j, broadcast
      v[j] = T(0);
      if (w[j] != 0 && ((w[j] >> lane) & 1ull)) v[j] = __builtin_nontemporal_load(p + j * 64);   // selected rows only (never past nrows)
    }
#pragma unroll
    for (int j = 0; j < 16; j++)
      if ((w[j] >> lane) & 1ull) { acc = comb<A>(acc, (A)v[j], op); cnt++; }
  }
  acc = wave_reduce<A>(acc, op);
  cnt = wave_sum64(cnt);
  if (lane == 0) { sh[wib] = acc; shc[wib] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    A r = sh[0]; uint64_t c = shc[0];
    for (int i = 1; i < kWavesPerBlock; i++) { r = comb<A>(r, sh[i], op); c += shc[i]; }
    partials[blockIdx.x] = r; pcounts[blockIdx.x] = c;
  }
}
template <typename A>
__global__ __launch_bounds__(kBlock) void k_reduce_final(const A* __restrict__ partials, const uint64_t* __restrict__ pcounts, int n, int op,
                                                         A* __restrict__ result, uint64_t* __restrict__ rcount) {
  __shared__ A sh[kWavesPerBlock];
  __shared__ uint64_t shc[kWavesPerBlock];
  A acc = red_identity<A>(op); uint64_t cnt = 0;
  // fixed association: thread t folds partials t, t+256, ... then a fixed tree
  for (int i = threadIdx.x; i < n; i += kBlock) { acc = comb<A>(acc, partials[i], op); cnt += pcounts[i]; }
  acc = wave_reduce<A>(acc, op); cnt = wave_sum64(cnt);
  if (lane_id() == 0) { sh[threadIdx.x >> 6] = acc; shc[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    A r = sh[0]; uint64_t c = shc[0];
    for (int i = 1; i < kWavesPerBlock; i++) { r = comb<A>(r, sh[i], op); c += shc[i]; }
    *result = r; *rcount = c;
  }
}

size_t reduce_scratch_bytes() { return (size_t)kRedBlocks * 16 + 64; }

template <typename T>
static void launch_reduce_t(hipStream_t s, const uint64_t* bitmap, const void* col, int op, int64_t nrows, void* partials, void* result) {
  using A = typename Acc<T>::type;
  const int64_t nwords = (nrows + 63) / 64;
  int grid = (int)((nwords + kWavesPerBlock - 1) / kWavesPerBlock);
  if (grid > kRedBlocks) grid = kRedBlocks;
  if (grid < 1) grid = 1;
  A* pv = (A*)partials; uint64_t* pc = (uint64_t*)((char*)partials + (size_t)kRedBlocks * 8);
  hipLaunchKernelGGL((k_reduce_partial<T>), dim3(grid), dim3(kBlock), 0, s, bitmap, (const T*)col, op, nwords, pv, pc);
  hipLaunchKernelGGL((k_reduce_final<A>), dim3(1), dim3(kBlock), 0, s, (const A*)pv, (const uint64_t*)pc, grid, op, (A*)result,
                     (uint64_t*)((char*)result + 8));
}

void launch_reduce(hipStream_t s, const uint64_t* bitmap, const void* col, int32_t dtype, int op, int64_t nrows, void* partials, void* result) {
  switch (dtype) {
    case DFDB_I8:  launch_reduce_t<int8_t>(s, bitmap, col, op, nrows, partials, result); break;
    case DFDB_I16: launch_reduce_t<int16_t>(s, bitmap, col, op, nrows, partials, result); break;
    case DFDB_I32: launch_reduce_t<int32_t>(s, bitmap, col, op, nrows, partials, result); break;
    case DFDB_I64: launch_reduce_t<int64_t>(s, bitmap, col, op, nrows, partials, result); break;
    case DFDB_U8: case DFDB_BOOL: launch_reduce_t<uint8_t>(s, bitmap, col, op, nrows, partials, result); break;
    case DFDB_U16: launch_reduce_t<uint16_t>(s, bitmap, col, op, nrows, partials, result); break;
    case DFDB_U32: launch_reduce_t<uint32_t>(s, bitmap, col, op, nrows, partials, result); break;
    case DFDB_U64: launch_reduce_t<uint64_t>(s, bitmap, col, op, nrows, partials, result); break;
    case DFDB_F32: launch_reduce_t<float>(s, bitmap, col, op, nrows, partials, result); break;
    default:       launch_reduce_t<double>(s, bitmap, col, op, nrows, partials, result); break;
  }
}

}  // namespace dfdb
