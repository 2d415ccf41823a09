// k_select_experiment.hip — K1S: single-pass  `x OP c`  ->  bitmap + tile counts + tile prefix + ROW INDICES  (gfx950).
//
// AN EXPERIMENT THAT LOST (round 2): not part of libdfdb_hip.so.  tools/bench_select.hip includes this file and times it against the shipped
// three-launch form; profiles/r2_single_pass_select.txt has the numbers (1.76-2.27 ms against 1.48-1.55 ms per 1e9 rows at 10 %) and DESIGN.md §4
// says why: the look-back waits on quads that finish up to 100 us out of order, and even with the answer handed in (DFDB_SELECT_NOLOOK) the index
// stores interleaved with the read stream cost more than K2 does on its own.
//
// The three-launch form of selection(x -> x OP c) -> indices (K1 k_scan_cmp, the count scan, K2 k_compact_indices) writes the
// bitmap, reads it back, and waits for two kernel boundaries.  Here one launch does all of it: what the reference's loop body does per
// block — evaluate the broadcast, then append the block's LogicalIndex to the result (src/tables/selection.jl:133-166,
// src/tables/materialization.jl:33-37) — with the running row count carried between blocks by a decoupled look-back instead of by
// program order.
//
//   * a wave takes the next 4096-row group (four K1 tiles) from a global ticket counter: groups are numbered in the order they START,
//     so every group a wave can wait for belongs to a wave that is already running (no dependence on dispatch order or residency);
//   * the group is scanned exactly as K1 does (16 coalesced 512-B loads in flight per tile, ballot = bitmap word), the bitmap word,
//     the tile counts leave as in K1, and the group's selected count is published as an AGGREGATE descriptor;
//   * look-back: the 64 lanes read the 64 preceding descriptors at once, add aggregates down to the nearest group that has published an
//     inclusive PREFIX, then publish this group's own PREFIX (flag and value share one 64-bit word: no fence between them);
//   * the indices leave at the group's global offset in table order: word by word, the selected lanes of a word store a contiguous run.
//
//   algorithmic bytes / row: 8 (column) + 1/8 (bitmap) + 12/1024 (tile count + prefix) + 8 sigma (indices) + 8/4096 (descriptor)
#include <algorithm>
#include <type_traits>
#include "device_utils.hpp"
#include "kernels.hpp"
#include "../include/dfdb_ir.h"

namespace dfdb {

namespace {
constexpr int kBlock = 256;           // 4 waves
constexpr int kWavesPerBlock = 4;
constexpr int64_t kTile = 1024;
constexpr int kWordsPerTile = 16;
constexpr int64_t kGroup = 4096;
constexpr uint64_t kAgg = 1ull << 62, kPre = 2ull << 62, kVal = (1ull << 62) - 1ull;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr int kStateHead = 16;        // the ticket has a 128-byte line to itself

template <int OP, typename T>
__device__ __forceinline__ bool cmp_op(T x, T c) {
  if constexpr (OP == CMP_EQ) return x == c;
  else if constexpr (OP == CMP_NE) return x != c;
  else if constexpr (OP == CMP_LT) return x < c;
  else if constexpr (OP == CMP_LE) return x <= c;
  else if constexpr (OP == CMP_GT) return x > c;
  else return x >= c;
}
__device__ __forceinline__ uint32_t rank_in(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ uint64_t read_lane64(uint64_t v, int l) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
}
__device__ __forceinline__ uint64_t first_lane64(uint64_t v) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}
}  // namespace

// EMIT 0: word by word straight to HBM (contiguous run per store).  EMIT 1: K2's form — 16-bit positions staged in LDS (8 KB per wave),
// then full 512-byte stores.
//
// Work distribution: a workgroup draws one ticket per 65 536 rows (agent-scope atomics on ONE address retire one per ~14 ns on this chip —
// measured: a ticket per 4096-row group took 3.4 ms per 1e9 rows, all of it the counter); its four waves take a QUAD of four consecutive
// groups (16 384 rows) each, and the look-back runs over quads.  A quad's aggregate depends only on its own loads, and every quad before it
// was ticketed earlier or belongs to a lower wave of the same workgroup: whoever a wave waits for is already running.
template <typename T, int OP, int EMIT>
__global__ __launch_bounds__(kBlock, EMIT == 1 ? 4 : 6) void k_scan_select(const T* __restrict__ col, T c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ tile_counts,
                                                        uint64_t* __restrict__ prefix, int64_t* __restrict__ out, int64_t out_cap, int64_t row_base,
                                                        int64_t nrows, int64_t ntiles, uint64_t* __restrict__ state, int wt_store) {
  __shared__ uint16_t pos_sh[EMIT == 1 ? kWavesPerBlock : 1][EMIT == 1 ? kGroup : 1];
  __shared__ unsigned long long ticket_sh;
  const int lane = lane_id();
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t ngroups = (ntiles + 3) / 4;
  uint64_t* desc = state + kStateHead;
#ifdef DFDB_SELECT_STATS
  unsigned long long st_spins = 0, st_steps = 0, st_t0 = __builtin_readcyclecounter(), st_look = 0, st_tick = 0, st_emit = 0;
  const int64_t nq_pad = (ngroups + 3) / 4 + 4;
  uint64_t* dbg_pub = desc + nq_pad; uint64_t* dbg_wait = dbg_pub + nq_pad; uint64_t* dbg_spin0 = dbg_wait + nq_pad; uint64_t* dbg_done = dbg_spin0 + nq_pad;
#endif
  // The quad scanned in the PREVIOUS trip (its words and group totals) waits here while the next quad's loads run: by the time the wave
  // comes back to it, the quads before it have long published (a wave that looks back right after its own loads spends a third of its
  // time spinning on the slowest of the ~5000 quads in flight: measured).
  uint64_t P0 = 0, P1 = 0, P2 = 0, P3 = 0; uint32_t PG0 = 0, PG1 = 0, PG2 = 0, PG3 = 0; int64_t pq = -1;
  for (;;) {
#ifdef DFDB_SELECT_STATS
    const unsigned long long st_a = __builtin_readcyclecounter();
#endif
    __syncthreads();
    if (threadIdx.x == 0) ticket_sh = __hip_atomic_fetch_add((unsigned long long*)state, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int64_t chunk = (int64_t)first_lane64(ticket_sh);                     // (in SGPRs: everything derived from it is wave-uniform control flow)
#ifdef DFDB_SELECT_STATS
    st_tick += __builtin_readcyclecounter() - st_a;
#endif
    const bool more = chunk * 16 < ngroups;                                    // (workgroup-uniform)
    const int64_t q = chunk * 4 + wib;                                         // this wave's quad
    const int64_t g0 = q * 4;
    const bool have = more && g0 < ngroups;
    // (the four groups' words and totals live in named registers, picked by the wave-uniform i: an unrolled loop over arrays lets
    // the compiler hoist all 256 loads of the quad and spill)
    uint64_t W0 = 0, W1 = 0, W2 = 0, W3 = 0; uint32_t GT0 = 0, GT1 = 0, GT2 = 0, GT3 = 0;
    if (have) {
#pragma unroll 1
      for (int i = 0; i < 4; i++) {
        const int64_t g = g0 + i;
        if (g >= ngroups) break;                                               // (wave-uniform)
        const int64_t t0 = g * 4;
        uint64_t myword = 0;
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
          const int64_t tile = t0 + k;
          if (tile >= ntiles) break;                                           // (wave-uniform)
          const int64_t base = tile * kTile;
          const T* p = col + base + lane;
          const int l0 = 16 * k;
          if (base + kTile <= nrows) {
            T v[kWordsPerTile];
#pragma unroll
            for (int j = 0; j < kWordsPerTile; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
            for (int j = 0; j < kWordsPerTile; j++) {
              const uint64_t m = __ballot(cmp_op<OP, T>(v[j], c));
              if (lane == l0 + j) myword = m;
            }
          } else {
#pragma unroll
            for (int j = 0; j < kWordsPerTile; j++) {
              const int64_t row = base + j * 64 + lane;
              bool r = false;
              if (row < nrows) r = cmp_op<OP, T>(p[j * 64], c);
              const uint64_t m = __ballot(r);
              if (lane == l0 + j) myword = m;
            }
          }
        }
        uint32_t cnt = (uint32_t)__popcll(myword);
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);      // every 16-lane group adds up its own tile
        if (wt_store) __hip_atomic_store(&bitmap[g * 64 + lane], myword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else bitmap[g * 64 + lane] = myword;
        if ((lane & 15) == 0 && t0 + (lane >> 4) < ntiles) tile_counts[t0 + (lane >> 4)] = cnt;
        const uint32_t gt = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 0) + (uint32_t)__builtin_amdgcn_readlane((int)cnt, 16) +
                            (uint32_t)__builtin_amdgcn_readlane((int)cnt, 32) + (uint32_t)__builtin_amdgcn_readlane((int)cnt, 48);
        if (i == 0) { W0 = myword; GT0 = gt; } else if (i == 1) { W1 = myword; GT1 = gt; }
        else if (i == 2) { W2 = myword; GT2 = gt; } else { W3 = myword; GT3 = gt; }
      }
      // the quad's aggregate (quad 0: already its inclusive prefix)
      const uint64_t total = (uint64_t)GT0 + GT1 + GT2 + GT3;
      if (lane == 0) __hip_atomic_store(&desc[q], (q == 0 ? kPre : kAgg) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef DFDB_SELECT_STATS
      if (lane == 0) dbg_pub[q] = __builtin_amdgcn_s_memrealtime();
#endif
    }

    if (pq >= 0) {
      // ---- decoupled look-back for the PREVIOUS quad over the quads that started before it
#ifdef DFDB_SELECT_STATS
      const unsigned long long st_b = __builtin_readcyclecounter();
#endif
      const uint64_t ptotal = (uint64_t)PG0 + PG1 + PG2 + PG3;
      uint64_t excl = 0;
#ifdef DFDB_SELECT_NOLOOK
      excl = first_lane64(prefix[pq * 16]);     // upper-bound experiment: the prefix array already holds the answer
      if (false) {
#else
      if (pq > 0) {
#endif
        int64_t at = pq - 1;
        uint32_t spins = 0;
        for (;;) {
          const int64_t idx = at - lane;
          const uint64_t d = idx >= 0 ? __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kPre;   // before quad 0: a prefix of 0
          const uint32_t flag = (uint32_t)(d >> 62);
          const uint64_t pm = __ballot(flag == 2u), im = __ballot(flag == 0u);
          const uint64_t first = pm & (0ull - pm);                             // nearest quad with an inclusive prefix
          const uint64_t need = pm ? ((first << 1) - 1ull) : ~0ull;            // that lane and the ones nearer
#ifdef DFDB_SELECT_STATS
          st_steps++; if (im & need) st_spins++;
          if ((im & need) && spins == 0 && lane == 0) { dbg_spin0[pq] = __builtin_amdgcn_s_memrealtime(); dbg_wait[pq] = (uint64_t)(at - (63 - __builtin_clzll(im & need))); }
#endif
          if (im & need) {                                                     // one of them has not published yet
            if (++spins > kSpinLimit) { if (lane == 0) state[1] = 1; break; }  // (never seen; the launch reports it instead of hanging)
            __builtin_amdgcn_s_sleep(1); continue;
          }
          // aggregates are <= 2^14 each: 32-bit sum; the prefix itself is added from its lane
          const uint32_t a = ((need >> lane) & 1ull) && flag == 1u ? (uint32_t)d : 0u;
          excl += wave_sum(a);
          if (pm) { excl += read_lane64(d, __builtin_ctzll(pm)) & kVal; break; }
          at -= 64;
        }
        if (lane == 0) __hip_atomic_store(&desc[pq], kPre | (excl + ptotal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#ifdef DFDB_SELECT_STATS
      const unsigned long long st_c = __builtin_readcyclecounter(); st_look += st_c - st_b;
      if (lane == 0) dbg_done[pq] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
      for (int i = 0; i < 4; i++) {
        const int64_t g = pq * 4 + i;
        if (g >= ngroups) break;
        const int64_t t0 = g * 4;
        const uint64_t myword = i == 0 ? P0 : i == 1 ? P1 : i == 2 ? P2 : P3;
        const uint32_t gti = i == 0 ? PG0 : i == 1 ? PG1 : i == 2 ? PG2 : PG3;
        // ---- per-tile exclusive prefix (what the count scan writes), the total after the last group
        const uint32_t pc = (uint32_t)__popcll(myword);
        uint32_t cn = pc;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) cn += __shfl_xor(cn, d, 64);
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cn, 0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)cn, 16),
                       c2 = (uint32_t)__builtin_amdgcn_readlane((int)cn, 32);
        {
          const int k = lane >> 4;
          const uint64_t tex = excl + (k > 0 ? c0 : 0u) + (k > 1 ? c1 : 0u) + (k > 2 ? c2 : 0u);
          if ((lane & 15) == 0 && t0 + k < ntiles) prefix[t0 + k] = tex;
          if (g == ngroups - 1 && lane == 0) prefix[ntiles] = excl + gti;
        }
        if (out != nullptr && gti != 0) {
          // ---- indices, table order
          const int64_t row1 = row_base + g * kGroup + 1;
          if constexpr (EMIT == 0) {
            int64_t o = (int64_t)excl;
#pragma unroll 4
            for (int j = 0; j < 64; j++) {
              const uint64_t m = read_lane64(myword, j);
              if (m == 0) continue;
              if ((m >> lane) & 1ull) { const int64_t z = o + rank_in(m); if (z < out_cap) out[z] = row1 + j * 64 + lane; }
              o += __popcll(m);
            }
          } else {
            uint16_t* pos = pos_sh[wib];
            const uint32_t incl = wave_incl_scan(pc);
            uint32_t o = incl - pc;
            const uint32_t lbase = (uint32_t)lane << 6;
            uint64_t w = myword;
            while (w) {
              const int b = __builtin_ctzll(w);
              w &= w - 1;
              pos[o++] = (uint16_t)(lbase + (uint32_t)b);
            }
            wave_lds_fence();
            for (uint32_t k = lane; k < gti; k += 64) {
              const int64_t z = (int64_t)excl + k;
              if (z < out_cap) out[z] = row1 + pos[k];
            }
            wave_lds_fence();
          }
        }
        excl += gti;
      }
#ifdef DFDB_SELECT_STATS
      st_emit += __builtin_readcyclecounter() - st_c;
#endif
    }
    if (!more) break;
    P0 = W0; P1 = W1; P2 = W2; P3 = W3; PG0 = GT0; PG1 = GT1; PG2 = GT2; PG3 = GT3; pq = have ? q : -1;
  }
#ifdef DFDB_SELECT_STATS
  if (lane == 0) {
    atomicAdd((unsigned long long*)&state[2], st_spins); atomicAdd((unsigned long long*)&state[3], st_steps);
    atomicAdd((unsigned long long*)&state[4], __builtin_readcyclecounter() - st_t0); atomicAdd((unsigned long long*)&state[5], st_look);
    atomicAdd((unsigned long long*)&state[6], st_tick); atomicAdd((unsigned long long*)&state[7], st_emit); atomicAdd((unsigned long long*)&state[8], 1ull);
  }
#endif
}

static int g_select_emit = 0;
void set_select_emit(int v) { g_select_emit = v; }

size_t scan_select_state_bytes(int64_t nrows) { return
#ifdef DFDB_SELECT_STATS
      5 *
#endif
      (size_t)((((nrows + kTile - 1) / kTile + 3) / 4 + 3) / 4 + 4 + kStateHead) * 8; }

template <typename T, int OP>
static void launch_select_t(hipStream_t s, const void* col, uint64_t cbits, uint64_t* bitmap, uint32_t* tc, uint64_t* prefix, int64_t* out, int64_t out_cap,
                            int64_t row_base, int64_t nrows, uint64_t* state, int wt_store) {
  const int64_t ntiles = (nrows + kTile - 1) / kTile;
  const int64_t ngroups = (ntiles + 3) / 4;
  int64_t blocks = (ngroups + 15) / 16;
  if (blocks > 2048) blocks = 2048;
  const T c = from_bits<T>(cbits);
  if (g_select_emit == 1)
    hipLaunchKernelGGL((k_scan_select<T, OP, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, s, (const T*)col, c, bitmap, tc, prefix, out, out_cap, row_base, nrows, ntiles, state, wt_store);
  else
    hipLaunchKernelGGL((k_scan_select<T, OP, 0>), dim3((unsigned)blocks), dim3(kBlock), 0, s, (const T*)col, c, bitmap, tc, prefix, out, out_cap, row_base, nrows, ntiles, state, wt_store);
}
template <typename T>
static void launch_select_op(hipStream_t s, const void* col, int op, uint64_t cbits, uint64_t* bitmap, uint32_t* tc, uint64_t* prefix, int64_t* out, int64_t out_cap,
                             int64_t row_base, int64_t nrows, uint64_t* state, int wt_store) {
  switch (op) {
    case CMP_EQ: launch_select_t<T, CMP_EQ>(s, col, cbits, bitmap, tc, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
    case CMP_NE: launch_select_t<T, CMP_NE>(s, col, cbits, bitmap, tc, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
    case CMP_LT: launch_select_t<T, CMP_LT>(s, col, cbits, bitmap, tc, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
    case CMP_LE: launch_select_t<T, CMP_LE>(s, col, cbits, bitmap, tc, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
    case CMP_GT: launch_select_t<T, CMP_GT>(s, col, cbits, bitmap, tc, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
    default:     launch_select_t<T, CMP_GE>(s, col, cbits, bitmap, tc, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
  }
}

bool scan_select_supports(int32_t dtype) { return dtype == DFDB_I64 || dtype == DFDB_U64 || dtype == DFDB_F64; }

// `state` (scan_select_state_bytes) is zeroed here, on the stream, before the kernel
void launch_scan_select(hipStream_t s, const void* col, int32_t dtype, int op, uint64_t cbits, uint64_t* bitmap, uint32_t* tile_counts, uint64_t* prefix,
                        int64_t* out, int64_t out_cap, int64_t row_base, int64_t nrows, uint64_t* state, int wt_store) {
  if (nrows <= 0) { (void)hipMemsetAsync(prefix, 0, 8, s); return; }
  (void)hipMemsetAsync(state, 0, scan_select_state_bytes(nrows), s);
  switch (dtype) {
    case DFDB_I64: launch_select_op<int64_t>(s, col, op, cbits, bitmap, tile_counts, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
    case DFDB_U64: launch_select_op<uint64_t>(s, col, op, cbits, bitmap, tile_counts, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
    default:       launch_select_op<double>(s, col, op, cbits, bitmap, tile_counts, prefix, out, out_cap, row_base, nrows, state, wt_store); break;
  }
}

}  // namespace dfdb
