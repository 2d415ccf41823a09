// k_decode.hip — K7: LZ4 block decode on gfx950, K8: Union{T,Missing} bodies, String bodies.
//
// Replaces LZ4_decompress_safe as called per (column, block) by BlockStream.read_block
// (src/io/BlockStreams.jl:101-119) and read_block_body! for nullable and String columns
// (src/io/blocks.jl:46-71).  The LZ4 *block* format is serial inside a block, so parallelism comes from
// the blocks: one wavefront decodes one block (15 259 blocks per 1e9-row column).
//   k_lz4_decode: superbatch of 8 windows of 64 input bytes: the candidate decode of all 512 positions up front, a branch-free chain
//   walk over the real sequence starts, one start bit + one 8-byte record per sequence, far sources prefetched from HBM, bytes
//   produced in output order (see the kernel).  Recent output lives in an LDS ring; only sources that left the ring cost a fence.
// A match whose source overlaps its destination (offset < length) is a periodic pattern: byte k of the
// match equals source byte k mod offset, all of which precede the write pointer, so it is also copied in
// parallel.
#include <type_traits>
#include "device_utils.hpp"
#include "kernels.hpp"

namespace dfdb {

constexpr int kBlock = 256;

__device__ __forceinline__ uint64_t ld_u64_unaligned(const uint8_t* p) {
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  return *(const u64u*)p;
}

// ---- K7: register-window parse for the one-sequence path + superbatch execution ------------------------------------------
// v1 measured 26 GB/s on 8-byte integer columns (one sequence per ~8 output bytes): ~2800 cycles per sequence, all of
// it dependent L2/HBM round trips (token, literals, offset, a fence, the match source).  v3 keeps the whole
// per-sequence dependency chain on chip:
//   * the compressed stream is staged through a 4 KB per-wave LDS buffer in 2 KB chunks; the NEXT chunk is always in
//     flight in registers (global loads issued one chunk ahead), so staging never waits on memory in steady state;
//   * the parser reads tokens / lengths / offsets from a 512-byte REGISTER window (8 bytes per lane, refreshed from the
//     staging buffer every ~400 consumed bytes) with v_readlane: scalar work, no memory latency;
//   * the last 8 KB of output live in a per-wave LDS ring.  For a sequence with literals + match <= 64 bytes every
//     lane produces ONE output byte with ONE ds_read_u8 whose address points either into the staging buffer (a literal,
//     or a match byte that falls inside this sequence's own literals) or into the ring (earlier output), then ONE
//     ds_write_b8.  The LDS queue is in order per wave, so consecutive sequences need no fence;
//   * the ring is flushed to HBM in coalesced dword stores every 2 KB; nothing in the chain waits for a store.
// Longer runs take the same steps 64 bytes at a time; a match that reaches further back than the ring is copied from
// HBM behind a fence (rare on columnar data; costs what every v1 sequence cost).
#ifdef DFDB_LZ4_PROF   // tools/bench_lz4.hip only: cycles per phase of block 0 (s_memtime also drains the LDS queue: phase boundaries only)
__device__ unsigned long long g_lz4_prof[32];
#define LZ4_PROF(k) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const uint64_t pf_t = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); pf_acc[k] += pf_t - pf_t0; pf_t0 = pf_t; }
#define LZ4_COUNT(k, n) { pf_acc[k] += (n); }
#else
#define LZ4_PROF(k)
#define LZ4_COUNT(k, n)
#endif

#ifndef DFDB_LZ4_PIPE_U
#define DFDB_LZ4_PIPE_U 4
#endif
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// SCAN = 1: decode fused with the first predicate of the scan (the reference's loop body decodes a block and evaluates
// the selection over it in one iteration, src/io/blocksiterator.jl:98-121): every 512 decoded bytes of an 8-byte column pass the
// comparison `value OP c` on their way from the LDS ring to HBM and leave their 64-bit mask word — the bitmap and the per-1024-row
// counts K1 would have produced from a second pass over the decoded column.
// the same without a branch: `sel` says which of {less, equal, greater, unordered} satisfy the operator (bits 0..3), the three orderings are all computed and the
// column's type picks one by select — uniform branches are scalar instructions, and the CU's four SIMDs share one scalar unit (this runs once per 512 decoded bytes)
__device__ __forceinline__ uint32_t lz_op_sel(int op) {
  return op == CMP_EQ ? 2u : op == CMP_NE ? (1u | 4u | 8u) : op == CMP_LT ? 1u : op == CMP_LE ? (1u | 2u) : op == CMP_GT ? 4u : (4u | 2u);
}
__device__ __forceinline__ bool lz_cmp8_sel(uint64_t v, uint64_t c, int dtype, uint32_t sel) {
  const double a = __builtin_bit_cast(double, v), b = __builtin_bit_cast(double, c);
  const bool isf = dtype == DFDB_F64, isu = dtype == DFDB_U64;
  const bool lt = isf ? a < b : (isu ? v < c : (int64_t)v < (int64_t)c);
  const bool gt = isf ? a > b : (isu ? v > c : (int64_t)v > (int64_t)c);
  const bool eq = isf ? a == b : v == c;
  const bool un = !(lt || gt || eq);                                  // a NaN on either side of a Float64 comparison
  return ((sel & 1u) && lt) || ((sel & 2u) && eq) || ((sel & 4u) && gt) || ((sel & 8u) && un);
}
__device__ __forceinline__ bool lz_cmp8(uint64_t v, uint64_t c, int dtype, int op) {
  int r;                                                         // -1 / 0 / 1, 2 = unordered
  if (dtype == DFDB_F64) { const double a = __builtin_bit_cast(double, v), b = __builtin_bit_cast(double, c); r = (a != a || b != b) ? 2 : (a < b ? -1 : (a > b ? 1 : 0)); }
  else if (dtype == DFDB_U64) r = v < c ? -1 : (v > c ? 1 : 0);
  else r = (int64_t)v < (int64_t)c ? -1 : ((int64_t)v > (int64_t)c ? 1 : 0);
  switch (op) {
    case CMP_EQ: return r == 0;
    case CMP_NE: return r != 0;
    case CMP_LT: return r == -1;
    case CMP_LE: return r == -1 || r == 0;
    case CMP_GT: return r == 1;
    default:     return r == 1 || r == 0;
  }
}

// PIPE = 1: TWO waves per block.  Wave 0 parses (candidates, walk, dense records: everything up to the records and the start bitmap of a
// superbatch), wave 1 produces (far prefetch, byte production, flush) one superbatch behind, out of double-buffered records in LDS.  A block's
// latency becomes max(parse, produce) instead of their sum — what matters when there are fewer blocks than wave slots (a streamed chunk, a small
// table: at 1 526 blocks the chip is a quarter full and a block takes as long as it takes).  Sequences only the one-sequence path handles are
// executed by wave 0 once wave 1 has drained; ownership of the ring and of the output position passes through LDS control words.
// OCC: waves per SIMD the register allocation must leave room for; FARMAX: far-source slots per superbatch (24 bytes of LDS each)
// INDEX: a column that keeps its LZ4 blocks in HBM is decoded again and again (dfdb_table_decode_resident, decode_on_scan), and where its sequences
// start never changes.  INDEX = 1 records that while decoding — one bit per compressed byte, set where a sequence starts — and INDEX = 2 decodes WITH
// it: the start bits of the superbatch's windows come out of a 64-dword register window of the index (three v_readlane + a funnel shift per window)
// instead of the candidate decode of every position, the next^2 / next^4 / next^8 tables, the chain walk and the fill-in (phases 1 and 2: 40 % of the
// vector and a third of the LDS instructions of a superbatch); which sequences the batch takes is decided where their fields are decoded anyway (3b).
// HIST = 1 (round 5, SURVEY.md §8f-2: "without writing decoded blocks to HBM"): the decoded bytes never become a column.  What leaves the LDS ring goes to a
// per-WAVE history ring of 64 KB in a scratch buffer (`dst` + workgroup * kHistStride) instead of to the block's place in a decoded array: an LZ4 offset is at
// most 65 535 and block-local (BlockStreams.jl:110: one LZ4_decompress_safe per block), so the last 64 KB of a block's output are all a match can ever reach.
// Far sources (behind the LDS ring) are fetched from that history — a few hundred MB that thousands of waves keep rewriting, i.e. cache-resident lines rather
// than 8 GB of column the decoder used to write and then read 24 bytes at a time —, the first 32 bytes of a lap are mirrored behind the ring so that a 24-byte
// fetch near the lap's end stays contiguous, and blocks are handed to waves by a ticket counter (the grid is the resident waves, not the blocks: every
// workgroup owns ONE ring for its whole life).  With SCAN the predicate still sees every byte on its way out (bitmap + tile counts are the only output:
// the reference's loop body decodes into two reusable buffers and keeps nothing either, BlockStreams.jl:9-15,101-119, blocksiterator.jl:98-121); without
// SCAN it is a validating decode (status only), which is how a compressed-only column is checked and its sequence-start index recorded at load time.
constexpr uint32_t kHistBytes = 65536u, kHistStride = 65536u + 64u;
template <int WAVES, int kRing, int kStage, int kBatchBytes, int W, int SCAN, int PIPE, int OCC = (PIPE ? 4 : 5), int FARMAX = 64, int INDEX = 0, int HIST = 0>
__global__ __launch_bounds__(PIPE ? 128 : WAVES * 64, OCC) void k_lz4_decode(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                               const Lz4Block* __restrict__ blocks, int32_t nblocks, int32_t* __restrict__ status,
                                                               LzScan sc, uint32_t* __restrict__ index) {
  static_assert(INDEX != 1 || !PIPE, "the two-wave pipeline does not record the index (it reads it: PIPE with INDEX = 2)");
  static_assert(!HIST || (!PIPE && WAVES == 1), "the history-ring form is one wave per workgroup");
  // where output position p of the current block lives behind `out`: its place in the decoded array, or its slot of the wave's history ring
  auto HO = [](uint32_t p) -> uint32_t { return HIST ? (p & (kHistBytes - 1u)) : p; };
  // PIPE with INDEX = 2: with the index the parser wave is the short one, so it also fetches the superbatch's far sources (into a per-slot area): the
  // producer is left with byte production and the flush, and the two waves are balanced again
  constexpr bool kParserFar = PIPE && INDEX == 2;
  constexpr int kFarAreas = kParserFar ? 2 : 1;
  // one array per wave, staging buffer first and the output ring behind it: a byte of either is ONE ds_read_u8 off the same base
  constexpr int kFarMax = FARMAX;                // v5: matches per superbatch whose source has left the ring (fetched from HBM up front)
  constexpr uint32_t kWords = kBatchBytes / 32;  // start-bit words per superbatch
  static_assert(!PIPE || (WAVES == 2 && !SCAN), "the two-wave pipeline is its own configuration");
  constexpr int NW = PIPE ? 1 : WAVES;           // sets of LDS arrays per workgroup
  constexpr int NS = PIPE ? 2 : 1;               // record slots (PIPE: one being parsed into, one being produced from)
  __shared__ __attribute__((aligned(16))) uint8_t lds_sh[NW][kStage + kRing + kFarAreas * kFarMax * 24 + 64];   // (+ 64 bytes nobody reads: where lanes past a superbatch's last byte put theirs)
  constexpr uint32_t kDump = kStage + kRing + kFarAreas * kFarMax * 24;
  static_assert(kFarAreas * kFarMax * 24 <= kStage, "far bytes are addressed as ((j + B) & (kStage - 1)) | (kStage + kRing)");
  __shared__ uint32_t fard_sh[NW][NS][kFarMax + 1];   // (+ a slot nobody reads)
  __shared__ uint32_t ctl_sh[PIPE ? 16 : 1];     // PIPE control words (below)
  constexpr int kSeqMax = 21 * W + 3;            // a 64-byte window starts at most 21 sequences (>= 3 input bytes each)
  __shared__ uint32_t bits_sh[NW][kBatchBytes / 32 + 2];   // + two words that stay zero
  __shared__ uint2 bitsx_sh[NW][NS][kBatchBytes / 32 + 8];     // {start-bit word, starts before it - 1}; the tail stays {0, -1}
  __shared__ uint2 info_sh[NW][NS][kSeqMax + 1];               // entry 0 is a dummy: production's ordinal -1 (a row past the last byte) reads it
  const uint32_t lane = (uint32_t)lane_id();
  // (one wave per workgroup in the shipped configuration: every LDS array then sits at a compile-time address that folds into the ds instructions' offset field)
  const int wib = (WAVES == 1 || PIPE) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = PIPE ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;     // PIPE: 0 parses, 1 produces
  uint8_t* lds = lds_sh[wib];
  uint8_t* stage = lds;
  uint8_t* ring = lds + kStage;
  uint32_t* bits = bits_sh[wib];
  uint2* bitsx = bitsx_sh[wib][0];
  if (lane < 8 && role == 0) { for (int sl = 0; sl < NS; sl++) bitsx_sh[wib][sl][kBatchBytes / 32 + lane] = make_uint2(0u, 0xffffffffu); }
  const uint32_t lane_below = (2u << (lane & 31u)) - 1u;         // bits 0 .. lane mod 32
  uint2* info = info_sh[wib][0] + 1;
  uint32_t* fard = fard_sh[wib][0];
  auto use_slot = [&](int sl) { bitsx = bitsx_sh[wib][sl]; info = info_sh[wib][sl] + 1; fard = fard_sh[wib][sl]; };
  if (lane < 2 && role == 0) bits[kBatchBytes / 32 + lane] = 0;
  // PIPE control words.  PUB: superbatches published by the parser; CONS: consumed by the producer; HT / HNF [slot]: output bytes and far sources of
  // the superbatch in that slot; H_OP / H_FL: the output position and the flushed position, written by whoever owned the ring last; HAND: bumped by
  // the parser after it used the ring itself; END: nothing more will be published; ERR: either wave gave up (a bounded spin ran out)
  enum { C_PUB = 0, C_CONS = 1, C_HT = 2, C_HNF = 4, C_OP = 6, C_FL = 7, C_HAND = 8, C_END = 9, C_ERR = 10 };
  volatile uint32_t* ctl = ctl_sh;
  constexpr uint32_t kSpinLimit = 1u << 24;
  const int64_t wgid = PIPE ? (int64_t)blockIdx.x : 0;
  // the ring must keep every byte that is not in HBM yet: flush this often (a v5 superbatch adds up to kBatchBytes on top)
  constexpr uint32_t kFlush = kRing >= 4096 ? 1024u : 512u;
  static_assert(kStage == kRing, "production addresses staging buffer, ring and far bytes as ((j + B) & (kStage - 1)) | O");
  constexpr uint32_t kFA = SCAN ? 511u : 255u;                    // flushes end on multiples of kFA + 1 bytes (SCAN: whole 64-row mask words)
  static_assert((kFlush + kFA + 1 + kBatchBytes <= kRing && (kWords == 64 || kWords == 32 || kWords == 16) && 64 * W + 344 <= kStage / 2 && kFarMax <= 64),
                "v5: superbatch output must fit the ring behind the unflushed bytes; one window alone never exceeds the budget");
  const int64_t wave = PIPE ? wgid : (int64_t)blockIdx.x * WAVES + wib;
  const int64_t nwaves = PIPE ? (int64_t)gridDim.x : (int64_t)gridDim.x * WAVES;
  constexpr int kChunk = kStage / 2;            // two chunks staged, a third in flight
  constexpr int kNF = kChunk / 512;             // 8-byte words per lane of the chunk in flight
  static_assert(kChunk >= 1024 && kChunk % 512 == 0, "the 512-byte register window plus one sequence's look-ahead must fit behind ip");
  // HIST: blocks by ticket (sc.ticket, a device word the launcher zeroes): whichever wave is free takes the next block, its ring goes with it
  for (int64_t b = HIST ? -1 : wave; ; b += nwaves) {
    if (HIST) {
      // the next block that needs decoding.  A later conjunct / stage (and_existing = 1: the mask already holds survivors) does not decode a block none of
      // whose tiles kept a row — the reference's late materialization at block granularity, blocksiterator.jl:111-113; its mask words and counts stay
      // as they are: zero (and_existing = 2: AND without the skip, an A/B knob).  (An inner loop of its own, not a `continue` of the block loop: the compiler
      // made an endless loop of that form — the ticket was taken once and block 0 decoded for ever.)
      for (;;) {
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(sc.ticket, 1u);
        b = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)tk, 0);
        if (b >= nblocks || !(SCAN && sc.and_existing == 1)) break;
        const int64_t tile0 = blocks[b].dst_off >> 13, nt = ((int64_t)blocks[b].dst_len / 8 + 1023) / 1024;      // (8-byte rows: 8192 bytes per 1024-row tile)
        uint32_t any = 0;
        for (int64_t k = lane; k < nt; k += 64) any |= sc.counts[tile0 + k];
        if (__ballot(any != 0u) != 0) break;
        if (lane == 0) status[b] = 0;
      }
    }
    if (b >= nblocks) break;
    const Lz4Block blk = blocks[b];
    const uint8_t* in = src + blk.src_off;
    uint8_t* out = HIST ? dst + (size_t)blockIdx.x * kHistStride : dst + blk.dst_off;
    const uint32_t in_len = (uint32_t)blk.src_len, out_len = (uint32_t)blk.dst_len;
    uint32_t ip = 0, op = 0, flushed = 0;
    uint32_t cb = 0;                   // staged: input bytes [cb, cb + 4096); invariant cb <= ip < cb + 2048
    uint32_t wbase = 0, wlo = 0, whi = 0;
    int err = 0;
    uint32_t extstops = 0;             // v5: tokens with a length of 15 met by the one-sequence path
    uint32_t iw = 0; uint64_t ibase = ~0ull;   // INDEX 2: index dwords [ibase / 32 + lane], ibase = the window's first bit (a multiple of 32; ~0: nothing loaded)
#ifdef DFDB_LZ4_PROF
    uint64_t pf_acc[32] = {}; uint64_t pf_t0 = __builtin_readcyclecounter(); const uint64_t pf_start = pf_t0;
#endif
    uint64_t f[kNF];                   // the chunk in flight: input bytes [cb + kStage, cb + kStage + kChunk)

    auto chunk_load = [&](uint32_t pos) {          // global -> registers (bytes past in_len are never consumed)
      const uint32_t g = pos + lane * (8 * kNF);
#pragma unroll
      for (int i = 0; i < kNF; i++) f[i] = g + 8 * i < in_len ? ld_u64_unaligned(in + g + 8 * i) : 0ull;
    };
    auto chunk_store = [&](uint32_t pos) {         // registers -> staging slot of input position pos (a multiple of 2048)
      uint64_t* d = (uint64_t*)(stage + (pos & (kStage - 1)) + lane * (8 * kNF));
#pragma unroll
      for (int i = 0; i < kNF; i++) d[i] = f[i];
    };
    auto window_load = [&](uint32_t pos8) {        // 512 input bytes from pos8 (multiple of 8), 8 per lane
      wbase = pos8;
      const uint2 w = *(const uint2*)(stage + ((pos8 + lane * 8) & (kStage - 1)));
      wlo = w.x; whi = w.y;
    };
    // make input bytes [p, p + 72) parseable: advance the staging buffer and refresh the register window as needed
    auto advance = [&](uint32_t p) {
      // (PIPE: the producer still reads the PREVIOUS superbatch's literals out of the staging buffer, so 1 KB behind p stays staged)
      while (p >= cb + kChunk + (PIPE ? 1024u : 0u)) {   // p left the first staged chunk: recycle its slot for the chunk in flight
        chunk_store(cb + kStage);                  // (waits for those loads: issued a whole chunk ago)
        cb += kChunk;
        chunk_load(cb + kStage);
        if (wbase < cb) { wbase = 0xffffffffu; }   // window no longer backed: force a reload when it is next used
      }
    };
    auto ensure = [&](uint32_t p) {
      advance(p);
      if (p < wbase || p - wbase > 432u) window_load(p & ~7u);
    };
    // the same without recycling staging slots: used while the literals of the current sequence still sit in the staging
    // buffer (every position a <= 64-literal sequence can touch is < sequence start + 400 < cb + 4096)
    auto window_only = [&](uint32_t p) {
      if (p < wbase || p - wbase > 432u) window_load(p & ~7u);
    };
    // 8 input bytes at p (wave-uniform), little-endian; requires wbase <= p, p - wbase <= 496
    auto fetch64 = [&](uint32_t p) -> uint64_t {
      const uint32_t d = p - wbase, q = d >> 3, sh = (d & 7u) * 8u;
      const uint64_t lo = (uint64_t)rl(whi, q) << 32 | rl(wlo, q);
      const uint64_t hi = (uint64_t)rl(whi, q + 1) << 32 | rl(wlo, q + 1);
      return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
    };
    auto byte_at = [&](uint32_t p) -> uint32_t {
      const uint32_t d = p - wbase, q = d >> 3, sh = (d & 7u) * 8u;
      const uint64_t lo = (uint64_t)rl(whi, q) << 32 | rl(wlo, q);
      return (uint32_t)(lo >> sh) & 255u;
    };
    // SCAN: mask words of this block so far, the current tile's words (word k of the tile in lane k) and its selected count
    uint32_t sc_words = 0, sc_tile_count = 0; uint64_t sc_myword = 0;
    // The comparison as MASK ARITHMETIC (round 5).  Written per value with the column's type and the operator as run-time selects, the compiler made uniform
    // branches of every one of them: ~90 scalar instructions per 512 decoded bytes on the CU's one scalar unit — a third of all the kernel issues.  Instead every
    // 8-byte value gets an order-preserving unsigned image (signed: sign bit flipped; Float64: -0.0 folded into +0.0, negatives complemented; the two forms blended
    // with a mask VALUE, not a branch), the constant the same image once per block, and `less`, `greater` and `NaN` leave the vector unit as wave masks (the ballot
    // is free); which of {less, equal, greater, unordered} the operator accepts is four more mask values.  An interval term ANDs a second such mask in.
    struct ScKey { uint64_t ck, S_lt, S_eq, S_gt, S_un, CN; };
    const uint64_t sc_FM = SCAN && sc.dtype == DFDB_F64 ? ~0ull : 0ull, sc_XM = SCAN && sc.dtype == DFDB_U64 ? 0ull : 0x8000000000000000ull;
    auto sc_key_of = [&](uint64_t c, int op) -> ScKey {
      const uint32_t sel = lz_op_sel(op);
      ScKey k;
      const bool cnan = sc_FM && (c & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
      const uint64_t cz = c == 0x8000000000000000ull ? 0ull : c;
      k.ck = sc_FM ? (cz ^ ((uint64_t)((int64_t)cz >> 63) | 0x8000000000000000ull)) : (c ^ sc_XM);
      k.S_lt = (sel & 1u) ? ~0ull : 0ull; k.S_eq = (sel & 2u) ? ~0ull : 0ull; k.S_gt = (sel & 4u) ? ~0ull : 0ull; k.S_un = (sel & 8u) ? ~0ull : 0ull;
      k.CN = cnan ? ~0ull : 0ull;                                   // a NaN constant: every row compares unordered
      return k;
    };
    const ScKey sc_k1 = SCAN ? sc_key_of(sc.cbits, sc.op) : ScKey{}, sc_k2 = SCAN && sc.op2 >= 0 ? sc_key_of(sc.cbits2, sc.op2) : ScKey{};
    auto sc_mask = [&](uint64_t kimg, uint64_t mnan, const ScKey& k) -> uint64_t {      // the rows of this 64-row word the term accepts
      const uint64_t mlt = __ballot(kimg < k.ck), mgt = __ballot(kimg > k.ck);
      const uint64_t un = mnan | k.CN, ord = ~un;
      return (mlt & ord & k.S_lt) | (mgt & ord & k.S_gt) | (~(mlt | mgt) & ord & k.S_eq) | (un & k.S_un);
    };
    const int64_t sc_word0 = blk.dst_off / 512;                   // the block's first mask word (host: the block starts on a 1024-row tile)
    // ring -> HBM, bytes [flushed, upto)
    auto flush_to = [&](uint32_t upto) {
      if (SCAN) {
        // `flushed` is a multiple of 512 here; a group shorter than 512 bytes can only be the block's last
#pragma unroll 1
        for (uint32_t g = flushed; g < upto; g += 512u) {
          const bool have = g + lane * 8u + 8u <= upto;
          const uint64_t v = *(const uint64_t*)(ring + ((g + lane * 8u) & (kRing - 1)));      // (always inside the ring: only the ballot needs `have`)
          if (g + 512u <= upto) {
            ((uint64_t*)(out + HO(g)))[lane] = v;                                          // a whole group leaves as one 512-byte store (8-byte columns: `out` is 8-aligned)
            if (HIST && HO(g) == 0u && lane < 4u) *(uint64_t*)(out + kHistBytes + lane * 8u) = v;   // the lap's first 32 bytes again, behind the ring
          }
          const uint64_t vz = v == 0x8000000000000000ull ? 0ull : v;
          const uint64_t kimg = ((vz ^ ((uint64_t)((int64_t)vz >> 63) | 0x8000000000000000ull)) & sc_FM) | ((v ^ sc_XM) & ~sc_FM);
          const uint64_t mnan = __ballot((v & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) & sc_FM;
          uint64_t m = sc_mask(kimg, mnan, sc_k1);
          if (sc.op2 >= 0) m &= sc_mask(kimg, mnan, sc_k2);
          if (g + 512u > upto) m &= __ballot(have);                  // (a block's last, shorter group)
          if (lane == (sc_words & 15u)) sc_myword = m;
          sc_tile_count += (uint32_t)__builtin_popcountll(m);
          sc_words++;
          if ((sc_words & 15u) == 0u) {                            // a 1024-row tile is complete: one 128-byte line of bitmap + its count
            const int64_t w0 = sc_word0 + sc_words - 16u;
            if (sc.and_existing) {                                 // (wave-uniform) the words AND what the mask held; the count is theirs
              if (lane < 16u) sc_myword &= (sc.bitmap + w0)[lane];
              uint32_t c = lane < 16u ? (uint32_t)__builtin_popcountll(sc_myword) : 0u;
              for (int d = 8; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
              sc_tile_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            }
            if (lane < 16u) (sc.bitmap + w0)[lane] = sc_myword;
            if (lane == 0u) sc.counts[w0 >> 4] = sc_tile_count;
            sc_tile_count = 0;
          }
        }
        flushed += (upto - flushed) & ~511u;                               // (those bytes are in HBM; what is left is a block's last, shorter group)
      }
      if (!SCAN && (flushed & 7u) == 0 && (((uintptr_t)out) & 7u) == 0) {        // whole 512-byte steps, 8 bytes per lane
        const uint32_t n512 = (upto - flushed) & ~511u;
#pragma unroll 1                                                         // (one or two trips: an unrolled form spends more scalar instructions on its trip count than the loop has work)
        for (uint32_t o0 = 0; o0 < n512; o0 += 512) {
          const uint32_t o = o0 + lane * 8;
          const uint64_t v = *(const uint64_t*)(ring + ((flushed + o) & (kRing - 1)));
          *(uint64_t*)(out + HO(flushed + o)) = v;
          if (HIST && HO(flushed + o0) == 0u && lane < 4u) *(uint64_t*)(out + kHistBytes + lane * 8u) = v;
        }
        flushed += n512;
      }
      if ((flushed & 3u) == 0 && (((uintptr_t)out) & 3u) == 0) {
        const uint32_t n4 = (upto - flushed) & ~3u, n256 = n4 & ~255u;
#pragma unroll 1
        for (uint32_t o0 = 0; o0 < n256; o0 += 256) {                    // whole 256-byte steps: every lane stores (a wave-uniform loop: no exec bookkeeping)
          const uint32_t o = o0 + lane * 4;
          const uint32_t v = *(const uint32_t*)(ring + ((flushed + o) & (kRing - 1)));
          *(uint32_t*)(out + HO(flushed + o)) = v;
          if (HIST && HO(flushed + o0) == 0u && lane < 8u) *(uint32_t*)(out + kHistBytes + lane * 4u) = v;
        }
        if (n256 != n4) {                                                // (a block's last flush)
          const uint32_t o = n256 + lane * 4;
          if (o < n4) { const uint32_t v = *(const uint32_t*)(ring + ((flushed + o) & (kRing - 1))); *(uint32_t*)(out + HO(flushed + o)) = v; }
        }
        flushed += n4;
      }
      if (flushed != upto) { for (uint32_t o = flushed + lane; o < upto; o += 64) out[HO(o)] = ring[o & (kRing - 1)]; }
      flushed = upto;
    };

    // far prefetch + byte production + flush of ONE superbatch out of the current record slot (info / bitsx / fard): T output bytes, nfar far sources
    auto produce = [&](uint32_t T, uint32_t nfar) {
          LZ4_COUNT(6, 1); LZ4_COUNT(7, (T + 63) / 64); LZ4_COUNT(10, nfar);
#ifdef DFDB_LZ4_SKIP_FAR      // tools/bench_lz4 experiment only: what the far round trip costs (the output is WRONG without it)
      if (false) {
#else
      if (nfar && !kParserFar) {
#endif
        // far sources were flushed before this superbatch began (they lie > kRing - 64 - kBatchBytes behind op and at most
        // kFlush + 256 bytes are ever unflushed): 24 bytes each, HBM/L2 -> LDS, ONE memory round trip for the whole superbatch
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane < nfar) {
          const uint32_t so = fard[lane];
          uint64_t* d = (uint64_t*)(lds + kStage + kRing + lane * 24u);
          // (24 bytes whatever the match's length: near a block's end the last of them lie past its output — in the next block's, or in the >= 32 bytes
          //  of slack every caller leaves behind the last block (kernels.hpp) — and are never used: the bytes a sequence copies all precede `op`)
          const uint8_t* sp = out + HO(so);          // (HIST: at most 24 bytes past the lap's end: the mirror)
          const uint64_t a = ld_u64_unaligned(sp), b2 = ld_u64_unaligned(sp + 8), c2 = ld_u64_unaligned(sp + 16);
          d[0] = a; d[1] = b2; d[2] = c2;
        }
        wave_lds_fence();
        LZ4_PROF(5);
      }
      // ---- phase 5
      constexpr int U = DFDB_LZ4_PIPE_U;                               // rows per trip
      for (uint32_t c = 0; c < T; c += 64u * U) {
        uint32_t ORD[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint2 e = bitsx[(c >> 5) + 2u * (uint32_t)u + (lane >> 5)];          // (a row past T reads the {0, -1} tail)
          ORD[u] = e.y + (uint32_t)__builtin_popcount(e.x & lane_below);
        }
        uint2 INF[U];
#pragma unroll
        for (int u = 0; u < U; u++) INF[u] = info[(int32_t)ORD[u]];                   // (rows past T: ordinal -1 or the last sequence's: any record, the byte is never written)
        LZ4_PROF(16);
        uint32_t R[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint32_t j = c + 64u * (uint32_t)u + lane;
          const uint2 inf = INF[u];
          const uint32_t litend = inf.x & 0xffffu, lbo = inf.x >> 16, mbo = inf.y & 0xffffu, off7 = inf.y >> 16;
          const bool is_lit = j < litend;
          const uint32_t x = is_lit ? lbo : mbo;
          const uint32_t addr = ((j + x) & (uint32_t)(kStage - 1)) | (x & ~(uint32_t)(kStage - 1));
          const bool inrow = !is_lit && off7 <= lane;                  // the source byte is made by a lower lane of this very row
          R[u] = inrow ? lane - off7 : (0x80000000u | addr);           // (pointers only ever go down: lane 0 is always a root, rows past T included)
        }
        LZ4_PROF(17);
        for (;;) {                                                     // pointer doubling: R[j] = R[R[j]] (always a lower lane)
          uint32_t unres = 0;                                          // bit 31 set: some row of this lane still holds a pointer
#pragma unroll
          for (int u = 0; u < U; u++) unres |= ~R[u];
          if (__ballot((unres >> 31) != 0u) == 0) break;
#pragma unroll
          for (int u = 0; u < U; u++) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((R[u] & 63u) << 2), (int)R[u]);
            if ((R[u] >> 31) == 0u) R[u] = t;
          }
          LZ4_COUNT(8, 1);
        }
        LZ4_PROF(18);
#pragma unroll
        for (int u = 0; u < U; u++) {                                  // in row order: a later row may copy bytes an earlier row of this trip wrote
          const uint32_t j = c + 64u * (uint32_t)u + lane;
          const uint8_t v = lds[R[u] & 0xffffu];
          // (a select instead of a branch: every divergent `if` is an exec save / restore on the CU's one scalar unit, and these run eight times per superbatch)
          lds[j < T ? (uint32_t)kStage + ((op + j) & (uint32_t)(kRing - 1)) : kDump + lane] = v;
        }
      }
      LZ4_PROF(19);
      op += T;
      LZ4_PROF(2);
      if (op - flushed >= kFlush) flush_to(op & ~kFA);
      LZ4_PROF(3);
    };
    uint32_t npub = 0; int pslot = 0; bool owned = !PIPE;   // PIPE parser: superbatches published, the slot being parsed into, whether it owns the ring right now
    if (PIPE) {
      use_slot(0);
      if (role == 0 && lane == 0) { for (int c = 0; c < 16; c++) ctl[c] = 0; }
      __syncthreads();
      if (role == 1) {
        // ---- the producer: superbatch after superbatch as the parser publishes them
        uint32_t done = 0, hand = 0;
        for (;;) {
          uint32_t spins = 0, pub;
          while ((pub = ctl[C_PUB]) <= done && !ctl[C_END]) { __builtin_amdgcn_s_sleep(2); if (++spins > kSpinLimit || ctl[C_ERR]) { ctl[C_ERR] = 1; break; } }
          if (ctl[C_ERR]) break;
          if (pub <= done) { if (ctl[C_PUB] <= done) break; else continue; }       // END and nothing left
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (ctl[C_HAND] != hand) { hand = ctl[C_HAND]; op = ctl[C_OP]; flushed = ctl[C_FL]; }
          const int sl = (int)(done & 1u);
          use_slot(sl);
          produce(ctl[C_HT + sl], ctl[C_HNF + sl]);
          ctl[C_OP] = op; ctl[C_FL] = flushed;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          done++; ctl[C_CONS] = done;
        }
        __syncthreads();
        continue;                                    // next block
      }
    }
    // prime the staging buffer: chunks 0 and 1 in LDS, chunk 2 in flight
    chunk_load(0); chunk_store(0);
    chunk_load(kChunk); chunk_store(kChunk);
    chunk_load(kStage);
    window_load(0);

    while (ip < in_len) {                          // every quantity that steers control flow is wave-uniform
      if (PIPE && owned) {                           // the parser used the ring itself (one-sequence path): give it back, with the positions
        ctl[C_OP] = op; ctl[C_FL] = flushed;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        ctl[C_HAND] = ctl[C_HAND] + 1u;
        owned = false;
      }
      {
        // ---- a SUPERBATCH of W 64-byte windows of input, in five phases that each expose their parallelism to the hardware.  What shaped
        // them (cycle probes in tools/bench_lz4, rocprofv3 instruction counters): one wave retires this code at ~10 cycles per instruction,
        // a block's time is the length of its dependent chains, and the expensive links are (a) VALU -> SGPR -> anything (v_readlane / ballot
        // feeding scalar code or a lane select: ~50 cycles) and (b) an LDS round trip (~55).  So: few of those links per sequence, many
        // independent ones in flight together, everything else in the vector domain.
        //  1. candidates: lane l decodes the token at window start + l AS IF a sequence began there (literal count, match length, where the
        //     next sequence would start).  All W windows together: the two dependent LDS round trips are paid once.
        //  2. walk: the chain of REAL starts.  A start whose successor leaves the window points at itself, so walks park on the last start.
        //     next^2, next^4, next^8 come from three rounds of W independent ds_bpermute; the serial walk then visits every eighth start
        //     (three v_readlane hops per window instead of up to 22) and three forward ds_permute levels mark the starts in between.
        //  3. dense records: every start writes its candidate to LDS at its ordinal among the starts (mbcnt of the start mask); the <= 171
        //     sequences are then finished 64 per row — output position by one DPP scan per ROW (not per window), far-source slots, validity,
        //     the budget cut at SEQUENCE granularity — leaving an 8-byte record each and one START BIT at its first output byte.
        //  4. sources that have left the ring are fetched from HBM, 24 bytes each, one memory round trip for the whole superbatch.
        //  5. production, IN OUTPUT ORDER, four 64-byte rows per trip: a lane finds its sequence as the rank of its position among the start
        //     bits (bitmap word + exclusive prefix side by side in LDS: one read, one v_bcnt), gathers the record — four 16-bit fields that
        //     make the byte's LDS address ((j + B) & mask) | O for literal / ring / far byte alike, or name the lower lane of the row that
        //     makes it — and pointers are collapsed by ds_bpermute doubling; the rows' gathers are in flight together, only the final byte
        //     fetches are ordered.
        LZ4_PROF(4);
        advance(ip);
        uint32_t T = 0;                  // output bytes accepted
        uint32_t nseq = 0;               // sequences found by the walk
        uint32_t consumed = 0;           // input bytes they cover = where the next sequence starts, relative to ip
        bool nonsimple = false;          // the walk ended on a sequence the batch does not take
        bool bad = false;
        uint32_t nfar = 0;
        // Two forms of the candidate phase.  The plain one takes sequences whose lengths fit the token (<= 14 literals, <= 18 match
        // bytes).  The EXT one also takes a length of 15 that continues in ONE more byte (a 255 there — lengths >= 270 / 274 — still
        // leaves the sequence to the one-sequence path): two more LDS reads per candidate, sequences that jump over whole windows.
        // A block switches to it for good once the one-sequence path has met two such tokens.
        uint64_t RECM[W];                // INDEX 1: the start bits the walk found (the EXT walk may run on past a sequence phase 3b rejects: only the accepted prefix is real)
        auto windows = [&](auto extc) {
          constexpr bool EXT = decltype(extc)::value;
          uint32_t NX[W];
          // ---- phase 1: only WHERE the next sequence would start (and whether the batch can take this one at all) is decoded for every
          // position; lengths and offsets are decoded later, 64 real sequences per row instead of 64 positions per window
          uint32_t tok[W], e1[W];
#pragma unroll
          for (int w = 0; w < W; w++) {
            const uint32_t pos = ip + 64u * (uint32_t)w + lane;
            tok[w] = stage[pos & (kStage - 1)];
            if (EXT) e1[w] = stage[(pos + 1) & (kStage - 1)];
          }
#pragma unroll
          for (int w = 0; w < W; w++) {
            const uint32_t pos = ip + 64u * (uint32_t)w + lane;
            const uint32_t token = tok[w];
            if (!EXT) {
              const uint32_t lit = token >> 4;
              const bool simple = token < 0xf0u && (token & 15u) != 15u && pos + lit + 3u < in_len;
              NX[w] = simple ? lane + 3u + lit : 1023u;   // where the following sequence starts, window-relative; 1023: a sequence the batch does not take
            } else {
              const bool le = token >= 0xf0u, me = (token & 15u) == 15u;
              const uint32_t lit = le ? 15u + e1[w] : token >> 4;
              const uint32_t end = pos + 3u + (le ? 1u : 0u) + lit + (me ? 1u : 0u);
              const bool simple = !(le && e1[w] == 255u) && end < in_len;
              NX[w] = simple ? end - (ip + 64u * (uint32_t)w) : 1023u;             // <= 63 + 274
            }
          }
          bits[lane & (kBatchBytes / 32 - 1)] = 0;
          LZ4_PROF(0);
          // ---- phase 2a: next^2, next^4 and next^8 of every candidate, all windows together (three rounds of W independent ds_bpermute).
          // A start whose successor lies outside the window (or that the batch does not take) points at itself, so do its powers: walks park there.
          // (level by level, each level's W gathers issued back to back and waited for once: see the pins in phase 2c)
          static_assert(W == 8 || W == 4 || W == 2, "the pins below name eight, four or two registers");
#define DFDB_PIN8(a) do { if constexpr (W == 8) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2 % W]), "+v"(a[3 % W]), "+v"(a[4 % W]), "+v"(a[5 % W]), "+v"(a[6 % W]), "+v"(a[7 % W])); \
                          else if constexpr (W == 4) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2 % W]), "+v"(a[3 % W])); \
                          else asm volatile("" : "+v"(a[0]), "+v"(a[1])); } while (0)
          uint32_t X8[W], P2[W], P4[W];
#pragma unroll
          for (int w = 0; w < W; w++) {
            const uint32_t nxp = NX[w] < 64u ? NX[w] : lane;
            P2[w] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nxp << 2), (int)nxp);
            // one register per window from here on: exit value (10 bits) | next^2 | next | taken | next^4
            NX[w] |= nxp << 16 | (NX[w] != 1023u ? 1u << 22 : 0u);
          }
          DFDB_PIN8(P2);
#pragma unroll
          for (int w = 0; w < W; w++) { P4[w] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(P2[w] << 2), (int)P2[w]); NX[w] |= P2[w] << 10; }
          DFDB_PIN8(P4);
#pragma unroll
          for (int w = 0; w < W; w++) { X8[w] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(P4[w] << 2), (int)P4[w]); NX[w] |= P4[w] << 23; }
          DFDB_PIN8(X8);
#undef DFDB_PIN8
          // ---- phase 2b: the serial part.  A hop is a VALU -> SGPR -> VALU round trip (v_readlane with a lane select that the previous
          // v_readlane produced: ~50 cycles on this part), and a block's latency is made of them, so the walk visits every EIGHTH start only:
          // three hops reach the last of the <= 22 starts of a window.
          uint64_t V[W];
          uint32_t p = 0;                  // next sequence start, relative to the window being walked
#pragma unroll
          for (int w = 0; w < W; w++) V[w] = 0;
#pragma unroll
          for (int w = 0; w < W; w++) {
            if (EXT && p >= 64u) { p -= 64u; consumed = 64u * (uint32_t)(w + 1) + p; continue; }   // a long sequence jumped over this window
            uint64_t v = 0;
#pragma unroll
            for (int u = 0; u < 3; u++) {
              asm("s_bitset1_b64 %0, %1" : "+s"(v) : "s"(p));
              p = rl(X8[w], p);
            }
            V[w] = v;                                                            // p: the last start of the window
            const uint32_t exitp = rl(NX[w], p) & 1023u;
            nonsimple = exitp == 1023u;
            if (nonsimple) { consumed = 64u * (uint32_t)w + p; break; }
            p = exitp - 64u; consumed = 64u * (uint32_t)(w + 1) + p;
          }
          LZ4_PROF(13);
          // ---- phases 2c + 3a: the starts between the visited ones (three forward permutes: the visited starts mark their fourth
          // successors, all of those their second, all of those their first; a lane that is no source sends to lane 0, which no hop can
          // target: a hop moves at least three positions), then every start drops its POSITION at its ordinal.
          // (in the vector domain until the one ballot per window that the ordinals need: flags are 0 / 1 registers, a permute address is
          //  flag * 4 * next — a ballot that feeds scalar code that feeds an inverse ballot costs two ~50-cycle crossings per level)
          // Level by level over ALL windows: the W permutes of a level are issued back to back and waited for once (the empty asm pins that
          // order: the compiler would otherwise sink every permute to its use and pay W x 3 LDS round trips one after another).
          const uint32_t ge3 = lane >= 3u ? 1u : 0u;
          static_assert(W == 8 || W == 4 || W == 2, "the pins below name eight, four or two registers");
          uint32_t FL[W], RP[W];
#define DFDB_PIN8(a) do { if constexpr (W == 8) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2 % W]), "+v"(a[3 % W]), "+v"(a[4 % W]), "+v"(a[5 % W]), "+v"(a[6 % W]), "+v"(a[7 % W])); \
                          else if constexpr (W == 4) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2 % W]), "+v"(a[3 % W])); \
                          else asm volatile("" : "+v"(a[0]), "+v"(a[1])); } while (0)
#pragma unroll
          for (int w = 0; w < W; w++) {
            FL[w] = __builtin_amdgcn_inverse_ballot_w64(V[w]) ? 1u : 0u;
            RP[w] = (uint32_t)__builtin_amdgcn_ds_permute((int)(((NX[w] >> 21) & 0xfcu) * FL[w]), (int)FL[w]);      // 4 * next^4
          }
          DFDB_PIN8(RP);
#pragma unroll
          for (int w = 0; w < W; w++) {
            FL[w] |= RP[w] & ge3;
            RP[w] = (uint32_t)__builtin_amdgcn_ds_permute((int)(((NX[w] >> 8) & 0xfcu) * FL[w]), (int)FL[w]);       // 4 * next^2
          }
          DFDB_PIN8(RP);
#pragma unroll
          for (int w = 0; w < W; w++) {
            FL[w] |= RP[w] & ge3;
            RP[w] = (uint32_t)__builtin_amdgcn_ds_permute((int)(((NX[w] >> 14) & 0xfcu) * FL[w]), (int)FL[w]);      // 4 * next
          }
          DFDB_PIN8(RP);
#undef DFDB_PIN8
          uint64_t MASK[W];
#pragma unroll
          for (int w = 0; w < W; w++) {
            FL[w] = (FL[w] | (RP[w] & ge3)) & (NX[w] >> 22) & 1u;                 // bit 22: a sequence the batch takes (a parked start that it does not take is marked too)
            MASK[w] = __ballot(FL[w] != 0u);
          }
          if (INDEX == 1) {
#pragma unroll
            for (int w = 0; w < W; w++) RECM[w] = MASK[w];                      // recorded once phase 3b has said which of them the batch takes
          }
          // (nseq <= kSeqMax by construction: starts are >= 3 input bytes apart, ceil(64 W / 3) of them at most)
#pragma unroll
          for (int w = 0; w < W; w++) {
            const uint32_t ord = nseq + __builtin_amdgcn_mbcnt_hi((uint32_t)(MASK[w] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)MASK[w], 0u));
            info[(int32_t)(FL[w] != 0u ? ord : 0xffffffffu)].x = 64u * (uint32_t)w + lane;   // where the sequence starts, relative to ip (no start: the dummy record; a select, not a branch)
            nseq += (uint32_t)__builtin_popcountll(MASK[w]);
          }
          LZ4_PROF(14);
        };
        const bool ext_mode = extstops >= 2u;
        if constexpr (INDEX == 2) {
          // ---- phases 1-3a from the index: the start bits of the W windows, every start's position at its ordinal
          bits[lane & (kBatchBytes / 32 - 1)] = 0;
          const uint64_t gbit = (uint64_t)blk.src_off + ip;                 // the index bit of input position ip
          if (gbit < ibase || gbit + 64u * W + 96u > ibase + 2048u) { ibase = gbit & ~31ull; iw = (index + (ibase >> 5))[lane]; }
          const uint32_t o0 = (uint32_t)(gbit - ibase);
          // every lane fetches the dword that holds ITS bit (ds_bpermute out of the register window: no LDS memory, and the W fetches are in flight
          // together) — the scalar form (three v_readlane and a 64-bit funnel shift per window) cost ~25 scalar instructions and a VALU -> SGPR
          // crossing per window, on the one scalar unit the CU's four SIMDs share
          uint32_t WD[W];
#pragma unroll
          for (int w = 0; w < W; w++) {
            const uint32_t bp = o0 + 64u * (uint32_t)w + lane;
            WD[w] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((bp >> 3) & ~3u), (int)iw);
          }
          const uint32_t left = in_len - ip;                                   // input bytes from ip on: bits at or past in_len are the next block's
#pragma unroll
          for (int w = 0; w < W; w++) {
            const uint32_t rel = 64u * (uint32_t)w + lane, bp = o0 + rel;
            const bool st = ((WD[w] >> (bp & 31u)) & 1u) != 0u && rel < left;
            const uint64_t m = __ballot(st);
            const uint32_t ord = nseq + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            info[(int32_t)(st && ord < (uint32_t)kSeqMax ? ord : 0xffffffffu)].x = rel;      // (no start here: the dummy entry in front of the records takes it)
            nseq += (uint32_t)__builtin_popcountll(m);
          }
          if (nseq > (uint32_t)kSeqMax || nseq == 0u) { err = 9; break; }    // not an index of this stream
        } else { if (ext_mode) windows(std::true_type{}); else windows(std::false_type{}); }
        LZ4_PROF(11);
        if (nseq) {
          wave_lds_fence();
          // ---- phase 3b: the sequences, dense, 64 per row
          uint32_t nacc = 0;
          for (uint32_t r0 = 0; r0 < nseq; r0 += 64u) {
            const uint32_t k = r0 + lane;
            const bool valid = k < nseq;
            // the sequence at that position, decoded now that it is known to be one (the candidate phase only found where it ends)
            const uint32_t spos = ip + (valid ? info[k].x : 0u);
            const uint32_t token = stage[spos & (kStage - 1)];
            const bool le = ext_mode && token >= 0xf0u, me = ext_mode && (token & 15u) == 15u;
            const uint32_t e1 = ext_mode ? (uint32_t)stage[(spos + 1) & (kStage - 1)] : 0u;
            const uint32_t lit = valid ? (le ? 15u + e1 : token >> 4) : 0u;
            const uint32_t opos = spos + 1u + (le ? 1u : 0u) + lit;             // the 2-byte offset field
            const uint32_t offset = (uint32_t)stage[opos & (kStage - 1)] | (uint32_t)stage[(opos + 1) & (kStage - 1)] << 8;
            const uint32_t m1 = ext_mode ? (uint32_t)stage[(opos + 2) & (kStage - 1)] : 0u;
            const uint32_t ml = valid ? 4u + (me ? 15u + m1 : (token & 15u)) : 0u;
            const uint32_t inpos = spos - ip + 1u + (le ? 1u : 0u);             // its literals, relative to ip
            // EXT form only: a length that continues past its one extension byte, or a far source longer than the 24 prefetched bytes,
            // belongs to the one-sequence path: the superbatch ends in front of it
            // INDEX 2: nothing has looked at this sequence yet — what the candidate phase decides per position (can the batch take it at all?) is decided here
            const bool rej_idx = INDEX == 2 && (ext_mode ? ((le && e1 == 255u) || !(opos + 2u + (me ? 1u : 0u) < in_len))
                                                         : (token >= 0xf0u || (token & 15u) == 15u || !(spos + lit + 3u < in_len)));
            const bool reject = rej_idx || (ext_mode && ((me && m1 == 255u) || (offset + 64u > (uint32_t)kRing && ml > 24u)));
            const uint32_t tot = lit + ml;                                       // (0 for a lane past the last sequence)
            const uint32_t incl = T + wave_incl_scan(tot);
            const uint32_t ostart = incl - tot;
            const bool far = valid && offset + 64u > (uint32_t)kRing;
            const uint64_t farmask = __ballot(far);
            const uint32_t fo = nfar + __builtin_amdgcn_mbcnt_hi((uint32_t)(farmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)farmask, 0u));
            // the budget cut: output bytes and far slots both grow with the ordinal, so the accepted sequences are a prefix
            const bool ok = valid && !reject;
            const uint64_t okmask = __ballot(ok) | ~__ballot(valid);             // the sequences before the first rejected one
            const bool before = okmask == ~0ull || lane < (uint32_t)__builtin_ctzll(~okmask);
            const bool acc = valid && before && incl <= (uint32_t)kBatchBytes && fo + (far ? 1u : 0u) <= (uint32_t)kFarMax;
            const uint32_t na = (uint32_t)__builtin_popcountll(__ballot(acc));
            if (acc) {
              bad = bad || offset == 0u || offset > op + ostart + lit;
              fard[far ? fo : (uint32_t)kFarMax] = op + ostart + lit - offset;  // where its source starts in the block's output (not far: the slot nobody reads)
              // the record production reads, as 16-bit fields.  A byte j of the sequence comes from LDS address ((j + B) & (kStage - 1)) | O
              // with (B, O) = the literal pair below its literal end and the match pair from there on:
              //   literal  B = ip + inpos - ostart          O = 0                 (staging buffer)
              //   match    B = op - offset                  O = kStage            (ring; valid when the source precedes the row)
              //   far      B = 24 fo - (ostart + lit)       O = kStage + kRing    (prefetched source bytes)
              // off7 = the match distance when it can fall inside a 64-byte row (else 127): lane - off7 is then the lane that makes the byte
              const uint32_t litend = ostart + lit;
              const uint32_t lbo = (ip + inpos - ostart) & (uint32_t)(kStage - 1);
              // (both forms computed, one kept by a mask: as a conditional the compiler turns it into two exec-masked regions, four scalar instructions per row of sequences)
              const uint32_t mfar = ((fo * 24u + (kParserFar ? (uint32_t)pslot * (uint32_t)(kFarMax * 24) : 0u) - litend) & (uint32_t)(kStage - 1)) | (uint32_t)(kStage + kRing);
              const uint32_t mnear = ((op - offset) & (uint32_t)(kStage - 1)) | (uint32_t)kStage;
              const uint32_t fsel = 0u - (uint32_t)far;
              const uint32_t mbo = (mfar & fsel) | (mnear & ~fsel);
              const uint32_t off7 = far || offset > 127u ? 127u : offset;
              info[k] = make_uint2(litend | lbo << 16, mbo | off7 << 16);
              atomicOr(&bits[ostart >> 5], 1u << (ostart & 31u));
            }
            if (na) T = rl(incl, na - 1u);
            nacc += na;
            nfar += (uint32_t)__builtin_popcountll(farmask & (na >= 64u ? ~0ull : ((1ull << na) - 1ull)));
            const uint32_t nvalid = nseq - r0 < 64u ? nseq - r0 : 64u;
            if (INDEX == 2 && na == nvalid) consumed = rl(opos + 2u + (me ? 1u : 0u) - ip, nvalid - 1u);   // where the sequence after the last one starts
            if (na < nvalid) {                                                   // cut here: the next superbatch starts with sequence r0 + na
              consumed = rl(spos - ip, na);
              nonsimple = ((~okmask >> na) & 1ull) != 0;                          // stopped by a sequence only the one-sequence path takes
              break;
            }
          }
          (void)nacc;
        }
        LZ4_PROF(12);
        if (INDEX == 1 && lane == 0) {
          // record: the start bits of the sequences this superbatch takes — the walked starts in front of `consumed` (accepted sequences are a prefix in
          // position order; the start at `consumed` is the next superbatch's, or the one-sequence path's, to record)
#pragma unroll
          for (int w = 0; w < W; w++) {
            uint64_t m = RECM[w];
            const uint32_t lim = consumed > 64u * (uint32_t)w ? consumed - 64u * (uint32_t)w : 0u;
            if (lim < 64u) m &= (1ull << lim) - 1ull;
            if (m == 0) continue;
            const uint64_t gb = (uint64_t)blk.src_off + ip + 64u * (uint32_t)w;
            const uint32_t sh = (uint32_t)(gb & 63ull);
            unsigned long long* iw64 = (unsigned long long*)index + (gb >> 6);
            atomicOr(iw64, (unsigned long long)(m << sh));
            if (sh) atomicOr(iw64 + 1, (unsigned long long)(m >> (64u - sh)));
          }
        }
        if (T) {
          if (__ballot(bad) != 0 || T > out_len - op) { err = 5; break; }
          wave_lds_fence();
          LZ4_COUNT(9, nseq);
          // rank of an output position among the start bits = its sequence's ordinal.  Kept in the vector domain (a v_readlane of the bitmap
          // followed by scalar popcounts cost a VALU -> SGPR round trip of ~50 cycles per row): word and exclusive prefix sit side by side in LDS
          {
            constexpr uint32_t kW = kWords < 32u ? kWords : 32u;    // (kWords == 64 has never been used: one bitsx entry per lane at most)
            const uint32_t w = bits[lane & (kWords - 1)];
            const uint32_t cw = (uint32_t)__builtin_popcount(w);
            const uint32_t incl = wave_incl_scan(lane < kW ? cw : 0u);
            if (lane < kW) bitsx[lane] = make_uint2(w, incl - cw - 1u);
          }
          wave_lds_fence();
          if constexpr (kParserFar) {
           if (nfar) {
            // the far sources of this superbatch, fetched by the PARSER: they lie more than kRing - 64 bytes behind their sequence, i.e. more than
            // kRing - 64 - 2 kBatchBytes behind the superbatch the producer can still be working on (it has finished the one before that: the wait below),
            // and everything older than kFlush + 256 bytes behind a finished superbatch's end is in HBM (the producer's release, our acquire at the wait)
            static_assert(kRing - 64 - 2 * kBatchBytes >= (kRing >= 4096 ? 1024 : 512) + 256, "parser-side far prefetch needs the sources flushed");
            if (lane < nfar) {
              const uint32_t so = fard[lane];
              uint64_t* d = (uint64_t*)(lds + kStage + kRing + (uint32_t)pslot * (uint32_t)(kFarMax * 24) + lane * 24u);
              const uint64_t a = ld_u64_unaligned(out + so), b2 = ld_u64_unaligned(out + so + 8), c2 = ld_u64_unaligned(out + so + 16);
              d[0] = a; d[1] = b2; d[2] = c2;
            }
            wave_lds_fence();
           }
          }
          if (PIPE) {
            // hand the superbatch to the producer and go on parsing; the slot after next is free once the producer has consumed the one before
            ctl[C_HT + pslot] = T; ctl[C_HNF + pslot] = nfar;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            npub++; ctl[C_PUB] = npub;
            op += T; ip += consumed;
            pslot ^= 1; use_slot(pslot);
            { uint32_t spins = 0; while (npub >= 2u && ctl[C_CONS] + 1u < npub) { __builtin_amdgcn_s_sleep(2); if (++spins > kSpinLimit || ctl[C_ERR]) { err = 7; break; } } }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (err) break;
            if (!nonsimple) continue;
          } else {
            produce(T, nfar);
            ip += consumed;
            if (!nonsimple) continue;
          }
        }
      }
      if (PIPE) {
        // a sequence only the one-sequence path takes: wait until the producer has drained, take the ring over
        uint32_t spins = 0;
        while (ctl[C_CONS] != npub) { __builtin_amdgcn_s_sleep(2); if (++spins > kSpinLimit || ctl[C_ERR]) { err = 7; break; } }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (err) break;
        flushed = ctl[C_FL];
        if (ctl[C_OP] != op) { err = 8; break; }
        owned = true;
      }
      ensure(ip);
      if (INDEX == 1 && lane == 0) { const uint64_t gb = (uint64_t)blk.src_off + ip; atomicOr((unsigned long long*)index + (gb >> 6), 1ull << (gb & 63ull)); }
      const uint64_t t64 = fetch64(ip);
      const uint32_t token = (uint32_t)t64 & 255u;
      ip++;
      uint32_t lit = token >> 4;
      if ((lit == 15u || (token & 15u) == 15u)) extstops++;
      if (lit == 15) {
        uint32_t bb;
        do { if (ip >= in_len) { err = 1; break; } ensure(ip); bb = byte_at(ip); ip++; lit += bb; } while (bb == 255);
        if (err) break;
      }
      if (lit > in_len - ip || lit > out_len - op) { err = 2; break; }
      const uint32_t lit_ip = ip;                  // the literals are input bytes [lit_ip, lit_ip + lit)
      const bool last = ip + lit >= in_len;        // the last sequence is literals only
      uint32_t offset = 0, ml = 0;
      // the match fields of a sequence with > 64 literals or with match-length extension bytes (up to 2056 of them) are
      // parsed AFTER its literals have left the staging buffer: walking them may recycle staging slots
      const bool defer = lit > 64u || (token & 15u) == 15u;
      if (!last) {
        const uint32_t mp = ip + lit;              // position of the 2-byte offset
        if (mp + 2 > in_len) { err = 3; break; }
        if (!defer) {
          if (lit <= 5) offset = (uint32_t)(t64 >> (8u * (1u + lit))) & 0xffffu;   // still inside the 8 bytes already fetched
          else { window_only(mp); offset = (uint32_t)fetch64(mp) & 0xffffu; }
          ml = (token & 15u) + 4u;
          if (offset == 0 || offset > op + lit || ml > out_len - op - lit) { err = 5; break; }
          ip = mp + 2;
        }
      } else ip += lit;

      if (!defer && lit + ml <= 64u && offset + 64u <= (uint32_t)kRing) {
        // ---- fast path: the whole sequence is <= 64 bytes; lane k makes output byte op + k
        const uint32_t total = lit + ml, mbase = op + lit;
        if (total) {
          const uint32_t k = lane;
          uint32_t j = k - lit;                                          // index inside the match
          if (offset < 64u && offset != 0) j = j % offset;               // periodic source (overlapping match)
          const uint32_t s = mbase - offset + j;                         // absolute output position of the source byte
          const bool from_in = k < lit || s >= op;                       // a literal, or a match byte inside this sequence's literals
          const uint32_t in_pos = k < lit ? lit_ip + k : lit_ip + (s - op);
          const uint8_t* a = from_in ? stage + (in_pos & (kStage - 1)) : ring + (s & (kRing - 1));
          if (k < total) { const uint8_t v = *a; ring[(op + k) & (kRing - 1)] = v; }
          op += total;
        }
      } else {
        // ---- general path: 64 bytes at a time
        uint32_t rem = lit, lp = lit_ip;
        while (rem) {                                                    // literals: staging buffer -> ring
          ensure(lp);
          const uint32_t n = rem < 64u ? rem : 64u;
          if (lane < n) ring[(op + lane) & (kRing - 1)] = stage[(lp + lane) & (kStage - 1)];
          lp += n; op += n; rem -= n;
          if (op - flushed >= kFlush) flush_to(op & ~kFA);
        }
        if (!last) {
          if (defer) {                                                   // match fields parsed now: the literals are out of the staging buffer
            uint32_t mp = lp;
            ensure(mp); offset = (uint32_t)fetch64(mp) & 0xffffu; mp += 2;
            ml = token & 15u;
            if (ml == 15) {
              uint32_t bb;
              do { if (mp >= in_len) { err = 4; break; } ensure(mp); bb = byte_at(mp); mp++; ml += bb; } while (bb == 255);
              if (err) break;
            }
            ml += 4;
            if (offset == 0 || offset > op || ml > out_len - op) { err = 5; break; }
            ip = mp;
          }
          if (offset + 64u <= (uint32_t)kRing) {                         // source inside the ring
            uint32_t done = 0;
            while (done < ml) {
              const uint32_t n = ml - done < 64u ? ml - done : 64u;
              const uint32_t j = offset < 64u ? lane % offset : lane;    // chunk start - offset + (i mod offset): always already written
              const uint32_t s = op - offset + j;
              if (lane < n) { const uint8_t v = ring[s & (kRing - 1)]; ring[(op + lane) & (kRing - 1)] = v; }
              op += n; done += n;
              if (op - flushed >= kFlush) flush_to(op & ~kFA);
            }
          } else if (SCAN || HIST) {
            // far match, SCAN form: the bytes must pass the ring (and the predicate) like all others, so they come back from HBM 64 at a
            // time (the source lies more than kRing - 64 bytes back: always flushed, never inside the step)
            uint32_t done = 0, fenced = 0xffffffffu;
            while (done < ml) {
              const uint32_t n = ml - done < 64u ? ml - done : 64u;
              if (fenced != flushed) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_s_waitcnt(0); fenced = flushed; }
              if (lane < n) { const uint8_t v = __builtin_nontemporal_load(out + HO(op - offset + lane)); ring[(op + lane) & (kRing - 1)] = v; }
              op += n; done += n;
              if (op - flushed >= kFlush) flush_to(op & ~kFA);
            }
          } else {                                                       // far match: through HBM, v1 style
            flush_to(op);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            const uint8_t* m = out + op - offset;                        // offset > 8128 > any 64-byte step: never overlaps a step
            for (uint32_t k = lane; k < ml; k += 64) {
              const uint8_t v = __builtin_nontemporal_load(m + k);
              out[op + k] = v;
              if (ml - (k - lane) <= (uint32_t)kRing) ring[(op + k) & (kRing - 1)] = v;   // keep the ring current (its last 8 KB)
            }
            if (ml >= (uint32_t)kRing - 64u) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
            op += ml; flushed = op;
          }
        }
      }
      if (op - flushed >= kFlush) flush_to(op & ~kFA);
      if (last) break;
    }
    if (PIPE) {
      if (!owned && !err) {                       // take the ring back for the epilogue
        uint32_t spins = 0;
        while (ctl[C_CONS] != npub) { __builtin_amdgcn_s_sleep(2); if (++spins > kSpinLimit || ctl[C_ERR]) { err = 7; break; } }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (!err) { flushed = ctl[C_FL]; if (ctl[C_OP] != op) err = 8; }
      }
      if (err) ctl[C_ERR] = 1;
      ctl[C_END] = 1;                              // the producer leaves its loop
    }
    if (!err && op != out_len) err = 6;         // @assert size == sizes.origin "decompression error" (:112)
    if (!err) flush_to(op);
    if (SCAN && !err && (sc_words & 15u) != 0u) {                 // the column's last, shorter block: a partial tile
      const bool mine = lane < (sc_words & 15u);
      if (sc.and_existing) {
        if (mine) sc_myword &= sc.bitmap[sc_word0 + (sc_words & ~15u) + lane];
        uint32_t c = mine ? (uint32_t)__builtin_popcountll(sc_myword) : 0u;
        for (int d = 8; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        sc_tile_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
      }
      if (mine) sc.bitmap[sc_word0 + (sc_words & ~15u) + lane] = sc_myword;
      if (lane == 0u) sc.counts[(sc_word0 + sc_words) >> 4] = sc_tile_count;
    }
#ifdef DFDB_LZ4_PROF
    if (b == 0 && lane == 0) { pf_acc[15] = __builtin_readcyclecounter() - pf_start; for (int k = 0; k < 32; k++) g_lz4_prof[k] = pf_acc[k]; }
#endif
    if (lane == 0) status[b] = err;
    wave_lds_fence();
    if (PIPE) __syncthreads();                    // (the producer waits here too: the next block reuses every LDS array)
    (void)owned; (void)npub; (void)pslot;
  }
}

// Measured on 8-byte integer columns (one sequence per 8 output bytes, the worst case for a block-serial format): see DESIGN.md §4 K7.
// (The four earlier decoders — plain global round trips, LDS staging, register-window parser, one-window batches: 26-54 GB/s — were
// dropped from the library in round 2; git history and DESIGN.md §10 keep what they taught.)
// pipe: -1: by block count, 0: one wave per block, 1: two-wave pipeline (ctx option "lz4_pipeline", tools/bench_lz4)
void launch_lz4_decode(hipStream_t s, const uint8_t* src, uint8_t* dst, const Lz4Block* blocks, int32_t nblocks, int32_t* status, int pipe, uint32_t* index, int index_mode) {
  if (nblocks <= 0) return;
#ifndef DFDB_LZ4_HIST_ONLY      // (a build of the history-ring forms alone, for looking at their registers: not the library's)
  // latency-bound: give every block its own wave and let the CUs hold as many as they can
  int64_t g5 = nblocks; if (g5 > (1 << 20)) g5 = 1 << 20;
  // fewer blocks than the chip has wave slots: the two-wave pipeline shortens what matters then, a block's latency
  // Superbatch shape (round 3, tools/bench_lz4 pipe = 10 / 14 / 15 / 16 / 17, profiles/r3_lz4_harness.txt): 4 windows / 512 output bytes / 32 far slots
  // need 79 VGPRs and 5.9 KB of LDS -> 6 waves per SIMD, and decode 8-byte integer columns at 431-445 GB/s where round 2's 8 windows / 1024 bytes /
  // 64 far slots (91 VGPRs, 7.7 KB: 5 waves) gave 404-412; +4 ... +17 % on every body tried (1:n 396 -> 438, h mod 1000 370 -> 384, runs 364 -> 391,
  // String 400 -> 431, incompressible 945 -> 1076, zeros 1195 -> 1406 GB/s).  Asking for 7 waves (72 VGPRs, 24 far slots) gave 415, for 8 (64 VGPRs:
  // 4 spills) 396, 8 windows at 6 waves 428: past six waves per SIMD the LDS pipeline and instruction issue are what the waves share, not latency.
  if (pipe == 10) { hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 1024, 8, 0, 0>), dim3((unsigned)g5), dim3(64), 0, s, src, dst, blocks, nblocks, status, LzScan{}, nullptr); return; }
  if (pipe == 15) { hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 0, 0, 7, 24>), dim3((unsigned)g5), dim3(64), 0, s, src, dst, blocks, nblocks, status, LzScan{}, nullptr); return; }
  // (other shapes of the indexed form — 8 windows / 1024 bytes with 32 or 48 far slots, 7 waves per SIMD — measured within 1 % of this one: profiles/r3_lz4_index.txt)
  // the two-wave pipeline needs 128 VGPRs per wave: 4 waves per SIMD = 8 workgroups per CU = 2048 blocks resident at once, and a block takes ~3.3 ms
  // there however few there are (763 blocks 2.6 ms, 1526 3.3 ms, 2048 3.4 ms = 312 GB/s; 2560 blocks need a second round: 5.5 ms, where one wave per
  // block takes 5.3); superbatches of 4 windows in the pipeline are slower (1526 blocks: 214 vs 242 GB/s: twice the hand-offs)
  const bool pipeline = pipe == 1 || (pipe < 0 && nblocks <= 2048);
  if (pipeline && index && index_mode == 2)
    hipLaunchKernelGGL((k_lz4_decode<2, 4096, 4096, 1024, 8, 0, 1, 4, 64, 2>), dim3((unsigned)g5), dim3(128), 0, s, src, dst, blocks, nblocks, status, LzScan{}, index);
  else if (pipeline && !(index && index_mode == 1))                     // (a recording launch takes the one-wave form whatever the block count: once per column)
    hipLaunchKernelGGL((k_lz4_decode<2, 4096, 4096, 1024, 8, 0, 1>), dim3((unsigned)g5), dim3(128), 0, s, src, dst, blocks, nblocks, status, LzScan{}, nullptr);
  else if (index && index_mode == 2)
    hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 0, 0, 6, 32, 2>), dim3((unsigned)g5), dim3(64), 0, s, src, dst, blocks, nblocks, status, LzScan{}, index);
  else if (index && index_mode == 1)
    hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 0, 0, 6, 32, 1>), dim3((unsigned)g5), dim3(64), 0, s, src, dst, blocks, nblocks, status, LzScan{}, index);
  else
    hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 0, 0, 6, 32>), dim3((unsigned)g5), dim3(64), 0, s, src, dst, blocks, nblocks, status, LzScan{}, nullptr);
#endif
}
// whether launch_lz4_decode would take (record or read) an index for this many blocks under this pipeline setting
bool lz4_decode_takes_index(int32_t nblocks, int pipe) { (void)nblocks; return pipe != 10 && pipe != 15; }
// decode + `value OP c` over an 8-byte column in one pass: dst receives the decoded column, sc.bitmap / sc.counts what K1 would write
void launch_lz4_decode_scan(hipStream_t s, const uint8_t* src, uint8_t* dst, const Lz4Block* blocks, int32_t nblocks, int32_t* status, const LzScan& sc,
                            uint32_t* index, int index_mode) {
  if (nblocks <= 0) return;
#ifndef DFDB_LZ4_HIST_ONLY
  int64_t g5 = nblocks; if (g5 > (1 << 20)) g5 = 1 << 20;
  if (index && index_mode == 2) hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 1, 0, 6, 32, 2>), dim3((unsigned)g5), dim3(64), 0, s, src, dst, blocks, nblocks, status, sc, index);
  else if (index && index_mode == 1) hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 1, 0, 6, 32, 1>), dim3((unsigned)g5), dim3(64), 0, s, src, dst, blocks, nblocks, status, sc, index);
  else hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 1, 0, 6, 32>), dim3((unsigned)g5), dim3(64), 0, s, src, dst, blocks, nblocks, status, sc, nullptr);
#endif
}

// the history-ring forms (HIST): one workgroup = one wave = one 64-KB ring for the life of the launch; blocks by ticket
size_t lz4_hist_scratch_bytes(int waves) { return 256 + (size_t)waves * kHistStride + 256; }
int lz4_hist_default_waves(int compute_units) { return compute_units * 4 * 6; }      // 4 SIMDs x the 6 waves this kernel's 79 VGPRs / 5.9 KB of LDS allow
void launch_lz4_decode_hist(hipStream_t s, const uint8_t* src, uint8_t* scratch, int waves, const Lz4Block* blocks, int32_t nblocks, int32_t* status, const LzScan* scp,
                            uint32_t* index, int index_mode) {
  if (nblocks <= 0) return;
  if (waves > nblocks) waves = nblocks;
  if (waves < 1) waves = 1;
  (void)hipMemsetAsync(scratch, 0, 4, s);
  LzScan sc = scp ? *scp : LzScan{};
  sc.ticket = (uint32_t*)scratch;
  uint8_t* rings = scratch + 256;
  const dim3 g((unsigned)waves), b(64);
  if (!index) index_mode = 0;
  if (scp) {
    if (index_mode == 2) hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 1, 0, 6, 32, 2, 1>), g, b, 0, s, src, rings, blocks, nblocks, status, sc, index);
    else if (index_mode == 1) hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 1, 0, 6, 32, 1, 1>), g, b, 0, s, src, rings, blocks, nblocks, status, sc, index);
    else hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 1, 0, 6, 32, 0, 1>), g, b, 0, s, src, rings, blocks, nblocks, status, sc, nullptr);
  } else {
    if (index_mode == 2) hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 0, 0, 6, 32, 2, 1>), g, b, 0, s, src, rings, blocks, nblocks, status, sc, index);
    else if (index_mode == 1) hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 0, 0, 6, 32, 1, 1>), g, b, 0, s, src, rings, blocks, nblocks, status, sc, index);
    else hipLaunchKernelGGL((k_lz4_decode<1, 2048, 2048, 512, 4, 0, 0, 6, 32, 0, 1>), g, b, 0, s, src, rings, blocks, nblocks, status, sc, nullptr);
  }
}

// ---------------------------------------------------------------- K8: Union{T,Missing} bodies
// body = cld(rows,64) UInt64 chunks (bit i = row i missing) then rows*width values (blocks.jl:9-18,46-60).
// One workgroup per block: values are copied to the column at the block's row offset; the bits are
// re-packed at the block's global bit position (blocks need not start on a word boundary).
__global__ __launch_bounds__(kBlock) void k_unpack_nullable(const uint8_t* __restrict__ bodies, const int64_t* __restrict__ body_off,
                                                            const int64_t* __restrict__ row_off, const int64_t* __restrict__ rows_of, int width,
                                                            uint8_t* __restrict__ values, uint64_t* __restrict__ missing_bits) {
  const int b = blockIdx.x;
  const int64_t r0 = row_off[b], rows = rows_of[b];      // (the blocks need not be adjacent: a streamed chunk decodes only those with survivors)
  const uint8_t* body = bodies + body_off[b];
  const int64_t nchunks = (rows + 63) / 64;
  const uint64_t* chunks = (const uint64_t*)body;
  const uint8_t* vals = body + nchunks * 8;
  const int64_t nbytes = rows * width;
  uint8_t* vdst = values + r0 * width;
  for (int64_t k = threadIdx.x; k < nbytes; k += kBlock) vdst[k] = vals[k];
  if ((r0 & 63) == 0) {     // the usual case (block_size a multiple of 64): the block's chunks ARE the column's bitmap words
    uint64_t* mw = missing_bits + (r0 >> 6);
    for (int64_t k = threadIdx.x; k < nchunks; k += kBlock) {
      uint64_t w = chunks[k];
      const int64_t left = rows - k * 64;
      if (left < 64) w &= (1ull << left) - 1ull;                  // bits past the block's last row are not rows
      if (w) atomicOr((unsigned long long*)&mw[k], w);           // (the next block may share the last word when rows % 64 != 0)
    }
    return;
  }
  for (int64_t i = threadIdx.x; i < rows; i += kBlock) {
    if ((chunks[i >> 6] >> (i & 63)) & 1ull) {
      const int64_t g = r0 + i;
      atomicOr((unsigned long long*)&missing_bits[g >> 6], 1ull << (g & 63));
    }
  }
}
void launch_unpack_nullable(hipStream_t s, const uint8_t* bodies, const int64_t* body_off, const int64_t* row_off, const int64_t* rows_of, int32_t nblocks, int width,
                            uint8_t* values, uint64_t* missing_bits) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(k_unpack_nullable, dim3((unsigned)nblocks), dim3(kBlock), 0, s, bodies, body_off, row_off, rows_of, width, values, missing_bits);
}

// ---------------------------------------------------------------- String bodies
// body = Int32 datasize, rows x Int32 sizes, datasize bytes (blocks.jl:21-33,62-71)
__global__ __launch_bounds__(kBlock) void k_unpack_strings(const uint8_t* __restrict__ bodies, const int64_t* __restrict__ body_off,
                                                           const int64_t* __restrict__ row_off, const int64_t* __restrict__ rows_of, const int64_t* __restrict__ byte_off,
                                                           int32_t* __restrict__ sizes, uint8_t* __restrict__ bytes) {
  const int b = blockIdx.x;
  const int64_t r0 = row_off[b], rows = rows_of[b];
  const int64_t nb = byte_off[b + 1] - byte_off[b];
  const uint8_t* body = bodies + body_off[b];
  const int32_t* bs = (const int32_t*)(body + 4);
  const uint8_t* bd = body + 4 + rows * 4;
  for (int64_t i = threadIdx.x; i < rows; i += kBlock) sizes[r0 + i] = bs[i];
  uint8_t* d = bytes + byte_off[b];
  for (int64_t k = threadIdx.x; k < nb; k += kBlock) d[k] = bd[k];
}
void launch_unpack_strings(hipStream_t s, const uint8_t* bodies, const int64_t* body_off, const int64_t* row_off, const int64_t* rows_of, const int64_t* byte_off,
                           int32_t nblocks, int32_t* sizes, uint8_t* bytes) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(k_unpack_strings, dim3((unsigned)nblocks), dim3(kBlock), 0, s, bodies, body_off, row_off, rows_of, byte_off, sizes, bytes);
}

}  // namespace dfdb
