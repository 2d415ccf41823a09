// k_interp.hip — device interpreter for arbitrary IR expressions (the general form of K1 and of computed
// projection columns).  The reference JIT-fuses any Julia function over a block
// (src/tables/broadcast.jl:51-68,121-133); ahead-of-time HIP cannot, so a predicate or computed column
// that is not one of the specialised shapes (k_scan.hip, k_strings.hip) is compiled on the host to a small
// ACCUMULATOR program and run here.
//
// Machine: one lane = one row; a wave carries kW = 4 consecutive 64-row words (256 rows) through every
// instruction dispatch, so the wave-uniform decode + branch cost of an instruction is paid once per 256 rows
// and every column read is 4 independent coalesced loads per lane.  The accumulator A[kW] and the fetched
// operand B[kW] live in VGPRs; only sub-expression values that must outlive the evaluation of a sibling
// sub-tree are pushed to an LDS stack ([level][k][thread], conflict-free) whose depth the host computes, so
// the common `f(col, const) OP const` chains touch no LDS at all.  The host resolves every type decision
// (promotion, conversions, signedness, wrap width, which comparison kernel) into a handler id + flags; the
// device only switches on wave-uniform values.  The Boolean result leaves as a ballot word exactly like K1.
// Scalar semantics are Julia's: Int wraparound, exact Int-vs-Float comparison, `/` in floating point,
// rem/mod/div signs, DivideError / InexactError reported to the host.
//
// (Round-1 history: the first interpreter dispatched per 64 rows with an LDS register file: 193 SALU + 77
// VALU instructions per word, 0.73 TB/s.  profiles/r1_README.md has the counters.)
#include "device_utils.hpp"
#include "engine.hpp"

namespace dfdb {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int64_t kTile = 1024;
constexpr int kW = 4;            // 64-row words a wave carries through one instruction dispatch
constexpr int kGroups = 16 / kW;
constexpr int kMaxIns = 96;
constexpr int kMaxCols = 32;     // (a queue of fused predicate stages is ONE program: eight columns was reached by three ordinary stages)
constexpr int kMaxStr = 4;       // string columns whose bytes are read (need per-row byte offsets)
constexpr int kMaxLds = 8;       // stack levels + offset arrays: 8 KB of dynamic LDS each per workgroup

enum Handler : uint8_t {
  H_LOAD = 0,
  H_FADD, H_FSUB, H_FMUL, H_FDIV, H_FREM, H_FMOD, H_FIDIV, H_FMIN, H_FMAX, H_FNEG, H_FABS, H_FROUND32,
  H_IADD, H_ISUB, H_IMUL, H_IMIN_S, H_IMIN_U, H_IMAX_S, H_IMAX_U, H_IDIVOP, H_INEG, H_IABS, H_WRAP,
  H_BAND1, H_BOR1, H_AND, H_OR, H_XOR, H_NOT,
  H_CMP_FF, H_CMP_SS, H_CMP_UU, H_CMP_US, H_CMP_IF,
  H_INSET, H_CAST,
  // handlers that set the missing flag of their result themselves (everything above: flag = union of the operands' flags)
  H_STRCMP, H_STRPRE, H_STRSUF, H_ISMISS, H_AND3, H_OR3, H_COALESCE, H_ISMISSA
};
constexpr int kFirstOwnFlag = H_STRCMP;
enum BSrc : uint8_t { B_NONE = 0, B_IMM = 1, B_COL = 2, B_POP = 3 };
enum Cvt : uint8_t { CV_NONE = 0, CV_S2D, CV_U2D, CV_S2F, CV_U2F, CV_D2F };
enum : uint8_t { F_SWAP = 1,   // the fetched operand is the LEFT one: exchange A and B before the handler
                 F_PUSH = 4,   // A is live: push it before this instruction produces a fresh value
                 F_UNS = 8,    // unsigned compute type (integer division) / UInt64 integer side (H_CMP_IF)
                 F_FLIP = 16 };// string comparison with the constant on the left

struct IInstr {                // host-side form (the Compiler fills these fields)
  uint8_t h, bsrc, cva, cvb, flags, cmp /* DFIR_EQ.. or DFIR_IDIV/REM/MOD */, wsh /* 64 - bits of the integer result type */, wsg;
  int32_t slot;                // column slot (B_COL, string handlers, H_ISMISS)
  int32_t len;                 // pattern / set length
  uint64_t imm;                // immediate bits / pool offset
  uint8_t ta, tb, rt, so;      // generic handlers (H_CAST, H_INSET): operand/result dtypes; so: offset-array index of the string column
  int32_t aslot;               // 1 + column slot loaded into A before the operation (fused leaf load), 0 = A is the running value
  uint64_t imm2;               // type_min of the compute type (typemin ÷ -1 check)
};
// device form: the same fields packed into dwords so that the whole instruction arrives with ONE scalar load
// (sub-dword fields would be fetched with vector loads + s_waitcnt vmcnt(0): three exposed round trips per dispatch)
struct DInstr {
  uint32_t w0;                 // h | bsrc<<8 | cva<<16 | cvb<<24
  uint32_t w1;                 // flags | cmp<<8 | wsh<<16 | wsg<<24
  int32_t slot, len;
  uint64_t imm;
  uint32_t w2;                 // ta | tb<<8 | rt<<16 | so<<24
  int32_t aslot;
  uint64_t imm2;
};
static DInstr pack_instr(const IInstr& i) {
  DInstr d{};
  d.w0 = (uint32_t)i.h | (uint32_t)i.bsrc << 8 | (uint32_t)i.cva << 16 | (uint32_t)i.cvb << 24;
  d.w1 = (uint32_t)i.flags | (uint32_t)i.cmp << 8 | (uint32_t)i.wsh << 16 | (uint32_t)i.wsg << 24;
  d.slot = i.slot; d.len = i.len; d.imm = i.imm;
  d.w2 = (uint32_t)i.ta | (uint32_t)i.tb << 8 | (uint32_t)i.rt << 16 | (uint32_t)i.so << 24;
  d.aslot = i.aslot; d.imm2 = i.imm2;
  return d;
}
struct IColDesc {
  const void* data; const uint64_t* missing; const int64_t* tile_off; const uint8_t* bytes;
  int32_t dtype; int32_t wide;     // wide: an 8-byte column (Int64 / UInt64 / Float64): loaded as it is, no dtype switch
};
struct IProgram {
  int32_t n, ncols, result_dtype, nstr;
  int32_t nullable_result, pad0;     // the value is Union{T,Missing}: flags are written beside it (projection)
  const uint8_t* pool;               // string patterns and set elements
  int32_t str_slot[kMaxStr];         // column slots of the string columns that need byte offsets
  IColDesc cols[kMaxCols];
  DInstr ins[kMaxIns];
};

// ---------------------------------------------------------------- scalar helpers
__device__ __forceinline__ double bits_d(uint64_t x) { return __longlong_as_double((long long)x); }
__device__ __forceinline__ uint64_t d_bits(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ bool isf(int t) { return t == DFDB_F32 || t == DFDB_F64; }
__device__ __forceinline__ bool issigned(int t) { return t >= DFDB_I8 && t <= DFDB_I64; }

__device__ __forceinline__ int64_t wrap_to(int64_t x, int t) {
  switch (t) {
    case DFDB_I8: return (int8_t)x; case DFDB_I16: return (int16_t)x; case DFDB_I32: return (int32_t)x;
    case DFDB_U8: return (uint8_t)x; case DFDB_U16: return (uint16_t)x; case DFDB_U32: return (uint32_t)x;
    default: return x;
  }
}
__device__ __forceinline__ int64_t type_min(int t) {
  switch (t) { case DFDB_I8: return -128; case DFDB_I16: return -32768; case DFDB_I32: return -2147483648LL; case DFDB_I64: return INT64_MIN; }
  return 0;
}
// operand (64-bit register image of type t) as a float of compute type ct, one rounding from the source
__device__ __forceinline__ double as_float(uint64_t x, int t, int ct) {
  if (isf(t)) { const double d = bits_d(x); return ct == DFDB_F32 ? (double)(float)d : d; }
  if (ct == DFDB_F32) return t == DFDB_U64 ? (double)(float)x : (double)(float)(int64_t)x;
  return t == DFDB_U64 ? (double)x : (double)(int64_t)x;
}
// exact three-way comparisons; 2 = unordered
__device__ __forceinline__ int cmp_int_float(int64_t x, bool xu, double y) {
  if (y != y) return 2;
  if (xu) {
    if (y >= 18446744073709551616.0) return -1;
    if (y < 0.0) return 1;
    const uint64_t yi = (uint64_t)y; const uint64_t ux = (uint64_t)x;
    if (ux < yi) return -1;
    if (ux > yi) return 1;
    return (y - (double)yi) > 0.0 ? -1 : 0;
  }
  if (y >= 9223372036854775808.0) return -1;
  if (y < -9223372036854775808.0) return 1;
  const int64_t yi = (int64_t)y;
  if (x < yi) return -1;
  if (x > yi) return 1;
  const double fr = y - (double)yi;
  return fr > 0.0 ? -1 : (fr < 0.0 ? 1 : 0);
}
__device__ __forceinline__ int cmp3(uint64_t xa, int ta, uint64_t xb, int tb) {
  const bool fa = isf(ta), fb = isf(tb);
  if (fa && fb) { const double a = bits_d(xa), b = bits_d(xb); if (a != a || b != b) return 2; return a < b ? -1 : (a > b ? 1 : 0); }
  if (fa) { const int r = cmp_int_float((int64_t)xb, tb == DFDB_U64, bits_d(xa)); return r == 2 ? 2 : -r; }
  if (fb) return cmp_int_float((int64_t)xa, ta == DFDB_U64, bits_d(xb));
  const bool ua = ta == DFDB_U64, ub = tb == DFDB_U64;
  if (ua == ub) { if (ua) return xa < xb ? -1 : (xa > xb ? 1 : 0); const int64_t a = (int64_t)xa, b = (int64_t)xb; return a < b ? -1 : (a > b ? 1 : 0); }
  if (ua) { if ((int64_t)xb < 0) return 1; return xa < xb ? -1 : (xa > xb ? 1 : 0); }
  if ((int64_t)xa < 0) return -1;
  return xa < xb ? -1 : (xa > xb ? 1 : 0);
}
__device__ __forceinline__ bool cmp_result(int op, int c) {
  switch (op) {
    case DFIR_EQ: return c == 0; case DFIR_NE: return c != 0; case DFIR_LT: return c == -1;
    case DFIR_LE: return c == -1 || c == 0; case DFIR_GT: return c == 1; default: return c == 1 || c == 0;
  }
}
__device__ __forceinline__ double jl_fmin(double x, double y) { const double r = (x < y || (x == y && __builtin_signbit(x))) ? x : y; return (x != x || y != y) ? __builtin_nan("") : r; }
__device__ __forceinline__ double jl_fmax(double x, double y) { const double r = (x > y || (x == y && !__builtin_signbit(x))) ? x : y; return (x != x || y != y) ? __builtin_nan("") : r; }

__device__ __forceinline__ int str_cmp_dev(const uint8_t* a, int la, const uint8_t* b, int lb) {
  const int m = la < lb ? la : lb;
  for (int k = 0; k < m; k++) { const int d = (int)a[k] - (int)b[k]; if (d) return d < 0 ? -1 : 1; }
  return la < lb ? -1 : (la > lb ? 1 : 0);
}

// ---------------------------------------------------------------- out-of-line slow paths
// Everything with lane-divergent control flow (loops over bytes, 64-bit division, fmod, error reporting) is a real
// call: the dispatch loop then contains wave-uniform branches only, so LLVM's CFG structurizer leaves it alone and
// the handler switch stays a plain scalar compare-and-branch tree instead of a flag-driven state machine.
#define DFDB_SLOW __device__ __attribute__((noinline))
DFDB_SLOW double slow_frem(double a, double b) { return fmod(a, b); }
DFDB_SLOW double slow_fmod(double a, double b) {
  double v = fmod(a, b);
  if (v == 0.0) v = __builtin_copysign(v, b); else if ((v > 0.0) != (b > 0.0)) v += b;
  return v;
}
// err[0]: flags (1 DivideError, 2 InexactError); the two 64-bit words at err + 2: the SMALLEST row each kind happened on (the host decides from it
// whether the reference's block-by-block iteration would have reached that row at all: query.cpp error_is_reached)
__device__ __forceinline__ void flag_error(int* err, int code, uint64_t row) {
  atomicOr(err, code);
  atomicMin((unsigned long long*)(err + 2) + (code == 1 ? 0 : 1), (unsigned long long)row);
}
DFDB_SLOW double slow_fidiv(double a, double b) { return __builtin_rint((a - fmod(a, b)) / b); }
DFDB_SLOW uint64_t slow_idivop(int64_t a, int64_t b, int op, bool uns, int64_t tmin, bool alive, int* err, uint64_t row) {
  int64_t v = 0;
  if (b == 0) { if (alive) flag_error(err, 1, row); }
  else if (uns) v = op == DFIR_IDIV ? (int64_t)((uint64_t)a / (uint64_t)b) : (int64_t)((uint64_t)a % (uint64_t)b);
  else if (b == -1) { if (op == DFIR_IDIV) { if (a == tmin) { if (alive) flag_error(err, 1, row); } else v = -a; } }
  else if (op == DFIR_IDIV) v = a / b;
  else { v = a % b; if (op == DFIR_MOD && v != 0 && ((v < 0) != (b < 0))) v += b; }
  return (uint64_t)v;
}
DFDB_SLOW int slow_cmp_int_float(int64_t x, bool xu, double y) { return cmp_int_float(x, xu, y); }
DFDB_SLOW uint64_t slow_strop(int h_is_cmp, int suffix, const uint8_t* p, int len, const uint8_t* pat, int pl, int op, bool flip) {
  if (h_is_cmp) { const int sc = str_cmp_dev(p, len, pat, pl); return cmp_result(op, flip ? -sc : sc); }
  bool ok = len >= pl;
  if (ok) { if (suffix) p += len - pl; for (int i = 0; i < pl && ok; i++) ok = p[i] == pat[i]; }
  return ok;
}
DFDB_SLOW uint64_t slow_inset(uint64_t x, int ta, const uint64_t* set, int n, int tb) {
  bool hit = false;
  for (int i = 0; i < n && !hit; i++) hit = cmp3(x, ta, set[i], tb) == 0;
  return hit;
}
// DFIR_CAST = Julia's T(x) / convert(T, x): exact or InexactError (Int8(300), Int8(300.0), UInt64(-1), UInt64(-1.0), Int64(typemax(UInt64)),
// Bool(2) all throw; Float32(x) rounds).  The implicit promotions of arithmetic wrap instead (`a % T`, Base int.jl) and do not come here.
__device__ __forceinline__ bool int_fits(uint64_t x, bool src_unsigned, int rt) {
  int64_t lo, hi;                                   // [typemin, typemax] of the targets below 64 bits
  switch (rt) {
    case DFDB_I8: lo = -128; hi = 127; break;        case DFDB_I16: lo = -32768; hi = 32767; break;
    case DFDB_I32: lo = -2147483648LL; hi = 2147483647LL; break;
    case DFDB_U8: lo = 0; hi = 255; break;           case DFDB_U16: lo = 0; hi = 65535; break;
    case DFDB_U32: lo = 0; hi = 4294967295LL; break;
    case DFDB_I64: return !src_unsigned || (int64_t)x >= 0;      // a UInt64 above typemax(Int64)
    case DFDB_U64: return src_unsigned || (int64_t)x >= 0;       // a negative signed value
    default: return true;
  }
  if (src_unsigned) return x <= (uint64_t)hi;
  return (int64_t)x >= lo && (int64_t)x <= hi;
}
DFDB_SLOW uint64_t slow_cast(uint64_t xa, int ta, int rt, bool alive, int* err, uint64_t row) {
  if (isf(rt)) return d_bits(as_float(xa, ta, rt));
  if (isf(ta)) {   // Float -> Int / Bool: InexactError unless integral and inside the TARGET's range
    const double d = bits_d(xa);
    bool okr = d == __builtin_trunc(d);
    uint64_t v = 0;
    if (rt == DFDB_U64) { okr = okr && d >= 0.0 && d < 18446744073709551616.0; if (okr) v = (uint64_t)d; }
    else { okr = okr && d >= -9223372036854775808.0 && d < 9223372036854775808.0; if (okr) { v = (uint64_t)(int64_t)d; okr = rt == DFDB_BOOL || int_fits(v, false, rt); } }
    if (!okr && alive) flag_error(err, 2, row);
    if (!okr) v = 0;
    if (rt == DFDB_BOOL) { if (v > 1 && alive) flag_error(err, 2, row); return v != 0; }
    return (uint64_t)wrap_to((int64_t)v, rt);
  }
  if (rt == DFDB_BOOL) { if (xa > 1 && alive) flag_error(err, 2, row); return xa != 0; }
  if (ta != DFDB_BOOL && !int_fits(xa, ta >= DFDB_U8 && ta <= DFDB_U64, rt) && alive) flag_error(err, 2, row);
  return (uint64_t)wrap_to((int64_t)xa, rt);
}

// ---------------------------------------------------------------- the interpreter
#define EACH for (int k = 0; k < kW; k++)

// column reads: wave-uniform pointer to the tile's first row (SGPR pair) + a 32-bit byte offset per lane, so the
// load uses the saddr addressing mode and no 64-bit vector address arithmetic
template <typename T, bool FLT>
__device__ __forceinline__ void load_words(const void* data, int64_t base, const uint32_t (&idx)[kW], uint64_t (&B)[kW]) {
  // the pointer comes out of the program image (a generic pointer to the compiler): name the global address space,
  // otherwise these become flat_load with 64-bit vector addresses
  typedef const char __attribute__((address_space(1))) * gchar_p;
  typedef const T __attribute__((address_space(1))) * gT_p;
  gchar_p p = (gchar_p)((const T*)data + base);
#pragma unroll
  EACH {
    // keep the 32-bit offset arithmetic HERE (the empty asm stops the compiler from hoisting 64-bit copies of idx*1,
    // *2, *4, *8 out of the dispatch loop: 32 VGPRs) so that the load selects the SGPR-base + 32-bit-VGPR-offset form
    uint32_t o = idx[k];
    asm volatile("" : "+v"(o));
    const T v = __builtin_nontemporal_load((gT_p)(p + (uint32_t)(o * (uint32_t)sizeof(T))));
    if (FLT) B[k] = d_bits((double)v); else B[k] = (uint64_t)(int64_t)v;
  }
}
__device__ __forceinline__ void load_col(const IColDesc& c, int64_t base, const uint32_t (&idx)[kW], uint64_t (&B)[kW]) {
  if (c.wide) { load_words<uint64_t, false>(c.data, base, idx, B); return; }      // (one scalar test instead of a ten-way dtype tree)
  switch (c.dtype & DFDB_DTYPE_MASK) {
    case DFDB_I64: case DFDB_U64: case DFDB_F64: load_words<uint64_t, false>(c.data, base, idx, B); break;
    case DFDB_I32: case DFDB_STRING: load_words<int32_t, false>(c.data, base, idx, B); break;   // String: the size
    case DFDB_F32: load_words<float, true>(c.data, base, idx, B); break;
    case DFDB_U32: load_words<uint32_t, false>(c.data, base, idx, B); break;
    case DFDB_I16: load_words<int16_t, false>(c.data, base, idx, B); break;
    case DFDB_U16: load_words<uint16_t, false>(c.data, base, idx, B); break;
    case DFDB_I8:  load_words<int8_t, false>(c.data, base, idx, B); break;
    case DFDB_U8:  load_words<uint8_t, false>(c.data, base, idx, B); break;
    case DFDB_BOOL:
#pragma unroll
      EACH B[k] = ((const uint8_t*)c.data + base)[idx[k]] != 0;
      break;
  }
}
// missing flags of the rows just loaded from column c (1 = missing): one wave-uniform 64-bit word per 64 rows for a
// fixed-width column, the sign of the size for a String
__device__ __forceinline__ void load_missing(const IColDesc& c, int64_t base, int g, uint32_t lane, const uint64_t (&V)[kW], uint32_t (&M)[kW]) {
  if (!(c.dtype & DFDB_NULLABLE)) {
#pragma unroll
    EACH M[k] = 0;
  } else if ((c.dtype & DFDB_DTYPE_MASK) == DFDB_STRING) {
#pragma unroll
    EACH M[k] = (int64_t)V[k] < 0;
  } else {
#pragma unroll
    EACH { const uint64_t w = c.missing ? c.missing[(base >> 6) + g * kW + k] : 0ull; M[k] = (uint32_t)(w >> lane) & 1u; }
  }
}
__device__ __forceinline__ void convert(uint64_t (&X)[kW], int mode) {
  switch (mode) {
    case CV_S2D:
#pragma unroll
      EACH X[k] = d_bits((double)(int64_t)X[k]);
      break;
    case CV_U2D:
#pragma unroll
      EACH X[k] = d_bits((double)X[k]);
      break;
    case CV_S2F:
#pragma unroll
      EACH X[k] = d_bits((double)(float)(int64_t)X[k]);
      break;
    case CV_U2F:
#pragma unroll
      EACH X[k] = d_bits((double)(float)X[k]);
      break;
    case CV_D2F:
#pragma unroll
      EACH X[k] = d_bits((double)(float)bits_d(X[k]));
      break;
  }
}
// wrap a 64-bit image to the integer type of 64-sh bits (sh, sg wave-uniform)
__device__ __forceinline__ uint64_t wrapv(uint64_t x, int sh, bool sg) {
  return sg ? (uint64_t)((int64_t)(x << sh) >> sh) : ((x << sh) >> sh);
}
__device__ __forceinline__ bool cmp_pick(int op, bool lt, bool eq, bool un) {   // un: unordered (NaN)
  switch (op) {
    case DFIR_EQ: return eq; case DFIR_NE: return !eq; case DFIR_LT: return lt; case DFIR_LE: return lt || eq;
    case DFIR_GT: return !lt && !eq && !un; default: return !lt && !un;
  }
}

#define FLOAT_OP(EXPR)                                                        \
  {                                                                           \
    _Pragma("unroll") EACH {                                                  \
      const double a = bits_d(A[k]), b = bits_d(B[k]); double v; (void)b;     \
      EXPR;                                                                   \
      A[k] = d_bits(v);                                                       \
    }                                                                         \
  } break
#define INT_OP(EXPR)                                                          \
  {                                                                           \
    _Pragma("unroll") EACH {                                                  \
      uint64_t a = A[k], b = B[k], v;                                         \
      if (wsh) { a = wrapv(a, wsh, wsg); b = wrapv(b, wsh, wsg); }            \
      EXPR;                                                                   \
      A[k] = wsh ? wrapv(v, wsh, wsg) : v;                                    \
    }                                                                         \
  } break

template <int MODE, bool STR, bool NUL>   // MODE 0: predicate -> bitmap ; 1: computed column at the selected rows -> compacted output; NUL: Union{T,Missing} flags
__global__ __launch_bounds__(kBlock) void k_interp(const IProgram* __restrict__ prog, uint64_t* __restrict__ bitmap,
                                                   uint32_t* __restrict__ tile_counts, const uint64_t* __restrict__ prefix, void* __restrict__ out,
                                                   int64_t out_cap, int64_t nrows, int64_t ntiles, int and_existing, int* __restrict__ err,
                                                   int stack_levels, uint8_t* __restrict__ out_missing) {
  extern __shared__ uint64_t lds[];   // [stack level | offset array][k][thread]
  const int tid = threadIdx.x, lane = lane_id();
  // everything indexed by the tile is wave-uniform (SGPRs): say so, the compiler cannot see that tid>>6 is
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int nins = prog->n, rdt = prog->result_dtype, nstr = STR ? prog->nstr : 0;
  const uint8_t* pool = prog->pool;
  const bool masked = MODE == 1 || and_existing;
  uint32_t* ldsf = (uint32_t*)(lds + (size_t)(stack_levels + nstr) * kW * kBlock);   // NUL: missing flags of the pushed values
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * kTile;
    const uint32_t lastvalid = (uint32_t)(nrows - base < kTile ? nrows - base - 1 : kTile - 1);
    int64_t srun[kMaxStr];
#pragma unroll
    for (int s = 0; s < kMaxStr; s++) srun[s] = (STR && s < nstr) ? prog->cols[prog->str_slot[s]].tile_off[tile] : 0;
    uint32_t myword_lo = 0, myword_hi = 0, tile_cnt = 0;
    uint32_t run_sel = 0;
    // the tile's 16 incoming mask words: one coalesced load, handed out per word with v_readlane
    uint32_t tm_lo = ~0u, tm_hi = ~0u;
    if (masked && lane < 16) { const uint64_t w = bitmap[tile * 16 + lane]; tm_lo = (uint32_t)w; tm_hi = (uint32_t)(w >> 32); }
    for (int g = 0; g < kGroups; g++) {
      uint32_t idx[kW]; bool inb[kW]; uint64_t maskword[kW]; uint64_t anymask = 0;
#pragma unroll
      EACH {
        const uint32_t r = (uint32_t)((g * kW + k) * 64 + lane);
        inb[k] = r <= lastvalid;
        idx[k] = inb[k] ? r : lastvalid;
        maskword[k] = masked ? ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)tm_hi, g * kW + k) << 32 |
                                (uint32_t)__builtin_amdgcn_readlane((int)tm_lo, g * kW + k)) : ~0ull;
        anymask |= maskword[k];
      }
      // byte offsets of these rows in the string columns whose bytes are read (wave prefix sum of the sizes)
      if (STR) {
#pragma unroll
        for (int s = 0; s < kMaxStr; s++) {
          if (s < nstr) {
            const int32_t* sizes = (const int32_t*)prog->cols[prog->str_slot[s]].data;
#pragma unroll
            EACH {
              const int32_t sz = inb[k] ? (sizes + base)[idx[k]] : 0;
              const uint32_t c = sz > 0 ? (uint32_t)sz : 0u;
              const uint32_t incl = wave_incl_scan(c);
              lds[((stack_levels + s) * kW + k) * kBlock + tid] = (uint64_t)(srun[s] + (int64_t)(incl - c));
              srun[s] += (int64_t)__shfl(incl, 63, 64);
            }
          }
        }
      }
      if (masked && anymask == 0) {   // wave-uniform: nothing selected in these 256 rows
        continue;
      }
      uint64_t A[kW], B[kW];
      uint32_t Am[kW], Bm[kW];             // NUL: 1 = the value is missing
#pragma unroll
      EACH { A[k] = 0; B[k] = 0; Am[k] = 0; Bm[k] = 0; }
      int sp = 0;
      for (int pc = 0; pc < nins; pc++) {
        const DInstr& in = prog->ins[pc];
        const uint32_t w0 = in.w0, w1 = in.w1;
        const int in_h = w0 & 0xff, in_bsrc = (w0 >> 8) & 0xff, in_cva = (w0 >> 16) & 0xff, in_cvb = w0 >> 24;
        const int fl = w1 & 0xff, in_cmp = (w1 >> 8) & 0xff, wsh = (w1 >> 16) & 0xff; const bool wsg = (w1 >> 24) != 0;
        if (fl & F_PUSH) {
#pragma unroll
          EACH lds[(sp * kW + k) * kBlock + tid] = A[k];
          if (NUL) {
#pragma unroll
            EACH ldsf[(sp * kW + k) * kBlock + tid] = Am[k];
          }
          sp++;
        }
        if (in.aslot) {                                                   // fused leaf load: `col OP x` is one dispatch
          load_col(prog->cols[in.aslot - 1], base, idx, A);
          if (NUL) load_missing(prog->cols[in.aslot - 1], base, g, (uint32_t)lane, A, Am);
        }
        switch (in_bsrc) {
          case B_IMM: {
            const uint64_t v = in.imm;
#pragma unroll
            EACH { B[k] = v; Bm[k] = 0; }
          } break;
          case B_COL:
            load_col(prog->cols[in.slot], base, idx, B);
            if (NUL) load_missing(prog->cols[in.slot], base, g, (uint32_t)lane, B, Bm);
            break;
          case B_POP:
            sp--;
#pragma unroll
            EACH B[k] = lds[(sp * kW + k) * kBlock + tid];
            if (NUL) {
#pragma unroll
              EACH Bm[k] = ldsf[(sp * kW + k) * kBlock + tid];
            }
            break;
          default:
#pragma unroll
            EACH Bm[k] = 0;
            break;
        }
        if (fl & F_SWAP) {
#pragma unroll
          EACH { const uint64_t t = A[k]; A[k] = B[k]; B[k] = t; const uint32_t tm = Am[k]; Am[k] = Bm[k]; Bm[k] = tm; }
        }
        if (NUL && in_h != H_LOAD && in_h < kFirstOwnFlag) {              // Base methods propagate missing
#pragma unroll
          EACH Am[k] |= Bm[k];
        }
        if (w0 >> 16) {                                                     // (one test for the usual case: neither operand is converted)
          if (in_cva) convert(A, in_cva);
          if (in_cvb) convert(B, in_cvb);
        }
        // the handlers as a local function of the handler id: called with a CONSTANT id for the commonest ones (the switch folds to that one case), so that a compare or an
        // add is found after one or two scalar compares instead of the six levels of a 43-way compare tree — the dispatch loop is bound by the CU's one scalar unit
        if constexpr (NUL) {
          // (the kernels that carry missing flags are register-bound: wrapped in the local function below they lose a wave per SIMD and run 13-16 % slower)
          const int hh = in_h;
          switch (hh) {
#include "k_interp_handlers.inc"
          }
        } else {
        auto handler = [&](const int hh) __attribute__((always_inline)) {
        switch (hh) {
#include "k_interp_handlers.inc"
        }
        };
        if (in_h == H_CMP_SS) handler(H_CMP_SS);
        else if (in_h == H_CMP_FF) handler(H_CMP_FF);
        else if (in_h == H_IADD) handler(H_IADD);
        else if (in_h == H_IMUL) handler(H_IMUL);
        else if (in_h == H_ISUB) handler(H_ISUB);
        else if (in_h == H_FMUL) handler(H_FMUL);
        else if (in_h == H_FADD) handler(H_FADD);
        else if (in_h == H_AND) handler(H_AND);
        else if (in_h == H_OR) handler(H_OR);
        else handler(in_h);
        }
      }
#pragma unroll
      EACH {
        const int j = g * kW + k;
        if (MODE == 0) {
          uint64_t m = __ballot(inb[k] && (A[k] & 1ull));
          if (and_existing) m &= maskword[k];
          // m is wave-uniform: drop it into lane j of the tile's word vector, count it on the scalar unit
          myword_lo = write_lane(myword_lo, (uint32_t)m, j);
          myword_hi = write_lane(myword_hi, (uint32_t)(m >> 32), j);
          tile_cnt += (uint32_t)__popcll(m);
        } else {
          const uint32_t rank = (uint32_t)__popcll(maskword[k] & ((1ull << lane) - 1ull));
          const int64_t o = (int64_t)prefix[tile] + run_sel + rank;
          const bool alive = inb[k] && ((maskword[k] >> lane) & 1ull);
          const uint64_t res = A[k];
          if (alive && o < out_cap) {
            switch (rdt) {
              case DFDB_I8: case DFDB_U8: ((uint8_t*)out)[o] = (uint8_t)res; break;
              case DFDB_BOOL: ((uint8_t*)out)[o] = (uint8_t)(res & 1ull); break;
              case DFDB_I16: case DFDB_U16: ((uint16_t*)out)[o] = (uint16_t)res; break;
              case DFDB_I32: case DFDB_U32: ((uint32_t*)out)[o] = (uint32_t)res; break;
              case DFDB_F32: ((float*)out)[o] = (float)bits_d(res); break;
              default: ((uint64_t*)out)[o] = res; break;
            }
            if (NUL && out_missing) out_missing[o] = (uint8_t)Am[k];
          }
          run_sel += (uint32_t)__popcll(maskword[k]);
        }
      }
    }
    if (MODE == 0) {
      if (lane < 16) __hip_atomic_store(&bitmap[tile * 16 + lane], (uint64_t)myword_hi << 32 | myword_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // write-through: see k_scan_cmp
      if (lane == 0) tile_counts[tile] = tile_cnt;
    }
  }
}
#undef EACH
#undef CMP_CASES
#undef FLOAT_OP
#undef INT_OP

// ---------------------------------------------------------------- host: typed tree -> accumulator program
struct Compiler {
  const dfdb_table* t;
  IProgram prog{};
  std::vector<uint8_t> pool;
  std::vector<int> col_ord;   // slot -> table ordinal
  std::vector<IInstr> code;
  bool a_live = false;        // A holds a value that a later instruction still needs
  bool nul = false;           // some value of the program is Union{T,Missing}: the kernel carries missing flags
  int sp = 0, max_sp = 0;

  int slot_for(int ordinal) {
    for (size_t i = 0; i < col_ord.size(); i++) if (col_ord[i] == ordinal) return (int)i;
    if ((int)col_ord.size() >= kMaxCols) fail(DFDB_ERR_UNSUPPORTED, "expression references more than %d columns", kMaxCols);
    const Column& c = t->cols[(size_t)ordinal];
    if (!c.resident) fail(DFDB_ERR_ARGUMENT, "column %s is not resident on the device (dfdb_table_load it first)", c.name.c_str());
    IColDesc d{}; d.data = c.data.p; d.missing = c.missing.as<uint64_t>(); d.tile_off = (const int64_t*)c.tile_off.p; d.bytes = c.bytes.as<uint8_t>();
    d.dtype = c.dtype;
    { const int b = dt_base(c.dtype); d.wide = (b == DFDB_I64 || b == DFDB_U64 || b == DFDB_F64) ? 1 : 0; }
    prog.cols[col_ord.size()] = d;
    col_ord.push_back(ordinal);
    return (int)col_ord.size() - 1;
  }
  int offsets_for(int slot) {   // index of the per-row byte-offset array of a string column
    for (int i = 0; i < prog.nstr; i++) if (prog.str_slot[i] == slot) return i;
    if (prog.nstr >= kMaxStr) fail(DFDB_ERR_UNSUPPORTED, "expression reads the bytes of more than %d String columns", kMaxStr);
    prog.str_slot[prog.nstr] = slot;
    return prog.nstr++;
  }
  size_t pool_put(const void* p, size_t n) {
    while (pool.size() % 8) pool.push_back(0);
    const size_t off = pool.size();
    pool.insert(pool.end(), (const uint8_t*)p, (const uint8_t*)p + n);
    return off;
  }
  IInstr& ins(int h) {
    if ((int)code.size() >= kMaxIns) fail(DFDB_ERR_UNSUPPORTED, "expression too large for the device interpreter (%d instructions)", kMaxIns);
    code.emplace_back(); IInstr& in = code.back(); memset(&in, 0, sizeof in); in.h = (uint8_t)h; return in;
  }
  // an instruction that overwrites A without reading it: spill the live accumulator first
  IInstr& fresh(int h) {
    IInstr& in = ins(h);
    if (a_live) { in.flags |= F_PUSH; if (++sp > max_sp) max_sp = sp; }
    a_live = true;
    return in;
  }
  static uint64_t const_image(const Node& n) {   // 64-bit register image of a constant
    const int b = dt_base(n.dtype);
    if (b == DFDB_F32) { float f; memcpy(&f, &n.cbits, 4); double d = f; uint64_t u; memcpy(&u, &d, 8); return u; }
    if (b == DFDB_F64 || b == DFDB_I64 || b == DFDB_U64) return n.cbits;
    if (b == DFDB_BOOL) return n.cbits != 0;
    int64_t v = (int64_t)n.cbits;
    switch (b) { case DFDB_I8: v = (int8_t)v; break; case DFDB_I16: v = (int16_t)v; break; case DFDB_I32: v = (int32_t)v; break;
                 case DFDB_U8: v = (uint8_t)v; break; case DFDB_U16: v = (uint16_t)v; break; case DFDB_U32: v = (uint32_t)v; break; }
    return (uint64_t)v;
  }
  static bool hisf(int t) { return t == DFDB_F32 || t == DFDB_F64; }
  static bool hsigned(int t) { return t >= DFDB_I8 && t <= DFDB_I64; }
  // conversion of an operand image of type t to the float compute type ct (as_float on the device)
  static int cv_for(int t, int ct) {
    if (hisf(t)) return (ct == DFDB_F32 && t == DFDB_F64) ? CV_D2F : CV_NONE;
    if (ct == DFDB_F32) return t == DFDB_U64 ? CV_U2F : CV_S2F;
    return t == DFDB_U64 ? CV_U2D : CV_S2D;
  }
  static uint64_t host_convert(uint64_t x, int mode) {   // the same IEEE round-to-nearest conversions as the device's
    double d;
    switch (mode) {
      case CV_S2D: d = (double)(int64_t)x; break;
      case CV_U2D: d = (double)x; break;
      case CV_S2F: d = (double)(float)(int64_t)x; break;
      case CV_U2F: d = (double)(float)x; break;
      case CV_D2F: { double s; memcpy(&s, &x, 8); d = (double)(float)s; } break;
      default: return x;
    }
    uint64_t u; memcpy(&u, &d, 8); return u;
  }
  static void wrap_of(IInstr& in, int t) {   // result wrap of the integer type t
    int bits = 64; bool sg = hsigned(t);
    switch (t) { case DFDB_I8: case DFDB_U8: bits = 8; break; case DFDB_I16: case DFDB_U16: bits = 16; break;
                 case DFDB_I32: case DFDB_U32: bits = 32; break; case DFDB_BOOL: bits = 1; sg = false; break; }
    in.wsh = (uint8_t)(64 - bits); in.wsg = sg ? 1 : 0;
  }
  static int64_t type_min_of(int t) {
    switch (t) { case DFDB_I8: return -128; case DFDB_I16: return -32768; case DFDB_I32: return -2147483648LL; case DFDB_I64: return INT64_MIN; }
    return 0;
  }
  static int mirror(int op) {
    switch (op) { case DFIR_LT: return DFIR_GT; case DFIR_LE: return DFIR_GE; case DFIR_GT: return DFIR_LT; case DFIR_GE: return DFIR_LE; }
    return op;
  }
  static bool leaf(const Node& n) { return n.op == DFIR_CONST || (n.op == DFIR_COL && dt_base(n.dtype) != DFDB_STRING); }

  void load_leaf(const Node& n) {
    IInstr& in = fresh(H_LOAD);
    if (n.op == DFIR_COL) { in.bsrc = B_COL; in.slot = slot_for(n.col); }
    else { in.bsrc = B_IMM; in.imm = const_image(n); }
  }
  // operand fetch of a binary instruction: returns with A/B arranged so that after the optional swap A = left, B = right
  static bool col_leaf(const Node& n) { return n.op == DFIR_COL && dt_base(n.dtype) != DFDB_STRING; }
  // instruction h applied to the value of a: a column leaf is loaded by the instruction itself
  IInstr& op_on(const Node& a, int h) {
    if (col_leaf(a)) { IInstr& in = fresh(h); in.aslot = slot_for(a.col) + 1; return in; }
    eval(a);
    return ins(h);
  }
  void set_b_leaf(IInstr& in, const Node& l) {
    if (l.op == DFIR_COL) { in.bsrc = B_COL; in.slot = slot_for(l.col); } else { in.bsrc = B_IMM; in.imm = const_image(l); }
  }
  void fetch_operands(const Node& n, IInstr*& out, int h) {
    const Node &l = *n.a, &r = *n.b;
    if (l.op == DFIR_CONST && col_leaf(r)) {   // const OP col: A = the column (fused load), B = the constant, exchanged
      IInstr& in = fresh(h); in.aslot = slot_for(r.col) + 1;
      set_b_leaf(in, l);
      in.flags |= F_SWAP;
      out = &in;
    } else if (leaf(r)) {                // A = l (running value or fused column load), B = r
      IInstr& in = op_on(l, h);
      set_b_leaf(in, r);
      out = &in;
    } else if (leaf(l)) {                // A = r, B = l, exchanged on the device
      eval(r);
      IInstr& in = ins(h);
      set_b_leaf(in, l);
      in.flags |= F_SWAP;
      out = &in;
    } else {
      eval(l);
      eval(r);        // its first instruction pushes the value of l
      IInstr& in = ins(h);
      in.bsrc = B_POP; in.flags |= F_SWAP; sp--;
      out = &in;
    }
  }
  // conversions of (left, right) to the float compute type; a constant is converted here, once
  void set_conversions(IInstr& in, int ta, int tb, int ct) {
    const int ca = cv_for(ta, ct), cb = cv_for(tb, ct);
    if (in.bsrc == B_IMM) {
      const bool imm_is_left = (in.flags & F_SWAP) != 0;
      if (imm_is_left) { in.imm = host_convert(in.imm, ca); in.cvb = (uint8_t)cb; }
      else { in.imm = host_convert(in.imm, cb); in.cva = (uint8_t)ca; }
    } else { in.cva = (uint8_t)ca; in.cvb = (uint8_t)cb; }
  }

  // emit code that leaves the value of n in A
  void eval(const Node& n) {
    if (dt_nullable(n.dtype)) nul = true;
    if (leaf(n)) { load_leaf(n); return; }
    if (n.op == DFIR_COL) { IInstr& in = fresh(H_LOAD); in.bsrc = B_COL; in.slot = slot_for(n.col); return; }   // String column: its size
    if (n.op == DFIR_CONST_STR || n.op == DFIR_CONST_SET) fail(DFDB_ERR_UNSUPPORTED, "string/set constant in an unsupported position");
    const int rt = dt_base(n.dtype);
    const bool a_str = n.a && dt_base(n.a->dtype) == DFDB_STRING, b_str = n.b && dt_base(n.b->dtype) == DFDB_STRING;
    if (a_str || b_str) {   // string forms: column vs constant only
      if (n.op == DFIR_SIZEOF || n.op == DFIR_ISMISSING) {
        if (n.a->op != DFIR_COL) fail(DFDB_ERR_UNSUPPORTED, "sizeof/ismissing need a String column");
        eval(*n.a);                                    // the Int32 size; -1 = missing (the load also raises the missing flag)
        if (n.op == DFIR_SIZEOF) { IInstr& in = ins(H_IMAX_S); in.bsrc = B_IMM; in.imm = 0; }
        else { nul = true; ins(H_ISMISSA); }           // the flag becomes the (never missing) Bool value
        return;
      }
      const Node *cn = nullptr, *sn = nullptr; bool flipped = false;
      if (n.a->op == DFIR_COL && n.b->op == DFIR_CONST_STR) { cn = n.a.get(); sn = n.b.get(); }
      else if (n.a->op == DFIR_CONST_STR && n.b->op == DFIR_COL && n.op >= DFIR_EQ && n.op <= DFIR_GE) { cn = n.b.get(); sn = n.a.get(); flipped = true; }
      else fail(DFDB_ERR_UNSUPPORTED, "String expressions are limited to column-vs-constant comparisons, startswith, endswith, sizeof, ismissing");
      const int slot = slot_for(cn->col);
      const int so = offsets_for(slot);
      IInstr& in = fresh(n.op == DFIR_STARTSWITH ? H_STRPRE : (n.op == DFIR_ENDSWITH ? H_STRSUF : H_STRCMP));
      in.slot = slot; in.so = (uint8_t)so; in.cmp = (uint8_t)n.op; if (flipped) in.flags |= F_FLIP;
      in.len = (int32_t)sn->str.size(); in.imm = pool_put(sn->str.data(), sn->str.size());
      return;
    }
    if (n.op == DFIR_ISMISSING) {
      if (n.a->op == DFIR_COL) { IInstr& in = fresh(H_ISMISS); in.slot = slot_for(n.a->col); return; }
      nul = true;
      eval(*n.a);                                      // ismissing of a computed value: its flag becomes the value
      ins(H_ISMISSA);
      return;
    }
    if (n.op == DFIR_COALESCE) {
      IInstr* pin = nullptr; nul = true;
      fetch_operands(n, pin, H_COALESCE);
      return;
    }
    if (n.op == DFIR_IN_SET) {
      std::vector<uint64_t> vals;
      for (uint64_t v : n.b->set) { Node c; c.dtype = n.b->set_dtype; c.cbits = v; vals.push_back(const_image(c)); }
      IInstr& in = op_on(*n.a, H_INSET); in.ta = (uint8_t)dt_base(n.a->dtype); in.tb = (uint8_t)dt_base(n.b->set_dtype);
      in.len = (int32_t)vals.size(); in.imm = pool_put(vals.data(), vals.size() * 8);
      return;
    }
    const int ta = dt_base(n.a->dtype);
    if (!n.b) {   // unary
      if (n.op == DFIR_NOT) { op_on(*n.a, H_NOT); return; }
      if (n.op == DFIR_CAST) { IInstr& in = op_on(*n.a, H_CAST); in.ta = (uint8_t)ta; in.rt = (uint8_t)dt_base(n.cast_to); return; }
      if (n.op == DFIR_NEG || n.op == DFIR_ABS) {
        if (hisf(rt)) { IInstr& in = op_on(*n.a, n.op == DFIR_NEG ? H_FNEG : H_FABS); in.cva = (uint8_t)cv_for(ta, rt); return; }
        if (rt == DFDB_BOOL) { eval(*n.a); return; }
        IInstr& in = op_on(*n.a, n.op == DFIR_NEG ? H_INEG : (hsigned(rt) ? H_IABS : H_WRAP)); wrap_of(in, rt);
        return;
      }
      fail(DFDB_ERR_UNSUPPORTED, "unary operation 0x%x is not supported by the device interpreter", n.op);
    }
    const int tb = dt_base(n.b->dtype);
    IInstr* pin = nullptr;
    if ((n.op >= DFIR_ADD && n.op <= DFIR_MOD) || n.op == DFIR_MIN || n.op == DFIR_MAX) {
      int ct = rt;   // compute type: the promoted type; Float for `/`; Int for Bool ± Bool
      if (n.op == DFIR_DIV) { const int p = promote_num(n.a->dtype, n.b->dtype); ct = dt_isfloat(p) ? dt_base(p) : DFDB_F64; }
      else if ((n.op == DFIR_ADD || n.op == DFIR_SUB) && ta == DFDB_BOOL && tb == DFDB_BOOL) ct = DFDB_I64;
      if (hisf(ct)) {
        int h = H_FADD;
        switch (n.op) { case DFIR_ADD: h = H_FADD; break; case DFIR_SUB: h = H_FSUB; break; case DFIR_MUL: h = H_FMUL; break; case DFIR_DIV: h = H_FDIV; break;
                        case DFIR_REM: h = H_FREM; break; case DFIR_MOD: h = H_FMOD; break; case DFIR_IDIV: h = H_FIDIV; break; case DFIR_MIN: h = H_FMIN; break;
                        default: h = H_FMAX; break; }
        fetch_operands(n, pin, h);
        set_conversions(*pin, ta, tb, ct);
        if (ct == DFDB_F32) ins(H_FROUND32);
      } else if (ct == DFDB_BOOL) {
        fetch_operands(n, pin, (n.op == DFIR_MUL || n.op == DFIR_MIN) ? H_BAND1 : H_BOR1);
      } else {
        int h;
        switch (n.op) { case DFIR_ADD: h = H_IADD; break; case DFIR_SUB: h = H_ISUB; break; case DFIR_MUL: h = H_IMUL; break;
                        case DFIR_MIN: h = ct == DFDB_U64 ? H_IMIN_U : H_IMIN_S; break; case DFIR_MAX: h = ct == DFDB_U64 ? H_IMAX_U : H_IMAX_S; break;
                        default: h = H_IDIVOP; break; }
        fetch_operands(n, pin, h);
        wrap_of(*pin, ct);
        if (h == H_IDIVOP) { pin->cmp = (uint8_t)n.op; if (!hsigned(ct)) pin->flags |= F_UNS; pin->imm2 = (uint64_t)type_min_of(ct); }
      }
      return;
    }
    if (n.op >= DFIR_EQ && n.op <= DFIR_GE) {
      const bool fa = hisf(ta), fb = hisf(tb);
      int h, op = n.op; bool exchange = false, uns = false;
      if (fa && fb) h = H_CMP_FF;
      else if (fa) { h = H_CMP_IF; exchange = true; uns = tb == DFDB_U64; }        // float OP int  ==  int mirror(OP) float
      else if (fb) { h = H_CMP_IF; uns = ta == DFDB_U64; }
      else {
        const bool ua = ta == DFDB_U64, ub = tb == DFDB_U64;
        if (ua == ub) h = ua ? H_CMP_UU : H_CMP_SS;
        else { h = H_CMP_US; exchange = !ua; }
      }
      fetch_operands(n, pin, h);
      if (exchange) { pin->flags ^= F_SWAP; op = mirror(op); }
      pin->cmp = (uint8_t)op;
      if (uns) pin->flags |= F_UNS;
      return;
    }
    if (n.op >= DFIR_AND && n.op <= DFIR_XOR) {
      if (n.op != DFIR_XOR && rt == DFDB_BOOL && dt_nullable(n.dtype)) {   // Bool & / | over Union{Bool,Missing}: three-valued
        fetch_operands(n, pin, n.op == DFIR_AND ? H_AND3 : H_OR3);
        return;
      }
      fetch_operands(n, pin, n.op == DFIR_AND ? H_AND : (n.op == DFIR_OR ? H_OR : H_XOR));
      wrap_of(*pin, rt);
      return;
    }
    fail(DFDB_ERR_UNSUPPORTED, "operation 0x%x is not supported by the device interpreter", n.op);
  }

  void compile(const Node& root) {
    code.reserve(kMaxIns);   // IInstr& references handed out by ins() stay valid
    eval(root);
    prog.n = (int32_t)code.size();
    for (size_t i = 0; i < code.size(); i++) prog.ins[i] = pack_instr(code[i]);
    prog.ncols = (int32_t)col_ord.size();
    prog.result_dtype = dt_base(root.dtype);
    prog.nullable_result = dt_nullable(root.dtype) ? 1 : 0;
    if (prog.nullable_result) nul = true;
    for (int i = 0; i < prog.ncols; i++) if (dt_nullable(prog.cols[i].dtype) && !nul) { /* a nullable column read only through ismissing(col) */ }
    if (max_sp + prog.nstr > kMaxLds) fail(DFDB_ERR_UNSUPPORTED, "expression too deep for the device interpreter");
  }
};

static void run_interp(dfdb_query* q, const Node& root, int mode, bool and_existing, void* out, int64_t cap, uint8_t* out_missing) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  Compiler c; c.t = t; c.compile(root);
  // device copies: [IProgram][pool][err]
  const size_t pool_off = round_up((int64_t)sizeof(IProgram), 64), err_off = pool_off + (size_t)round_up((int64_t)c.pool.size() + 8, 64);
  DevBuf& db = q->tmp_a; db.ensure(err_off + 64);
  c.prog.pool = db.as<uint8_t>() + pool_off;
  std::vector<uint8_t> img(err_off + 64, 0);
  memcpy(img.data(), &c.prog, sizeof(IProgram));
  if (!c.pool.empty()) memcpy(img.data() + pool_off, c.pool.data(), c.pool.size());
  memset(img.data() + err_off + 8, 0xFF, 16);            // the smallest erroring rows: none yet
  HIP_CHECK(hipMemcpyAsync(db.p, img.data(), img.size(), hipMemcpyHostToDevice, s));
  stream_wait(q->t->ctx);
  const int64_t ntiles = ceil_div(t->nrows, kTile);
  if (ntiles == 0) return;
  int64_t grid = ceil_div(ntiles, kWavesPerBlock); if (grid > 16384) grid = 16384;
  int* derr = (int*)(db.as<uint8_t>() + err_off);
  const size_t lds_bytes = (size_t)(c.max_sp + c.prog.nstr) * kW * kBlock * sizeof(uint64_t) + (c.nul ? (size_t)c.max_sp * kW * kBlock * sizeof(uint32_t) : 0);
  {
    LaunchTimer lt(ctx, mode == 0 ? "interp_predicate" : "interp_project");
    const bool str = c.prog.nstr > 0;
#define DFDB_INTERP_LAUNCH(M, S, NL, AE)                                                                                                        \
    hipLaunchKernelGGL((k_interp<M, S, NL>), dim3((unsigned)grid), dim3(kBlock), lds_bytes, s, (const IProgram*)db.p, q->bitmap.as<uint64_t>(), \
                       q->tile_counts.as<uint32_t>(), q->prefix.as<uint64_t>(), out, cap, t->nrows, ntiles, AE, derr, c.max_sp, out_missing)
#define DFDB_INTERP_PICK(M, AE)                                                                                                     \
    do {                                                                                                                            \
      if (c.nul) { if (str) DFDB_INTERP_LAUNCH(M, true, true, AE); else DFDB_INTERP_LAUNCH(M, false, true, AE); }                  \
      else       { if (str) DFDB_INTERP_LAUNCH(M, true, false, AE); else DFDB_INTERP_LAUNCH(M, false, false, AE); }                \
    } while (0)
    if (mode == 0) DFDB_INTERP_PICK(0, and_existing ? 1 : 0); else DFDB_INTERP_PICK(1, 1);
#undef DFDB_INTERP_PICK
#undef DFDB_INTERP_LAUNCH
    HIP_CHECK(hipGetLastError());
  }
  struct { int flags, pad; uint64_t row[2]; } herr{0, 0, {~0ull, ~0ull}};
  HIP_CHECK(hipMemcpyAsync(&herr, derr, 24, hipMemcpyDeviceToHost, s));
  stream_wait(q->t->ctx);
  if (!herr.flags) return;
  if (mode == 0) {
    // a predicate: WHETHER the reference raises depends on whether its block-by-block iteration reaches the row (query_execute decides once every
    // stage has run: error_is_reached); the erroring rows count as not selected until then
    if (herr.flags & 1) q->err_row[0] = std::min(q->err_row[0], herr.row[0]);
    if (herr.flags & 2) q->err_row[1] = std::min(q->err_row[1], herr.row[1]);
    return;
  }
  if (q->proj_err) {          // one of several projection columns: query_materialize picks the error the reference's block-by-block, column-by-column order meets first
    if (herr.flags & 1) q->proj_err[0] = std::min(q->proj_err[0], herr.row[0]);
    if (herr.flags & 2) q->proj_err[1] = std::min(q->proj_err[1], herr.row[1]);
    return;
  }
  if ((herr.flags & 1) && (!(herr.flags & 2) || herr.row[0] <= herr.row[1])) fail(DFDB_ERR_DIVIDE, "DivideError: integer division error");
  fail(DFDB_ERR_ARGUMENT, "InexactError: conversion is not exact");
}

void run_interp_predicate(dfdb_query* q, const Node& pred, bool and_existing) { run_interp(q, pred, 0, and_existing, nullptr, 0, nullptr); }
// missing_dst: one byte per selected row (1 = missing) when the expression is Union{T,Missing}; may be null
void run_interp_project(dfdb_query* q, const Node& expr, void* dst, int64_t cap, uint8_t* missing_dst) { run_interp(q, expr, 1, true, dst, cap, missing_dst); }

}  // namespace dfdb
