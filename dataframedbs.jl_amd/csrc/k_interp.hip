// k_interp.hip — device interpreter for arbitrary IR expressions (the general form of K1 and of computed
// projection columns).  The reference JIT-fuses any Julia function over a block
// (src/tables/broadcast.jl:51-68,121-133); ahead-of-time HIP cannot, so a predicate or computed column
// that is not one of the specialised shapes (k_scan.hip, k_strings.hip) is compiled on the host to a small
// register program and run here: one lane = one row, a wave walks a 1024-row tile 64 rows at a time, the
// virtual registers live in LDS ([reg][thread], conflict-free), every instruction is dispatched with
// wave-uniform branches, and the Boolean result leaves as a ballot word exactly like K1.  Column reads stay
// fully coalesced (lane l reads row base+l).  Scalar semantics are Julia's: Int wraparound, exact
// Int-vs-Float comparison, `/` in floating point, rem/mod/div signs, DivideError reported to the host.
#include "device_utils.hpp"
#include "engine.hpp"

namespace dfdb {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int64_t kTile = 1024;
constexpr int kMaxRegs = 16;
constexpr int kMaxIns = 64;
constexpr int kMaxCols = 8;
constexpr int kOpMov = 0xF0;

enum OperandKind : uint8_t { K_REG = 0, K_COL = 1, K_IMM = 2 };

struct IInstr {
  uint8_t op, rt, ta, tb, ka, kb, dst, aux;
  int32_t a, b;          // register index / column slot / (b: pattern or set length)
  uint64_t imm_a, imm_b; // immediate bits / pool offset
};
struct IColDesc {
  const void* data; const uint64_t* missing; const int64_t* tile_off; const uint8_t* bytes;
  int32_t dtype; int32_t need_off;   // need_off: a byte-reading string op uses this column
};
struct IProgram {
  int32_t n, ncols, result_dtype, pad;
  const uint8_t* pool;               // string patterns and set elements
  IColDesc cols[kMaxCols];
  IInstr ins[kMaxIns];
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
__device__ __forceinline__ double jl_fmin(double x, double y) { if (x != x || y != y) return __builtin_nan(""); return (x < y || (x == y && __builtin_signbit(x))) ? x : y; }
__device__ __forceinline__ double jl_fmax(double x, double y) { if (x != x || y != y) return __builtin_nan(""); return (x > y || (x == y && !__builtin_signbit(x))) ? x : y; }

__device__ __forceinline__ uint64_t load_col(const IColDesc& c, int64_t row) {
  switch (c.dtype & DFDB_DTYPE_MASK) {
    case DFDB_I8:  return (uint64_t)(int64_t)((const int8_t*)c.data)[row];
    case DFDB_I16: return (uint64_t)(int64_t)((const int16_t*)c.data)[row];
    case DFDB_I32: return (uint64_t)(int64_t)((const int32_t*)c.data)[row];
    case DFDB_I64: case DFDB_U64: return ((const uint64_t*)c.data)[row];
    case DFDB_U8:  return ((const uint8_t*)c.data)[row];
    case DFDB_U16: return ((const uint16_t*)c.data)[row];
    case DFDB_U32: return ((const uint32_t*)c.data)[row];
    case DFDB_F32: return d_bits((double)((const float*)c.data)[row]);
    case DFDB_F64: return ((const uint64_t*)c.data)[row];
    case DFDB_BOOL: return ((const uint8_t*)c.data)[row] != 0;
    case DFDB_STRING: return (uint64_t)(int64_t)((const int32_t*)c.data)[row];   // the size
  }
  return 0;
}

__device__ __forceinline__ int str_cmp_dev(const uint8_t* a, int la, const uint8_t* b, int lb) {
  const int m = la < lb ? la : lb;
  for (int k = 0; k < m; k++) { const int d = (int)a[k] - (int)b[k]; if (d) return d < 0 ? -1 : 1; }
  return la < lb ? -1 : (la > lb ? 1 : 0);
}

// ---------------------------------------------------------------- the interpreter
template <int MODE>   // 0: predicate -> bitmap ; 1: computed column at the selected rows -> compacted output
__global__ __launch_bounds__(kBlock) void k_interp(const IProgram* __restrict__ prog, uint64_t* __restrict__ bitmap,
                                                   uint32_t* __restrict__ tile_counts, const uint64_t* __restrict__ prefix, void* __restrict__ out,
                                                   int64_t out_cap, int64_t nrows, int64_t ntiles, int and_existing, int* __restrict__ err) {
  __shared__ uint64_t R[kMaxRegs][kBlock];
  __shared__ int64_t SO[kMaxCols][kBlock];   // per-row byte offsets of the string columns (dynamic slot index -> LDS, not scratch)
  const int tid = threadIdx.x, lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (tid >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int nins = prog->n, ncols = prog->ncols, rdt = prog->result_dtype;
  const uint8_t* pool = prog->pool;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * kTile;
    int64_t srun[kMaxCols];
#pragma unroll
    for (int s = 0; s < kMaxCols; s++) srun[s] = (s < ncols && prog->cols[s].need_off) ? prog->cols[s].tile_off[tile] : 0;
    uint64_t myword = 0;
    uint32_t run_sel = 0;
    for (int j = 0; j < 16; j++) {
      const int64_t row = base + j * 64 + lane;
      const bool inb = row < nrows;
      const int64_t lrow = inb ? row : nrows - 1;
      const uint64_t maskword = (MODE == 1 || and_existing) ? bitmap[tile * 16 + j] : ~0ull;
      const bool alive = inb && ((maskword >> lane) & 1ull);
      // byte offsets of this row in the string columns that need them (wave prefix-sum of the sizes)
#pragma unroll
      for (int s = 0; s < kMaxCols; s++) {
        if (s < ncols && prog->cols[s].need_off) {
          const int32_t sz = inb ? ((const int32_t*)prog->cols[s].data)[row] : 0;
          const uint32_t c = sz > 0 ? (uint32_t)sz : 0u;
          const uint32_t incl = wave_incl_scan(c);
          SO[s][tid] = srun[s] + (int64_t)(incl - c);
          srun[s] += (int64_t)__shfl(incl, 63, 64);
        }
      }
      if (MODE == 1 && maskword == 0) continue;   // wave-uniform
      for (int pc = 0; pc < nins; pc++) {
        const IInstr& in = prog->ins[pc];
        const int op = in.op, ta = in.ta, tb = in.tb, rt = in.rt;
        uint64_t xa = in.ka == K_REG ? R[in.a][tid] : (in.ka == K_COL ? load_col(prog->cols[in.a], lrow) : in.imm_a);
        uint64_t xb = 0;
        const bool binary_val = in.kb != 0xff;
        if (binary_val) xb = in.kb == K_REG ? R[in.b][tid] : (in.kb == K_COL ? load_col(prog->cols[in.b], lrow) : in.imm_b);
        uint64_t r = 0;
        if (op == kOpMov) r = xa;
        else if (op >= DFIR_EQ && op <= DFIR_GE) {
          if (ta == DFDB_STRING) {   // string column (slot in.a) vs pattern at pool+imm_b, length in.b
            const IColDesc& c = prog->cols[in.a];
            const int len = (int64_t)xa > 0 ? (int)xa : 0;
            const int sc = str_cmp_dev(c.bytes + SO[in.a][tid], len, pool + in.imm_b, in.b);
            r = cmp_result(op, in.aux ? -sc : sc);   // aux: the constant was the left operand
          } else r = cmp_result(op, cmp3(xa, ta, xb, tb));
        } else if (op == DFIR_STARTSWITH || op == DFIR_ENDSWITH) {
          const IColDesc& c = prog->cols[in.a];
          const int len = (int64_t)xa > 0 ? (int)xa : 0, pl = in.b;
          bool ok = len >= pl;
          if (ok) { const uint8_t* p = c.bytes + SO[in.a][tid] + (op == DFIR_ENDSWITH ? len - pl : 0); for (int k = 0; k < pl && ok; k++) ok = p[k] == pool[in.imm_b + k]; }
          r = ok;
        } else if (op == DFIR_SIZEOF) r = (int64_t)xa > 0 ? xa : 0;
        else if (op == DFIR_ISMISSING) {
          const IColDesc& c = prog->cols[in.a];
          if ((c.dtype & DFDB_DTYPE_MASK) == DFDB_STRING) r = (int64_t)xa < 0;
          else r = c.missing ? ((c.missing[lrow >> 6] >> (lrow & 63)) & 1ull) : 0;
        } else if (op == DFIR_NOT) r = !(xa & 1ull);
        else if (op == DFIR_IN_SET) {
          const uint64_t* set = (const uint64_t*)(pool + in.imm_b);
          bool hit = false;
          for (int k = 0; k < in.b && !hit; k++) hit = cmp3(xa, ta, set[k], tb) == 0;
          r = hit;
        } else if (op == DFIR_CAST) {
          if (isf(rt)) r = d_bits(as_float(xa, ta, rt));
          else if (isf(ta)) {   // Float -> Int / Bool: InexactError unless integral and in range
            const double d = bits_d(xa);
            const bool okr = d == __builtin_trunc(d) && d >= -9223372036854775808.0 && d < 9223372036854775808.0;
            if (!okr && alive) atomicOr(err, 2);
            const int64_t v = okr ? (int64_t)d : 0;
            if (rt == DFDB_BOOL) { if (v != 0 && v != 1 && alive) atomicOr(err, 2); r = v != 0; } else r = (uint64_t)wrap_to(v, rt);
          } else if (rt == DFDB_BOOL) { if (xa > 1 && alive) atomicOr(err, 2); r = xa != 0; }
          else r = (uint64_t)wrap_to((int64_t)xa, rt);
        } else if (op == DFIR_NEG || op == DFIR_ABS) {
          if (isf(rt)) { const double d = as_float(xa, ta, rt); r = d_bits(op == DFIR_NEG ? -d : __builtin_fabs(d)); }
          else if (rt == DFDB_BOOL) r = xa;
          else { int64_t v = (int64_t)xa; if (op == DFIR_NEG || (issigned(rt) && v < 0)) v = (int64_t)(0 - (uint64_t)v); r = (uint64_t)wrap_to(v, rt); }
        } else if ((op >= DFIR_AND && op <= DFIR_XOR)) {
          const uint64_t v = op == DFIR_AND ? (xa & xb) : (op == DFIR_OR ? (xa | xb) : (xa ^ xb));
          r = rt == DFDB_BOOL ? (v & 1ull) : (uint64_t)wrap_to((int64_t)v, rt);
        } else {   // arithmetic: ADD SUB MUL DIV IDIV REM MOD MIN MAX
          const int ct = in.aux;   // compute type chosen on the host (promotion; Float for `/`)
          if (isf(ct)) {
            const double a = as_float(xa, ta, ct), b = as_float(xb, tb, ct);
            double v;
            switch (op) {
              case DFIR_ADD: v = a + b; break; case DFIR_SUB: v = a - b; break; case DFIR_MUL: v = a * b; break;
              case DFIR_DIV: v = a / b; break;
              case DFIR_REM: v = fmod(a, b); break;
              case DFIR_MOD: { v = fmod(a, b); if (v == 0.0) v = __builtin_copysign(v, b); else if ((v > 0.0) != (b > 0.0)) v += b; break; }
              case DFIR_IDIV: v = __builtin_rint((a - fmod(a, b)) / b); break;
              case DFIR_MIN: v = jl_fmin(a, b); break;
              default: v = jl_fmax(a, b); break;
            }
            if (ct == DFDB_F32) v = (double)(float)v;
            r = d_bits(v);
          } else if (ct == DFDB_BOOL) {   // Bool*Bool, min/max on Bool
            r = (op == DFIR_MUL || op == DFIR_MIN) ? (xa & xb & 1ull) : ((xa | xb) & 1ull);
          } else {
            const int64_t a = wrap_to((int64_t)xa, ct), b = wrap_to((int64_t)xb, ct);
            const bool uns = !issigned(ct);
            int64_t v = 0;
            switch (op) {
              case DFIR_ADD: v = (int64_t)((uint64_t)a + (uint64_t)b); break;
              case DFIR_SUB: v = (int64_t)((uint64_t)a - (uint64_t)b); break;
              case DFIR_MUL: v = (int64_t)((uint64_t)a * (uint64_t)b); break;
              case DFIR_MIN: v = (uns && ct == DFDB_U64) ? ((uint64_t)a < (uint64_t)b ? a : b) : (a < b ? a : b); break;
              case DFIR_MAX: v = (uns && ct == DFDB_U64) ? ((uint64_t)a > (uint64_t)b ? a : b) : (a > b ? a : b); break;
              default:   // IDIV REM MOD
                if (b == 0) { if (alive) atomicOr(err, 1); v = 0; }
                else if (uns) v = op == DFIR_IDIV ? (int64_t)((uint64_t)a / (uint64_t)b) : (int64_t)((uint64_t)a % (uint64_t)b);
                else if (b == -1) { if (op == DFIR_IDIV) { if (a == type_min(ct)) { if (alive) atomicOr(err, 1); v = 0; } else v = -a; } else v = 0; }
                else if (op == DFIR_IDIV) v = a / b;
                else { v = a % b; if (op == DFIR_MOD && v != 0 && ((v < 0) != (b < 0))) v += b; }
            }
            r = (uint64_t)wrap_to(v, ct);
          }
        }
        R[in.dst][tid] = r;
      }
      const uint64_t res = R[0][tid];
      if (MODE == 0) {
        uint64_t m = __ballot(inb && (res & 1ull));
        if (and_existing) m &= maskword;
        if (lane == j) myword = m;
      } else {
        const uint32_t rank = (uint32_t)__popcll(maskword & ((1ull << lane) - 1ull));
        const int64_t o = (int64_t)prefix[tile] + run_sel + rank;
        if (alive && o < out_cap) {
          switch (rdt) {
            case DFDB_I8: case DFDB_U8: ((uint8_t*)out)[o] = (uint8_t)res; break;
            case DFDB_BOOL: ((uint8_t*)out)[o] = (uint8_t)(res & 1ull); break;
            case DFDB_I16: case DFDB_U16: ((uint16_t*)out)[o] = (uint16_t)res; break;
            case DFDB_I32: case DFDB_U32: ((uint32_t*)out)[o] = (uint32_t)res; break;
            case DFDB_F32: ((float*)out)[o] = (float)bits_d(res); break;
            default: ((uint64_t*)out)[o] = res; break;
          }
        }
        run_sel += (uint32_t)__popcll(maskword);
      }
    }
    if (MODE == 0) {
      uint32_t cnt = lane < 16 ? (uint32_t)__popcll(myword) : 0u;
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
      if (lane < 16) bitmap[tile * 16 + lane] = myword;
      if (lane == 0) tile_counts[tile] = cnt;
    }
  }
}

// ---------------------------------------------------------------- host: tree -> register program
struct Operand { uint8_t kind; int32_t idx; uint64_t imm; int32_t dtype; };

struct Compiler {
  const dfdb_table* t;
  IProgram prog{};
  std::vector<uint8_t> pool;
  std::vector<int> col_ord;   // slot -> table ordinal

  int slot_for(int ordinal) {
    for (size_t i = 0; i < col_ord.size(); i++) if (col_ord[i] == ordinal) return (int)i;
    if ((int)col_ord.size() >= kMaxCols) fail(DFDB_ERR_UNSUPPORTED, "expression references more than %d columns", kMaxCols);
    const Column& c = t->cols[(size_t)ordinal];
    if (!c.resident) fail(DFDB_ERR_ARGUMENT, "column %s is not resident on the device (dfdb_table_load it first)", c.name.c_str());
    IColDesc d{}; d.data = c.data.p; d.missing = c.missing.as<uint64_t>(); d.tile_off = (const int64_t*)c.tile_off.p; d.bytes = c.bytes.as<uint8_t>();
    d.dtype = c.dtype; d.need_off = 0;
    prog.cols[col_ord.size()] = d;
    col_ord.push_back(ordinal);
    return (int)col_ord.size() - 1;
  }
  size_t pool_put(const void* p, size_t n) {
    while (pool.size() % 8) pool.push_back(0);
    const size_t off = pool.size();
    pool.insert(pool.end(), (const uint8_t*)p, (const uint8_t*)p + n);
    return off;
  }
  IInstr& push() {
    if (prog.n >= kMaxIns) fail(DFDB_ERR_UNSUPPORTED, "expression too large for the device interpreter (%d instructions)", kMaxIns);
    IInstr& in = prog.ins[prog.n++]; memset(&in, 0, sizeof in); in.kb = 0xff; return in;
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
  void set_a(IInstr& in, const Operand& o) { in.ka = o.kind; in.a = o.idx; in.imm_a = o.imm; in.ta = (uint8_t)dt_base(o.dtype); }
  void set_b(IInstr& in, const Operand& o) { in.kb = o.kind; in.b = o.idx; in.imm_b = o.imm; in.tb = (uint8_t)dt_base(o.dtype); }

  Operand emit(const Node& n, int depth) {
    if (depth >= kMaxRegs) fail(DFDB_ERR_UNSUPPORTED, "expression too deep for the device interpreter");
    if (n.op == DFIR_COL) return Operand{K_COL, slot_for(n.col), 0, n.dtype};
    if (n.op == DFIR_CONST) return Operand{K_IMM, 0, const_image(n), n.dtype};
    if (n.op == DFIR_CONST_STR || n.op == DFIR_CONST_SET) fail(DFDB_ERR_UNSUPPORTED, "string/set constant in an unsupported position");
    const int rt = dt_base(n.dtype);
    // string forms: column vs constant only
    const bool a_str = n.a && dt_base(n.a->dtype) == DFDB_STRING, b_str = n.b && dt_base(n.b->dtype) == DFDB_STRING;
    if (a_str || b_str) {
      if (n.op == DFIR_SIZEOF || n.op == DFIR_ISMISSING) {
        if (n.a->op != DFIR_COL) fail(DFDB_ERR_UNSUPPORTED, "sizeof/ismissing need a String column");
        IInstr& in = push(); in.op = (uint8_t)n.op; in.rt = (uint8_t)rt; in.dst = (uint8_t)depth;
        set_a(in, Operand{K_COL, slot_for(n.a->col), 0, n.a->dtype});
        return Operand{K_REG, depth, 0, n.dtype};
      }
      const Node *cn = nullptr, *sn = nullptr; bool flipped = false;
      if (n.a->op == DFIR_COL && n.b->op == DFIR_CONST_STR) { cn = n.a.get(); sn = n.b.get(); }
      else if (n.a->op == DFIR_CONST_STR && n.b->op == DFIR_COL && n.op >= DFIR_EQ && n.op <= DFIR_GE) { cn = n.b.get(); sn = n.a.get(); flipped = true; }
      else fail(DFDB_ERR_UNSUPPORTED, "String expressions are limited to column-vs-constant comparisons, startswith, endswith, sizeof, ismissing");
      const int slot = slot_for(cn->col);
      prog.cols[slot].need_off = 1;
      IInstr& in = push(); in.op = (uint8_t)n.op; in.rt = (uint8_t)rt; in.dst = (uint8_t)depth; in.aux = flipped ? 1 : 0;
      set_a(in, Operand{K_COL, slot, 0, cn->dtype});
      in.kb = K_IMM; in.tb = DFDB_STRING; in.b = (int32_t)sn->str.size(); in.imm_b = pool_put(sn->str.data(), sn->str.size());
      return Operand{K_REG, depth, 0, n.dtype};
    }
    if (n.op == DFIR_ISMISSING) {
      if (n.a->op != DFIR_COL) fail(DFDB_ERR_UNSUPPORTED, "ismissing needs a column");
      IInstr& in = push(); in.op = DFIR_ISMISSING; in.rt = DFDB_BOOL; in.dst = (uint8_t)depth;
      set_a(in, Operand{K_COL, slot_for(n.a->col), 0, n.a->dtype});
      return Operand{K_REG, depth, 0, n.dtype};
    }
    if (n.op == DFIR_IN_SET) {
      const Operand oa = emit(*n.a, depth);
      std::vector<uint64_t> vals;
      for (uint64_t v : n.b->set) { Node c; c.dtype = n.b->set_dtype; c.cbits = v; vals.push_back(const_image(c)); }
      IInstr& in = push(); in.op = DFIR_IN_SET; in.rt = DFDB_BOOL; in.dst = (uint8_t)depth; set_a(in, oa);
      in.kb = K_IMM; in.tb = (uint8_t)dt_base(n.b->set_dtype); in.b = (int32_t)vals.size(); in.imm_b = pool_put(vals.data(), vals.size() * 8);
      return Operand{K_REG, depth, 0, n.dtype};
    }
    const Operand oa = emit(*n.a, depth);
    Operand ob{}; const bool binary = (bool)n.b;
    if (binary) ob = emit(*n.b, depth + (oa.kind == K_REG ? 1 : 0));
    IInstr& in = push(); in.op = (uint8_t)n.op; in.rt = (uint8_t)rt; in.dst = (uint8_t)depth;
    set_a(in, oa);
    if (binary) set_b(in, ob);
    if (n.op == DFIR_CAST) in.aux = (uint8_t)dt_base(n.cast_to);
    if ((n.op >= DFIR_ADD && n.op <= DFIR_MOD) || n.op == DFIR_MIN || n.op == DFIR_MAX) {
      int ct = rt;
      if (n.op == DFIR_DIV) { const int p = promote_num(oa.dtype, ob.dtype); ct = dt_isfloat(p) ? p : DFDB_F64; }
      else if ((n.op == DFIR_ADD || n.op == DFIR_SUB) && dt_base(oa.dtype) == DFDB_BOOL && dt_base(ob.dtype) == DFDB_BOOL) ct = DFDB_I64;
      in.aux = (uint8_t)ct;
    }
    return Operand{K_REG, depth, 0, n.dtype};
  }

  void compile(const Node& root) {
    const Operand r = emit(root, 0);
    if (r.kind != K_REG) { IInstr& in = push(); in.op = (uint8_t)kOpMov; in.rt = (uint8_t)dt_base(root.dtype); in.dst = 0; set_a(in, r); }
    prog.ncols = (int32_t)col_ord.size();
    prog.result_dtype = dt_base(root.dtype);
  }
};

static void run_interp(dfdb_query* q, const Node& root, int mode, bool and_existing, void* out, int64_t cap) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  Compiler c; c.t = t; c.compile(root);
  // device copies: [IProgram][pool][err]
  const size_t pool_off = round_up((int64_t)sizeof(IProgram), 64), err_off = pool_off + (size_t)round_up((int64_t)c.pool.size() + 8, 64);
  DevBuf& db = q->tmp_a; db.ensure(err_off + 64);
  c.prog.pool = db.as<uint8_t>() + pool_off;
  std::vector<uint8_t> img(err_off + 64, 0);
  memcpy(img.data(), &c.prog, sizeof(IProgram));
  if (!c.pool.empty()) memcpy(img.data() + pool_off, c.pool.data(), c.pool.size());
  HIP_CHECK(hipMemcpyAsync(db.p, img.data(), img.size(), hipMemcpyHostToDevice, s));
  stream_wait(q->t->ctx);
  const int64_t ntiles = ceil_div(t->nrows, kTile);
  if (ntiles == 0) return;
  int64_t grid = ceil_div(ntiles, kWavesPerBlock); if (grid > 2048) grid = 2048;
  int* derr = (int*)(db.as<uint8_t>() + err_off);
  {
    LaunchTimer lt(ctx, mode == 0 ? "interp_predicate" : "interp_project");
    if (mode == 0) hipLaunchKernelGGL((k_interp<0>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const IProgram*)db.p, q->bitmap.as<uint64_t>(),
                                      q->tile_counts.as<uint32_t>(), q->prefix.as<uint64_t>(), out, cap, t->nrows, ntiles, and_existing ? 1 : 0, derr);
    else hipLaunchKernelGGL((k_interp<1>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const IProgram*)db.p, q->bitmap.as<uint64_t>(),
                            q->tile_counts.as<uint32_t>(), q->prefix.as<uint64_t>(), out, cap, t->nrows, ntiles, 1, derr);
  }
  int herr = 0;
  HIP_CHECK(hipMemcpyAsync(&herr, derr, 4, hipMemcpyDeviceToHost, s));
  stream_wait(q->t->ctx);
  if (herr & 1) fail(DFDB_ERR_DIVIDE, "DivideError: integer division error");
  if (herr & 2) fail(DFDB_ERR_ARGUMENT, "InexactError: conversion is not exact");
}

void run_interp_predicate(dfdb_query* q, const Node& pred, bool and_existing) { run_interp(q, pred, 0, and_existing, nullptr, 0); }
void run_interp_project(dfdb_query* q, const Node& expr, void* dst, int64_t cap) { run_interp(q, expr, 1, true, dst, cap); }

}  // namespace dfdb
