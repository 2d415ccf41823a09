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

#include "k_interp_device.inc"

static DInstr pack_instr(const IInstr& i) {
  DInstr d{};
  d.w0 = (uint32_t)i.h | (uint32_t)i.bsrc << 8 | (uint32_t)i.cva << 16 | (uint32_t)i.cvb << 24;
  d.w1 = (uint32_t)i.flags | (uint32_t)i.cmp << 8 | (uint32_t)i.wsh << 16 | (uint32_t)i.wsg << 24;
  d.slot = i.slot; d.len = i.len; d.imm = i.imm;
  d.w2 = (uint32_t)i.ta | (uint32_t)i.tb << 8 | (uint32_t)i.rt << 16 | (uint32_t)i.so << 24;
  d.aslot = i.aslot; d.imm2 = i.imm2;
  return d;
}
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
    IColDesc d{};
    d.data = c.comp_only ? column_data(const_cast<dfdb_table*>(t), const_cast<Column&>(c)) : c.data.p;      // (compressed-only: a whole-column decode for this call)
    d.missing = c.missing.as<uint64_t>(); d.tile_off = (const int64_t*)c.tile_off.p; d.bytes = c.bytes.as<uint8_t>();
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
  // the second tier (jit.cpp): this program shape compiled by hipRTC from the interpreter's own source — used as soon as the compiler is done with it
  std::shared_ptr<JitKernel> jk;
  const int64_t jit_mode = ctx_option(ctx, "jit", 1);
  if (jit_mode > 0 && t->nrows >= ctx_option(ctx, "jit_min_rows", (int64_t)1 << 22)) {
    JitShape sh;
    sh.mode = mode; sh.str = c.prog.nstr > 0; sh.nul = c.nul; sh.and_existing = mode == 0 ? (and_existing ? 1 : 0) : 1; sh.stack_levels = c.max_sp;
    sh.result_dtype = c.prog.result_dtype; sh.nstr = c.prog.nstr;
    for (int i = 0; i < c.prog.n; i++) {
      const DInstr& d = c.prog.ins[i];
      sh.w0.push_back(d.w0); sh.w1.push_back(d.w1); sh.w2.push_back(d.w2); sh.slot.push_back(d.slot); sh.aslot.push_back(d.aslot);
    }
    for (int i = 0; i < c.prog.ncols; i++) sh.col_dtype.push_back(c.prog.cols[i].dtype);
    for (int i = 0; i < 4; i++) sh.str_slot[i] = c.prog.str_slot[i];
    jk = jit_request(ctx, sh, jit_mode >= 2);
  }
  bool launched = false;
  if (jk) {
    const IProgram* a_prog = (const IProgram*)db.p; uint64_t* a_bm = q->bitmap.as<uint64_t>(); uint32_t* a_tc = q->tile_counts.as<uint32_t>();
    const uint64_t* a_px = q->prefix.as<uint64_t>(); void* a_out = out; int64_t a_cap = cap, a_nrows = t->nrows, a_nt = ntiles;
    int a_ae = mode == 0 ? (and_existing ? 1 : 0) : 1, a_sl = c.max_sp; int* a_err = derr; uint8_t* a_om = out_missing;
    void* args[] = {&a_prog, &a_bm, &a_tc, &a_px, &a_out, &a_cap, &a_nrows, &a_nt, &a_ae, &a_err, &a_sl, &a_om};
    LaunchTimer lt(ctx, mode == 0 ? "jit_predicate" : "jit_project");
    launched = jit_launch(*jk, ctx, (unsigned)grid, lds_bytes, args);
  }
  if (!launched) {
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
