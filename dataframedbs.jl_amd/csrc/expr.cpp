// expr.cpp — IR parsing, Julia result typing and kernel routing (see expr.hpp).
#include "expr.hpp"
#include <functional>
#include "engine.hpp"
#include <cmath>
#include <cstdarg>

namespace dfdb {

void fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  throw Error(code, buf);
}

static const char* kNames[] = {nullptr, "Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64",
                               "Float32", "Float64", "Bool", "String"};
static const int kWidth[] = {0, 1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 1, 0};
int dt_width(int32_t dt) { int b = dt_base(dt); return (b >= 1 && b <= 12) ? kWidth[b] : 0; }
std::string dt_name(int32_t dt) {  // ColumnTypes.typestring: columntypes/base.jl:78-126, complex.jl:1-8
  int b = dt_base(dt);
  if (b < 1 || b > 12) return "?";
  return dt_nullable(dt) ? std::string("Missing(") + kNames[b] + ")" : std::string(kNames[b]);
}
int32_t dt_parse(const std::string& s0) {
  std::string s = s0; int32_t flag = 0;
  if (s.size() > 9 && s.compare(0, 8, "Missing(") == 0 && s.back() == ')') { flag = DFDB_NULLABLE; s = s.substr(8, s.size() - 9); }
  for (int b = 1; b <= 12; b++) if (s == kNames[b]) return b | flag;
  fail(DFDB_ERR_UNSUPPORTED, "UndefinedType: column type '%s' is outside the engine's dtype set", s0.c_str());
}

// read_block_body! is a memcpy for every isbits T (src/io/blocks.jl:37-44), so Date / DateTime / Time / Char columns are
// integer columns to the hot path (days / milliseconds / nanoseconds / the UInt32 holding the UTF-8 bytes); the host side
// reinterprets values and lowers constants (columntypes/base.jl:108-126 and complex.jl name the type strings)
static const struct { const char* name; int dtype; } kAlias[] = {{"Date", DFDB_I64}, {"DateTime", DFDB_I64}, {"Time", DFDB_I64}, {"Char", DFDB_U32}};
int32_t dt_parse_ex(const std::string& s0, std::string* logical) {
  std::string s = s0; int32_t flag = 0;
  if (logical) logical->clear();
  if (s.size() > 9 && s.compare(0, 8, "Missing(") == 0 && s.back() == ')') { flag = DFDB_NULLABLE; s = s.substr(8, s.size() - 9); }
  for (const auto& a : kAlias) if (s == a.name) { if (logical) *logical = a.name; return a.dtype | flag; }
  return dt_parse(s0);
}
std::string dt_type_string(int32_t dt, const std::string& logical) {
  if (logical.empty()) return dt_name(dt);
  return dt_nullable(dt) ? "Missing(" + logical + ")" : logical;
}

int promote_num(int a, int b) {   // Julia promote_type restricted to the column dtypes
  a = dt_base(a); b = dt_base(b);
  if (a == DFDB_BOOL && b == DFDB_BOOL) return DFDB_BOOL;
  if (a == DFDB_BOOL) return b;
  if (b == DFDB_BOOL) return a;
  if (a == DFDB_F64 || b == DFDB_F64) return DFDB_F64;
  if (a == DFDB_F32 || b == DFDB_F32) return DFDB_F32;
  const int sa = dt_width(a), sb = dt_width(b);
  if (dt_issigned(a) == dt_issigned(b)) return sa >= sb ? a : b;
  if (sa != sb) return sa > sb ? a : b;
  return dt_issigned(a) ? b : a;
}

// result dtype; UNSUPPORTED for absent methods.  An operand of type Union{T,Missing} makes the result Union{R,Missing}
// (every Base method on the path propagates missing; `&` / `|` with three-valued logic), except ismissing (Bool) and
// coalesce (nullable only if its LAST argument is).  A selection function must still return plain Bool (selection.jl:52-55).
static int infer_base(int op, int ta, int tb);
static int infer(int op, int ta, int tb) {
  if (op == DFIR_ISMISSING) return DFDB_BOOL;
  if (op == DFIR_COALESCE) {
    if (dt_base(ta) != dt_base(tb) || !dt_isnum(dt_base(ta)))
      fail(DFDB_ERR_UNSUPPORTED, "coalesce(%s, %s): the result would be a Union of two value types", dt_name(ta).c_str(), dt_name(tb).c_str());
    return dt_base(ta) | (dt_nullable(tb) ? DFDB_NULLABLE : 0);
  }
  const int r = infer_base(op, ta, tb);
  return (dt_nullable(ta) || (tb && dt_nullable(tb))) ? (r | DFDB_NULLABLE) : r;
}
static int infer_base(int op, int ta, int tb) {
  const int a = dt_base(ta), b = tb ? dt_base(tb) : 0;
  auto nomethod = [&]() -> int { fail(DFDB_ERR_UNSUPPORTED, "no method for IR op 0x%02x on (%s, %s)", op, dt_name(ta).c_str(), tb ? dt_name(tb).c_str() : "-"); };
  switch (op) {
    case DFIR_ADD: case DFIR_SUB:
      if (!dt_isnum(a) || !dt_isnum(b)) return nomethod();
      if (a == DFDB_BOOL && b == DFDB_BOOL) return DFDB_I64;
      return promote_num(a, b);
    case DFIR_MUL: case DFIR_MIN: case DFIR_MAX:
      if (!dt_isnum(a) || !dt_isnum(b)) return nomethod();
      return promote_num(a, b);
    case DFIR_DIV: {
      if (!dt_isnum(a) || !dt_isnum(b)) return nomethod();
      const int p = promote_num(a, b);
      return dt_isfloat(p) ? p : DFDB_F64;
    }
    case DFIR_IDIV: case DFIR_REM: case DFIR_MOD: {
      if (!dt_isnum(a) || !dt_isnum(b)) return nomethod();
      const int p = promote_num(a, b);
      if (p == DFDB_BOOL) return nomethod();
      return p;
    }
    case DFIR_NEG: if (!dt_isnum(a)) return nomethod(); return a == DFDB_BOOL ? DFDB_I64 : a;
    case DFIR_ABS: if (!dt_isnum(a)) return nomethod(); return a;
    case DFIR_EQ: case DFIR_NE: case DFIR_LT: case DFIR_LE: case DFIR_GT: case DFIR_GE:
      if (dt_isnum(a) && dt_isnum(b)) return DFDB_BOOL;
      if (a == DFDB_STRING && b == DFDB_STRING) return DFDB_BOOL;
      return nomethod();
    case DFIR_AND: case DFIR_OR: case DFIR_XOR:
      if (a == DFDB_BOOL && b == DFDB_BOOL) return DFDB_BOOL;
      if ((dt_isint(a) || a == DFDB_BOOL) && (dt_isint(b) || b == DFDB_BOOL)) return promote_num(a, b);
      return nomethod();
    case DFIR_NOT: if (a != DFDB_BOOL) return nomethod(); return DFDB_BOOL;
    case DFIR_IN_SET: if (!dt_isnum(a)) return nomethod(); return DFDB_BOOL;
    case DFIR_STARTSWITH: case DFIR_ENDSWITH: if (a != DFDB_STRING || b != DFDB_STRING) return nomethod(); return DFDB_BOOL;
    case DFIR_SIZEOF: if (a != DFDB_STRING) return nomethod(); return DFDB_I64;
  }
  return nomethod();
}

NodePtr Node::clone() const {
  auto n = std::make_unique<Node>();
  n->op = op; n->dtype = dtype; n->col = col; n->cbits = cbits; n->str = str; n->set = set; n->set_dtype = set_dtype; n->cast_to = cast_to;
  if (a) n->a = a->clone();
  if (b) n->b = b->clone();
  return n;
}
NodePtr make_and(NodePtr a, NodePtr b) {
  auto n = std::make_unique<Node>();
  n->op = DFIR_AND; n->dtype = DFDB_BOOL; n->a = std::move(a); n->b = std::move(b);
  return n;
}
void required_columns(const Node& n, std::vector<int>& out) {
  if (n.op == DFIR_COL) { for (int c : out) if (c == n.col) return; out.push_back(n.col); return; }
  if (n.a) required_columns(*n.a, out);
  if (n.b) required_columns(*n.b, out);
}
void flatten_and(const Node& n, std::vector<const Node*>& out) {
  if (n.op == DFIR_AND && dt_base(n.dtype) == DFDB_BOOL && n.a && n.b && dt_base(n.a->dtype) == DFDB_BOOL && dt_base(n.b->dtype) == DFDB_BOOL) {
    flatten_and(*n.a, out); flatten_and(*n.b, out);
  } else out.push_back(&n);
}

NodePtr parse_ir(const dfdb_table& t, const uint8_t* ir, size_t len) {
  std::vector<NodePtr> st;
  size_t pos = 0;
  auto need = [&](size_t n) { if (pos + n > len) fail(DFDB_ERR_ARGUMENT, "truncated IR"); };
  auto pop = [&]() -> NodePtr { if (st.empty()) fail(DFDB_ERR_ARGUMENT, "IR stack underflow"); NodePtr n = std::move(st.back()); st.pop_back(); return n; };
  while (pos < len) {
    const int op = ir[pos++];
    auto n = std::make_unique<Node>();
    n->op = op;
    if (st.size() > 60) fail(DFDB_ERR_ARGUMENT, "IR stack overflow");
    switch (op) {
      case DFIR_COL: {
        need(4); uint32_t c; memcpy(&c, ir + pos, 4); pos += 4;
        if (c >= t.cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %u out of range", c);
        n->col = (int)c; n->dtype = t.cols[c].dtype; break;
      }
      case DFIR_CONST: {
        need(9); n->dtype = ir[pos]; memcpy(&n->cbits, ir + pos + 1, 8); pos += 9;
        if (!dt_isnum(n->dtype) || dt_nullable(n->dtype)) fail(DFDB_ERR_ARGUMENT, "bad constant dtype %d", n->dtype);
        break;
      }
      case DFIR_CONST_STR: {
        need(4); uint32_t l; memcpy(&l, ir + pos, 4); pos += 4; need(l);
        n->str.assign((const char*)ir + pos, l); pos += l; n->dtype = DFDB_STRING; break;
      }
      case DFIR_CONST_SET: {
        need(5); n->set_dtype = ir[pos]; uint32_t cnt; memcpy(&cnt, ir + pos + 1, 4); pos += 5; need(8ull * cnt);
        if (!dt_isnum(n->set_dtype)) fail(DFDB_ERR_ARGUMENT, "bad set dtype");
        n->set.resize(cnt); if (cnt) memcpy(n->set.data(), ir + pos, 8ull * cnt); pos += 8ull * cnt; n->dtype = 0; break;
      }
      case DFIR_NEG: case DFIR_ABS: case DFIR_NOT: case DFIR_ISMISSING: case DFIR_SIZEOF: case DFIR_CAST: {
        if (op == DFIR_CAST) { need(1); n->cast_to = ir[pos++]; }
        n->a = pop();
        if (n->a->op == DFIR_CONST_SET) fail(DFDB_ERR_ARGUMENT, "a set is only valid as the second argument of in");
        if (op == DFIR_CAST) {
          if (!dt_isnum(n->a->dtype) || !dt_isnum(n->cast_to) || dt_nullable(n->cast_to))
            fail(DFDB_ERR_UNSUPPORTED, "unsupported conversion %s -> %s", dt_name(n->a->dtype).c_str(), dt_name(n->cast_to).c_str());
          n->dtype = n->cast_to | (dt_nullable(n->a->dtype) ? DFDB_NULLABLE : 0);
        } else {
          n->dtype = infer(op, n->a->dtype, 0);
        }
        break;
      }
      default: {
        const bool known = (op >= DFIR_ADD && op <= DFIR_MAX) || (op >= DFIR_EQ && op <= DFIR_GE) || (op >= DFIR_AND && op <= DFIR_XOR) ||
                           op == DFIR_IN_SET || op == DFIR_STARTSWITH || op == DFIR_ENDSWITH || op == DFIR_COALESCE;
        if (!known) fail(DFDB_ERR_UNSUPPORTED, "unknown IR opcode 0x%02x", op);
        n->b = pop(); n->a = pop();
        if (op == DFIR_IN_SET) {
          if (n->b->op != DFIR_CONST_SET || n->a->op == DFIR_CONST_SET) fail(DFDB_ERR_ARGUMENT, "in needs (value, set)");
          n->dtype = infer(op, n->a->dtype, 0);
        } else {
          if (n->a->op == DFIR_CONST_SET || n->b->op == DFIR_CONST_SET) fail(DFDB_ERR_ARGUMENT, "a set is only valid as the second argument of in");
          if ((op == DFIR_STARTSWITH || op == DFIR_ENDSWITH) && n->b->op != DFIR_CONST_STR) fail(DFDB_ERR_UNSUPPORTED, "startswith/endswith need a constant pattern");
          n->dtype = infer(op, n->a->dtype, n->b->dtype);
        }
        break;
      }
    }
    st.push_back(std::move(n));
  }
  if (st.size() != 1) fail(DFDB_ERR_ARGUMENT, "IR must leave exactly one value (left %zu)", st.size());
  if (st[0]->op == DFIR_CONST_SET) fail(DFDB_ERR_ARGUMENT, "IR result cannot be a set");
  return std::move(st[0]);
}

// ---------------------------------------------------------------- routing to specialised kernels
static int cmp_from_ir(int op) {
  switch (op) { case DFIR_EQ: return CMP_EQ; case DFIR_NE: return CMP_NE; case DFIR_LT: return CMP_LT; case DFIR_LE: return CMP_LE;
                case DFIR_GT: return CMP_GT; case DFIR_GE: return CMP_GE; }
  return -1;
}
static int flip(int op) {  // c OP x  ==  x flip(OP) c
  switch (op) { case CMP_LT: return CMP_GT; case CMP_LE: return CMP_GE; case CMP_GT: return CMP_LT; case CMP_GE: return CMP_LE; }
  return op;
}

struct IntRange { __int128 lo, hi; };
static IntRange int_range(int dt) {
  switch (dt_base(dt)) {
    case DFDB_I8: return {-128, 127}; case DFDB_I16: return {-32768, 32767};
    case DFDB_I32: return {-(__int128)2147483648LL, 2147483647}; case DFDB_I64: return {(__int128)INT64_MIN, (__int128)INT64_MAX};
    case DFDB_U8: return {0, 255}; case DFDB_U16: return {0, 65535}; case DFDB_U32: return {0, 4294967295LL};
    case DFDB_BOOL: return {0, 1};
    default: return {0, (__int128)UINT64_MAX};
  }
}
static uint64_t int_bits(__int128 v) { return (uint64_t)v; }

// make the term constant-true / constant-false for every value of an integer column type
static void const_term(ScanTerm& term, int coldt, bool value) {
  IntRange r = int_range(coldt);
  term.op = value ? CMP_GE : CMP_LT;   // x >= Tmin is always true, x < Tmin always false
  term.cbits = int_bits(r.lo);
}

// integer column vs integer constant c (as int128)
static void int_vs_int(ScanTerm& term, int coldt, int op, __int128 c) {
  IntRange r = int_range(coldt);
  if (c > r.hi) { const_term(term, coldt, op == CMP_LT || op == CMP_LE || op == CMP_NE); return; }
  if (c < r.lo) { const_term(term, coldt, op == CMP_GT || op == CMP_GE || op == CMP_NE); return; }
  term.op = op; term.cbits = int_bits(c);
}

// A value the scan kernels (and the transforming gather) can make from ONE column on the fly: the column itself (pre 0), rem(col, m) for a signed
// integer column and an integer m other than 0 / 1 / -1 (pre 1: the DivideError of m = 0 stays with the interpreter), col * k + d with at most one
// multiplication on the column and one addition / subtraction after it — all in wrapping Int64 (pre 2: a signed integer column, integer constants) or all
// in Float64 with a rounding after each step, never fused (pre 3: any numeric column) — and col / k, Julia's Float64 division (pre 4).  `out` gets the
// pre / pre_magic / pre_shift / pre_d fields of a ScanTerm; `coln` the column node.
bool match_column_transform(const Node* e, ScanTerm& out, const Node*& coln) {
  out.pre = 0; out.pre_shift = 0; out.pre_magic = 0; out.pre_d = 0;
  if (e->op == DFIR_COL) { coln = e; return true; }
  if (!e->a || !e->b || dt_nullable(e->dtype)) return false;
  // rem(col, m)
  if (e->op == DFIR_REM && e->a->op == DFIR_COL && e->b->op == DFIR_CONST && dt_base(e->dtype) == DFDB_I64) {
    coln = e->a.get();
    const int cdt = dt_base(coln->dtype), mdt = dt_base(e->b->dtype);
    const bool sint = cdt == DFDB_I8 || cdt == DFDB_I16 || cdt == DFDB_I32 || cdt == DFDB_I64;
    const bool mint = mdt == DFDB_I8 || mdt == DFDB_I16 || mdt == DFDB_I32 || mdt == DFDB_I64;
    if (!sint || !mint || dt_nullable(coln->dtype)) return false;
    const int64_t m = (int64_t)e->b->cbits;
    if (m == 0 || m == 1 || m == -1) return false;
    const uint64_t d = m < 0 ? 0ull - (uint64_t)m : (uint64_t)m;          // |m|, 2 <= d <= 2^63
    // unsigned division by the invariant d, branch-free form: q = mulhi(magic, x); floor(x / d) = (((x - q) >> 1) + q) >> shift
    const int fl = 63 - __builtin_clzll(d);
    uint64_t magic; int shift;
    if ((d & (d - 1)) == 0) { magic = 0; shift = fl - 1; }
    else {
      const unsigned __int128 num = (unsigned __int128)1 << (64 + fl);
      unsigned __int128 pm = num / d; const uint64_t rem = (uint64_t)(num % d);
      pm += pm;
      const uint64_t twice = rem + rem;
      if (twice >= d || twice < rem) pm += 1;
      magic = (uint64_t)pm + 1; shift = fl;
    }
    out.pre = 1; out.pre_magic = magic; out.pre_shift = shift; out.pre_d = d;
    return true;
  }
  auto const_is_int = [](const Node* k) { const int d = dt_base(k->dtype); return d == DFDB_I8 || d == DFDB_I16 || d == DFDB_I32 || d == DFDB_I64; };
  auto const_as_double = [](const Node* k, double& o) {
    const int d = dt_base(k->dtype);
    if (d == DFDB_F64) { memcpy(&o, &k->cbits, 8); return true; }
    if (d == DFDB_F32) { float f; memcpy(&f, &k->cbits, 4); o = (double)f; return true; }
    if (d == DFDB_I8 || d == DFDB_I16 || d == DFDB_I32 || d == DFDB_I64) { o = (double)(int64_t)k->cbits; return true; }
    return false;
  };
  // col / k
  if (e->op == DFIR_DIV && e->a->op == DFIR_COL && e->b->op == DFIR_CONST && dt_base(e->dtype) == DFDB_F64 && !dt_nullable(e->a->dtype) && dt_isnum(e->a->dtype) &&
      dt_base(e->a->dtype) != DFDB_BOOL && dt_base(e->a->dtype) != DFDB_F32) {
    double kv;
    if (!const_as_double(e->b.get(), kv)) return false;
    coln = e->a.get(); out.pre = 4; memcpy(&out.pre_magic, &kv, 8);
    return true;
  }
  // col * k + d
  // (fadd starts at -0.0, the additive identity of IEEE arithmetic: x + (-0.0) == x for every x, -0.0 included; +0.0 would turn a -0.0 product into +0.0)
  struct Affine { const Node* col = nullptr; bool flt = false; __int128 imul = 1, iadd = 0; double fmul = 1.0, fadd = -0.0; bool has_add = false, has_mul = false; };
  std::function<bool(const Node*, Affine&, int)> affine = [&](const Node* x0, Affine& A, int depth) -> bool {
    if (x0->op == DFIR_COL) { A.col = x0; return !dt_nullable(x0->dtype); }
    if (depth > 2 || !x0->a || !x0->b || dt_nullable(x0->dtype)) return false;
    const int rt = dt_base(x0->dtype);
    if (rt != DFDB_I64 && rt != DFDB_F64) return false;
    const bool flt = rt == DFDB_F64;
    const Node *x = nullptr, *k = nullptr; bool k_left = false;
    if (x0->b->op == DFIR_CONST) { x = x0->a.get(); k = x0->b.get(); } else if (x0->a->op == DFIR_CONST) { x = x0->b.get(); k = x0->a.get(); k_left = true; } else return false;
    if (x0->op == DFIR_MUL) {
      if (x->op != DFIR_COL || !affine(x, A, depth + 1) || A.has_mul || A.has_add) return false;   // the multiplication touches the column itself
      if (flt) { double kv; if (!const_as_double(k, kv)) return false; A.flt = true; A.fmul = kv; }
      else { if (!const_is_int(k)) return false; A.imul = (__int128)(int64_t)k->cbits; }
      A.has_mul = true; return true;
    }
    if (x0->op != DFIR_ADD && x0->op != DFIR_SUB) return false;
    if (!affine(x, A, depth + 1) || A.has_add) return false;
    if (x->op != DFIR_COL && (dt_base(x->dtype) == DFDB_F64) != flt) return false;             // (col * k) in one type, the sum in another: not one form
    if (flt) {
      double kv; if (!const_as_double(k, kv)) return false;
      if (!A.flt && A.has_mul) return false;
      A.flt = true;
      if (x0->op == DFIR_ADD) A.fadd = kv;
      else if (!k_left) A.fadd = -kv;                       // x - k  =  x + (-k)   (exact: negation does not round)
      else { A.fmul = -A.fmul; A.fadd = kv; }               // k - x  =  (-x) + k
    } else {
      if (!const_is_int(k) || A.flt) return false;
      const __int128 kv = (__int128)(int64_t)k->cbits;
      if (x0->op == DFIR_ADD) A.iadd = kv; else if (!k_left) A.iadd = -kv; else { A.imul = -A.imul; A.iadd = kv; }
    }
    A.has_add = true; return true;
  };
  if (e->op != DFIR_MUL && e->op != DFIR_ADD && e->op != DFIR_SUB) return false;
  Affine aff;
  if (!affine(e, aff, 0)) return false;
  const int cdt = dt_base(aff.col->dtype);
  const bool sint = cdt == DFDB_I8 || cdt == DFDB_I16 || cdt == DFDB_I32 || cdt == DFDB_I64;
  const bool isflt = dt_base(e->dtype) == DFDB_F64;
  if (isflt != aff.flt) return false;
  if (!isflt && !sint) return false;
  if (isflt && !(dt_isnum(cdt) && cdt != DFDB_BOOL)) return false;
  coln = aff.col;
  if (isflt) { out.pre = 3; memcpy(&out.pre_magic, &aff.fmul, 8); memcpy(&out.pre_d, &aff.fadd, 8); }
  else { out.pre = 2; out.pre_magic = (uint64_t)aff.imul; out.pre_d = (uint64_t)aff.iadd; }
  return true;
}

bool match_simple_term(const Node& n, const dfdb_table& t, ScanTerm& term, int& ordinal) {
  if (n.op == DFIR_COL && n.dtype == DFDB_BOOL) {   // a Bool column as the selection itself (DFColumn{Bool}: view.jl:60-72): its bytes != 0
    term.col = nullptr; term.dtype = DFDB_BOOL; term.op = CMP_NE; term.cbits = 0; ordinal = n.col;
    return true;
  }
  int op = cmp_from_ir(n.op);
  if (op < 0 || !n.a || !n.b) return false;
  const Node *coln = nullptr, *cn = nullptr;
  // <column transform> OP const, either way round (match_column_transform: the column itself, rem, col * k + d, col / k)
  if (n.b->op == DFIR_CONST && match_column_transform(n.a.get(), term, coln)) cn = n.b.get();
  else if (n.a->op == DFIR_CONST && match_column_transform(n.b.get(), term, coln)) { cn = n.a.get(); op = flip(op); }
  else { term.pre = 0; return false; }
  const bool remn = term.pre == 1;
  const int coldt = dt_base(coln->dtype);
  const int ct = remn || term.pre == 2 ? (int)DFDB_I64 : (term.pre >= 3 ? (int)DFDB_F64 : coldt), kt = dt_base(cn->dtype);   // ct: the type the comparison happens in
  if (dt_nullable(coln->dtype) || !dt_isnum(ct) || ct == DFDB_BOOL) return false;
  (void)t;
  term.col = nullptr; term.dtype = coldt; ordinal = coln->col;
  if (dt_isint(ct)) {
    if (dt_isint(kt) || kt == DFDB_BOOL) {
      __int128 c = (kt == DFDB_U64) ? (__int128)(uint64_t)cn->cbits : (__int128)(int64_t)cn->cbits;
      if (kt == DFDB_BOOL) c = cn->cbits ? 1 : 0;
      int_vs_int(term, ct, op, c); return true;
    }
    // Int column vs Float constant: Julia compares exactly -> move to an equivalent integer threshold
    double d; if (kt == DFDB_F32) { float f; memcpy(&f, &cn->cbits, 4); d = f; } else memcpy(&d, &cn->cbits, 8);
    if (d != d) { const_term(term, ct, op == CMP_NE); return true; }
    if (d >= 1.8446744073709552e19) { const_term(term, ct, op == CMP_LT || op == CMP_LE || op == CMP_NE); return true; }
    if (d < -9.2233720368547758e18) { const_term(term, ct, op == CMP_GT || op == CMP_GE || op == CMP_NE); return true; }
    const double fl = std::floor(d);
    const __int128 fi = (__int128)fl;
    if (fl == d) { int_vs_int(term, ct, op, fi); return true; }
    switch (op) {   // d strictly between fi and fi+1
      case CMP_EQ: const_term(term, ct, false); return true;
      case CMP_NE: const_term(term, ct, true); return true;
      case CMP_LT: case CMP_LE: int_vs_int(term, ct, CMP_LE, fi); return true;
      default: int_vs_int(term, ct, CMP_GE, fi + 1); return true;
    }
  }
  // float column
  if (ct == DFDB_F64) {
    double d;
    if (kt == DFDB_F64) memcpy(&d, &cn->cbits, 8);
    else if (kt == DFDB_F32) { float f; memcpy(&f, &cn->cbits, 4); d = f; }
    else if (kt == DFDB_BOOL) d = cn->cbits ? 1.0 : 0.0;
    else {
      if (kt == DFDB_U64) { uint64_t u = cn->cbits; if (u > (1ull << 53)) return false; d = (double)u; }
      else { int64_t i = (int64_t)cn->cbits; if (i > (1ll << 53) || i < -(1ll << 53)) return false; d = (double)i; }
    }
    term.op = op; memcpy(&term.cbits, &d, 8); return true;
  }
  if (ct == DFDB_F32) {
    float f;
    if (kt == DFDB_F32) memcpy(&f, &cn->cbits, 4);
    else if (kt == DFDB_F64) { double d; memcpy(&d, &cn->cbits, 8); f = (float)d; if ((double)f != d && d == d) return false; }
    else if (kt == DFDB_BOOL) f = cn->cbits ? 1.f : 0.f;
    else { int64_t i = (int64_t)cn->cbits; if (kt == DFDB_U64 || i > (1 << 24) || i < -(1 << 24)) return false; f = (float)i; }
    term.op = op; term.cbits = 0; memcpy(&term.cbits, &f, 4); return true;
  }
  return false;
}

// allow_nullable: the caller evaluates `coalesce(term, false)` — K5 gives exactly that over a Union{String,Missing} column (a missing row selects nothing)
bool match_string_term(const Node& n, const dfdb_table& t, int& ordinal, int& mode, std::string& pat, bool allow_nullable) {
  (void)t;
  if (!n.a || !n.b) return false;
  const Node *coln = nullptr, *cn = nullptr;
  if (n.a->op == DFIR_COL && n.b->op == DFIR_CONST_STR) { coln = n.a.get(); cn = n.b.get(); }
  else if (n.a->op == DFIR_CONST_STR && n.b->op == DFIR_COL && (n.op == DFIR_EQ || n.op == DFIR_NE)) { coln = n.b.get(); cn = n.a.get(); }
  else return false;
  if (dt_base(coln->dtype) != DFDB_STRING || (dt_nullable(coln->dtype) && !allow_nullable)) return false;
  switch (n.op) {
    case DFIR_EQ: mode = 0; break; case DFIR_NE: mode = 1; break;
    case DFIR_STARTSWITH: mode = 2; break; case DFIR_ENDSWITH: mode = 3; break;
    default: return false;
  }
  ordinal = coln->col; pat = cn->str; return true;
}

}  // namespace dfdb
