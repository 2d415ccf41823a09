/* orc_expr.c — expression trees of the CPU oracle: IR parsing, Julia result-type inference and
 * gather-then-evaluate execution.  TEST INFRASTRUCTURE ONLY (see oracle.h).  Restates:
 *   src/tables/broadcast.jl:2-49    (ColRef / BlockBroadcasting, result type, required_columns)
 *   src/tables/broadcast.jl:96-133  (_extract_for_eval! gather, eval_on_range)
 * Scalar semantics are Julia Base's (promotion, exact Int/Float comparison, rem/mod/div, wraparound).
 */
#include "orc_internal.h"
#include <math.h>

/* ---------------------------------------------------------------- arena */
void* arena_alloc(arena_t* a, size_t n) {
  n = (n + 63) & ~(size_t)63;
  if (!a->base) { a->cap = 1u << 22; a->base = (uint8_t*)malloc(a->cap); a->used = 0; }
  if (a->used + n <= a->cap) { void* p = a->base + a->used; a->used += n; return p; }
  /* spill: individually malloc'ed, released at the next reset (which also grows the main chunk) */
  void** blk = (void**)malloc(n + 64);
  if (!blk) return NULL;
  blk[0] = a->overflow; blk[1] = (void*)n; a->overflow = blk;
  return (uint8_t*)blk + 64;
}
void arena_reset(arena_t* a) {
  size_t extra = 0;
  while (a->overflow) { void** blk = (void**)a->overflow; a->overflow = blk[0]; extra += (size_t)blk[1] + 64; free(blk); }
  if (extra) { free(a->base); a->cap = (a->cap + extra) * 2; a->base = (uint8_t*)malloc(a->cap); }
  a->used = 0;
}
void arena_free(arena_t* a) { arena_reset(a); free(a->base); a->base = NULL; a->cap = 0; }

/* ---------------------------------------------------------------- Julia promotion */
static int int_size(int b) { return dt_width(b); }
static int promote_num(int a, int b) { /* promote_type for the numeric dtypes */
  a = dt_base(a); b = dt_base(b);
  if (a == DFDB_BOOL && b == DFDB_BOOL) return DFDB_BOOL;
  if (a == DFDB_BOOL) return b;
  if (b == DFDB_BOOL) return a;
  if (a == DFDB_F64 || b == DFDB_F64) return DFDB_F64;
  if (a == DFDB_F32 || b == DFDB_F32) return DFDB_F32;
  int sa = int_size(a), sb = int_size(b);
  if (dt_issigned(a) == dt_issigned(b)) return sa >= sb ? a : b;
  if (sa != sb) return sa > sb ? a : b;   /* larger wins */
  return dt_issigned(a) ? b : a;          /* tie -> unsigned */
}

/* result type of op over operand types; -1 = no method (UNSUPPORTED).  An operand of type Union{T,Missing} makes the result
 * Union{R,Missing} (every Base method on the path propagates missing; `&`/`|` do so with three-valued logic), except
 * ismissing (Bool) and coalesce (the first non-missing argument: nullable only if the LAST argument is). */
static int infer_base(int op, int a, int b);
static int infer(int op, int ta, int tb) {
  int na = dt_nullable(ta), nb = tb ? dt_nullable(tb) : 0;
  int a = dt_base(ta), b = tb ? dt_base(tb) : 0;
  if (op == DFIR_ISMISSING) return DFDB_BOOL;
  if (op == DFIR_COALESCE) { if (a != b || !dt_isnum(a)) return -1; return a | (nb ? DFDB_NULLABLE : 0); }
  int r = infer_base(op, a, b);
  return (r >= 0 && (na || nb)) ? (r | DFDB_NULLABLE) : r;
}
static int infer_base(int op, int a, int b) {
  switch (op) {
    case DFIR_ADD: case DFIR_SUB:
      if (!dt_isnum(a) || !dt_isnum(b)) return -1;
      if (a == DFDB_BOOL && b == DFDB_BOOL) return DFDB_I64; /* true + true == 2 */
      return promote_num(a, b);
    case DFIR_MUL: case DFIR_MIN: case DFIR_MAX:
      if (!dt_isnum(a) || !dt_isnum(b)) return -1;
      return promote_num(a, b);
    case DFIR_DIV: {
      if (!dt_isnum(a) || !dt_isnum(b)) return -1;
      int p = promote_num(a, b);
      return dt_isfloat(p) ? p : DFDB_F64;
    }
    case DFIR_IDIV: case DFIR_REM: case DFIR_MOD: {
      if (!dt_isnum(a) || !dt_isnum(b)) return -1;
      int p = promote_num(a, b);
      return p == DFDB_BOOL ? -1 : p;
    }
    case DFIR_NEG: if (!dt_isnum(a)) return -1; return a == DFDB_BOOL ? DFDB_I64 : a;
    case DFIR_ABS: if (!dt_isnum(a)) return -1; return a;
    case DFIR_EQ: case DFIR_NE: case DFIR_LT: case DFIR_LE: case DFIR_GT: case DFIR_GE:
      if (dt_isnum(a) && dt_isnum(b)) return DFDB_BOOL;
      if (a == DFDB_STRING && b == DFDB_STRING) return DFDB_BOOL;
      return -1;
    case DFIR_AND: case DFIR_OR: case DFIR_XOR:
      if (a == DFDB_BOOL && b == DFDB_BOOL) return DFDB_BOOL;
      if ((dt_isint(a) || a == DFDB_BOOL) && (dt_isint(b) || b == DFDB_BOOL)) return promote_num(a, b);
      return -1;
    case DFIR_NOT: return a == DFDB_BOOL ? DFDB_BOOL : -1;
    case DFIR_IN_SET: return dt_isnum(a) ? DFDB_BOOL : -1;
    case DFIR_STARTSWITH: case DFIR_ENDSWITH: return (a == DFDB_STRING && b == DFDB_STRING) ? DFDB_BOOL : -1;
    case DFIR_SIZEOF: return a == DFDB_STRING ? DFDB_I64 : -1;
  }
  return -1;
}

/* ---------------------------------------------------------------- parsing */
void expr_free(node_t* n) {
  if (!n) return;
  expr_free(n->a); expr_free(n->b);
  free(n->str); free(n->set_i); free(n->set_f); free(n);
}
node_t* expr_clone(const node_t* n) {
  if (!n) return NULL;
  node_t* c = (node_t*)malloc(sizeof *c); *c = *n;
  if (n->str) { c->str = (uint8_t*)malloc((size_t)n->slen + 1); memcpy(c->str, n->str, (size_t)n->slen); }
  if (n->set_i) { c->set_i = (int64_t*)malloc(8 * (size_t)n->nset); memcpy(c->set_i, n->set_i, 8 * (size_t)n->nset); }
  if (n->set_f) { c->set_f = (double*)malloc(8 * (size_t)n->nset); memcpy(c->set_f, n->set_f, 8 * (size_t)n->nset); }
  c->a = expr_clone(n->a); c->b = expr_clone(n->b);
  return c;
}
node_t* expr_and(node_t* a, node_t* b) {
  node_t* n = (node_t*)calloc(1, sizeof *n);
  n->op = DFIR_AND; n->dtype = DFDB_BOOL; n->a = a; n->b = b; return n;
}
void expr_required(const node_t* n, int32_t* ords, int* count, int cap) {
  if (!n) return;
  if (n->op == DFIR_COL) {
    for (int i = 0; i < *count; i++) if (ords[i] == n->col) return;
    if (*count < cap) ords[(*count)++] = n->col;
    return;
  }
  expr_required(n->a, ords, count, cap); expr_required(n->b, ords, count, cap);
}

static double const_as_f64(int dt, const uint8_t* p) {
  int64_t i; memcpy(&i, p, 8);
  switch (dt_base(dt)) {
    case DFDB_F64: { double d; memcpy(&d, p, 8); return d; }
    case DFDB_F32: { float f; memcpy(&f, p, 4); return (double)f; }
    case DFDB_U64: return (double)(uint64_t)i;
    default: return (double)i;
  }
}

int expr_parse(const orc_table* t, const uint8_t* ir, size_t len, node_t** out) {
  node_t* stack[64]; int sp = 0; size_t pos = 0; int rc = 0;
#define FAIL(code, ...) do { rc = orc_fail(code, __VA_ARGS__); goto done; } while (0)
  while (pos < len) {
    int op = ir[pos++];
    node_t* n = (node_t*)calloc(1, sizeof *n);
    n->op = op;
    if (sp >= 62) { free(n); FAIL(ORC_ERR_ARGUMENT, "IR stack overflow"); }
    switch (op) {
      case DFIR_COL: {
        uint32_t c; if (pos + 4 > len) { free(n); FAIL(ORC_ERR_ARGUMENT, "truncated IR"); }
        memcpy(&c, ir + pos, 4); pos += 4;
        if ((int)c >= t->ncols) { free(n); FAIL(ORC_ERR_KEY, "column ordinal %u out of range", c); }
        n->col = (int)c; n->dtype = t->cols[c].dtype; stack[sp++] = n; break;
      }
      case DFIR_CONST: {
        if (pos + 9 > len) { free(n); FAIL(ORC_ERR_ARGUMENT, "truncated IR"); }
        n->cdtype = ir[pos]; n->dtype = ir[pos];
        if (!dt_isnum(n->dtype)) { free(n); FAIL(ORC_ERR_ARGUMENT, "bad const dtype"); }
        memcpy(&n->ci, ir + pos + 1, 8); n->cf = const_as_f64(n->dtype, ir + pos + 1); pos += 9;
        stack[sp++] = n; break;
      }
      case DFIR_CONST_STR: {
        uint32_t l; if (pos + 4 > len) { free(n); FAIL(ORC_ERR_ARGUMENT, "truncated IR"); }
        memcpy(&l, ir + pos, 4); pos += 4;
        if (pos + l > len) { free(n); FAIL(ORC_ERR_ARGUMENT, "truncated IR"); }
        n->str = (uint8_t*)malloc((size_t)l + 1); memcpy(n->str, ir + pos, l); n->slen = (int32_t)l; pos += l;
        n->dtype = DFDB_STRING; stack[sp++] = n; break;
      }
      case DFIR_CONST_SET: {
        uint32_t cnt; if (pos + 5 > len) { free(n); FAIL(ORC_ERR_ARGUMENT, "truncated IR"); }
        n->set_dtype = ir[pos]; memcpy(&cnt, ir + pos + 1, 4); pos += 5;
        if (pos + 8ull * cnt > len) { free(n); FAIL(ORC_ERR_ARGUMENT, "truncated IR"); }
        n->nset = (int32_t)cnt; n->set_i = (int64_t*)malloc(8 * (size_t)cnt + 8); n->set_f = (double*)malloc(8 * (size_t)cnt + 8);
        for (uint32_t k = 0; k < cnt; k++) { memcpy(&n->set_i[k], ir + pos + 8 * k, 8); n->set_f[k] = const_as_f64(n->set_dtype, ir + pos + 8 * k); }
        pos += 8ull * cnt; n->dtype = 0; stack[sp++] = n; break;
      }
      case DFIR_NEG: case DFIR_ABS: case DFIR_NOT: case DFIR_ISMISSING: case DFIR_SIZEOF: case DFIR_CAST: {
        if (op == DFIR_CAST) { if (pos + 1 > len) { free(n); FAIL(ORC_ERR_ARGUMENT, "truncated IR"); } n->cast_to = ir[pos++]; }
        if (sp < 1) { free(n); FAIL(ORC_ERR_ARGUMENT, "IR stack underflow"); }
        n->a = stack[--sp];
        int rt = op == DFIR_CAST ? ((dt_isnum(n->a->dtype) && dt_isnum(n->cast_to) && !dt_nullable(n->cast_to)) ? (n->cast_to | (dt_nullable(n->a->dtype) ? DFDB_NULLABLE : 0)) : -1)
                                 : infer(op, n->a->dtype, 0);
        stack[sp++] = n;
        if (rt < 0) FAIL(ORC_ERR_UNSUPPORTED, "no method for op 0x%02x on %s", op, dt_name(n->a->dtype));
        n->dtype = rt; break;
      }
      default: {
        if (!((op >= DFIR_ADD && op <= DFIR_MAX) || (op >= DFIR_EQ && op <= DFIR_GE) || (op >= DFIR_AND && op <= DFIR_XOR) ||
              op == DFIR_IN_SET || op == DFIR_STARTSWITH || op == DFIR_ENDSWITH || op == DFIR_COALESCE)) { free(n); FAIL(ORC_ERR_UNSUPPORTED, "unknown IR opcode 0x%02x", op); }
        if (sp < 2) { free(n); FAIL(ORC_ERR_ARGUMENT, "IR stack underflow"); }
        n->b = stack[--sp]; n->a = stack[--sp];
        int rt;
        if (op == DFIR_IN_SET) rt = (n->b->op == DFIR_CONST_SET) ? infer(op, n->a->dtype, 0) : -1;
        else if (n->a->op == DFIR_CONST_SET || n->b->op == DFIR_CONST_SET) rt = -1;
        else rt = infer(op, n->a->dtype, n->b->dtype);
        if ((op == DFIR_STARTSWITH || op == DFIR_ENDSWITH) && n->b->op != DFIR_CONST_STR) rt = -1;
        stack[sp++] = n;
        if (rt < 0) FAIL(ORC_ERR_UNSUPPORTED, "no method for op 0x%02x", op);
        n->dtype = rt; break;
      }
    }
  }
  if (sp != 1) FAIL(ORC_ERR_ARGUMENT, "IR must leave exactly one value (left %d)", sp);
  if (stack[0]->op == DFIR_CONST_SET) FAIL(ORC_ERR_ARGUMENT, "IR result cannot be a set");
  *out = stack[0]; return 0;
done:
  for (int i = 0; i < sp; i++) expr_free(stack[i]);
  return rc;
#undef FAIL
}

int orc_expr_result_type(orc_table* t, const uint8_t* ir, size_t len, int32_t* dtype) {
  node_t* n; int rc = expr_parse(t, ir, len, &n); if (rc) return rc;
  *dtype = n->dtype; expr_free(n); return 0;
}
int orc_expr_required_columns(orc_table* t, const uint8_t* ir, size_t len, int32_t* ordinals, int cap) {
  node_t* n; int rc = expr_parse(t, ir, len, &n); if (rc) return -rc;
  int cnt = 0; expr_required(n, ordinals, &cnt, cap); expr_free(n); return cnt;
}

/* ---------------------------------------------------------------- scalar semantics */
static inline int64_t wrap_int(int64_t x, int dt) {
  switch (dt) {
    case DFDB_I8: return (int8_t)x;   case DFDB_I16: return (int16_t)x; case DFDB_I32: return (int32_t)x;
    case DFDB_U8: return (uint8_t)x;  case DFDB_U16: return (uint16_t)x; case DFDB_U32: return (uint32_t)x;
    default: return x;
  }
}
static inline int64_t int_min_of(int dt) {
  switch (dt) { case DFDB_I8: return -128; case DFDB_I16: return -32768; case DFDB_I32: return -2147483648LL; case DFDB_I64: return INT64_MIN; }
  return 0;
}
/* three-way compare, 2 = unordered.  Int vs Float is mathematically exact (Julia Base float.jl). */
static inline int cmp_ff(double x, double y) { if (x != x || y != y) return 2; return x < y ? -1 : (x > y ? 1 : 0); }
static inline int cmp_if(int64_t x, double y) {
  if (y != y) return 2;
  if (y >= 9223372036854775808.0) return -1;
  if (y < -9223372036854775808.0) return 1;
  int64_t yi = (int64_t)y;
  if (x < yi) return -1;
  if (x > yi) return 1;
  double fr = y - (double)yi;
  return fr > 0 ? -1 : (fr < 0 ? 1 : 0);
}
static inline int cmp_uf(uint64_t x, double y) {
  if (y != y) return 2;
  if (y >= 18446744073709551616.0) return -1;
  if (y < 0) return 1;
  uint64_t yi = (uint64_t)y;
  if (x < yi) return -1;
  if (x > yi) return 1;
  double fr = y - (double)yi;
  return fr > 0 ? -1 : 0;
}
static inline int cmp_ii(int64_t x, int xu, int64_t y, int yu) { /* xu/yu: value is a UInt64 bit pattern */
  if (xu == yu) { if (xu) return (uint64_t)x < (uint64_t)y ? -1 : ((uint64_t)x > (uint64_t)y ? 1 : 0); return x < y ? -1 : (x > y ? 1 : 0); }
  if (xu) { if (y < 0) return 1; return (uint64_t)x < (uint64_t)y ? -1 : ((uint64_t)x > (uint64_t)y ? 1 : 0); }
  if (x < 0) return -1;
  return (uint64_t)x < (uint64_t)y ? -1 : ((uint64_t)x > (uint64_t)y ? 1 : 0);
}
static inline int cmp_to_bool(int op, int c) {
  switch (op) {
    case DFIR_EQ: return c == 0;
    case DFIR_NE: return c != 0;
    case DFIR_LT: return c == -1;
    case DFIR_LE: return c == -1 || c == 0;
    case DFIR_GT: return c == 1;
    default:      return c == 1 || c == 0;
  }
}

/* ---------------------------------------------------------------- evaluation */
typedef struct {
  const colbuf_t* bufs; const int32_t* idx; int64_t n; arena_t* ar; int err;
  int64_t erow[2];   /* the first evaluated element each kind of error happened on ([0] DivideError, [1] InexactError): Julia throws the one of the earlier row */
} ectx_t;
static void flag_err(ectx_t* c, int code, int64_t k) {
  c->err = code;
  int kind = code == ORC_ERR_DIVIDE ? 0 : 1;
  if (k < c->erow[kind]) c->erow[kind] = k;
}

static int vec_alloc(ectx_t* c, vec_t* v, int dtype, int64_t n) {
  memset(v, 0, sizeof *v); v->dtype = dtype; v->n = n;
  size_t cnt = (size_t)(n > 0 ? n : 1);
  if (dt_isint(dtype)) v->i = (int64_t*)arena_alloc(c->ar, cnt * 8);
  else if (dt_isfloat(dtype)) v->f = (double*)arena_alloc(c->ar, cnt * 8);
  else v->b = (uint8_t*)arena_alloc(c->ar, cnt);
  if (!v->i && !v->f && !v->b) return orc_fail(ORC_ERR_NOMEM, "oracle arena exhausted");
  return 0;
}

/* _extract_for_eval! (broadcast.jl:96-118): copy the column at the selected positions into a private
 * contiguous buffer (straight copy when every row is selected) */
static int gather_col(ectx_t* c, const colbuf_t* b, vec_t* v) {
  int bt = dt_base(b->dtype); int64_t n = c->n; const int32_t* idx = c->idx;
  if (bt == DFDB_STRING) {
    memset(v, 0, sizeof *v); v->dtype = b->dtype; v->n = n; v->scol = b;
    if (dt_nullable(b->dtype)) { /* a missing String is a size of -1 (FlatStringsVectors.jl:80-85) */
      v->miss = (uint8_t*)arena_alloc(c->ar, (size_t)(n > 0 ? n : 1));
      if (!v->miss) return orc_fail(ORC_ERR_NOMEM, "oracle arena exhausted");
      for (int64_t k = 0; k < n; k++) v->miss[k] = b->sizes[idx ? idx[k] : k] < 0;
    }
    return 0;
  }
  int rc = vec_alloc(c, v, bt, n); if (rc) return rc;
  v->dtype = b->dtype;
#define GATHER(T, dst) do { const T* s = (const T*)b->data; if (!idx) for (int64_t k = 0; k < n; k++) dst[k] = s[k]; else for (int64_t k = 0; k < n; k++) dst[k] = s[idx[k]]; } while (0)
  switch (bt) {
    case DFDB_I8: GATHER(int8_t, v->i); break;   case DFDB_I16: GATHER(int16_t, v->i); break;
    case DFDB_I32: GATHER(int32_t, v->i); break; case DFDB_I64: GATHER(int64_t, v->i); break;
    case DFDB_U8: GATHER(uint8_t, v->i); break;  case DFDB_U16: GATHER(uint16_t, v->i); break;
    case DFDB_U32: GATHER(uint32_t, v->i); break; case DFDB_U64: GATHER(int64_t, v->i); break;
    case DFDB_F32: GATHER(float, v->f); break;   case DFDB_F64: GATHER(double, v->f); break;
    case DFDB_BOOL: GATHER(uint8_t, v->b); break;
  }
#undef GATHER
  if (dt_nullable(b->dtype)) {
    v->miss = (uint8_t*)arena_alloc(c->ar, (size_t)(n > 0 ? n : 1));
    if (!v->miss) return orc_fail(ORC_ERR_NOMEM, "oracle arena exhausted");
    if (!idx) memcpy(v->miss, b->missing, (size_t)n); else for (int64_t k = 0; k < n; k++) v->miss[k] = b->missing[idx[k]];
  }
  return 0;
}

/* numeric conversion of an operand to compute type `to` (Julia convert on promotion) */
static int conv(ectx_t* c, const vec_t* s, int to, vec_t* d) {
  int from = dt_base(s->dtype);
  if (from == to) { *d = *s; return 0; }
  int64_t n = s->is_const ? 1 : s->n;
  int rc = vec_alloc(c, d, to, n); if (rc) return rc;
  d->is_const = s->is_const; d->n = s->n; d->miss = s->miss;
  const uint8_t* ms = s->miss;
  if (dt_isint(to)) {
    if (dt_isint(from)) for (int64_t k = 0; k < n; k++) d->i[k] = wrap_int(s->i[k], to);
    else if (from == DFDB_BOOL) for (int64_t k = 0; k < n; k++) d->i[k] = s->b[k];
    else for (int64_t k = 0; k < n; k++) { /* Float -> Int: InexactError unless integral */
      double x = s->f[k];
      if (ms && ms[k]) { d->i[k] = 0; continue; }
      /* (only an explicit T(x) converts Float -> Int: promotion goes the other way; the range is the TARGET's) */
      int ok = x == trunc(x);
      if (to == DFDB_U64) ok = ok && x >= 0.0 && x < 18446744073709551616.0;
      else ok = ok && x >= -9223372036854775808.0 && x < 9223372036854775808.0;
      if (!ok) { flag_err(c, ORC_ERR_ARGUMENT, k); d->i[k] = 0; }
      else d->i[k] = to == DFDB_U64 ? (int64_t)(uint64_t)x : wrap_int((int64_t)x, to);
    }
  } else if (dt_isfloat(to)) {
    if (dt_isint(from)) {
      if (to == DFDB_F32) { if (from == DFDB_U64) for (int64_t k = 0; k < n; k++) d->f[k] = (double)(float)(uint64_t)s->i[k]; else for (int64_t k = 0; k < n; k++) d->f[k] = (double)(float)s->i[k]; }
      else { if (from == DFDB_U64) for (int64_t k = 0; k < n; k++) d->f[k] = (double)(uint64_t)s->i[k]; else for (int64_t k = 0; k < n; k++) d->f[k] = (double)s->i[k]; }
    } else if (from == DFDB_BOOL) for (int64_t k = 0; k < n; k++) d->f[k] = s->b[k];
    else for (int64_t k = 0; k < n; k++) d->f[k] = to == DFDB_F32 ? (double)(float)s->f[k] : s->f[k];
  } else { /* to Bool */
    if (dt_isint(from)) for (int64_t k = 0; k < n; k++) { if (s->i[k] != 0 && s->i[k] != 1 && !(ms && ms[k])) flag_err(c, ORC_ERR_ARGUMENT, k); d->b[k] = s->i[k] != 0; }
    else for (int64_t k = 0; k < n; k++) { if (s->f[k] != 0 && s->f[k] != 1 && !(ms && ms[k])) flag_err(c, ORC_ERR_ARGUMENT, k); d->b[k] = s->f[k] != 0; }
  }
  return 0;
}

static inline double jl_fmod_mod(double x, double y) { /* Base.mod(x::Float, y) */
  double r = fmod(x, y);
  if (r == 0) return copysign(r, y);
  if ((r > 0) != (y > 0)) return r + y;
  return r;
}
static inline double jl_fmin(double x, double y) { if (x != x || y != y) return NAN; return (x < y || (x == y && signbit(x))) ? x : y; }
static inline double jl_fmax(double x, double y) { if (x != x || y != y) return NAN; return (x > y || (x == y && !signbit(x))) ? x : y; }

static int eval(const node_t* nd, ectx_t* c, vec_t* out);

static int str_elem(const vec_t* v, ectx_t* c, int64_t k, const uint8_t** p, int32_t* len) {
  if (v->scol) { int64_t r = c->idx ? c->idx[k] : k; *p = v->scol->sdata + v->scol->offsets[r]; *len = v->scol->sizes[r]; return 0; }
  *p = v->cstr; *len = v->cstr_len; return 0;
}
static int str_cmp(const uint8_t* a, int32_t la, const uint8_t* b, int32_t lb) { /* Base.cmp(::String, ::String) = memcmp then length */
  int32_t m = la < lb ? la : lb;
  int r = m > 0 ? memcmp(a, b, (size_t)m) : 0;
  if (r) return r < 0 ? -1 : 1;
  return la < lb ? -1 : (la > lb ? 1 : 0);
}

/* missing flags of a result: the union of its operands' (constants are never missing) */
static int miss_union(ectx_t* c, const vec_t* a, const vec_t* b, uint8_t** out) {
  const uint8_t* ma = a ? a->miss : NULL; const uint8_t* mb = b ? b->miss : NULL;
  *out = NULL;
  if (!ma && !mb) return 0;
  uint8_t* m = (uint8_t*)arena_alloc(c->ar, (size_t)(c->n > 0 ? c->n : 1));
  if (!m) return orc_fail(ORC_ERR_NOMEM, "oracle arena exhausted");
  for (int64_t k = 0; k < c->n; k++) m[k] = (uint8_t)((ma ? ma[k] : 0) | (mb ? mb[k] : 0));
  *out = m; return 0;
}

static int eval_binary_core(const node_t* nd, ectx_t* c, vec_t* out, vec_t* pva, vec_t* pvb);
static int eval_binary(const node_t* nd, ectx_t* c, vec_t* out) {
  vec_t va, vb; int rc;
  memset(&va, 0, sizeof va); memset(&vb, 0, sizeof vb);
  if (nd->op == DFIR_COALESCE) { /* coalesce(a, b): a where it is not missing, else b */
    if ((rc = eval(nd->a, c, &va)) || (rc = eval(nd->b, c, &vb))) return rc;
    int rt = dt_base(nd->dtype); int64_t n = c->n;
    if ((rc = vec_alloc(c, out, rt, n))) return rc;
    out->dtype = nd->dtype;
    int64_t sa = va.is_const ? 0 : 1, sb = vb.is_const ? 0 : 1;
    for (int64_t k = 0; k < n; k++) {
      int am = va.miss ? va.miss[k] : 0;
      if (out->i) out->i[k] = am ? vb.i[k * sb] : va.i[k * sa];
      else if (out->f) out->f[k] = am ? vb.f[k * sb] : va.f[k * sa];
      else out->b[k] = am ? vb.b[k * sb] : va.b[k * sa];
    }
    if (va.miss && vb.miss) {
      out->miss = (uint8_t*)arena_alloc(c->ar, (size_t)(n > 0 ? n : 1));
      if (!out->miss) return orc_fail(ORC_ERR_NOMEM, "oracle arena exhausted");
      for (int64_t k = 0; k < n; k++) out->miss[k] = va.miss[k] & vb.miss[k];
    }
    return 0;
  }
  if ((rc = eval_binary_core(nd, c, out, &va, &vb))) return rc;
  if (!out->miss && (nd->op < DFIR_AND || nd->op > DFIR_OR || dt_base(nd->dtype) != DFDB_BOOL))   /* Bool & / | set their own flags */
    rc = miss_union(c, &va, nd->op == DFIR_IN_SET ? NULL : &vb, &out->miss);
  return rc;
}

static int eval_binary_core(const node_t* nd, ectx_t* c, vec_t* out, vec_t* pva, vec_t* pvb) {
#define va (*pva)
#define vb (*pvb)
  int rc;
  if ((rc = eval(nd->a, c, &va))) return rc;
  if (nd->op == DFIR_IN_SET) {
    const node_t* s = nd->b; int64_t n = c->n;
    if ((rc = vec_alloc(c, out, DFDB_BOOL, n))) return rc;
    int at = dt_base(va.dtype); int sflt = dt_isfloat(s->set_dtype); int su = dt_base(s->set_dtype) == DFDB_U64;
    for (int64_t k = 0; k < n; k++) {
      int64_t ka = va.is_const ? 0 : k; int hit = 0;
      for (int32_t j = 0; j < s->nset && !hit; j++) {
        int cm;
        if (dt_isfloat(at)) cm = sflt ? cmp_ff(va.f[ka], s->set_f[j]) : (su ? -cmp_uf((uint64_t)s->set_i[j], va.f[ka]) : -cmp_if(s->set_i[j], va.f[ka]));
        else { int64_t x = at == DFDB_BOOL ? va.b[ka] : va.i[ka];
               cm = sflt ? (at == DFDB_U64 ? cmp_uf((uint64_t)x, s->set_f[j]) : cmp_if(x, s->set_f[j])) : cmp_ii(x, at == DFDB_U64, s->set_i[j], su); }
        if (cm == 2 || cm == -2) cm = 2;
        hit = cm == 0;
      }
      out->b[k] = (uint8_t)hit;
    }
    return 0;
  }
  if ((rc = eval(nd->b, c, &vb))) return rc;
  int64_t n = c->n; int op = nd->op;
  int ta = dt_base(va.dtype), tb = dt_base(vb.dtype);
  int64_t sa = va.is_const ? 0 : 1, sb = vb.is_const ? 0 : 1;

  /* ---- strings ---- */
  if (ta == DFDB_STRING) {
    if ((rc = vec_alloc(c, out, DFDB_BOOL, n))) return rc;
    for (int64_t k = 0; k < n; k++) {
      const uint8_t *pa, *pb; int32_t la, lb;
      str_elem(&va, c, k, &pa, &la); str_elem(&vb, c, k, &pb, &lb);
      if (la < 0) la = 0;
      if (lb < 0) lb = 0;
      int r;
      if (op == DFIR_STARTSWITH) r = la >= lb && (lb == 0 || memcmp(pa, pb, (size_t)lb) == 0);
      else if (op == DFIR_ENDSWITH) r = la >= lb && (lb == 0 || memcmp(pa + la - lb, pb, (size_t)lb) == 0);
      else r = cmp_to_bool(op, str_cmp(pa, la, pb, lb));
      out->b[k] = (uint8_t)r;
    }
    return 0;
  }

  /* ---- comparisons ---- */
  if (op >= DFIR_EQ && op <= DFIR_GE) {
    if ((rc = vec_alloc(c, out, DFDB_BOOL, n))) return rc;
    uint8_t* o = out->b;
    /* the fused fast loops Julia's broadcast compiles for same-type operands */
    if (ta == DFDB_I64 && tb == DFDB_I64) {
      const int64_t *x = va.i, *y = vb.i;
#define LOOP(OPR) for (int64_t k = 0; k < n; k++) o[k] = x[k * sa] OPR y[k * sb]
      switch (op) { case DFIR_EQ: LOOP(==); break; case DFIR_NE: LOOP(!=); break; case DFIR_LT: LOOP(<); break;
                    case DFIR_LE: LOOP(<=); break; case DFIR_GT: LOOP(>); break; default: LOOP(>=); }
      return 0;
    }
    if (ta == DFDB_F64 && tb == DFDB_F64) {
      const double *x = va.f, *y = vb.f;
      switch (op) { case DFIR_EQ: LOOP(==); break; case DFIR_NE: LOOP(!=); break; case DFIR_LT: LOOP(<); break;
                    case DFIR_LE: LOOP(<=); break; case DFIR_GT: LOOP(>); break; default: LOOP(>=); }
#undef LOOP
      return 0;
    }
    for (int64_t k = 0; k < n; k++) {
      int64_t ka = k * sa, kb = k * sb; int cm;
      int fa = dt_isfloat(ta), fb = dt_isfloat(tb);
      if (fa && fb) cm = cmp_ff(va.f[ka], vb.f[kb]);
      else if (fa) { int64_t y = tb == DFDB_BOOL ? vb.b[kb] : vb.i[kb]; int r = tb == DFDB_U64 ? cmp_uf((uint64_t)y, va.f[ka]) : cmp_if(y, va.f[ka]); cm = r == 2 ? 2 : -r; }
      else if (fb) { int64_t x = ta == DFDB_BOOL ? va.b[ka] : va.i[ka]; cm = ta == DFDB_U64 ? cmp_uf((uint64_t)x, vb.f[kb]) : cmp_if(x, vb.f[kb]); }
      else { int64_t x = ta == DFDB_BOOL ? va.b[ka] : va.i[ka]; int64_t y = tb == DFDB_BOOL ? vb.b[kb] : vb.i[kb]; cm = cmp_ii(x, ta == DFDB_U64, y, tb == DFDB_U64); }
      o[k] = (uint8_t)cmp_to_bool(op, cm);
    }
    return 0;
  }

  /* ---- logic ---- */
  int rt = dt_base(nd->dtype);
  if (op >= DFIR_AND && op <= DFIR_XOR && rt == DFDB_BOOL) {
    if ((rc = vec_alloc(c, out, DFDB_BOOL, n))) return rc;
    const uint8_t *x = va.b, *y = vb.b; uint8_t* o = out->b;
    if ((va.miss || vb.miss) && op != DFIR_XOR) { /* three-valued logic: false & missing = false, true | missing = true */
      uint8_t* m = (uint8_t*)arena_alloc(c->ar, (size_t)(n > 0 ? n : 1));
      if (!m) return orc_fail(ORC_ERR_NOMEM, "oracle arena exhausted");
      for (int64_t k = 0; k < n; k++) {
        int am = va.miss ? va.miss[k] : 0, bm = vb.miss ? vb.miss[k] : 0, a = x[k * sa] & 1, b = y[k * sb] & 1;
        int decided = op == DFIR_AND ? ((!am && !a) || (!bm && !b)) : ((!am && a) || (!bm && b));   /* one known operand settles it */
        m[k] = (uint8_t)((am | bm) && !decided);
        o[k] = (uint8_t)(decided ? (op == DFIR_OR) : (m[k] ? 0 : (op == DFIR_AND ? (a & b) : (a | b))));
      }
      out->miss = m;
      return 0;
    }
    if (op == DFIR_AND) for (int64_t k = 0; k < n; k++) o[k] = x[k * sa] & y[k * sb];
    else if (op == DFIR_OR) for (int64_t k = 0; k < n; k++) o[k] = x[k * sa] | y[k * sb];
    else for (int64_t k = 0; k < n; k++) o[k] = x[k * sa] ^ y[k * sb];
    return 0;
  }

  /* ---- arithmetic: convert both operands to the promoted compute type ---- */
  int ct = rt;
  if (op == DFIR_DIV) { int p = promote_num(ta, tb); ct = dt_isfloat(p) ? p : DFDB_F64; }
  if (op == DFIR_MUL && ta == DFDB_BOOL && tb == DFDB_BOOL) { /* true*true isa Bool */
    if ((rc = vec_alloc(c, out, DFDB_BOOL, n))) return rc;
    for (int64_t k = 0; k < n; k++) out->b[k] = va.b[k * sa] & vb.b[k * sb];
    return 0;
  }
  if ((op == DFIR_MIN || op == DFIR_MAX) && rt == DFDB_BOOL) {
    if ((rc = vec_alloc(c, out, DFDB_BOOL, n))) return rc;
    for (int64_t k = 0; k < n; k++) out->b[k] = op == DFIR_MIN ? (va.b[k * sa] & vb.b[k * sb]) : (va.b[k * sa] | vb.b[k * sb]);
    return 0;
  }
  vec_t xa, xb;
  if ((rc = conv(c, &va, ct, &xa))) return rc;
  if ((rc = conv(c, &vb, ct, &xb))) return rc;
  if ((rc = vec_alloc(c, out, rt, n))) return rc;
  if (dt_isfloat(ct)) {
    const double *x = xa.f, *y = xb.f; double* o = out->f; int f32 = ct == DFDB_F32;
    for (int64_t k = 0; k < n; k++) {
      double a = x[k * sa], b = y[k * sb], r;
      switch (op) {
        case DFIR_ADD: r = a + b; break; case DFIR_SUB: r = a - b; break; case DFIR_MUL: r = a * b; break;
        case DFIR_DIV: r = a / b; break;
        case DFIR_REM: r = fmod(a, b); break;
        case DFIR_MOD: r = jl_fmod_mod(a, b); break;
        case DFIR_IDIV: r = nearbyint((a - fmod(a, b)) / b); break; /* div(x,y) = round((x - rem(x,y))/y) */
        case DFIR_MIN: r = jl_fmin(a, b); break;
        default: r = jl_fmax(a, b); break;
      }
      o[k] = f32 ? (double)(float)r : r; /* double arithmetic then one rounding == Float32 arithmetic for + - * / */
    }
    return 0;
  }
  /* integers (wraparound like Julia native ints) */
  const int64_t *x = xa.i, *y = xb.i; int64_t* o = out->i; int uns = !dt_issigned(ct);
  if (op == DFIR_ADD && ct == DFDB_I64) { for (int64_t k = 0; k < n; k++) o[k] = (int64_t)((uint64_t)x[k * sa] + (uint64_t)y[k * sb]); return 0; }
  if (op == DFIR_MUL && ct == DFDB_I64) { for (int64_t k = 0; k < n; k++) o[k] = (int64_t)((uint64_t)x[k * sa] * (uint64_t)y[k * sb]); return 0; }
  for (int64_t k = 0; k < n; k++) {
    int64_t a = x[k * sa], b = y[k * sb], r = 0;
    switch (op) {
      case DFIR_ADD: r = (int64_t)((uint64_t)a + (uint64_t)b); break;
      case DFIR_SUB: r = (int64_t)((uint64_t)a - (uint64_t)b); break;
      case DFIR_MUL: r = (int64_t)((uint64_t)a * (uint64_t)b); break;
      case DFIR_AND: r = a & b; break; case DFIR_OR: r = a | b; break; case DFIR_XOR: r = a ^ b; break;
      case DFIR_MIN: r = uns && ct == DFDB_U64 ? ((uint64_t)a < (uint64_t)b ? a : b) : (a < b ? a : b); break;
      case DFIR_MAX: r = uns && ct == DFDB_U64 ? ((uint64_t)a > (uint64_t)b ? a : b) : (a > b ? a : b); break;
      case DFIR_REM: case DFIR_MOD: case DFIR_IDIV:
        if ((xa.miss && xa.miss[k * sa]) || (xb.miss && xb.miss[k * sb])) { r = 0; break; }   /* missing ÷ x is missing, not an error */
        if (b == 0) { flag_err(c, ORC_ERR_DIVIDE, k); r = 0; break; }
        if (uns) {
          uint64_t ua = (uint64_t)a, ub = (uint64_t)b;
          r = op == DFIR_IDIV ? (int64_t)(ua / ub) : (int64_t)(ua % ub);
        } else if (b == -1) {
          if (op == DFIR_IDIV) { if (a == int_min_of(ct)) { flag_err(c, ORC_ERR_DIVIDE, k); r = 0; } else r = -a; }
          else r = 0;
        } else if (op == DFIR_IDIV) r = a / b;
        else { r = a % b; if (op == DFIR_MOD && r != 0 && ((r < 0) != (b < 0))) r += b; }
        break;
    }
    o[k] = wrap_int(r, ct);
  }
  return 0;
#undef va
#undef vb
}

static int eval(const node_t* nd, ectx_t* c, vec_t* out) {
  int rc;
  switch (nd->op) {
    case DFIR_COL: return gather_col(c, &c->bufs[nd->col], out);
    case DFIR_CONST: {
      if ((rc = vec_alloc(c, out, dt_base(nd->cdtype), 1))) return rc;
      out->is_const = 1; out->n = c->n;
      if (out->i) out->i[0] = wrap_int(nd->ci, dt_base(nd->cdtype)); else if (out->f) out->f[0] = nd->cf; else out->b[0] = nd->ci != 0;
      return 0;
    }
    case DFIR_CONST_STR: memset(out, 0, sizeof *out); out->dtype = DFDB_STRING; out->is_const = 1; out->n = c->n; out->cstr = nd->str; out->cstr_len = nd->slen; return 0;
    case DFIR_NOT: {
      vec_t a; if ((rc = eval(nd->a, c, &a))) return rc;
      if ((rc = vec_alloc(c, out, DFDB_BOOL, c->n))) return rc;
      for (int64_t k = 0; k < c->n; k++) out->b[k] = !a.b[a.is_const ? 0 : k];
      out->miss = a.miss;
      return 0;
    }
    case DFIR_ISMISSING: {
      if (nd->a->op != DFIR_COL) { /* ismissing of a computed value */
        vec_t a; if ((rc = eval(nd->a, c, &a))) return rc;
        if ((rc = vec_alloc(c, out, DFDB_BOOL, c->n))) return rc;
        for (int64_t k = 0; k < c->n; k++) out->b[k] = a.miss ? a.miss[k] : 0;
        return 0;
      }
      const colbuf_t* b = &c->bufs[nd->a->col];
      if ((rc = vec_alloc(c, out, DFDB_BOOL, c->n))) return rc;
      for (int64_t k = 0; k < c->n; k++) {
        int64_t r = c->idx ? c->idx[k] : k;
        out->b[k] = dt_base(b->dtype) == DFDB_STRING ? (b->sizes[r] < 0) : (dt_nullable(b->dtype) ? b->missing[r] : 0);
      }
      return 0;
    }
    case DFIR_SIZEOF: {
      vec_t a; if ((rc = eval(nd->a, c, &a))) return rc;
      if ((rc = vec_alloc(c, out, DFDB_I64, c->n))) return rc;
      for (int64_t k = 0; k < c->n; k++) { const uint8_t* p; int32_t l; str_elem(&a, c, k, &p, &l); out->i[k] = l < 0 ? 0 : l; }
      out->miss = a.miss;
      return 0;
    }
    case DFIR_NEG: case DFIR_ABS: {
      vec_t a, x; if ((rc = eval(nd->a, c, &a))) return rc;
      int rt = dt_base(nd->dtype);
      if ((rc = conv(c, &a, rt, &x))) return rc;
      int64_t n = x.is_const ? 1 : c->n;
      if ((rc = vec_alloc(c, out, rt, n))) return rc;
      out->is_const = x.is_const; out->n = c->n; out->miss = a.miss;
      if (rt == DFDB_BOOL) { for (int64_t k = 0; k < n; k++) out->b[k] = x.b[k]; return 0; }
      if (dt_isfloat(rt)) for (int64_t k = 0; k < n; k++) out->f[k] = nd->op == DFIR_NEG ? -x.f[k] : fabs(x.f[k]);
      else for (int64_t k = 0; k < n; k++) {
        int64_t v = x.i[k];
        if (nd->op == DFIR_NEG || (dt_issigned(rt) && v < 0)) v = (int64_t)(0 - (uint64_t)v);
        out->i[k] = wrap_int(v, rt);
      }
      return 0;
    }
    case DFIR_CAST: {
      /* T(x) / convert(T, x): exact or InexactError (Base: Int8(300), Int8(300.0), UInt64(-1), UInt64(-1.0), Int64(typemax(UInt64)) throw);
       * conv() itself wraps, which is what the implicit promotions of arithmetic do (`a % T`, Base int.jl), so the range is checked here */
      vec_t a; if ((rc = eval(nd->a, c, &a))) return rc;
      {
        const int from = dt_base(a.dtype), to = dt_base(nd->cast_to);
        const int64_t na = a.is_const ? 1 : a.n;
        if (dt_isint(to) && from != to && from != DFDB_BOOL) {
          int64_t lo = int_min_of(to), hi;
          switch (to) { case DFDB_I8: hi = 127; break; case DFDB_I16: hi = 32767; break; case DFDB_I32: hi = 2147483647LL; break;
                        case DFDB_U8: hi = 255; break; case DFDB_U16: hi = 65535; break; case DFDB_U32: hi = 4294967295LL; break; default: hi = INT64_MAX; }
          for (int64_t k = 0; k < na; k++) {
            if (a.miss && a.miss[k]) continue;
            int ok;
            if (dt_isfloat(from)) {
              double x = a.f[k];
              if (to == DFDB_U64) ok = x == trunc(x) && x >= 0.0 && x < 18446744073709551616.0;
              else if (to == DFDB_I64) ok = x == trunc(x) && x >= -9223372036854775808.0 && x < 9223372036854775808.0;
              else ok = x == trunc(x) && x >= (double)lo && x <= (double)hi;
            } else if (from == DFDB_U64) {
              uint64_t u = (uint64_t)a.i[k];
              ok = to == DFDB_U64 ? 1 : u <= (uint64_t)hi;
            } else {
              int64_t v = a.i[k];
              ok = to == DFDB_U64 ? v >= 0 : (v >= lo && v <= hi);
            }
            if (!ok) flag_err(c, ORC_ERR_ARGUMENT, k);
          }
        }
      }
      vec_t x; if ((rc = conv(c, &a, dt_base(nd->cast_to), &x))) return rc;
      *out = x; out->dtype = nd->dtype; out->miss = a.miss;
      return 0;
    }
    default: return eval_binary(nd, c, out);
  }
}

int expr_eval(const node_t* nd, const colbuf_t* bufs, const int32_t* idx, int64_t n, arena_t* ar, vec_t* out) {
  ectx_t c = {bufs, idx, n, ar, 0, {INT64_MAX, INT64_MAX}};
  vec_t v; int rc = eval(nd, &c, &v); if (rc) return rc;
  if (n == 0) c.err = 0;   /* no row reached this evaluation: a broadcast over nothing calls nothing, so even a constant sub-expression that always throws
                              (`UInt16(-42.0)`) does not (found by tests/test_gpu_fuzz.py: the engine was right) */
  if (c.err && c.erow[0] <= c.erow[1]) return orc_fail(ORC_ERR_DIVIDE, "DivideError: integer division error");   /* the earlier row's error (same row: the division) */
  if (c.err) return orc_fail(ORC_ERR_ARGUMENT, "InexactError in conversion");
  if (v.is_const && !v.scol && !v.cstr) { /* broadcast a scalar result to n elements */
    vec_t w; int rt = dt_base(v.dtype); ectx_t c2 = c;
    if ((rc = vec_alloc(&c2, &w, rt, n))) return rc;
    for (int64_t k = 0; k < n; k++) { if (w.i) w.i[k] = v.i[0]; else if (w.f) w.f[k] = v.f[0]; else w.b[k] = v.b[0]; }
    v = w;
  }
  *out = v; return 0;
}
