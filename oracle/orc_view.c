/* orc_view.c — DFView algebra, SelectionExecutor, ProjectionExecutor, BlocksIterator and
 * materialize/nrow of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see oracle.h).  Restates:
 *   src/tables/selection.jl:4-60     (SelectionQueue + composition rules)
 *   src/tables/selection.jl:68-196   (RangeToProcess, SelectionExecutor, apply, skip_if_can, is_finished)
 *   src/tables/projection.jl:1-154   (Projection, ProjectionExecutor.eval_on_range)
 *   src/io/blocksiterator.jl:20-145  (BlocksIterator data / size readers, late materialization)
 *   src/tables/view.jl:60-118,183-206 (selection(), projection(), required_columns, nrow)
 *   src/tables/materialization.jl:27-52 (materialize)
 *   src/tables/column.jl:102-126     (left-to-right element iteration, used by sum)
 */
#include "orc_internal.h"
#include <time.h>

enum { ST_RANGE = 0, ST_INTEGER = 1, ST_INDICES = 2, ST_PRED = 3 };

typedef struct {
  int kind;
  int64_t start, step, stop; /* ST_RANGE: Julia a:s:b with b normalised to the last element; n = length */
  int64_t n;
  int64_t* idx;              /* ST_INDICES / ST_INTEGER (n = 1) in caller order */
  int64_t* sorted;           /* ascending copy for `in` */
  node_t* pred;
  /* RangeToProcess state (selection.jl:68-75) */
  int64_t offset, first, last;
} stage_t;

typedef struct { char name[128]; node_t* expr; } projcol_t;

struct orc_view {
  orc_table* t;
  int nstages; stage_t* stages;
  int nproj; projcol_t* proj;
};

/* ---------------------------------------------------------------- ranges */
static int64_t range_len(int64_t a, int64_t s, int64_t b) {
  if (s > 0) return b < a ? 0 : (b - a) / s + 1;
  return b > a ? 0 : (a - b) / (-s) + 1;
}
static int cmp_i64(const void* a, const void* b) { int64_t x = *(const int64_t*)a, y = *(const int64_t*)b; return x < y ? -1 : x > y; }

static void stage_free(stage_t* s) { free(s->idx); free(s->sorted); expr_free(s->pred); memset(s, 0, sizeof *s); }
static void stage_finish(stage_t* s) { /* RangeToProcess(range): first = minimum, last = maximum */
  s->offset = 0;
  if (s->kind == ST_RANGE) {
    s->n = range_len(s->start, s->step, s->stop);
    if (s->n > 0) s->stop = s->start + (s->n - 1) * s->step;
    s->first = s->step > 0 ? s->start : s->stop;
    s->last = s->step > 0 ? s->stop : s->start;
    if (s->n == 0) { s->first = INT64_MAX; s->last = INT64_MIN; } /* minimum(empty) throws in Julia; never matches here */
  } else if (s->kind == ST_INDICES || s->kind == ST_INTEGER) {
    free(s->sorted); s->sorted = (int64_t*)malloc(8 * (size_t)(s->n ? s->n : 1));
    memcpy(s->sorted, s->idx, 8 * (size_t)s->n);
    qsort(s->sorted, (size_t)s->n, 8, cmp_i64);
    s->first = s->n ? s->sorted[0] : INT64_MAX; s->last = s->n ? s->sorted[s->n - 1] : INT64_MIN;
  }
}
static int stage_contains(const stage_t* s, int64_t v) {
  if (s->kind == ST_RANGE) {
    if (s->n == 0 || v < s->first || v > s->last) return 0;
    int64_t st = s->step > 0 ? s->step : -s->step;
    return (v - s->first) % st == 0;
  }
  int64_t lo = 0, hi = s->n;
  while (lo < hi) { int64_t m = (lo + hi) / 2; if (s->sorted[m] < v) lo = m + 1; else hi = m; }
  return lo < s->n && s->sorted[lo] == v;
}
/* element k (1-based) of a range-like stage, with Julia bounds checking */
static int stage_elem(const stage_t* s, int64_t k, int64_t* out) {
  if (k < 1 || k > s->n) return orc_fail(ORC_ERR_BOUNDS, "BoundsError: attempt to access %lld-element range at index [%lld]", (long long)s->n, (long long)k);
  *out = s->kind == ST_RANGE ? s->start + (k - 1) * s->step : s->idx[k - 1];
  return 0;
}

/* ---------------------------------------------------------------- view */
int orc_view_new(orc_table* t, orc_view** out) {
  orc_view* v = (orc_view*)calloc(1, sizeof *v);
  v->t = t;
  v->nproj = t->ncols; v->proj = (projcol_t*)calloc((size_t)t->ncols + 1, sizeof(projcol_t));
  for (int i = 0; i < t->ncols; i++) { /* full_table_projection: view.jl:43-48 */
    snprintf(v->proj[i].name, sizeof v->proj[i].name, "%s", t->cols[i].name);
    node_t* n = (node_t*)calloc(1, sizeof *n); n->op = DFIR_COL; n->col = i; n->dtype = t->cols[i].dtype;
    v->proj[i].expr = n;
  }
  *out = v; return 0;
}
void orc_view_free(orc_view* v) {
  if (!v) return;
  for (int i = 0; i < v->nstages; i++) stage_free(&v->stages[i]);
  for (int i = 0; i < v->nproj; i++) expr_free(v->proj[i].expr);
  free(v->stages); free(v->proj); free(v);
}
int orc_view_nstages(orc_view* v) { return v->nstages; }
int orc_view_stage(orc_view* v, int i, int* kind, int64_t* start, int64_t* step, int64_t* stop, int64_t* n) {
  if (i < 0 || i >= v->nstages) return orc_fail(ORC_ERR_BOUNDS, "stage %d out of range", i);
  stage_t* s = &v->stages[i];
  if (kind) *kind = s->kind;
  if (start) *start = s->kind == ST_RANGE ? s->start : (s->n ? s->idx[0] : 0);
  if (step) *step = s->step;
  if (stop) *stop = s->kind == ST_RANGE ? s->stop : (s->n ? s->idx[s->n - 1] : 0);
  if (n) *n = s->n;
  return 0;
}

/* add(q, elem) with _new_queue's rules (selection.jl:39-49,57-60) */
static int push_stage(orc_view* v, stage_t* ns) {
  stage_t* last = v->nstages ? &v->stages[v->nstages - 1] : NULL;
  int new_is_range = ns->kind != ST_PRED;
  if (last && last->kind != ST_PRED && new_is_range) {
    /* range∘range collapses to old[elem] (selection.jl:40) */
    stage_t r; memset(&r, 0, sizeof r); int rc = 0;
    if (last->kind == ST_INTEGER) { /* Number indexing: only x[1] exists */
      if (!(ns->kind == ST_INTEGER && ns->idx[0] == 1)) { stage_free(ns); return orc_fail(ORC_ERR_BOUNDS, "BoundsError: indexing a scalar selection"); }
      stage_free(ns); return 0;
    }
    if (ns->kind == ST_INTEGER) {
      int64_t e; rc = stage_elem(last, ns->idx[0], &e);
      if (!rc) { r.kind = ST_INTEGER; r.n = 1; r.idx = (int64_t*)malloc(8); r.idx[0] = e; }
    } else if (last->kind == ST_RANGE && ns->kind == ST_RANGE) {
      if (ns->n == 0) { r.kind = ST_RANGE; r.start = last->start; r.step = last->step * ns->step; r.stop = r.start - r.step; }
      else {
        int64_t e0, e1; rc = stage_elem(last, ns->start, &e0); if (!rc) rc = stage_elem(last, ns->stop, &e1);
        if (!rc) { r.kind = ST_RANGE; r.start = e0; r.step = last->step * ns->step; r.stop = e1; }
      }
    } else {
      r.kind = ST_INDICES; r.n = ns->n; r.idx = (int64_t*)malloc(8 * (size_t)(ns->n ? ns->n : 1));
      for (int64_t k = 0; k < ns->n && !rc; k++) {
        int64_t pos = ns->kind == ST_RANGE ? ns->start + k * ns->step : ns->idx[k];
        rc = stage_elem(last, pos, &r.idx[k]);
      }
    }
    stage_free(ns);
    if (rc) { stage_free(&r); return rc; }
    stage_free(last); *last = r; stage_finish(last);
    return 0;
  }
  if (last && last->kind == ST_PRED && !new_is_range) { /* predicate∘predicate fuses with & (selection.jl:44-47) */
    last->pred = expr_and(last->pred, ns->pred); ns->pred = NULL; stage_free(ns); return 0;
  }
  v->stages = (stage_t*)realloc(v->stages, sizeof(stage_t) * (size_t)(v->nstages + 1));
  v->stages[v->nstages++] = *ns;
  return 0;
}

int orc_view_add_range(orc_view* v, int64_t start, int64_t step, int64_t stop) {
  if (step == 0) return orc_fail(ORC_ERR_ARGUMENT, "ArgumentError: step cannot be zero");
  stage_t s; memset(&s, 0, sizeof s); s.kind = ST_RANGE; s.start = start; s.step = step; s.stop = stop; stage_finish(&s);
  return push_stage(v, &s);
}
int orc_view_add_integer(orc_view* v, int64_t i) {
  stage_t s; memset(&s, 0, sizeof s); s.kind = ST_INTEGER; s.n = 1; s.idx = (int64_t*)malloc(8); s.idx[0] = i; stage_finish(&s);
  return push_stage(v, &s);
}
int orc_view_add_indices(orc_view* v, const int64_t* idx, int64_t n) {
  stage_t s; memset(&s, 0, sizeof s); s.kind = ST_INDICES; s.n = n; s.idx = (int64_t*)malloc(8 * (size_t)(n ? n : 1));
  memcpy(s.idx, idx, 8 * (size_t)n); stage_finish(&s);
  return push_stage(v, &s);
}
int orc_view_add_predicate(orc_view* v, const uint8_t* ir, size_t len) {
  node_t* n; int rc = expr_parse(v->t, ir, len, &n); if (rc) return rc;
  if (n->dtype != DFDB_BOOL) { expr_free(n); return orc_fail(ORC_ERR_ARGUMENT, "Function for selection must have Bool result type"); } /* selection.jl:52-55 */
  stage_t s; memset(&s, 0, sizeof s); s.kind = ST_PRED; s.pred = n;
  return push_stage(v, &s);
}

int orc_view_set_projection(orc_view* v, int n, const char* const* names, const uint8_t* const* irs, const size_t* lens) {
  projcol_t* np = (projcol_t*)calloc((size_t)n + 1, sizeof(projcol_t)); int rc = 0;
  for (int i = 0; i < n && !rc; i++) {
    for (int j = 0; j < i; j++) if (strcmp(names[j], names[i]) == 0) rc = orc_fail(ORC_ERR_ARGUMENT, "Duplicated column %s", names[i]); /* projection.jl:25-28 */
    if (!rc) { snprintf(np[i].name, sizeof np[i].name, "%s", names[i]); rc = expr_parse(v->t, irs[i], lens[i], &np[i].expr); }
  }
  if (rc) { for (int i = 0; i < n; i++) expr_free(np[i].expr); free(np); return rc; }
  for (int i = 0; i < v->nproj; i++) expr_free(v->proj[i].expr);
  free(v->proj); v->proj = np; v->nproj = n; return 0;
}
int orc_view_ncols(orc_view* v) { return v->nproj; }
int orc_view_coltype(orc_view* v, int i, int32_t* dtype) {
  if (i < 0 || i >= v->nproj) return orc_fail(ORC_ERR_BOUNDS, "projection column %d out of range", i);
  *dtype = v->proj[i].expr->dtype; return 0;
}

static int sel_required(orc_view* v, int32_t* o, int cap) { int c = 0; for (int i = 0; i < v->nstages; i++) if (v->stages[i].kind == ST_PRED) expr_required(v->stages[i].pred, o, &c, cap); return c; }
static int proj_required(orc_view* v, int32_t* o, int cap) { int c = 0; for (int i = 0; i < v->nproj; i++) expr_required(v->proj[i].expr, o, &c, cap); return c; }
int orc_view_required_columns(orc_view* v, int32_t* ordinals, int cap) { /* view.jl:183-190: unique([proj..., sel...]) */
  int c = proj_required(v, ordinals, cap);
  int32_t tmp[256]; int sc = sel_required(v, tmp, 256);
  for (int i = 0; i < sc; i++) { int dup = 0; for (int j = 0; j < c; j++) dup |= ordinals[j] == tmp[i]; if (!dup && c < cap) ordinals[c++] = tmp[i]; }
  return c;
}

/* ---------------------------------------------------------------- SelectionExecutor */
struct orc_selexec {
  orc_table* t; int nstages; stage_t* stages; /* private copies carrying the mutable offsets */
  int32_t* index; size_t index_cap;           /* positions of the trues (Base.LogicalIndex) */
  arena_t arena;
  colbuf_t* ext;                              /* orc_selexec_apply's borrowed block */
};

static orc_selexec* selexec_make(orc_view* v) {
  orc_selexec* e = (orc_selexec*)calloc(1, sizeof *e);
  e->t = v->t; e->nstages = v->nstages; e->stages = (stage_t*)calloc((size_t)v->nstages + 1, sizeof(stage_t));
  for (int i = 0; i < v->nstages; i++) {
    stage_t* s = &e->stages[i]; *s = v->stages[i];
    s->idx = NULL; s->sorted = NULL; s->pred = expr_clone(v->stages[i].pred);
    if (v->stages[i].idx) { s->idx = (int64_t*)malloc(8 * (size_t)(s->n ? s->n : 1)); memcpy(s->idx, v->stages[i].idx, 8 * (size_t)s->n); }
    stage_finish(s);
  }
  return e;
}
static void selexec_destroy(orc_selexec* e) {
  if (!e) return;
  for (int i = 0; i < e->nstages; i++) stage_free(&e->stages[i]);
  free(e->stages); free(e->index); arena_free(&e->arena);
  if (e->ext) { for (int i = 0; i < e->t->ncols; i++) free(e->ext[i].offsets); free(e->ext); }
  free(e);
}
int orc_selexec_new(orc_view* v, orc_selexec** out) { *out = selexec_make(v); return 0; }
void orc_selexec_free(orc_selexec* e) { selexec_destroy(e); }

static int64_t build_index(orc_selexec* e, const uint8_t* mask, int64_t rows) { /* Base.LogicalIndex(mask) */
  if ((size_t)rows > e->index_cap) { e->index_cap = (size_t)rows * 2; e->index = (int32_t*)realloc(e->index, e->index_cap * 4); }
  int64_t n = 0;
  for (int64_t k = 0; k < rows; k++) { e->index[n] = (int32_t)k; n += mask[k]; }
  return n;
}

/* apply (selection.jl:161-167) = fill!(mask, true) then _apply_to_block per stage */
static int selexec_apply(orc_selexec* e, int64_t rows, const colbuf_t* bufs, uint8_t* mask, int64_t* count) {
  memset(mask, 1, (size_t)rows);                                   /* fill!(s.range_buffer, 1) :163 */
  for (int si = 0; si < e->nstages; si++) {
    stage_t* s = &e->stages[si];
    int64_t n = build_index(e, mask, rows);                        /* index = Base.LogicalIndex(range) :95,134 */
    if (s->kind != ST_PRED) {                                      /* range stage: selection.jl:94-111 */
      /* inblock_part = intersect((1:n) .+ offset, range) .- offset; survivors outside it are cleared */
      for (int64_t i = 0; i < n; i++) if (!stage_contains(s, s->offset + i + 1)) mask[e->index[i]] = 0;
      s->offset += n;                                              /* :107 */
    } else {                                                       /* predicate stage: selection.jl:133-157 */
      arena_reset(&e->arena);
      vec_t r; int rc = expr_eval(s->pred, bufs, n == rows ? NULL : e->index, n, &e->arena, &r); if (rc) return rc;
      if (n == rows) memcpy(mask, r.b, (size_t)rows);              /* :137-140 */
      else for (int64_t i = 0; i < n; i++) mask[e->index[i]] = r.b[i]; /* :142-146 */
    }
    if (rows == 0) break;                                          /* isempty(range) ? view(range, 1:0) : … */
  }
  int64_t c = 0; for (int64_t k = 0; k < rows; k++) c += mask[k];  /* LogicalIndex counts the trues :166 */
  *count = c; return 0;
}
static int selexec_isonly_range(orc_selexec* e) { for (int i = 0; i < e->nstages; i++) if (e->stages[i].kind == ST_PRED) return 0; return 1; } /* :169-175 */
int orc_selexec_skip_if_can(orc_selexec* e, int64_t size) { /* selection.jl:177-190: first stage only */
  if (e->nstages == 0 || e->stages[0].kind == ST_PRED) return 0;
  stage_t* s = &e->stages[0];
  if (s->first - s->offset > size) { s->offset += size; return 1; }
  return 0;
}
int orc_selexec_is_finished(orc_selexec* e) { /* selection.jl:192-196 */
  for (int i = 0; i < e->nstages; i++) if (e->stages[i].kind != ST_PRED && e->stages[i].last <= e->stages[i].offset) return 1;
  return 0;
}
int orc_selexec_apply(orc_selexec* e, int64_t rows, const void* const* cols, uint8_t* mask, int64_t* n) {
  if (!e->ext) e->ext = (colbuf_t*)calloc((size_t)e->t->ncols + 1, sizeof(colbuf_t));
  for (int i = 0; i < e->t->ncols; i++) {
    colbuf_t* b = &e->ext[i]; int64_t* keep = b->offsets;
    memset(b, 0, sizeof *b); b->offsets = keep;
    b->dtype = e->t->cols[i].dtype; b->rows = rows; b->external = 1;
    if (cols && cols[i]) {
      if (dt_base(b->dtype) == DFDB_STRING) return orc_fail(ORC_ERR_UNSUPPORTED, "selexec_apply takes fixed-width blocks only");
      b->data = (uint8_t*)cols[i];
    }
  }
  return selexec_apply(e, rows, e->ext, mask, n);
}

/* ---------------------------------------------------------------- BlocksIterator */
typedef struct {
  orc_view* v; orc_selexec* sel;
  int nreq; int32_t req[256];          /* streams in required_columns order */
  int nsel; int32_t selc[256];         /* sel_cols */
  int nprojc; int32_t projc[256];      /* proj_cols = setdiff(required(projection), sel_cols) */
  stream_t* streams; colbuf_t* bufs;   /* indexed by table ordinal */
  uint8_t* mask; size_t mask_cap;
  int64_t rows_before;                 /* table rows in blocks already passed (for select_indices) */
  int64_t block_rows;                  /* rows of the block just yielded */
  int64_t count;                       /* trues of the block just yielded */
  int size_reader;
  arena_t arena;
} iter_t;

static int has_ord(const int32_t* a, int n, int32_t x) { for (int i = 0; i < n; i++) if (a[i] == x) return 1; return 0; }

static int iter_open(iter_t* it, orc_view* v, int size_reader) { /* blocksiterator.jl:20-66 */
  memset(it, 0, sizeof *it);
  it->v = v; it->size_reader = size_reader; it->sel = selexec_make(v);
  orc_table* t = v->t;
  it->nsel = sel_required(v, it->selc, 256);
  int32_t pr[256]; int npr = proj_required(v, pr, 256);
  if (it->nsel == 0 && npr > 0) { it->selc[0] = pr[0]; it->nsel = 1; }        /* :30 / :49 (quirk Q5) */
  if (size_reader) { it->nprojc = 0; it->nreq = it->nsel; memcpy(it->req, it->selc, sizeof(int32_t) * (size_t)it->nsel); }
  else {
    for (int i = 0; i < npr; i++) if (!has_ord(it->selc, it->nsel, pr[i])) it->projc[it->nprojc++] = pr[i]; /* :31-33 */
    it->nreq = orc_view_required_columns(v, it->req, 256);
  }
  it->streams = (stream_t*)calloc((size_t)t->ncols + 1, sizeof(stream_t));
  it->bufs = (colbuf_t*)calloc((size_t)t->ncols + 1, sizeof(colbuf_t));
  for (int i = 0; i < t->ncols; i++) it->bufs[i].dtype = t->cols[i].dtype;
  for (int i = 0; i < it->nreq; i++) { col_t* c = &t->cols[it->req[i]]; stream_init(&it->streams[it->req[i]], c->image.p, c->image.n, c->data_off); }
  return 0;
}
static void iter_close(iter_t* it) {
  if (!it->v) return;
  orc_table* t = it->v->t;
  for (int i = 0; i < t->ncols; i++) { stream_free(&it->streams[i]); colbuf_free(&it->bufs[i]); }
  free(it->streams); free(it->bufs); free(it->mask); selexec_destroy(it->sel); arena_free(&it->arena);
  it->v = NULL;
}

/* skipblocks (blocksiterator.jl:69-78): true = stop */
static int iter_skipblocks(iter_t* it, int* stop) {
  stream_t* first = &it->streams[it->req[0]];
  while (!stream_eof(first)) {
    if (orc_selexec_is_finished(it->sel)) { *stop = 1; return 0; }
    if (!orc_selexec_skip_if_can(it->sel, it->v->t->block_size)) { *stop = 0; return 0; }
    orc_sizestats st = {0, 0, 0};
    for (int i = 0; i < it->nreq; i++) { int rc = stream_skip_block(&it->streams[it->req[i]], &st); if (rc) return rc; }
    it->rows_before += st.rows;
  }
  *stop = 1; return 0;
}

/* one step of Base.iterate (blocksiterator.jl:98-145); *done = 1 at the end. Yields only blocks with survivors. */
static int iter_next(iter_t* it, int* done) {
  for (;;) {
    int stop = 1;
    if (it->nreq > 0) { int rc = iter_skipblocks(it, &stop); if (rc) return rc; }
    if (stop) { *done = 1; return 0; }
    it->rows_before += it->block_rows; it->block_rows = 0;
    orc_sizestats sz = {0, 0, 0}; int rc;
    int only_range = it->size_reader && selexec_isonly_range(it->sel);          /* :135 */
    for (int i = 0; i < it->nsel; i++) {
      rc = only_range ? stream_skip_block(&it->streams[it->selc[i]], &sz) : stream_read_block(&it->streams[it->selc[i]], &it->bufs[it->selc[i]], &sz);
      if (rc) return rc;
    }
    int64_t rows = sz.rows;
    if ((size_t)rows > it->mask_cap) { it->mask_cap = (size_t)rows * 2 + 64; it->mask = (uint8_t*)realloc(it->mask, it->mask_cap); }
    rc = selexec_apply(it->sel, rows, it->bufs, it->mask, &it->count); if (rc) return rc; /* :111 / :137 */
    it->block_rows = rows;
    if (it->count == 0) {                                                        /* late materialization :112-113 (Q6) */
      orc_sizestats st; for (int i = 0; i < it->nprojc; i++) { rc = stream_skip_block(&it->streams[it->projc[i]], &st); if (rc) return rc; }
      continue;
    }
    for (int i = 0; i < it->nprojc; i++) { rc = stream_read_block(&it->streams[it->projc[i]], &it->bufs[it->projc[i]], &sz); if (rc) return rc; } /* :115 */
    *done = 0; return 0;
  }
}

/* ---------------------------------------------------------------- nrow */
int orc_nrow(orc_view* v, int64_t* n) { /* view.jl:192-206 */
  iter_t it; iter_open(&it, v, 1);
  int64_t res = 0; int rc = 0, done = 0;
  for (;;) { rc = iter_next(&it, &done); if (rc || done) break; res += it.count; }
  iter_close(&it); *n = res; return rc;
}

/* ---------------------------------------------------------------- materialize */
typedef struct { bytes_t data, bytes, missing; int64_t count; } accum_t;

/* ProjectionExecutor.eval_on_range (projection.jl:128-154) + append! (materialization.jl:33-37) */
static int project_block(iter_t* it, accum_t* acc) {
  orc_view* v = it->v; int64_t rows = it->block_rows, cnt = it->count;
  orc_selexec* e = it->sel;
  int64_t n = build_index(e, it->mask, rows);
  const int32_t* idx = n == rows ? NULL : e->index;
  (void)cnt;
  for (int p = 0; p < v->nproj; p++) {
    node_t* ex = v->proj[p].expr; accum_t* a = &acc[p]; int rc;
    if (ex->op == DFIR_COL) { /* ColProjExec: buffer .= data[name][range] */
      colbuf_t* b = &it->bufs[ex->col];
      if (dt_base(b->dtype) == DFDB_STRING) { /* FlatStringsVector gather (FlatStringsVectors.jl:136-157) */
        if ((rc = bytes_reserve(&a->data, a->data.n + (size_t)n * 4))) return rc;
        int32_t* so = (int32_t*)(a->data.p + a->data.n);
        for (int64_t k = 0; k < n; k++) {
          int64_t r = idx ? idx[k] : k; int32_t s = b->sizes[r]; so[k] = s;
          if (s > 0) { if ((rc = bytes_append(&a->bytes, b->sdata + b->offsets[r], (size_t)s))) return rc; }
        }
        a->data.n += (size_t)n * 4;
      } else {
        int w = dt_width(b->dtype);
        if ((rc = bytes_reserve(&a->data, a->data.n + (size_t)n * w))) return rc;
        uint8_t* d = a->data.p + a->data.n;
        if (!idx) memcpy(d, b->data, (size_t)n * w);
        else if (w == 8) { const int64_t* s = (const int64_t*)b->data; int64_t* o = (int64_t*)d; for (int64_t k = 0; k < n; k++) o[k] = s[idx[k]]; }
        else for (int64_t k = 0; k < n; k++) memcpy(d + k * w, b->data + (size_t)idx[k] * w, (size_t)w);
        a->data.n += (size_t)n * w;
        if (dt_nullable(b->dtype)) {
          if ((rc = bytes_reserve(&a->missing, a->missing.n + (size_t)n))) return rc;
          for (int64_t k = 0; k < n; k++) a->missing.p[a->missing.n + k] = b->missing[idx ? idx[k] : k];
          a->missing.n += (size_t)n;
        }
      }
    } else { /* BroadcastExecutor (computed column) */
      arena_reset(&it->arena);
      vec_t r; if ((rc = expr_eval(ex, it->bufs, idx, n, &it->arena, &r))) return rc;
      int w = dt_width(ex->dtype);
      if ((rc = bytes_reserve(&a->data, a->data.n + (size_t)n * w))) return rc;
      uint8_t* d = a->data.p + a->data.n;
      switch (dt_base(ex->dtype)) {
#define PUT(T, src) do { T* o = (T*)d; for (int64_t k = 0; k < n; k++) o[k] = (T)src[k]; } while (0)
        case DFDB_I8: PUT(int8_t, r.i); break;   case DFDB_I16: PUT(int16_t, r.i); break; case DFDB_I32: PUT(int32_t, r.i); break;
        case DFDB_I64: PUT(int64_t, r.i); break; case DFDB_U8: PUT(uint8_t, r.i); break;  case DFDB_U16: PUT(uint16_t, r.i); break;
        case DFDB_U32: PUT(uint32_t, r.i); break; case DFDB_U64: PUT(int64_t, r.i); break;
        case DFDB_F32: PUT(float, r.f); break;   case DFDB_F64: PUT(double, r.f); break;  case DFDB_BOOL: PUT(uint8_t, r.b); break;
#undef PUT
        default: return orc_fail(ORC_ERR_UNSUPPORTED, "computed String columns are outside the IR");
      }
      a->data.n += (size_t)n * w;
      if (dt_nullable(ex->dtype)) { /* Union{R,Missing} result: the flags travel beside the values */
        if ((rc = bytes_reserve(&a->missing, a->missing.n + (size_t)n))) return rc;
        for (int64_t k = 0; k < n; k++) a->missing.p[a->missing.n + k] = r.miss ? r.miss[k] : 0;
        a->missing.n += (size_t)n;
      }
    }
    a->count += n;
  }
  return 0;
}

static int materialize_impl(orc_view* v, orc_outcol* outs, int ncols, int count_pass) {
  if (ncols != v->nproj) return orc_fail(ORC_ERR_ARGUMENT, "expected %d output columns", v->nproj);
  int rc = 0; int64_t rows = 0;
  accum_t* acc = (accum_t*)calloc((size_t)v->nproj + 1, sizeof(accum_t));
  if (count_pass) { /* rows = nrow(v); sizehint! (materialization.jl:29-32, quirk Q8) */
    rc = orc_nrow(v, &rows);
    for (int p = 0; p < v->nproj && !rc; p++) {
      int w = dt_width(v->proj[p].expr->dtype); if (w == 0) w = 4;
      rc = bytes_reserve(&acc[p].data, (size_t)rows * w + 8);
    }
  }
  iter_t it; memset(&it, 0, sizeof it);
  if (!rc) {
    iter_open(&it, v, 0);
    int done = 0;
    for (;;) { rc = iter_next(&it, &done); if (rc || done) break; rc = project_block(&it, acc); if (rc) break; }
    iter_close(&it);
  }
  for (int p = 0; p < v->nproj; p++) {
    if (rc) { free(acc[p].data.p); free(acc[p].bytes.p); free(acc[p].missing.p); continue; }
    outs[p].dtype = v->proj[p].expr->dtype; outs[p].count = acc[p].count;
    outs[p].data = acc[p].data.p; outs[p].bytes = acc[p].bytes.p; outs[p].nbytes = (int64_t)acc[p].bytes.n; outs[p].missing = acc[p].missing.p;
  }
  free(acc); return rc;
}
int orc_materialize(orc_view* v, orc_outcol* outs, int ncols) { return materialize_impl(v, outs, ncols, 1); }
int orc_materialize_nocount(orc_view* v, orc_outcol* outs, int ncols) { return materialize_impl(v, outs, ncols, 0); }
void orc_outcols_free(orc_outcol* outs, int ncols) { for (int i = 0; i < ncols; i++) { free(outs[i].data); free(outs[i].bytes); free(outs[i].missing); memset(&outs[i], 0, sizeof outs[i]); } }

/* ---------------------------------------------------------------- indices / bitmap / sums */
int orc_select_indices(orc_view* v, int64_t* out, int64_t cap, int64_t* n) {
  iter_t it; iter_open(&it, v, 1);
  int64_t tot = 0; int rc = 0, done = 0;
  for (;;) {
    rc = iter_next(&it, &done); if (rc || done) break;
    for (int64_t k = 0; k < it.block_rows; k++) if (it.mask[k]) { if (out && tot < cap) out[tot] = it.rows_before + k + 1; tot++; }
  }
  iter_close(&it); *n = tot; return rc;
}
int orc_select_bitmap(orc_view* v, uint64_t* out, int64_t nwords) {
  memset(out, 0, (size_t)nwords * 8);
  iter_t it; iter_open(&it, v, 1);
  int rc = 0, done = 0;
  for (;;) {
    rc = iter_next(&it, &done); if (rc || done) break;
    for (int64_t k = 0; k < it.block_rows; k++) if (it.mask[k]) { int64_t r = it.rows_before + k; if ((r >> 6) < nwords) out[r >> 6] |= 1ull << (r & 63); }
  }
  iter_close(&it); return rc;
}
static int sum_impl(orc_view* v, int col, double* fo, int64_t* io) {
  if (col < 0 || col >= v->nproj) return orc_fail(ORC_ERR_BOUNDS, "projection column out of range");
  orc_outcol* outs = (orc_outcol*)calloc((size_t)v->nproj, sizeof(orc_outcol));
  int rc = orc_materialize_nocount(v, outs, v->nproj);
  if (!rc) {
    int dt = dt_base(outs[col].dtype); int64_t n = outs[col].count; double fs = 0; int64_t is = 0;
    if (dt == DFDB_F64) { const double* p = (const double*)outs[col].data; for (int64_t k = 0; k < n; k++) fs += p[k]; }
    else if (dt == DFDB_F32) { const float* p = (const float*)outs[col].data; float s = 0; for (int64_t k = 0; k < n; k++) s += p[k]; fs = s; }
    else if (dt == DFDB_I64 || dt == DFDB_U64) { const int64_t* p = (const int64_t*)outs[col].data; for (int64_t k = 0; k < n; k++) is = (int64_t)((uint64_t)is + (uint64_t)p[k]); fs = (double)is; }
    else if (dt == DFDB_I32) { const int32_t* p = (const int32_t*)outs[col].data; for (int64_t k = 0; k < n; k++) is += p[k]; fs = (double)is; }
    else if (dt == DFDB_BOOL || dt == DFDB_U8) { const uint8_t* p = (const uint8_t*)outs[col].data; for (int64_t k = 0; k < n; k++) is += p[k]; fs = (double)is; }
    else rc = orc_fail(ORC_ERR_UNSUPPORTED, "sum over this dtype not restated");
    if (fo) *fo = fs;
    if (io) *io = is;
    orc_outcols_free(outs, v->nproj);
  }
  free(outs); return rc;
}
int orc_sum_f64(orc_view* v, int col, double* out) { return sum_impl(v, col, out, NULL); }
int orc_sum_i64(orc_view* v, int col, int64_t* out) { return sum_impl(v, col, NULL, out); }

/* ---------------------------------------------------------------- cpu_baseline leg */
int orc_bench_scan(orc_view* v, int64_t* out, int64_t cap, int64_t* nsel, double* seconds) {
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  int rc = orc_select_indices(v, out, cap, nsel);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}
