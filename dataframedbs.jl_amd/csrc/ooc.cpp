// ooc.cpp — out-of-core execution behind the ordinary query entry points (ooc.hpp).
//
// What the reference's consumers do block by block over a BlocksIterator, done chunk by chunk over the block stream of stream.cpp:
//   nrow(v)              view.jl:192-206 + blocksiterator.jl:46-66,123-145  (BlockRowsIterator: the selection's columns only)      -> ooc_count
//   materialize(v)       materialization.jl:27-40 (count pre-pass for sizehint!, then append! per block)                          -> ooc_materialize
//   sum / minimum / ...  Base.iterate(::DFColumn) column.jl:102-126                                                               -> ooc_aggregate
//   unique(col)          Base.unique over the same iteration (docs/src/index.md:171-182,479-487)                                  -> ooc_unique
//   groupreduce          aggregate.jl:1-36 (completed as dfdb_query_groupreduce completes it)                                     -> ooc_groupreduce
// HBM holds the stream's chunks (ctx option "stream_slots" x "ooc_chunk_blocks" blocks per required column), never the table.
#include "ooc.hpp"
#include <algorithm>
#include <cmath>

namespace dfdb {

// ------------------------------------------------------------------ folding and the by-key merge (shared with group.cpp)
template <class T> static T fold_t(T a, T b, int op) { return op == DFDB_AGG_MIN ? std::min(a, b) : (op == DFDB_AGG_MAX ? std::max(a, b) : (T)(a + b)); }
double fold_f64(double x, double y, int op) {
  if (op == DFDB_AGG_SUM) return x + y;
  if (std::isnan(x) || std::isnan(y)) return NAN;
  if (x == y) return op == DFDB_AGG_MIN ? (std::signbit(x) ? x : y) : (std::signbit(x) ? y : x);
  return op == DFDB_AGG_MIN ? std::min(x, y) : std::max(x, y);
}
uint64_t fold_bits(uint64_t a, uint64_t b, int dt, int op) {
  if (dt == DFDB_F64) {
    double x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8);
    const double r = fold_f64(x, y, op);
    uint64_t o; memcpy(&o, &r, 8); return o;
  }
  if (dt == DFDB_U64) return op == DFDB_AGG_SUM ? a + b : fold_t<uint64_t>(a, b, op);
  return op == DFDB_AGG_SUM ? a + b : (uint64_t)fold_t<int64_t>((int64_t)a, (int64_t)b, op);
}

// isequal as a byte string: the missing flag, then the value's bytes with every NaN folded onto one (isequal(NaN, -NaN); -0.0 and 0.0 stay apart)
static std::string merge_key(int32_t kdt, const GroupPart& p, int64_t j, int64_t& byte_off) {
  const bool miss = !p.key_missing.empty() && p.key_missing[(size_t)j];
  std::string k(1, miss ? '\1' : '\0');
  if (dt_base(kdt) == DFDB_STRING) {
    int32_t sz; memcpy(&sz, p.key_data.data() + (size_t)j * 4, 4);
    if (sz < 0) { k[0] = '\1'; return k; }
    k.append((const char*)p.key_bytes.data() + byte_off, (size_t)sz); byte_off += sz;
    return k;
  }
  if (miss) return k;                                    // (the bytes under a missing flag are garbage: quirk Q11)
  const int w = dt_width(kdt);
  const uint8_t* v = p.key_data.data() + (size_t)j * w;
  if (dt_base(kdt) == DFDB_F64) { double d; memcpy(&d, v, 8); if (std::isnan(d)) { k.append("NaN"); return k; } }
  if (dt_base(kdt) == DFDB_F32) { float f; memcpy(&f, v, 4); if (std::isnan(f)) { k.append("NaN"); return k; } }
  k.append((const char*)v, (size_t)w);
  return k;
}

void GroupMerger::add(GroupMerged& m, const GroupPart& p) {
  const int32_t kdt = m.key_dtype;
  const bool is_str = dt_base(kdt) == DFDB_STRING;
  const int w = is_str ? 4 : dt_width(kdt);
  const bool rows = !p.first_rows.empty();
  int64_t boff = 0;
  for (int64_t j = 0; j < p.ng; j++) {                     // part order = table order: a key keeps the place of its first appearance
    const int64_t b0 = boff;
    const std::string k = merge_key(kdt, p, j, boff);
    auto it = slot.find(k);
    if (it == slot.end()) {
      slot.emplace(k, m.ng++);
      m.key_data.insert(m.key_data.end(), p.key_data.begin() + j * w, p.key_data.begin() + (j + 1) * w);
      m.key_missing.push_back(k[0] == '\1' ? 1 : 0);
      if (is_str) m.key_bytes.insert(m.key_bytes.end(), p.key_bytes.begin() + b0, p.key_bytes.begin() + boff);
      m.counts.push_back(p.counts[(size_t)j]); m.vals.push_back(p.vals[(size_t)j]);
      if (rows) m.first_rows.push_back(p.first_rows[(size_t)j]);
      continue;
    }
    const size_t s = (size_t)it->second;
    m.counts[s] += p.counts[(size_t)j];
    const uint64_t a = m.vals[s], b = p.vals[(size_t)j];
    if (m.op == DFDB_AGG_COUNT) m.vals[s] = a + b;
    else m.vals[s] = fold_bits(a, b, m.kind == 2 ? DFDB_F64 : (m.kind == 1 ? DFDB_U64 : DFDB_I64), m.op);   // wrapping Int sums, Float64 sums of the parts' sums, NaN-propagating min / max
  }
}

void fetch_group_part(dfdb_query* q, int32_t key_p, int64_t ng, int64_t kb, bool with_rows, GroupPart& part) {
  const int32_t kdt = q->proj[(size_t)key_p].expr->dtype;
  const bool is_str = dt_base(kdt) == DFDB_STRING;
  part.ng = ng;
  part.key_data.resize((size_t)ng * (size_t)(is_str ? 4 : dt_width(kdt)));
  if (dt_nullable(kdt) && !is_str) part.key_missing.assign((size_t)ng, 0);
  part.key_bytes.resize((size_t)kb);
  part.counts.assign((size_t)ng, 0); part.vals.assign((size_t)ng, 0);
  if (with_rows) {                                         // between groupreduce and its fetch q's selection IS the first occurrences (dfdb.h)
    part.first_rows.assign((size_t)ng, 0);
    if (ng > 0) { int64_t got = 0; query_select_indices(q, part.first_rows.data(), ng, DFDB_MEM_HOST, &got); if (got != ng) fail(DFDB_ERR_DEVICE, "groupreduce: %lld first rows for %lld groups", (long long)got, (long long)ng); }
  }
  dfdb_outcol o{}; o.memkind = DFDB_MEM_HOST; o.data = part.key_data.data(); o.bytes = part.key_bytes.data(); o.bytes_cap = kb;
  o.missing = part.key_missing.empty() ? nullptr : part.key_missing.data();
  std::vector<int64_t> vi((size_t)ng); std::vector<double> vf((size_t)ng);
  query_groupreduce_fetch(q, &o, part.counts.data(), vi.data(), vf.data());   // (puts the full selection back)
  for (int64_t j = 0; j < ng; j++) { if (q->gr_kind == 2) memcpy(&part.vals[(size_t)j], &vf[(size_t)j], 8); else part.vals[(size_t)j] = (uint64_t)vi[(size_t)j]; }
}

void merged_fetch(const GroupMerged& m, dfdb_outcol* keys, int64_t* counts, int64_t* vals_i, double* vals_f) {
  if (keys) {
    if (keys->memkind != DFDB_MEM_HOST) fail(DFDB_ERR_ARGUMENT, "merged keys are written to host buffers");
    const bool is_str = dt_base(m.key_dtype) == DFDB_STRING;
    keys->dtype = m.key_dtype; keys->count = m.ng; keys->nbytes = (int64_t)m.key_bytes.size();
    if (m.ng > 0) {
      if (!keys->data) fail(DFDB_ERR_ARGUMENT, "the key column has no data buffer");
      memcpy(keys->data, m.key_data.data(), m.key_data.size());
      if (keys->missing) memcpy(keys->missing, m.key_missing.data(), (size_t)m.ng);
      if (is_str && !m.key_bytes.empty()) {
        if ((int64_t)m.key_bytes.size() > keys->bytes_cap || !keys->bytes) fail(DFDB_ERR_ARGUMENT, "the key column needs %zu string bytes, capacity is %lld", m.key_bytes.size(), (long long)keys->bytes_cap);
        memcpy(keys->bytes, m.key_bytes.data(), m.key_bytes.size());
      }
    }
  }
  for (int64_t j = 0; j < m.ng; j++) {
    if (counts) counts[j] = m.counts[(size_t)j];
    const uint64_t b = m.vals[(size_t)j];
    double d; memcpy(&d, &b, 8);
    if (m.kind == 2) { if (vals_f) vals_f[j] = d; if (vals_i) vals_i[j] = (int64_t)d; }
    else { if (vals_i) vals_i[j] = (int64_t)b; if (vals_f) vals_f[j] = m.kind == 1 ? (double)b : (double)(int64_t)b; }
  }
}

// ------------------------------------------------------------------ which queries stream
static void view_columns(const dfdb_query* q, std::vector<int>& sel, std::vector<int>& all) {
  for (const Stage& st : q->stages) if (st.kind == ST_PRED) required_columns(*st.pred, sel);
  all = sel;
  for (const ProjCol& p : q->proj) required_columns(*p.expr, all);
}
bool query_out_of_core(const dfdb_query* q) {
  const dfdb_table* t = q->t;
  if (!t || t->path.empty() || q->stream_owned) return false;
  std::vector<int> sel, all;
  view_columns(q, sel, all);
  if (all.empty()) {
    // no column is needed (range stages only and an empty projection or one of constants): the row count comes from the first column, like the stream's
    if (t->cols.empty()) return false;
    return !t->cols[0].resident && t->nrows < 0;
  }
  for (int o : all) if (!t->cols[(size_t)o].resident) return true;
  return false;
}

static OocState& state(dfdb_query* q) {
  if (!q->ooc) q->ooc = std::make_shared<OocState>();
  if (q->ooc->str_bytes.size() != q->proj.size()) q->ooc->str_bytes.assign(q->proj.size(), -1);
  return *q->ooc;
}
void ooc_reset(dfdb_query* q) { if (q->ooc) { const dfdb_sizestats keep = q->ooc->read; q->ooc = std::make_shared<OocState>(); q->ooc->read = keep; } }

// the caller's view re-stated with another projection (and, for a narrowed query, another selection): what a pass streams
struct TempQuery {
  dfdb_query q;
  TempQuery(const dfdb_query* src, bool with_stages, int nstages = -1) {
    q.t = src->t;
    if (with_stages)
      for (const Stage& st : src->stages) {
        if (nstages >= 0 && (int)q.stages.size() >= nstages) break;
        Stage c; c.kind = st.kind; c.start = st.start; c.step = st.step; c.stop = st.stop; c.n = st.n; c.idx = st.idx; c.stage_base = st.stage_base;
        if (st.pred) c.pred = st.pred->clone();
        q.stages.push_back(std::move(c));
      }
  }
  void project(const ProjCol& p) { q.proj.push_back(ProjCol{p.name, p.expr->clone()}); }
  void project_column(const dfdb_table* t, int o) {
    auto n = std::make_unique<Node>(); n->op = DFIR_COL; n->col = o; n->dtype = t->cols[(size_t)o].dtype;
    q.proj.push_back(ProjCol{t->cols[(size_t)o].name, std::move(n)});
  }
};

// one pass of the block stream over `tq`; fn(chunk) for every chunk in table order
struct StreamPass {
  dfdb_stream* s = nullptr;
  dfdb_query* owner;
  StreamPass(dfdb_query* owner_, dfdb_query* tq) : owner(owner_) {
    const int64_t cb = ctx_option(tq->t->ctx, "ooc_chunk_blocks", 512);
    stream_open(tq, cb > 0 ? cb : 512, &s);
  }
  dfdb_query* next() { int64_t rows = 0, first = 0; return stream_next(s, &rows, &first); }
  ~StreamPass() {
    if (!s) return;
    try {
      dfdb_sizestats st{0, 0, 0};
      stream_read_stats(s, -1, &st);
      OocState& o = state(owner);
      o.read.rows += st.rows; o.read.compressed += st.compressed; o.read.uncompressed += st.uncompressed;
    } catch (...) {}
    try { stream_close(s); } catch (...) {}
  }
  StreamPass(const StreamPass&) = delete;
  StreamPass& operator=(const StreamPass&) = delete;
};

// a query narrowed by dfdb_query_unique: its selection is the first occurrences' table rows
static void narrowed_view(dfdb_query* q, TempQuery& tq) {
  OocState& o = state(q);
  Stage st; st.kind = ST_INDICES; st.idx = o.merged.first_rows;
  query_add_stage(&tq.q, std::move(st));
}

// ------------------------------------------------------------------ count
// The row counter reads the selection's columns only — or, when the queue holds no predicate, the FIRST projection column (blocksiterator.jl:46-66) —
// never a projection-only column.  With dfdb_query_hint_materialize on, the pass also sizes the projected String columns (their sizes are read for the
// blocks that kept a row: late materialization), so that count + materialize stay two passes like the reference's (materialization.jl:29-37).
int64_t ooc_count(dfdb_query* q) {
  OocState& o = state(q);
  if (o.narrowed) return o.merged.ng;
  if (o.count >= 0) return o.count;
  const dfdb_table* t = q->t;
  std::vector<int> sel, all;
  view_columns(q, sel, all);
  TempQuery tq(q, true);
  std::vector<int> sized;                             // projection columns of q sized in this pass -> their position in tq's projection
  std::vector<int> pos;
  // (no column needed at all: a projection of constants iterates nothing — `isempty(it.streams)`, blocksiterator.jl:101 —, an EMPTY projection counts the
  // selection off the first column's block sizes: both are the stream's own rules, so the projection goes over as it is)
  if (all.empty()) { for (const ProjCol& p : q->proj) tq.project(p); }
  else tq.project_column(t, sel.empty() ? all[0] : sel[0]);
  if (q->hint_materialize && !all.empty())
    for (size_t p = 0; p < q->proj.size(); p++) {
      const Node& e = *q->proj[p].expr;
      if (dt_base(e.dtype) != DFDB_STRING || e.op != DFIR_COL) continue;
      bool dup = false;
      for (size_t k = 0; k < tq.q.proj.size(); k++) if (tq.q.proj[k].expr->op == DFIR_COL && tq.q.proj[k].expr->col == e.col) { sized.push_back((int)p); pos.push_back((int)k); dup = true; break; }
      if (dup) continue;
      sized.push_back((int)p); pos.push_back((int)tq.q.proj.size());
      tq.project(q->proj[p]);
    }
  int64_t total = 0;
  std::vector<int64_t> sb(sized.size(), 0);
  {
    StreamPass pass(q, &tq.q);
    while (dfdb_query* c = pass.next()) {
      const int64_t n = query_count(c, -1);
      total += n;
      if (n > 0) for (size_t k = 0; k < sized.size(); k++) sb[k] += query_string_bytes(c, pos[k]);
    }
  }
  for (size_t k = 0; k < sized.size(); k++) o.str_bytes[(size_t)sized[k]] = sb[k];
  return o.count = total;
}

void ooc_select_indices(dfdb_query* q, int64_t* out, int64_t cap, int32_t memkind, int64_t* n) {
  OocState& o = state(q);
  if (o.narrowed) {
    const int64_t m = std::min<int64_t>(cap, o.merged.ng);
    if (n) *n = o.merged.ng;
    if (m <= 0) return;
    if (memkind == DFDB_MEM_DEVICE) { HIP_CHECK(hipMemcpyAsync(out, o.merged.first_rows.data(), (size_t)m * 8, hipMemcpyHostToDevice, q->t->ctx->stream)); HIP_CHECK(hipStreamSynchronize(q->t->ctx->stream)); }
    else memcpy(out, o.merged.first_rows.data(), (size_t)m * 8);
    return;
  }
  const dfdb_table* t = q->t;
  std::vector<int> sel, all;
  view_columns(q, sel, all);
  TempQuery tq(q, true);
  if (all.empty()) { for (const ProjCol& p : q->proj) tq.project(p); }
  else tq.project_column(t, sel.empty() ? all[0] : sel[0]);
  int64_t total = 0;
  {
    StreamPass pass(q, &tq.q);
    while (dfdb_query* c = pass.next()) {
      const int64_t k = query_count(c, -1);
      if (k > 0 && total < cap) {
        int64_t got = 0;
        query_select_indices(c, out + total, cap - total, memkind, memkind == DFDB_MEM_DEVICE ? nullptr : &got);
        if (memkind == DFDB_MEM_DEVICE) HIP_CHECK(hipStreamSynchronize(c->t->ctx->stream));   // (the chunk's buffers are reused by the next one)
      }
      total += k;
    }
  }
  o.count = total;
  if (n) *n = total;
}

// ------------------------------------------------------------------ materialize
static int64_t width_of(const Node& e) { return dt_base(e.dtype) == DFDB_STRING ? 4 : dt_width(e.dtype); }

// every chunk's rows appended to the caller's buffers at the running offsets (append!(res, bl), materialization.jl:33-37)
static void stream_materialize(dfdb_query* q, dfdb_query* tq, dfdb_outcol* outs, int32_t ncols, int64_t row_cap) {
  std::vector<int64_t> boff((size_t)ncols, 0);
  int64_t rows = 0;
  int32_t dts_set = 0;
  {
    StreamPass pass(q, tq);
    std::vector<dfdb_outcol> so((size_t)ncols);
    while (dfdb_query* c = pass.next()) {
      const int64_t n = query_count(c, -1);
      if (n == 0) continue;
      if (row_cap >= 0 && rows + n > row_cap) fail(DFDB_ERR_BOUNDS, "BoundsError: the view holds more rows than were counted (%lld): the table's files changed between dfdb_count and dfdb_materialize", (long long)row_cap);
      for (int32_t p = 0; p < ncols; p++) {
        dfdb_outcol x = outs[p];
        const Node& e = *tq->proj[(size_t)p].expr;
        if (x.data) x.data = (char*)x.data + rows * width_of(e);
        if (x.bytes) { x.bytes += boff[(size_t)p]; x.bytes_cap = outs[p].bytes_cap - boff[(size_t)p]; }
        if (x.missing) x.missing += rows;
        so[(size_t)p] = x;
      }
      query_materialize(c, so.data(), ncols);
      HIP_CHECK(hipStreamSynchronize(c->t->ctx->stream));   // device outputs are queued on the CHUNK's stream: done before the chunk goes
      for (int32_t p = 0; p < ncols; p++) { boff[(size_t)p] += so[(size_t)p].nbytes; outs[p].dtype = so[(size_t)p].dtype; }
      dts_set = 1;
      rows += n;
    }
  }
  for (int32_t p = 0; p < ncols; p++) {
    if (!dts_set) outs[p].dtype = tq->proj[(size_t)p].expr->dtype;
    outs[p].count = rows; outs[p].nbytes = boff[(size_t)p];
  }
  OocState& o = state(q);
  if (!o.narrowed) { o.count = rows; for (int32_t p = 0; p < ncols; p++) if (dt_base(tq->proj[(size_t)p].expr->dtype) == DFDB_STRING) o.str_bytes[(size_t)p] = boff[(size_t)p]; }
}

int64_t ooc_string_bytes(dfdb_query* q, int32_t i) {
  if (i < 0 || (size_t)i >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", i);
  const Node& e = *q->proj[(size_t)i].expr;
  if (dt_base(e.dtype) != DFDB_STRING) return 0;
  if (e.op != DFIR_COL) fail(DFDB_ERR_UNSUPPORTED, "computed String columns are outside the IR");
  OocState& o = state(q);
  if (o.narrowed && o.merged_col == e.col) return (int64_t)o.merged.key_bytes.size();
  if (!o.narrowed && o.str_bytes[(size_t)i] >= 0) return o.str_bytes[(size_t)i];
  // a sizing pass of its own: the selection with this one column as its projection
  TempQuery tq(q, !o.narrowed);
  if (o.narrowed) narrowed_view(q, tq);
  tq.project(q->proj[(size_t)i]);
  int64_t total = 0, rows = 0;
  {
    StreamPass pass(q, &tq.q);
    while (dfdb_query* c = pass.next()) { const int64_t n = query_count(c, -1); rows += n; if (n > 0) total += query_string_bytes(c, 0); }
  }
  if (!o.narrowed) { o.str_bytes[(size_t)i] = total; o.count = rows; }
  return total;
}

void ooc_materialize(dfdb_query* q, dfdb_outcol* outs, int32_t ncols) {
  if (ncols != (int32_t)q->proj.size()) fail(DFDB_ERR_ARGUMENT, "ArgumentError: view has %zu columns, %d outputs given", q->proj.size(), ncols);
  OocState& o = state(q);
  if (o.narrowed) {
    // unique(col): the key column comes out of the merged first occurrences; any other column is read at those rows
    bool only_key = true;
    for (const ProjCol& p : q->proj) only_key = only_key && p.expr->op == DFIR_COL && p.expr->col == o.merged_col;
    if (only_key) {
      for (int32_t p = 0; p < ncols; p++) {
        if (outs[p].memkind != DFDB_MEM_HOST) { only_key = false; break; }
      }
    }
    if (only_key) { for (int32_t p = 0; p < ncols; p++) merged_fetch(o.merged, &outs[p], nullptr, nullptr, nullptr); return; }
    TempQuery tq(q, false);
    narrowed_view(q, tq);
    for (const ProjCol& p : q->proj) tq.project(p);
    stream_materialize(q, &tq.q, outs, ncols, o.merged.ng);
    return;
  }
  if (ncols == 0) return;
  TempQuery tq(q, true);
  for (const ProjCol& p : q->proj) tq.project(p);
  stream_materialize(q, &tq.q, outs, ncols, o.count);
}

void ooc_materialize_column(dfdb_query* q, int32_t p, dfdb_outcol* o) {
  if (p < 0 || (size_t)p >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", p);
  OocState& st = state(q);
  TempQuery tq(q, !st.narrowed);
  if (st.narrowed) narrowed_view(q, tq);
  tq.project(q->proj[(size_t)p]);
  const int64_t cap = st.narrowed ? st.merged.ng : st.count;
  // (stream_materialize keeps what it learns under the projection index of the query it streams: a one-column pass must not overwrite column 0's string bytes)
  const std::vector<int64_t> keep = st.str_bytes;
  stream_materialize(q, &tq.q, o, 1, cap);
  state(q).str_bytes = keep;
}

// ------------------------------------------------------------------ aggregates
void ooc_aggregate(dfdb_query* q, int32_t op, int32_t i, int64_t* out_i, double* out_f) {
  if (op == DFDB_AGG_COUNT) { const int64_t n = ooc_count(q); if (out_i) *out_i = n; if (out_f) *out_f = (double)n; return; }
  if (i < 0 || (size_t)i >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", i);
  const Node& e = *q->proj[(size_t)i].expr;
  if (!dt_isnum(e.dtype) || dt_nullable(e.dtype)) fail(DFDB_ERR_UNSUPPORTED, "aggregate over %s is not supported", dt_name(e.dtype).c_str());
  OocState& o = state(q);
  TempQuery tq(q, !o.narrowed);
  if (o.narrowed) narrowed_view(q, tq);
  tq.project(q->proj[(size_t)i]);
  const int b = dt_base(e.dtype);
  const int adt = dt_isfloat(b) ? DFDB_F64 : (b == DFDB_U64 ? DFDB_U64 : DFDB_I64);    // the accumulator query_aggregate_device reduces in
  uint64_t acc = 0; bool have = false; int64_t rows = 0;
  {
    StreamPass pass(q, &tq.q);
    while (dfdb_query* c = pass.next()) {
      const int64_t n = query_count(c, -1);
      if (n == 0) continue;                              // (minimum / maximum of an empty chunk would raise)
      rows += n;
      int64_t vi = 0; double vf = 0;
      query_aggregate(c, op, 0, &vi, &vf);
      uint64_t bits; if (adt == DFDB_F64) memcpy(&bits, &vf, 8); else bits = (uint64_t)vi;
      acc = have ? fold_bits(acc, bits, adt, op) : bits;  // chunk order = block order = the reference's left-to-right order at chunk granularity
      have = true;
    }
  }
  if (!o.narrowed) o.count = rows;
  if (!have) {
    if (op != DFDB_AGG_SUM) fail(DFDB_ERR_ARGUMENT, "ArgumentError: reducing over an empty collection is not allowed");
    acc = 0;                                             // (0 and 0.0 share a bit pattern)
  }
  if (adt == DFDB_F64) { double d; memcpy(&d, &acc, 8); if (out_f) *out_f = d; if (out_i) *out_i = (int64_t)d; }
  else { if (out_i) *out_i = (int64_t)acc; if (out_f) *out_f = adt == DFDB_U64 ? (double)acc : (double)(int64_t)acc; }
}

// ------------------------------------------------------------------ unique / groupreduce
static void stream_groupreduce(dfdb_query* q, int32_t key_p, int32_t val_p, int32_t op, bool with_rows) {
  if (key_p < 0 || (size_t)key_p >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", key_p);
  const Node& ke = *q->proj[(size_t)key_p].expr;
  if (ke.op != DFIR_COL) fail(DFDB_ERR_UNSUPPORTED, "unique / groupreduce by a computed column: materialise it as a column first (dfdb_table_add_from_query)");
  if (op != DFDB_AGG_COUNT && op != DFDB_AGG_SUM && op != DFDB_AGG_MIN && op != DFDB_AGG_MAX) fail(DFDB_ERR_ARGUMENT, "unknown statistic %d", op);
  int kind = 0;
  if (op != DFDB_AGG_COUNT) {
    if (val_p < 0 || (size_t)val_p >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", val_p);
    const Node& ve = *q->proj[(size_t)val_p].expr;
    if (ve.op != DFIR_COL || !dt_isnum(ve.dtype) || dt_nullable(ve.dtype)) fail(DFDB_ERR_UNSUPPORTED, "groupreduce over %s: a plain numeric column is needed", dt_name(ve.dtype).c_str());
    const int b = dt_base(ve.dtype); kind = dt_isfloat(b) ? 2 : (dt_issigned(b) ? 0 : 1);
  }
  OocState& o = state(q);
  if (o.narrowed) fail(DFDB_ERR_ARGUMENT, "ArgumentError: the query's selection is narrowed by dfdb_query_unique: dfdb_query_reset it first");
  TempQuery tq(q, true);
  tq.project(q->proj[(size_t)key_p]);
  if (op != DFDB_AGG_COUNT) tq.project(q->proj[(size_t)val_p]);
  GroupMerged m; m.key_dtype = ke.dtype; m.kind = kind; m.op = op; m.with_stats = !with_rows;
  GroupMerger mg;
  int64_t rows = 0;
  {
    StreamPass pass(q, &tq.q);
    while (dfdb_query* c = pass.next()) {
      const int64_t n = query_count(c, -1);
      if (n == 0) continue;
      rows += n;
      int64_t ng = 0, kb = 0;
      query_groupreduce(c, 0, op == DFDB_AGG_COUNT ? -1 : 1, op, &ng, &kb);
      GroupPart part;
      fetch_group_part(c, 0, ng, kb, with_rows, part);
      mg.add(m, part);
    }
  }
  m.valid = true;
  o.count = rows;
  o.merged = std::move(m);
  o.merged_col = ke.col;
}

void ooc_unique(dfdb_query* q, int32_t p) {
  stream_groupreduce(q, p, -1, DFDB_AGG_COUNT, true);
  OocState& o = state(q);
  o.narrowed = true; o.gr_pending = false;
}

void ooc_groupreduce(dfdb_query* q, int32_t key_p, int32_t val_p, int32_t op, int64_t* ngroups, int64_t* key_bytes) {
  stream_groupreduce(q, key_p, val_p, op, false);
  OocState& o = state(q);
  o.gr_pending = true;
  if (ngroups) *ngroups = o.merged.ng;
  if (key_bytes) *key_bytes = (int64_t)o.merged.key_bytes.size();
}

void ooc_groupreduce_fetch(dfdb_query* q, dfdb_outcol* keys, int64_t* counts, int64_t* vals_i, double* vals_f) {
  OocState& o = state(q);
  if (!o.gr_pending || !o.merged.valid) fail(DFDB_ERR_ARGUMENT, "ArgumentError: dfdb_query_groupreduce has not been called (or the query was executed, reset or changed since)");
  merged_fetch(o.merged, keys, counts, vals_i, vals_f);
  o.merged = GroupMerged{}; o.gr_pending = false; o.merged_col = -1;
}

// ------------------------------------------------------------------ what a multi-GPU group asks of a shard that is not resident (group.cpp)
// survivors of stages [0, nstages) over the shard's block window: the operand of the all-gather between a predicate stage and a range stage after it
int64_t ooc_count_prefix(dfdb_query* q, int nstages) {
  const dfdb_table* t = q->t;
  TempQuery tq(q, true, nstages);
  std::vector<int> sel, all;
  view_columns(&tq.q, sel, all);
  if (!sel.empty()) tq.project_column(t, sel[0]);
  else { std::vector<int> s2, a2; view_columns(q, s2, a2); if (!a2.empty()) tq.project_column(t, a2[0]); }
  int64_t total = 0;
  StreamPass pass(q, &tq.q);
  while (dfdb_query* c = pass.next()) total += query_count(c, -1);
  return total;
}
// {value bits, selected rows} of sum / min / max over projection column i, the identity of `op` when nothing is selected (what query_aggregate_device
// leaves in red_result); returns the accumulator dtype
int ooc_aggregate_bits(dfdb_query* q, int32_t op, int32_t i, uint64_t out[2]) {
  if (i < 0 || (size_t)i >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", i);
  const Node& e = *q->proj[(size_t)i].expr;
  if (!dt_isnum(e.dtype) || dt_nullable(e.dtype)) fail(DFDB_ERR_UNSUPPORTED, "aggregate over %s is not supported", dt_name(e.dtype).c_str());
  const int b = dt_base(e.dtype);
  const int adt = dt_isfloat(b) ? DFDB_F64 : (b == DFDB_U64 ? DFDB_U64 : DFDB_I64);
  const int64_t n = ooc_count(q);
  if (n == 0) {
    uint64_t id = 0;
    if (op == DFDB_AGG_MIN) { if (adt == DFDB_F64) { const double d = INFINITY; memcpy(&id, &d, 8); } else id = adt == DFDB_U64 ? ~0ull : (uint64_t)INT64_MAX; }
    if (op == DFDB_AGG_MAX) { if (adt == DFDB_F64) { const double d = -INFINITY; memcpy(&id, &d, 8); } else id = adt == DFDB_U64 ? 0ull : (uint64_t)INT64_MIN; }
    out[0] = id; out[1] = 0;
    return adt;
  }
  int64_t vi = 0; double vf = 0;
  ooc_aggregate(q, op, i, &vi, &vf);
  if (adt == DFDB_F64) memcpy(&out[0], &vf, 8); else out[0] = (uint64_t)vi;
  out[1] = (uint64_t)n;
  return adt;
}
// the shard's groups as ONE part (its chunks merged in chunk order), for the merge across the ranks
int ooc_group_part(dfdb_query* q, int32_t key_p, int32_t val_p, int32_t op, GroupPart& part) {
  stream_groupreduce(q, key_p, val_p, op, false);
  OocState& o = state(q);
  GroupMerged& m = o.merged;
  part.ng = m.ng; part.key_data = std::move(m.key_data); part.key_bytes = std::move(m.key_bytes); part.counts = std::move(m.counts); part.vals = std::move(m.vals);
  const int32_t kdt = m.key_dtype;
  if (dt_nullable(kdt) && dt_base(kdt) != DFDB_STRING) part.key_missing = std::move(m.key_missing);
  const int kind = m.kind;
  o.merged = GroupMerged{}; o.merged_col = -1;
  return kind;
}

// ------------------------------------------------------------------ dfdb_query_prepare
// Only the columns the view needs are opened (view.jl:183-190, blocksiterator.jl:20-33).  They become resident when they fit: decoded if the decoded arrays
// (plus the load's staging of the compressed bytes) fit the budget, compressed-only (LZ4 blocks in HBM, K7 decodes inside the scan) if those fit and every
// missing column is a plain fixed-width one, and otherwise they stay on disk and the entry points stream.
int32_t query_prepare(dfdb_query* q) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx;
  if (t->path.empty()) return 0;
  std::vector<int> sel, all;
  view_columns(q, sel, all);
  if (all.empty() && !t->cols.empty() && t->nrows < 0) all.push_back(0);
  std::vector<int32_t> need;
  for (int o : all) if (!t->cols[(size_t)o].resident) need.push_back(o);
  if (need.empty()) return 0;
  HIP_CHECK(hipSetDevice(ctx->device));
  // two bounds: what the TABLE may hold (ctx option "hbm_budget_mb"; 0 = no bound of its own) and what the device has free right now (the load's staging of
  // one column file and, for the compressed-only form, the context's history rings come out of that too)
  size_t free_b = 0, total_b = 0;
  HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  const int64_t free_now = (int64_t)((double)free_b * 0.8);
  int64_t budget = ctx_option(ctx, "hbm_budget_mb", 0) << 20;
  if (budget > 0) { int64_t d = 0, k = 0; table_resident_bytes(t, -1, &d, &k); budget -= d + k; }
  else budget = INT64_MAX;
  int64_t dec = 0, comp = 0, comp_max = 0; bool plain = true;
  for (int32_t o : need) {
    dfdb_sizestats st{0, 0, 0};
    table_column_stats(t, o, &st);
    dec += st.uncompressed; comp += st.compressed; comp_max = std::max(comp_max, st.compressed);
    const Column& c = t->cols[(size_t)o];
    plain = plain && dt_base(c.dtype) != DFDB_STRING && !dt_nullable(c.dtype);
  }
  auto unload = [&] { for (int32_t o : need) { Column& c = t->cols[(size_t)o]; if (!c.resident) { c.data.release(); c.bytes.release(); c.missing.release(); c.tile_off.release(); c.comp.release(); c.comp_blocks.release(); c.comp_status.release(); c.comp_index.release(); c.comp_nblocks = 0; c.comp_only = false; } } };
  const int64_t kc_old = ctx_option(ctx, "keep_compressed", 0);
  struct Restore { dfdb_ctx* c; int64_t v; ~Restore() { c->options["keep_compressed"] = v; } } restore{ctx, kc_old};
  if (dec <= budget && dec + comp_max + (64 << 20) <= free_now) {           // decoded arrays (+ the staging of one column file at a time)
    if (kc_old == 2) ctx->options["keep_compressed"] = 0;
    try { table_load(t, need.data(), (int32_t)need.size(), 0, -1, nullptr); return 1; }
    catch (const Error& e) { if (e.code != DFDB_ERR_NOMEM) throw; (void)hipGetLastError(); unload(); }
  }
  const int64_t held = comp + comp / 8;                                      // the blocks + their sequence-start index
  int waves = 0; const bool rings = ctx->hist.p != nullptr; (void)waves;
  if (plain && held <= budget && held + comp_max + (rings ? 0 : (int64_t)lz4_hist_scratch_bytes(lz4_hist_default_waves(ctx->prop.multiProcessorCount))) + (64 << 20) <= free_now) {
    ctx->options["keep_compressed"] = 2;
    try { table_load(t, need.data(), (int32_t)need.size(), 0, -1, nullptr); return 2; }
    catch (const Error& e) { if (e.code != DFDB_ERR_NOMEM) throw; (void)hipGetLastError(); unload(); }
  }
  return 3;
}

}  // namespace dfdb
