// query.cpp — DFView on the device: SelectionQueue composition, stage execution, count, indices,
// materialize, aggregates.
//
// Replaces (paths under /root/reference): SelectionQueue add/_new_queue src/tables/selection.jl:37-60,
// SelectionExecutor.apply :161-167 with its range :94-111 and predicate :133-157 stages, the block loop of
// src/io/blocksiterator.jl:98-145, nrow src/tables/view.jl:192-206, ProjectionExecutor.eval_on_range
// src/tables/projection.jl:128-154 and materialize src/tables/materialization.jl:27-52.
//
// The reference walks blocks serially and carries a per-stage `offset`; here every stage is ONE pass over
// the whole resident column range, the mask is a packed bitmap in HBM, and the cross-block offset becomes
// an exclusive scan of per-tile popcounts.  The selection is evaluated once (no count pre-pass, quirk Q8):
// dfdb_count reads the scan total, dfdb_materialize reuses the same bitmap.
#include "engine.hpp"
#include "ooc.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>

namespace dfdb {

// launchers living in k_interp.hip / k_strings.hip that take engine-level descriptions
void run_interp_predicate(dfdb_query* q, const Node& pred, bool and_existing);
void run_interp_project(dfdb_query* q, const Node& expr, void* dst, int64_t cap, uint8_t* missing_dst);

// ---------------------------------------------------------------- ranges
static int64_t range_len(int64_t a, int64_t s, int64_t b) {
  if (s > 0) return b < a ? 0 : (b - a) / s + 1;
  return b > a ? 0 : (a - b) / (-s) + 1;
}
int64_t Stage::first() const {
  if (kind == ST_RANGE) return n == 0 ? INT64_MAX : (step > 0 ? start : stop);
  return idx.empty() ? INT64_MAX : *std::min_element(idx.begin(), idx.end());
}
int64_t Stage::last() const {
  if (kind == ST_RANGE) return n == 0 ? INT64_MIN : (step > 0 ? stop : start);
  return idx.empty() ? INT64_MIN : *std::max_element(idx.begin(), idx.end());
}
int64_t Stage::elem(int64_t k) const {
  if (k < 1 || k > n) fail(DFDB_ERR_BOUNDS, "BoundsError: attempt to access %lld-element selection at index [%lld]", (long long)n, (long long)k);
  return kind == ST_RANGE ? start + (k - 1) * step : idx[(size_t)k - 1];
}
static void normalise(Stage& s) {
  if (s.kind == ST_RANGE) {
    if (s.step == 0) fail(DFDB_ERR_ARGUMENT, "ArgumentError: step cannot be zero");
    s.n = range_len(s.start, s.step, s.stop);
    if (s.n > 0) s.stop = s.start + (s.n - 1) * s.step;
  } else if (s.kind != ST_PRED) s.n = (int64_t)s.idx.size();
}

// add(q, elem) with _new_queue's rules (selection.jl:39-49)
void query_add_stage(dfdb_query* q, Stage&& ns) {
  normalise(ns);
  q->executed_stages = -1; q->count = -1; q->prefix_valid = false; q->gr_state = 0;
  Stage* last = q->stages.empty() ? nullptr : &q->stages.back();
  const bool new_is_range = ns.kind != ST_PRED;
  if (last && last->kind != ST_PRED && new_is_range) {   // range∘range collapses to old[elem] (:40)
    Stage r;
    if (last->kind == ST_INTEGER) {                       // Number indexing: only x[1]
      if (!(ns.kind == ST_INTEGER && ns.idx[0] == 1)) fail(DFDB_ERR_BOUNDS, "BoundsError: indexing a scalar selection");
      return;
    }
    if (ns.kind == ST_INTEGER) { r.kind = ST_INTEGER; r.idx = {last->elem(ns.idx[0])}; }
    else if (last->kind == ST_RANGE && ns.kind == ST_RANGE) {
      r.kind = ST_RANGE; r.step = last->step * ns.step;
      if (ns.n == 0) { r.start = last->start; r.stop = r.start - r.step; }
      else { r.start = last->elem(ns.start); r.stop = last->elem(ns.stop); }
    } else {
      r.kind = ST_INDICES; r.idx.resize((size_t)ns.n);
      for (int64_t k = 0; k < ns.n; k++) r.idx[(size_t)k] = last->elem(ns.kind == ST_RANGE ? ns.start + k * ns.step : ns.idx[(size_t)k]);
    }
    normalise(r);
    *last = std::move(r);
    return;
  }
  if (last && last->kind == ST_PRED && !new_is_range) {   // predicate∘predicate fuses with & (:44-47)
    last->pred = make_and(std::move(last->pred), std::move(ns.pred));
    return;
  }
  q->stages.push_back(std::move(ns));
}

// ---------------------------------------------------------------- device state
static size_t padded_words(int64_t nrows) { return (size_t)(round_up(nrows > 0 ? nrows : 1, kCTileRows) / 64 + 64); }

static void ensure_state(dfdb_query* q) {
  dfdb_table* t = q->t;
  if (t->nrows < 0) fail(DFDB_ERR_ARGUMENT, "table has no resident columns: call dfdb_table_load / add_column first");
  const int64_t nrows = t->nrows;
  if (q->bitmap_rows != nrows) {
    const size_t nw = padded_words(nrows);
    const int64_t ntiles = ceil_div(nrows, kTileRows);
    q->bitmap.ensure(nw * 8);
    HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, nw * 8, t->ctx->stream));   // pad words stay zero forever
    q->tile_counts.ensure((size_t)(ntiles + 8) * 4);
    q->prefix.ensure((size_t)(ntiles + 8) * 8);
    HIP_CHECK(hipMemsetAsync(q->prefix.p, 0, (size_t)(ntiles + 8) * 8, t->ctx->stream));
    q->scan_scratch.ensure(scan_counts_scratch_bytes(ntiles));
    q->bitmap_rows = nrows;
    q->executed_stages = -1;
  }
}

// ---------------------------------------------------------------- placement calibration
// K1 reads a column at ~7 TB/s while writing 1/64 of that as the bitmap.  How fast the pair runs depends on WHERE the two allocations sit
// relative to each other in physical memory (tools/bench_offset: every 8-GB column timed against every one of twelve bitmap allocations —
// 1.18 ms for some pairs, 1.32 ms for others, repeatably; offsets INSIDE an allocation change nothing), presumably DRAM bank conflicts between
// the read stream and the open bitmap rows under the address hash.  Neither hipMalloc nor the virtual-memory API lets the caller choose
// physical placement, so the engine measures: the first time a column of >= 2^26 rows is the target of a fresh-mask scan, the scan is timed
// against a few candidate bitmap allocations spread over free HBM (spacer allocations in between, released
// afterwards) and the fastest stays with the column.  Queries that scan the column borrow it (one at a time; others use their own).
// ctx option "placement_calibrate" = 1 asks for it (default 0).  One-time cost per column: 27 scans (~35 ms per 1e9 rows) plus allocating and releasing
// the spacers, 0.03-1.4 s measured (profiles/r2_placement_cost.txt) — worth it for a column that stays resident and is scanned thousands of times, not
// for a short session, which is why it is opt-in; stream slots never calibrate.  "placement_spacer_mb" / "placement_candidates" size the search (round 3: no
// spacers by default — once the column itself is re-placed they change nothing, tools/r3_draws_spacer.sh, and releasing 8 x 12 GB of them was up to 5 s of the one-time cost).
void query_return_mask(dfdb_query* q) {
  if (q->mask_from < 0 || !q->t) { q->mask_from = -1; return; }
  Column& c = q->t->cols[(size_t)q->mask_from];
  std::swap(q->bitmap, c.mask_pref); c.mask_lent = false; q->mask_from = -1;
  q->executed_stages = -1; q->count = -1; q->prefix_valid = false;
}
// Round 3: it is the COLUMN's allocation that decides most of it.  Six 8-GB columns with the same contents in one process (tools/r3_column_placement.py): one
// scans at 1.19-1.20 ms whatever bitmap it is paired with, one at 1.25-1.30, four at 1.28-1.32 — the nine bitmap candidates of one column differ by 1-4 %, the
// columns by 9 %.  So the calibration first RE-PLACES the column: a few fresh allocations of its size (all held until the choice is made: a freed one would be
// handed out again), the data copied device to device, the scan timed on each against the query's own bitmap; the fastest allocation becomes the column.
// "placement_column_candidates" (default 8, at most 16; 0 = bitmaps only).  Costs candidates x (the column's size of free HBM + a copy + 3 scans), once per column.
template <class Launch>
static void place_mask(dfdb_query* q, int ordinal, Launch&& launch /* (uint64_t* bitmap, int64_t rows, const void* column) */) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const int64_t nrows = t->nrows;
  if (nrows < ((int64_t)1 << 26) || ctx_option(ctx, "placement_calibrate", 0) == 0) return;
  if (q->mask_from == ordinal) return;
  query_return_mask(q);
  Column& c = t->cols[(size_t)ordinal];
  const size_t bytes = padded_words(nrows) * 8;
  if (!c.mask_calibrated) {
    c.mask_calibrated = true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    const auto wall0 = std::chrono::steady_clock::now();
    const int kCand = 8;                    // candidate bitmaps (round 6: the A/B knobs of the search — "placement_candidates", "placement_spacer_mb", "placement_count_candidates" — are constants now)
    const size_t spacer = 0;              // (round 2 held 12 GB between the candidates: seconds to release, nothing gained once the column is re-placed)
    if (free_b < (size_t)kCand * (bytes + spacer) + ((size_t)4 << 30)) return;      // not enough room to look around: keep the query's own
    std::vector<DevBuf> cand((size_t)kCand), space((size_t)kCand);
    try {
      for (int k = 0; k < kCand; k++) { cand[(size_t)k].ensure(bytes); if (spacer >= ((size_t)64 << 20)) space[(size_t)k].ensure(spacer); }
    } catch (const Error&) { return; }
    const int64_t sample = nrows;      // the WHOLE column: how a bitmap allocation pairs with the first eighth says little about the rest (measured)
    struct EventPair {                 // (a HIP_CHECK that throws inside time_on must not leak them: ADVICE r3)
      hipEvent_t a = nullptr, b = nullptr;
      ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_CHECK(hipEventCreate(&ev.a)); HIP_CHECK(hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    const void* colp = c.data.p;
    auto time_on = [&](uint64_t* bm) {
      float best = 1e30f;
      for (int r = 0; r < 3; r++) {
        HIP_CHECK(hipEventRecord(e0, s));
        launch(bm, sample, colp);
        HIP_CHECK(hipEventRecord(e1, s));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
      }
      return best;
    };
    // ---- the column first
    const int kCol = (int)std::min<int64_t>(16, std::max<int64_t>(0, ctx_option(ctx, "placement_column_candidates", 8)));
    if (kCol > 0 && dt_base(c.dtype) != DFDB_STRING && c.data.bytes > 0 && free_b > (size_t)(kCol + 1) * c.data.bytes + (size_t)kCand * (bytes + spacer) + ((size_t)8 << 30)) {
      std::vector<DevBuf> ccand((size_t)kCol);
      float tcol = time_on(q->bitmap.as<uint64_t>()), tcol_worst = tcol; int cbest = -1;
      for (int k = 0; k < kCol; k++) {
        try { ccand[(size_t)k].ensure(c.data.bytes); } catch (const Error&) { break; }
        HIP_CHECK(hipMemcpyAsync(ccand[(size_t)k].p, c.data.p, c.data.bytes, hipMemcpyDeviceToDevice, s));
        colp = ccand[(size_t)k].p;
        const float tk = time_on(q->bitmap.as<uint64_t>());
        if (tk < tcol) { tcol = tk; cbest = k; }
        if (tk > tcol_worst) tcol_worst = tk;
      }
      auto& pcb = ctx->prof["placement_column_best_us"]; pcb.launches++; pcb.ms += tcol * 1e3;
      auto& pcw = ctx->prof["placement_column_worst_us"]; pcw.launches++; pcw.ms += tcol_worst * 1e3;
      HIP_CHECK(hipStreamSynchronize(s));
      if (cbest >= 0 && tcol < 0.985f * tcol_worst) std::swap(c.data, ccand[(size_t)cbest]);      // the column lives in the fastest allocation from now on
      colp = c.data.p;
      ccand.clear();                                      // the others (and the old one) go back
    }
    float tbest = time_on(q->bitmap.as<uint64_t>()), tworst = tbest; int kbest = -1;
    for (int k = 0; k < kCand; k++) {
      const float tk = time_on(cand[(size_t)k].as<uint64_t>());
      if (tk < tbest) { tbest = tk; kbest = k; }
      if (tk > tworst) tworst = tk;
    }
    // ---- the query's tile-count array last (4 bytes per 1024 rows: K1 writes it beside the bitmap).  Its placement is worth 1-2.5 % of the scan (eight candidates in each
    // of six processes: 1.195-1.236 ms), the search costs a few 4-MB allocations and 3 scans each: "placement_count_candidates" (default 4, 0 = leave it)
    const int kTc = 4;
    if (kTc > 0 && q->tile_counts.bytes > 0) {
      uint64_t* const bmx = kbest >= 0 ? cand[(size_t)kbest].as<uint64_t>() : q->bitmap.as<uint64_t>();
      std::vector<DevBuf> tcand((size_t)kTc);
      float tc_best = time_on(bmx), tc_worst = tc_best; int tc_k = -1;
      for (int k = 0; k < kTc; k++) {
        try { tcand[(size_t)k].ensure(q->tile_counts.bytes); } catch (const Error&) { break; }
        std::swap(q->tile_counts, tcand[(size_t)k]);
        const float tk = time_on(bmx);
        std::swap(q->tile_counts, tcand[(size_t)k]);
        if (tk < tc_best) { tc_best = tk; tc_k = k; }
        if (tk > tc_worst) tc_worst = tk;
      }
      auto& pcb = ctx->prof["placement_counts_best_us"]; pcb.launches++; pcb.ms += tc_best * 1e3;
      auto& pcw = ctx->prof["placement_counts_worst_us"]; pcw.launches++; pcw.ms += tc_worst * 1e3;
      HIP_CHECK(hipStreamSynchronize(s));
      if (tc_k >= 0) std::swap(q->tile_counts, tcand[(size_t)tc_k]);
    }
    c.mask_ms_best = tbest; c.mask_ms_worst = tworst;
    auto& pe = ctx->prof["placement_best_us"]; pe.launches++; pe.ms += tbest * 1e3;
    auto& pw = ctx->prof["placement_worst_us"]; pw.launches++; pw.ms += tworst * 1e3;
    if (kbest >= 0 && tbest < 0.985f * tworst) {       // a real difference, and the query's own allocation is not the winner
      c.mask_pref = std::move(cand[(size_t)kbest]);
      HIP_CHECK(hipMemsetAsync(c.mask_pref.p, 0, bytes, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));                 // candidates and spacers die here
    HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, bytes, s));
    cand.clear(); space.clear();                        // (freed inside the timed span: releasing the spacers is most of the cost)
    auto& pt = ctx->prof["placement_wall_us"]; pt.launches++;
    pt.ms += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - wall0).count();
  }
  if (c.mask_pref.p && !c.mask_lent && c.mask_pref.bytes >= bytes) {
    std::swap(q->bitmap, c.mask_pref); c.mask_lent = true; q->mask_from = ordinal;
  }
}

static uint64_t splitmix64_host(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

// (a compressed-only column — keep_compressed = 2 — that reaches a consumer through here gets a whole-column decode for the ABI call in progress:
//  table.cpp column_data; the paths that need none — K7's fused scan, the survivors' arena of the gathers — do not come through here)
static const Column& need_resident(dfdb_table* t, int ordinal) {
  Column& c = t->cols[(size_t)ordinal];
  if (!c.resident) fail(DFDB_ERR_ARGUMENT, "column %s is not resident on the device (dfdb_table_load it first)", c.name.c_str());
  if (c.comp_only) (void)column_data(t, c);
  return c;
}
// can K7 itself evaluate this simple term over a compressed-only column (HIST + SCAN: decode and compare in one pass, nothing decoded is kept)?
static bool comp_scannable(const dfdb_table* t, const ScanTerm& tm, int ord) {
  const Column& c = t->cols[(size_t)ord];
  return c.resident && c.comp_only && !c.data.p && tm.pre == 0 && (tm.dtype == DFDB_I64 || tm.dtype == DFDB_U64 || tm.dtype == DFDB_F64) && !dt_nullable(c.dtype) &&
         t->block_size % kTileRows == 0;
}

static void raise_reached_errors(dfdb_query* q, int nstages);
// The bitmap was changed outside query_execute (unique narrowed it to the first occurrences, groupreduce put the full selection back): everything derived
// from the PREVIOUS bitmap goes — the host count, and the survivors' arenas of compressed-only projection columns, which hold only the blocks that kept a
// row under the selection they were decoded for (a fetch restored the full selection and a later materialize gathered through the narrowed arena).
static void selection_changed(dfdb_query* q) {
  q->count = -1;
  for (auto& a : q->arenas) a.second.valid = false;
}
static void scan_prefix(dfdb_query* q) {
  dfdb_ctx* ctx = q->t->ctx;
  LaunchTimer lt(ctx, "scan_counts");
  launch_scan_counts(ctx->stream, q->tile_counts.as<uint32_t>(), q->prefix.as<uint64_t>(), ceil_div(q->t->nrows, kTileRows), q->scan_scratch.as<uint64_t>());
  q->prefix_valid = true;
}

// K9: does dictionary entry `e` satisfy the string term (mode 0 ==, 1 !=, 2 startswith, 3 endswith; k_strings.hip's modes)?
static bool dict_entry_matches(const std::string& e, int mode, const std::string& pat) {
  if (mode <= 1) { const bool r = e == pat; return mode == 1 ? !r : r; }
  if (mode == 2) return e.size() >= pat.size() && e.compare(0, pat.size(), pat) == 0;
  return e.size() >= pat.size() && e.compare(e.size() - pat.size(), pat.size(), pat) == 0;
}
static void run_dict_scan(dfdb_query* q, const Column& col, const std::vector<uint32_t>& lut, bool have) {
  dfdb_ctx* ctx = q->t->ctx; hipStream_t s = ctx->stream;
  DevBuf& lb = q->tmp_a; lb.ensure(lut.size() * 4 + 64);
  HIP_CHECK(hipMemcpyAsync(lb.p, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, s));
  stream_wait(ctx);                                      // `lut` is pageable host memory
  LaunchTimer lt(ctx, "dict_scan");
  launch_dict_scan(s, col.dict_codes.as<uint16_t>(), lb.as<uint32_t>(), (int32_t)lut.size(), q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), q->t->nrows, have);
}

// predicate stage = AND of its conjuncts, each routed to the cheapest kernel that is exact for it
static void run_predicate(dfdb_query* q, const Node& pred, bool first_stage, bool last_stage) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const int64_t nrows = t->nrows;
  std::vector<const Node*> conj; flatten_and(pred, conj);
  std::vector<const Node*> generic, strs; ScanTerms terms{}; terms.n = 0; terms.combine_or = 0;
  std::vector<ScanTerms> term_batches;
  std::vector<int> term_ords;
  // a conjunct that is itself a disjunction of simple terms — (a > c1) | (x < c2) — is one k_scan_terms pass with combine_or
  std::vector<ScanTerms> or_batches;
  auto match_or = [&](const Node& n, ScanTerms& out) {
    std::vector<const Node*> leaves, todo{&n};
    while (!todo.empty()) {
      const Node* x = todo.back(); todo.pop_back();
      if (x->op == DFIR_OR && x->a && x->b) { todo.push_back(x->b.get()); todo.push_back(x->a.get()); } else leaves.push_back(x);
    }
    out = ScanTerms{}; out.n = 0; out.combine_or = 1;
    for (const Node* l : leaves) {
      // in.(col, Ref([v1, v2, ...])) (test/broadcast.jl:63-71) with a few values is the disjunction col == v1 | col == v2 | ...
      if (l->op == DFIR_IN_SET && l->a && l->b && l->a->op == DFIR_COL && l->b->op == DFIR_CONST_SET && !l->b->set.empty()) {
        for (uint64_t bits : l->b->set) {
          Node eq; eq.op = DFIR_EQ; eq.dtype = DFDB_BOOL;
          eq.a = l->a->clone();
          eq.b = std::make_unique<Node>(); eq.b->op = DFIR_CONST; eq.b->dtype = l->b->set_dtype; eq.b->cbits = bits;
          ScanTerm tm; int ord;
          if (out.n == kMaxTerms || !match_simple_term(eq, *t, tm, ord)) return false;
          tm.col = need_resident(t, ord).data.p;
          out.t[out.n++] = tm;
        }
        continue;
      }
      ScanTerm tm; int ord;
      if (out.n == kMaxTerms || !match_simple_term(*l, *t, tm, ord)) return false;
      tm.col = need_resident(t, ord).data.p;
      out.t[out.n++] = tm;
    }
    return out.n >= 2;
  };
  // ismissing(col) / !ismissing(col) over a nullable fixed-width column: its missing bitmap is the mask
  // a disjunction (and negations) of string terms over ONE dictionary column — (brand == "sony") | (brand == "apple"), !startswith(brand, "x") — is a
  // Boolean function of the dictionary entry: one bit-table lookup of the codes instead of string compares in the interpreter
  struct DictLut { int ord; std::vector<uint32_t> lut; };
  std::vector<DictLut> dict_luts;
  std::function<bool(const Node&, int&)> dict_only = [&](const Node& n, int& ord) -> bool {
    if ((n.op == DFIR_OR || n.op == DFIR_AND || n.op == DFIR_XOR) && n.a && n.b) return dict_only(*n.a, ord) && dict_only(*n.b, ord);
    if (n.op == DFIR_NOT && n.a) return dict_only(*n.a, ord);
    int o, mode; std::string pat;
    if (!match_string_term(n, *t, o, mode, pat)) return false;
    if (ord >= 0 && o != ord) return false;
    if (t->cols[(size_t)o].dict_n <= 0 || !t->cols[(size_t)o].resident) return false;
    ord = o; return true;
  };
  std::function<bool(const Node&, const std::string&)> dict_eval = [&](const Node& n, const std::string& e) -> bool {
    if (n.op == DFIR_OR) return dict_eval(*n.a, e) || dict_eval(*n.b, e);
    if (n.op == DFIR_AND) return dict_eval(*n.a, e) && dict_eval(*n.b, e);
    if (n.op == DFIR_XOR) return dict_eval(*n.a, e) != dict_eval(*n.b, e);
    if (n.op == DFIR_NOT) return !dict_eval(*n.a, e);
    int o, mode; std::string pat; match_string_term(n, *t, o, mode, pat);
    return dict_entry_matches(e, mode, pat);
  };
  struct MissTerm { const uint64_t* bits; bool negate; };
  std::vector<MissTerm> miss;
  struct CompTerm { ScanTerm tm; int ord; };
  std::vector<CompTerm> comp_terms;
  auto match_missing = [&](const Node& n, MissTerm& out) {
    const Node* m = &n; bool neg = false;
    if (m->op == DFIR_NOT && m->a) { m = m->a.get(); neg = true; }
    if (m->op != DFIR_ISMISSING || !m->a || m->a->op != DFIR_COL) return false;
    const Column& col = need_resident(t, m->a->col);
    if (!dt_nullable(col.dtype) || dt_base(col.dtype) == DFDB_STRING || !col.missing.p) return false;
    out = MissTerm{col.missing.as<uint64_t>(), neg};
    return true;
  };
  for (const Node* c : conj) {
    ScanTerm tm; int ord; int mode; std::string pat;
    ScanTerms ob;
    MissTerm mt;
    if (match_missing(*c, mt)) { miss.push_back(mt); continue; }
    if (c->op == DFIR_OR || c->op == DFIR_NOT || c->op == DFIR_XOR) {
      int dord = -1;
      if (dict_only(*c, dord) && dord >= 0) {
        const Column& dc = t->cols[(size_t)dord];
        DictLut dl{dord, std::vector<uint32_t>((size_t)(dc.dict_n + 31) / 32, 0u)};
        for (int32_t k = 0; k < dc.dict_n; k++) if (dict_eval(*c, dc.dict_host[(size_t)k])) dl.lut[(size_t)k >> 5] |= 1u << (k & 31);
        dict_luts.push_back(std::move(dl));
        continue;
      }
    }
    if ((c->op == DFIR_OR || c->op == DFIR_IN_SET) && match_or(*c, ob)) { or_batches.push_back(ob); continue; }
    const bool simple = match_simple_term(*c, *t, tm, ord);      // (once: a failed match leaves `tm` half-written)
    if (simple && comp_scannable(t, tm, ord)) {
      // a compressed-only column: the decoder evaluates the term (a second comparison of the same column folds into an interval there too)
      bool folded = false;
      for (CompTerm& ct : comp_terms) if (ct.ord == ord && ct.tm.op2 < 0 && ct.tm.dtype == tm.dtype) { ct.tm.op2 = tm.op; ct.tm.cbits2 = tm.cbits; folded = true; break; }
      if (!folded) comp_terms.push_back(CompTerm{tm, ord});
      continue;
    }
    if (simple) {
      tm.col = need_resident(t, ord).data.p;
      // a second comparison of a column the current batch already compares folds into that term as an interval: `65 > x > 34`
      // (test/selection.jl:53) reads x once instead of twice
      bool folded = false;
      const size_t ord0 = term_ords.size() - (size_t)terms.n;
      for (int k = 0; k < terms.n && !folded; k++)
        if (term_ords[ord0 + (size_t)k] == ord && terms.t[k].op2 < 0 && terms.t[k].dtype == tm.dtype && terms.t[k].pre == tm.pre &&
            (tm.pre == 0 || (terms.t[k].pre_d == tm.pre_d && terms.t[k].pre_magic == tm.pre_magic))) { terms.t[k].op2 = tm.op; terms.t[k].cbits2 = tm.cbits; folded = true; }
      if (folded) continue;
      if (terms.n == kMaxTerms) { term_batches.push_back(terms); terms.n = 0; }
      terms.t[terms.n++] = tm; term_ords.push_back(ord);
    } else if (match_string_term(*c, *t, ord, mode, pat)) strs.push_back(c);
    else if (c->op == DFIR_COALESCE && c->a && c->b && c->b->op == DFIR_CONST && dt_base(c->b->dtype) == DFDB_BOOL && !dt_nullable(c->b->dtype) && c->b->cbits == 0 &&
             match_string_term(*c->a, *t, ord, mode, pat, true))
      strs.push_back(c->a.get());     // coalesce(s == "x", false) over a Union{String,Missing} column (the docs' real data set: index.md:264-272): K5's own answer, not the interpreter's
    else generic.push_back(c);
  }
  if (terms.n) term_batches.push_back(terms);
  bool have = !first_stage;   // does the bitmap already hold a mask to AND with?
  // generic conjuncts first, as ONE interpreter program over `c1 & c2 & ...`: Julia's fused `&` is not short-circuit
  // (BlockBroadcasting(&, (old, elem)), selection.jl:44-47), so every conjunct is evaluated — and may raise DivideError /
  // InexactError — on every row that REACHED this stage, not only on the rows an earlier conjunct kept
  if (generic.size() == 1) { run_interp_predicate(q, *generic[0], have); have = true; }
  else if (!generic.empty()) {
    NodePtr all = generic[0]->clone();
    for (size_t i = 1; i < generic.size(); i++) all = make_and(std::move(all), generic[i]->clone());
    run_interp_predicate(q, *all, have); have = true;
  }
  for (const DictLut& dl : dict_luts) { run_dict_scan(q, t->cols[(size_t)dl.ord], dl.lut, have); have = true; }
  for (const Node* c : strs) {
    int ord, mode; std::string pat; match_string_term(*c, *t, ord, mode, pat, true);
    const Column& col = need_resident(t, ord);
    const bool str_nullable = dt_nullable(col.dtype);
    // `col == "const"` is an AND-ed conjunct of this stage, and later stages only remove rows: every row the query finally selects holds exactly
    // `const` in this column, so its projection needs neither the column nor a capture (materialize_col: launch_fill_const_strings)
    if (mode == 0 && !str_nullable) { q->const_str_col = ord; q->const_str = pat; }
    if (col.dict_n > 0) {
      // K9: the column has a dictionary — the term is decided once per distinct string on the host, the rows are a bit-table lookup of their codes
      std::vector<uint32_t> lut((size_t)(col.dict_n + 31) / 32, 0u);
      for (int32_t k = 0; k < col.dict_n; k++) if (dict_entry_matches(col.dict_host[(size_t)k], mode, pat)) lut[(size_t)k >> 5] |= 1u << (k & 31);
      run_dict_scan(q, col, lut, have);
      have = true;
      continue;
    }
    DevBuf& pb = q->tmp_a; pb.ensure(pat.size() + 64);
    if (pat.size() > 16) { HIP_CHECK(hipMemcpyAsync(pb.p, pat.data(), pat.size(), hipMemcpyHostToDevice, s)); stream_wait(q->t->ctx); }   // (bytes past 16 are compared against the device copy)
    // capture (see dfdb_query.hint_materialize): only when this ONE launch produces the query's final mask and the column it reads is
    // itself projected — the match pass then keeps the selected rows' sizes and bytes and K6 never reads the column again
    StrCapture capture{nullptr, nullptr, nullptr};
    // (every other kind of conjunct runs AFTER this launch and narrows the mask: the capture would keep rows the query drops — found by tests/test_gpu_fuzz.py)
    bool do_cap = mode != 0 && !str_nullable && q->hint_materialize && q->stages.size() == 1 && !have && generic.empty() && term_batches.empty() && or_batches.empty() && miss.empty() && comp_terms.empty() &&
                  dict_luts.empty() && strs.size() == 1 && pat.size() <= 64;
    if (do_cap) {
      do_cap = false;
      for (const ProjCol& p : q->proj) if (p.expr->op == DFIR_COL && p.expr->col == ord) { do_cap = true; break; }
    }
    if (do_cap) {
      const int64_t nt = ceil_div(nrows, kTileRows);
      q->cap_str_sizes.ensure((size_t)nt * kTileRows * 4 + 256);
      q->cap_str_bytes.ensure((size_t)col.nbytes + 256);
      q->cap_str_tb.ensure((size_t)(nt + 8) * 4);
      capture = StrCapture{q->cap_str_sizes.as<int32_t>(), q->cap_str_bytes.as<uint8_t>(), q->cap_str_tb.as<uint32_t>()};
    }
    LaunchTimer lt(ctx, "str_match");
    launch_str_match(s, col.data.as<int32_t>(), (const int64_t*)col.tile_off.p, col.bytes.as<uint8_t>(), (const uint8_t*)pat.data(),
                     pb.as<uint8_t>(), (int32_t)pat.size(), mode, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), nrows, have,
                     do_cap ? &capture : nullptr, col.max_tile_bytes);
    if (do_cap) q->cap_str_col = ord;
    have = true;
  }
  for (const MissTerm& mt : miss) {
    LaunchTimer lt(ctx, "missing_mask");
    launch_missing_mask(s, mt.bits, mt.negate, have, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), nrows);
    have = true;
  }
  for (const ScanTerms& ob : or_batches) {
    LaunchTimer lt(ctx, "scan_terms");
    launch_scan_terms(s, ob, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), nrows, have, 0, nullptr);
    have = true;
  }
  // compressed-only columns: one K7 launch per term — every block decoded into the waves' history rings, the comparison applied to the bytes as they leave
  // the LDS ring, bitmap + tile counts the only output.  From the second mask on the words are AND-ed in and a block without a survivor
  // is not decoded at all (blocksiterator.jl:111-113 at block granularity).
  for (const CompTerm& ct : comp_terms) {
    Column& fc = t->cols[(size_t)ct.ord];
    LzScan sc{q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), ct.tm.cbits, ct.tm.dtype, ct.tm.op};
    sc.op2 = ct.tm.op2; sc.cbits2 = ct.tm.cbits2; sc.and_existing = have ? 1 : 0;
    int waves = 0; uint8_t* scratch = ctx_hist_scratch(ctx, &waves);
    const int imode = column_lz4_index(ctx, fc, true);
    LaunchTimer lt(ctx, "lz4_decode_scan_hist");
    prof_note(ctx, imode == 2 ? "lz4_decode_scan_hist.indexed" : imode == 1 ? "lz4_decode_scan_hist.recording" : "lz4_decode_scan_hist.plain");
    launch_lz4_decode_hist(s, fc.comp.as<uint8_t>(), scratch, waves, fc.comp_blocks.as<Lz4Block>(), (int32_t)fc.comp_nblocks, fc.comp_status.as<int32_t>(), &sc,
                           fc.comp_index.as<uint32_t>(), imode);
    if (std::find(q->comp_scanned.begin(), q->comp_scanned.end(), ct.ord) == q->comp_scanned.end()) q->comp_scanned.push_back(ct.ord);
    have = true;
  }
  // The launch that produces the query's final mask can do more with its LAST term (k_scan_terms EXTRA):
  //   sum     (dfdb_query_hint_aggregate): the hinted column is a term of the last batch of the last stage -> per-tile partial sums
  //   capture (dfdb_query_hint_materialize): a single-stage query of one batch of terms, one of them a projected 8-byte column
  int extra = 0, special = -1;
  if (!term_batches.empty() && !term_batches.back().combine_or) {
    ScanTerms& lb = term_batches.back();
    const size_t ord0 = term_ords.size() - (size_t)lb.n;           // term_ords of the last batch start here
    if ((q->hint_agg_op == DFDB_AGG_SUM || q->hint_agg_op == DFDB_AGG_MIN || q->hint_agg_op == DFDB_AGG_MAX) && last_stage && q->hint_agg_proj >= 0 &&
        (size_t)q->hint_agg_proj < q->proj.size()) {
      const Node& pe = *q->proj[(size_t)q->hint_agg_proj].expr;
      if (pe.op == DFIR_COL && !dt_nullable(pe.dtype))
        for (int k = 0; k < lb.n && special < 0; k++) {
          const int dt = lb.t[k].dtype;
          if (lb.t[k].pre) continue;                              // (a rem term compares rem(x, m): not the value to add up)
          if (term_ords[ord0 + (size_t)k] == pe.col && (dt == DFDB_I64 || dt == DFDB_U64 || dt == DFDB_F64)) {
            special = k; extra = q->hint_agg_op == DFDB_AGG_SUM ? 2 : (q->hint_agg_op == DFDB_AGG_MIN ? 3 : 4);
          }
        }
    }
    // capture: the last batch of the LAST stage is the launch that produces the query's final mask, whatever ran before it in this stage (generic
    // conjuncts, string and dictionary scans, missing masks, disjunctions all come first and only hand it a mask to AND with) or in earlier stages.
    // Every projected plain 8-byte column among its terms — up to two — is kept by the scan instead of gathered afterwards (round 4; round 1-3 captured
    // one column of a single-stage query whose only conjuncts were simple terms).
    int special2 = -1;
    if (!extra && q->hint_materialize && last_stage && ctx_option(ctx, "scan_capture", 2) > 0) {
      const int maxcap = (int)std::min<int64_t>(2, ctx_option(ctx, "scan_capture", 2));
      for (int k = 0; k < lb.n && (special < 0 || (special2 < 0 && maxcap >= 2)); k++) {
        const int dt = lb.t[k].dtype;
        if ((dt != DFDB_I64 && dt != DFDB_U64 && dt != DFDB_F64) || lb.t[k].pre) continue;
        if (special >= 0 && term_ords[ord0 + (size_t)k] == term_ords[ord0 + (size_t)special]) continue;     // (two terms of one column that did not fold)
        for (const ProjCol& p : q->proj) {
          bool wanted = p.expr->op == DFIR_COL && p.expr->col == term_ords[ord0 + (size_t)k] && !dt_nullable(p.expr->dtype);
          if (!wanted && p.expr->op != DFIR_COL && dt_width(p.expr->dtype) == 8 && !dt_nullable(p.expr->dtype)) {
            // a computed column that is a transform of this one column (`x * 2`, `a % 7`, `a * 3 + 1`): it is made from the captured values too
            ScanTerm tf; const Node* tcol = nullptr;
            wanted = match_column_transform(p.expr.get(), tf, tcol) && tf.pre != 0 && tcol->col == term_ords[ord0 + (size_t)k] && !dt_nullable(tcol->dtype);
          }
          if (wanted) { if (special < 0) { special = k; extra = 1; } else { special2 = k; extra = 5; } break; }
        }
      }
    }
    if (extra) {                                                     // the special term goes last (and the second captured one before it)
      std::swap(lb.t[special], lb.t[lb.n - 1]);
      std::swap(term_ords[ord0 + (size_t)special], term_ords[ord0 + (size_t)lb.n - 1]);
      if (extra == 5) {
        if (special2 == lb.n - 1) special2 = special;                // (it was sitting in the last place and has just been swapped away)
        std::swap(lb.t[special2], lb.t[lb.n - 2]);
        std::swap(term_ords[ord0 + (size_t)special2], term_ords[ord0 + (size_t)lb.n - 2]);
        q->cap_buf2.ensure((size_t)round_up(nrows, kTileRows) * 8 + 256);
      }
      if (extra == 1 || extra == 5) q->cap_buf.ensure((size_t)round_up(nrows, kTileRows) * 8 + 256);
      else q->agg_partials.ensure((size_t)(ceil_div(nrows, kTileRows) + 8) * 8);
    }
  }
  if (!have && !term_batches.empty() && first_stage) {
    // the launch that writes a fresh mask over the whole column: run it against the bitmap allocation this column pairs best with
    const ScanTerms tb0 = term_batches[0];
    const bool nt = true;                 // nontemporal column loads (the A/B knob "scan_nt" went in round 6: plain loads were never faster)
    const int wt0 = (ctx_option(ctx, "scan_wt_store", 1) ? 1 : 0) | (int)((ctx_option(ctx, "scan_narrow", 1) & 3) << 1);
    const void* const col_before = t->cols[(size_t)term_ords[0]].data.p;
    place_mask(q, term_ords[0], [&](uint64_t* bm, int64_t rows, const void* colp) {
      ScanTerms tbx = tb0;                                          // the calibration may be trying another allocation of the first term's column
      for (int k = 0; k < tbx.n; k++) if (tbx.t[k].col == col_before) tbx.t[k].col = colp;
      if (tbx.n == 1 && tbx.t[0].op2 < 0 && tbx.t[0].pre == 0) launch_scan_cmp(s, tbx.t[0].col, tbx.t[0].dtype, tbx.t[0].op, tbx.t[0].cbits, bm, q->tile_counts.as<uint32_t>(), rows, false, nt, nullptr, wt0);
      else launch_scan_terms(s, tbx, bm, q->tile_counts.as<uint32_t>(), rows, false, 0, nullptr, (int)ctx_option(ctx, "scan_pair", 1));
    });
    const void* const col_after = t->cols[(size_t)term_ords[0]].data.p;
    if (col_after != col_before) {                                  // the column was re-placed: every term of this stage that reads it follows
      for (ScanTerms& tbk : term_batches) for (int k = 0; k < tbk.n; k++) if (tbk.t[k].col == col_before) tbk.t[k].col = col_after;
      for (ScanTerms& tbk : or_batches) for (int k = 0; k < tbk.n; k++) if (tbk.t[k].col == col_before) tbk.t[k].col = col_after;
    }
  }
  for (size_t bi = 0; bi < term_batches.size(); bi++) {
    const ScanTerms& tb = term_batches[bi];
    const int ex = bi + 1 == term_batches.size() ? extra : 0;
    // decode -> scan fusion: a fresh-mask scan of ONE simple term over an 8-byte column that holds its LZ4 blocks in HBM
    // (ctx option keep_compressed at load time) decodes the blocks and evaluates the term in the same pass (K7 SCAN): what the reference's loop
    // body does per block (read_block! then apply the selection, blocksiterator.jl:98-121).  ctx option "decode_on_scan" = 1 asks for it;
    // the decoded column is (re)written on the way, so everything after this launch sees an ordinary resident column.
    if (tb.n == 1 && tb.t[0].op2 < 0 && tb.t[0].pre == 0 && ex == 0 && !have && first_stage && bi == 0 && ctx_option(ctx, "decode_on_scan", 0) != 0) {
      Column& fc = t->cols[(size_t)term_ords[0]];
      const int fdt = tb.t[0].dtype;
      if (fc.comp_nblocks > 0 && !dt_nullable(fc.dtype) && (fdt == DFDB_I64 || fdt == DFDB_U64 || fdt == DFDB_F64) && t->block_size % kTileRows == 0) {
        q->decoded_col = term_ords[0];                   // query_count looks at the blocks' statuses where it waits for the count anyway
        // few blocks (every one resident in the two-wave pipeline at once): the pipeline, then the ordinary scan of the decoded column, is the shorter way —
        // a block's latency is what a small launch pays, and the fused form is one wave per block (1 526 blocks: 328 GB/s + a 0.13-ms scan against ~210 GB/s fused)
        const int pipe = (int)ctx_option(ctx, "lz4_pipeline", -1);
        if (pipe == 1 || (pipe < 0 && fc.comp_nblocks <= 2048)) {
          const int dmode = column_lz4_index(ctx, fc, lz4_decode_takes_index((int32_t)fc.comp_nblocks, pipe));
          LaunchTimer lt(ctx, "lz4_decode");
          prof_note(ctx, dmode == 2 ? "lz4_decode.indexed" : dmode == 1 ? "lz4_decode.recording" : "lz4_decode.plain");
          launch_lz4_decode(s, fc.comp.as<uint8_t>(), fc.data.as<uint8_t>(), fc.comp_blocks.as<Lz4Block>(), (int32_t)fc.comp_nblocks, fc.comp_status.as<int32_t>(), pipe,
                            fc.comp_index.as<uint32_t>(), dmode);
        } else {
        const int imode = column_lz4_index(ctx, fc, true);
        LaunchTimer lt(ctx, "lz4_decode_scan");
        prof_note(ctx, imode == 2 ? "lz4_decode_scan.indexed" : imode == 1 ? "lz4_decode_scan.recording" : "lz4_decode_scan.plain");
        launch_lz4_decode_scan(s, fc.comp.as<uint8_t>(), fc.data.as<uint8_t>(), fc.comp_blocks.as<Lz4Block>(), (int32_t)fc.comp_nblocks, fc.comp_status.as<int32_t>(),
                               LzScan{q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), tb.t[0].cbits, fdt, tb.t[0].op}, fc.comp_index.as<uint32_t>(), imode);
        have = true;
        continue;
        }
      }
    }
    if (tb.n == 1 && tb.t[0].op2 < 0 && tb.t[0].pre == 0 && ex < 2 && !(ex == 1 && have)) {      // (k_scan_cmp captures over a fresh mask only)
      LaunchTimer lt(ctx, "scan_cmp");
      prof_note(ctx, ctx_option(ctx, "scan_wt_store", 1) ? "scan_cmp.wt_store" : "scan_cmp.plain_store");
      launch_scan_cmp(s, tb.t[0].col, tb.t[0].dtype, tb.t[0].op, tb.t[0].cbits, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), nrows, have,
                      true, ex == 1 ? q->cap_buf.p : nullptr,
                      (ctx_option(ctx, "scan_wt_store", 1) ? 1 : 0) | (int)((ctx_option(ctx, "scan_narrow", 1) & 3) << 1));
    } else {
      LaunchTimer lt(ctx, "scan_terms");
      const int pair = (int)ctx_option(ctx, "scan_pair", 1);
      if (pair && scan_pair_applies(tb, have)) prof_note(ctx, "scan_terms.pair");
      launch_scan_terms(s, tb, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), nrows, have, ex, (ex == 1 || ex == 5) ? q->cap_buf.p : ex >= 2 ? q->agg_partials.p : nullptr, pair,
                        ex == 5 ? q->cap_buf2.p : nullptr);
    }
    have = true;
  }
  if (extra == 1 || extra == 5) q->cap_col = term_ords.back();
  if (extra == 5) q->cap_col2 = term_ords[term_ords.size() - 2];
  if (extra >= 2 && extra <= 4) { q->agg_col = term_ords.back(); q->agg_dtype = term_batches.back().t[term_batches.back().n - 1].dtype; q->agg_op = q->hint_agg_op; }
}

static void run_range(dfdb_query* q, const Stage& st, bool first_stage) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  RangeSpec r{};
  if (st.kind == ST_RANGE) {
    r.kind = 0; r.first = st.first(); r.last = st.last(); r.step = st.step > 0 ? st.step : -st.step;
    if (st.n == 0) { r.first = 1; r.last = 0; r.step = 1; }
  } else {   // index vectors keep table order and collapse duplicates (quirk Q3): membership in the sorted unique list
    std::vector<int64_t> sorted(st.idx);
    std::sort(sorted.begin(), sorted.end());
    sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
    q->idx_sorted.ensure(sorted.size() * 8 + 64);
    if (!sorted.empty()) HIP_CHECK(hipMemcpyAsync(q->idx_sorted.p, sorted.data(), sorted.size() * 8, hipMemcpyHostToDevice, s));
    stream_wait(q->t->ctx);   // `sorted` is pageable host memory
    r.kind = 1; r.sorted = q->idx_sorted.as<int64_t>(); r.nsorted = (int64_t)sorted.size();
    r.first = sorted.empty() ? 1 : sorted.front(); r.last = sorted.empty() ? 0 : sorted.back(); r.step = 1;
  }
  if (!first_stage && !q->prefix_valid) scan_prefix(q);
  // a leading range stage numbers TABLE rows (row_base makes that global on a shard); later stages number
  // the survivors, continuing after the survivors that live on lower ranks (stage_base)
  const int64_t rank_base = first_stage ? t->row_base : st.stage_base;
  LaunchTimer lt(ctx, "range_stage");
  launch_range_stage(s, r, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), q->tile_counts.as<uint32_t>(), t->nrows, rank_base, first_stage);
  q->prefix_valid = false;
}

// A predicate hit DivideError / InexactError on some rows.  Julia raises it only if the reference's iteration EVALUATES the block such a row lives in
// (src/io/blocksiterator.jl:69-121): blocks wholly before the first element of a leading range stage are skipped unread (skip_if_can,
// selection.jl:177-190), and once ANY range-like stage has been fed its last element (`last <= offset`, is_finished :192-196, tested before every block)
// nothing more is read.  `offset` of stage k before a block = the survivors of stages [0, k) in the rows before that block (+ stage_base on a shard).
// The rows before the erroring block raise nothing (it is the SMALLEST erroring row), so those counts come from ordinary partial executions.
static int64_t selected_before(dfdb_query* q, int64_t row) {                 // set bits of q's current bitmap in rows [0, row)
  dfdb_ctx* ctx = q->t->ctx; hipStream_t s = ctx->stream;
  const int64_t tile = row / kTileRows, words = (row % kTileRows) / 64, rem = row % 64;
  uint64_t pre = 0, w[17] = {0};
  HIP_CHECK(hipMemcpyAsync(&pre, q->prefix.as<uint64_t>() + tile, 8, hipMemcpyDeviceToHost, s));
  if (row % kTileRows) HIP_CHECK(hipMemcpyAsync(w, q->bitmap.as<uint64_t>() + tile * 16, (size_t)(words + 1) * 8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  int64_t c = (int64_t)pre;
  for (int64_t k = 0; k < words; k++) c += __builtin_popcountll(w[k]);
  if (rem) c += __builtin_popcountll(w[words] & ((1ull << rem) - 1ull));
  return c;
}
// survivors of the current execution in every block of `block_size` rows.  Blocks that are a whole number of 1024-row tiles read one entry of the
// prefix scan per block (a strided copy); other block sizes (the reference's tests use 50 and 100) count the bits of the mask on the host.
void query_block_counts(dfdb_query* q, int64_t block_size, std::vector<int64_t>& counts) {
  dfdb_table* t = q->t; hipStream_t s = t->ctx->stream;
  const int64_t nrows = t->nrows < 0 ? 0 : t->nrows;
  const int64_t nb = ceil_div(nrows, block_size);
  counts.assign((size_t)nb, 0);
  if (nb == 0) return;
  if (!q->prefix_valid) scan_prefix(q);
  const int64_t ntiles = ceil_div(nrows, kTileRows);
  if (block_size % kTileRows == 0) {
    const int64_t tpb = block_size / kTileRows;
    std::vector<uint64_t> at((size_t)nb + 1, 0);
    HIP_CHECK(hipMemcpy2DAsync(at.data(), 8, q->prefix.p, (size_t)tpb * 8, 8, (size_t)nb, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&at[(size_t)nb], q->prefix.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (int64_t b = 0; b < nb; b++) counts[(size_t)b] = (int64_t)(at[(size_t)b + 1] - at[(size_t)b]);
    return;
  }
  const size_t nw = (size_t)ceil_div(nrows, 64);
  std::vector<uint64_t> bm(nw);
  HIP_CHECK(hipMemcpyAsync(bm.data(), q->bitmap.p, nw * 8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  for (int64_t b = 0; b < nb; b++) {
    const int64_t r0 = b * block_size, r1 = std::min(nrows, r0 + block_size);
    int64_t c = 0;
    for (int64_t r = r0; r < r1;) {
      const int64_t wi = r >> 6, lo = r & 63, hi = std::min<int64_t>(64, r1 - (wi << 6));
      uint64_t w = bm[(size_t)wi] >> lo;
      if (hi - lo < 64) w &= (1ull << (hi - lo)) - 1ull;
      c += __builtin_popcountll(w);
      r = (wi << 6) + hi;
    }
    counts[(size_t)b] = c;
  }
}

static bool error_is_reached(dfdb_query* q, uint64_t erow) {
  dfdb_table* t = q->t;
  const int64_t bs = t->block_size;
  const int64_t blk_row0 = ((int64_t)erow / bs) * bs;                          // first local row of the erroring block
  for (size_t k = 0; k < q->stages.size(); k++) {
    const Stage& st = q->stages[k];
    if (st.kind == ST_PRED) continue;
    if (k == 0) {
      // a leading range numbers table rows: blocks that end before its first element are skipped, blocks after its last are never read
      if (st.first() - (t->row_base + blk_row0) > bs) return false;         // _skip_if_can: elem.first - elem.offset > size_to_skip (the table's block size)
      if (st.last() <= t->row_base + blk_row0) return false;
      continue;
    }
    query_execute(q, (int)k);                                                  // survivors of stages [0, k): what stage k is fed
    if (st.last() <= st.stage_base + selected_before(q, blk_row0)) return false;
  }
  return true;
}
static void raise_reached_errors(dfdb_query* q, int nstages) {
  const uint64_t er[2] = {q->err_row[0], q->err_row[1]};
  bool any_range = false;
  for (const Stage& st : q->stages) any_range = any_range || st.kind != ST_PRED;
  int raise = -1;
  if (!any_range) raise = er[0] <= er[1] ? 0 : 1;                             // every block is evaluated: the error of the smaller row
  else {
    q->err_checking = true;
    try {
      // in row order: the first error the iteration reaches is the one Julia throws
      const int first = er[0] <= er[1] ? 0 : 1;
      for (int i = 0; i < 2 && raise < 0; i++) { const int kind = i == 0 ? first : 1 - first; if (er[kind] != ~0ull && error_is_reached(q, er[kind])) raise = kind; }
      if (raise < 0) query_execute(q, nstages);                                // not reached: the result stands (the erroring rows are beyond the finish)
    } catch (...) { q->err_checking = false; throw; }
    q->err_checking = false;
  }
  q->err_row[0] = q->err_row[1] = ~0ull;
  // (the row travels with the error: the shards of a multi-GPU group raise the error of the lowest GLOBAL row, group.cpp)
  if (raise == 0) { q->executed_stages = -1; throw Error(DFDB_ERR_DIVIDE, "DivideError: integer division error", (uint64_t)q->t->row_base + er[0]); }
  if (raise == 1) { q->executed_stages = -1; throw Error(DFDB_ERR_ARGUMENT, "InexactError: conversion is not exact", (uint64_t)q->t->row_base + er[1]); }
}

void query_execute(dfdb_query* q, int nstages) {
  ensure_state(q);
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx;
  if (nstages < 0 || nstages > (int)q->stages.size()) nstages = (int)q->stages.size();
  q->count = -1; q->prefix_valid = false; q->executed_stages = -1; q->cap_col = -1; q->cap_col2 = -1; q->cap_str_col = -1; q->agg_col = -1; q->const_str_col = -1;
  q->decoded_col = -1;
  q->comp_scanned.clear();
  for (auto& a : q->arenas) a.second.valid = false;      // the survivors' blocks of the previous selection
  q->gr_state = 0;             // a pending groupreduce belongs to the selection that is being replaced: its fetch must not restore the old one over this
  q->err_row[0] = q->err_row[1] = ~0ull;
  // A range-like stage that is EMPTY (an empty range, an empty index vector) finishes the reference's iteration before the first block is read:
  // is_finished (selection.jl:192-196: `last <= offset` for ANY range stage of the queue) is tested ahead of every block (blocksiterator.jl:69-78).
  // Nothing is evaluated — a predicate of another stage that would raise DivideError on some row never runs (found by the fuzz soak: the engine raised).
  for (const Stage& st : q->stages) {
    const bool empty = (st.kind == ST_RANGE && st.n == 0) || ((st.kind == ST_INDICES || st.kind == ST_INTEGER) && st.idx.empty());
    if (!empty) continue;
    HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, padded_words(t->nrows) * 8, ctx->stream));
    HIP_CHECK(hipMemsetAsync(q->tile_counts.p, 0, (size_t)ceil_div(t->nrows, kTileRows) * 4, ctx->stream));
    scan_prefix(q);
    q->executed_stages = nstages;
    return;
  }
  if (nstages == 0) {
    LaunchTimer lt(ctx, "fill_ones");
    launch_fill_ones(ctx->stream, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), t->nrows);
  }
  for (int i = 0; i < nstages; i++) {
    const Stage& st = q->stages[(size_t)i];
    if (st.kind == ST_PRED) { run_predicate(q, *st.pred, i == 0, i + 1 == (int)q->stages.size()); q->prefix_valid = false; }
    else run_range(q, st, i == 0);
  }
  // A view that needs NO column at all — range-like stages only and a projection of constants — iterates nothing in the reference: BlocksIterator
  // opens one stream per required column and `isempty(it.streams)` ends the iteration before the first block (blocksiterator.jl:101,125).  Same here.
  if (nstages == (int)q->stages.size()) {
    std::vector<int> req;
    for (const Stage& st : q->stages) if (st.kind == ST_PRED) required_columns(*st.pred, req);
    for (const ProjCol& p : q->proj) required_columns(*p.expr, req);
    if (req.empty() && !q->proj.empty()) {
      HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, padded_words(t->nrows) * 8, ctx->stream));
      HIP_CHECK(hipMemsetAsync(q->tile_counts.p, 0, (size_t)ceil_div(t->nrows, kTileRows) * 4, ctx->stream));
    }
  }
  scan_prefix(q);
  q->executed_stages = nstages;
  if (!q->err_checking && (q->err_row[0] != ~0ull || q->err_row[1] != ~0ull)) raise_reached_errors(q, nstages);
}

static void ensure_executed(dfdb_query* q) {
  if (q->executed_stages != (int)q->stages.size() || q->bitmap_rows != q->t->nrows) query_execute(q, -1);
}

int64_t query_count(dfdb_query* q, int nstages);
// ensure_executed for a consumer that hands results to the HOST (or builds on them: unique, groupreduce, a new column): an execution that decoded a column's
// resident LZ4 blocks on its way (decode_on_scan) is only as good as that decode, and the blocks' statuses are read — and a failed decode repeated once
// without the index — where query_count waits for the count (only dfdb_count did so).  Consumers that stay asynchronous on the device
// (dfdb_select_indices / _bitmap into device memory without a count, the group layer's device aggregates) are covered by the caller's next dfdb_count or
// dfdb_table_decode_status, as include/dfdb.h says.
static void ensure_executed_checked(dfdb_query* q) {
  ensure_executed(q);
  if (q->decoded_col >= 0 || !q->comp_scanned.empty()) (void)query_count(q, -1);
}

int64_t query_count(dfdb_query* q, int nstages) {
  if (nstages < 0) { ensure_executed(q); if (q->count >= 0) return q->count; }
  else query_execute(q, nstages);
  dfdb_ctx* ctx = q->t->ctx;
  const int64_t ntiles = ceil_div(q->t->nrows, kTileRows);
  HIP_CHECK(hipMemcpyAsync(ctx->pinned_scalar, q->prefix.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
  stream_wait(ctx);
  int64_t n = ctx->pinned_scalar[0];
  // An execution that decoded a column's resident LZ4 blocks on its way (decode_on_scan) has results only as good as that decode: a block K7 gave
  // up on (a damaged copy, or err 9 — a sequence-start index that is not this stream's) leaves stale mask words, tile counts and column bytes
  // behind.  The statuses are read here, where the host waits for the count anyway.  A bad block drops the index
  // (table_decode_status); ONE more execution decodes by parsing and records a new one; bad again -> the blocks themselves are damaged.
  if (!q->comp_scanned.empty()) {
    // the same for the compressed-only columns this execution decoded inside its scan (several terms may have: each column's statuses are its last launch's)
    const std::vector<int> cols = q->comp_scanned;
    bool again = false;
    for (int col : cols) again = table_decode_status(q->t, col) > 0 || again;
    if (again) {
      query_execute(q, nstages < 0 ? -1 : nstages);
      for (int col : q->comp_scanned) {
        const int64_t bad = table_decode_status(q->t, col);
        if (bad > 0) { q->executed_stages = -1; fail(DFDB_ERR_FORMAT, "column %s: %lld of its resident LZ4 blocks do not decode", q->t->cols[(size_t)col].name.c_str(), (long long)bad); }
      }
      HIP_CHECK(hipMemcpyAsync(ctx->pinned_scalar, q->prefix.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
      stream_wait(ctx);
      n = ctx->pinned_scalar[0];
    }
  }
  if (q->decoded_col >= 0) {
    const int col = q->decoded_col;
    if (table_decode_status(q->t, col) > 0) {
      query_execute(q, nstages < 0 ? -1 : nstages);
      int64_t bad = q->decoded_col >= 0 ? table_decode_status(q->t, col) : 0;
      if (bad > 0) { q->executed_stages = -1; fail(DFDB_ERR_FORMAT, "column %s: %lld of its resident LZ4 blocks do not decode", q->t->cols[(size_t)col].name.c_str(), (long long)bad); }
      HIP_CHECK(hipMemcpyAsync(ctx->pinned_scalar, q->prefix.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
      stream_wait(ctx);
      n = ctx->pinned_scalar[0];
    }
  }
  if (nstages < 0) q->count = n;
  return n;
}

void query_select_bitmap(dfdb_query* q, uint64_t* out, int32_t memkind) {
  if (memkind == DFDB_MEM_DEVICE) ensure_executed(q); else ensure_executed_checked(q);
  hipStream_t s = q->t->ctx->stream;
  const size_t bytes = (size_t)ceil_div(q->t->nrows, 64) * 8;
  if (!bytes) return;
  HIP_CHECK(hipMemcpyAsync(out, q->bitmap.p, bytes, memkind == DFDB_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
  if (memkind != DFDB_MEM_DEVICE) stream_wait(q->t->ctx);
}

void query_select_indices(dfdb_query* q, int64_t* out, int64_t cap, int32_t memkind, int64_t* n) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  ensure_executed(q);
  const int store = (int)ctx_option(ctx, "compact_store", kCompactStoreDefault);   // per context: two contexts may run different variants side by side
  static const char* const store_names[] = {"compact_indices.plain8", "compact_indices.nt8", "compact_indices.wt8", "compact_indices.nt16", "compact_indices.plain16",
                                            "compact_indices.nt16_lds4k", "compact_indices.plain16_lds4k"};
  prof_note(ctx, store_names[store >= 0 && store <= 6 ? store : 0]);
  const int gcap = 0;
  if (memkind == DFDB_MEM_DEVICE) {
    { LaunchTimer lt(ctx, "compact_indices");
      launch_compact_indices(s, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), out, t->nrows, t->row_base, cap, store, gcap); }
    if (n) *n = query_count(q, -1);
    return;
  }
  const int64_t cnt = query_count(q, -1);
  if (n) *n = cnt;
  const int64_t m = std::min(cnt, cap);
  if (m <= 0) return;
  q->tmp_b.ensure((size_t)m * 8);
  { LaunchTimer lt(ctx, "compact_indices");
    launch_compact_indices(s, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), q->tmp_b.as<int64_t>(), t->nrows, t->row_base, m, store, gcap); }
  HIP_CHECK(hipMemcpyAsync(out, q->tmp_b.p, (size_t)m * 8, hipMemcpyDeviceToHost, s));
  stream_wait(q->t->ctx);
}

// ---------------------------------------------------------------- compressed-only projection columns
// The source pointer of a gather over a fixed-width column.  A compressed-only column (keep_compressed = 2) has no decoded array: the blocks of the current
// selection that KEPT A ROW are decoded into an arena this query owns — at their natural offsets inside the span [first such block, last such block], so the
// gather kernels address it like the column itself through a shifted base — and the others are never touched (blocksiterator.jl:111-113: a block with an
// empty selection skips its projection columns).  Synchronises (the survivors per block are read on the host, like the reference's loop reads them).
static const void* gather_source(dfdb_query* q, int ord) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  Column& c = t->cols[(size_t)ord];
  if (!c.comp_only || c.data.p) return c.data.p;
  const int w = dt_width(c.dtype);
  dfdb_query::Arena& a = q->arenas[ord];
  if (!a.valid || a.from != c.comp.p) {                  // (a column that was loaded or compressed again since: other blocks)
    a.from = c.comp.p;
    std::vector<int64_t> counts;
    query_block_counts(q, t->block_size, counts);
    int64_t first = -1, last = -1;
    std::vector<Lz4Block> sub;
    for (int64_t b = 0; b < (int64_t)counts.size() && b < c.comp_nblocks; b++) if (counts[(size_t)b] > 0) { if (first < 0) first = b; last = b; }
    a.first_row = first < 0 ? 0 : first * t->block_size; a.nblocks = 0;
    if (first >= 0) {
      const int64_t base_off = c.comp_blocks_host[(size_t)first].dst_off;
      for (int64_t b = first; b <= last; b++) if (counts[(size_t)b] > 0) { Lz4Block x = c.comp_blocks_host[(size_t)b]; x.dst_off -= base_off; sub.push_back(x); }
      const Lz4Block& lb = c.comp_blocks_host[(size_t)last];
      a.buf.ensure((size_t)(lb.dst_off - base_off) + (size_t)lb.dst_len + 256);
      a.blocks.ensure(sub.size() * sizeof(Lz4Block)); a.status.ensure(sub.size() * 4);
      HIP_CHECK(hipMemcpyAsync(a.blocks.p, sub.data(), sub.size() * sizeof(Lz4Block), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemsetAsync(a.status.p, 0, sub.size() * 4, s));
      const int pipe = (int)ctx_option(ctx, "lz4_pipeline", -1);
      const int mode = column_lz4_index(ctx, c, lz4_decode_takes_index((int32_t)sub.size(), pipe));
      { LaunchTimer lt(ctx, "lz4_decode");
        prof_note(ctx, "lz4_decode.survivors");
        launch_lz4_decode(s, c.comp.as<uint8_t>(), a.buf.as<uint8_t>(), a.blocks.as<Lz4Block>(), (int32_t)sub.size(), a.status.as<int32_t>(), pipe, c.comp_index.as<uint32_t>(),
                          mode == 1 ? 0 : mode); }      // (a subset of the blocks cannot RECORD the column's index)
      if (mode == 1) c.comp_index_state = 0;
      std::vector<int32_t> st(sub.size());
      HIP_CHECK(hipMemcpyAsync(st.data(), a.status.p, st.size() * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));            // (also: `sub` is pageable host memory)
      int64_t bad = 0; for (int32_t v : st) bad += v != 0;
      if (bad && mode == 2) {                        // the index may be what is damaged: once more by parsing
        HIP_CHECK(hipMemsetAsync(a.status.p, 0, sub.size() * 4, s));
        launch_lz4_decode(s, c.comp.as<uint8_t>(), a.buf.as<uint8_t>(), a.blocks.as<Lz4Block>(), (int32_t)sub.size(), a.status.as<int32_t>(), pipe, nullptr, 0);
        HIP_CHECK(hipMemcpyAsync(st.data(), a.status.p, st.size() * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        bad = 0; for (int32_t v : st) bad += v != 0;
        if (!bad) { c.comp_index.release(); c.comp_index_state = 0; }
      }
      if (bad) fail(DFDB_ERR_FORMAT, "column %s: %lld of its resident LZ4 blocks do not decode", c.name.c_str(), (long long)bad);
      a.nblocks = (int64_t)sub.size();
    } else a.buf.ensure(256);
    a.valid = true;
  }
  // row r of the column lives at arena byte (r - first_row) * w: hand the kernels the base that makes `src + r * w` land there (never dereferenced below first_row)
  return (const void*)(a.buf.as<uint8_t>() - (intptr_t)a.first_row * w);
}

// ---------------------------------------------------------------- materialize
static bool string_captured(const dfdb_query* q, const Column& col) {
  return q->cap_str_col >= 0 && &q->t->cols[(size_t)q->cap_str_col] == &col && q->executed_stages == (int)q->stages.size();
}

// selected string bytes per 1024-row tile -> exclusive scan (output arena offsets); returns total
static int64_t string_out_offsets(dfdb_query* q, const Column& col, DevBuf& out_sizes_tmp, int32_t* out_sizes, int64_t cap, DevBuf& tile_off_out) {
  dfdb_ctx* ctx = q->t->ctx; hipStream_t s = ctx->stream;
  const int64_t nct = ceil_div(q->t->nrows, kTileRows);
  DevBuf& tb = q->tmp_c; tb.ensure((size_t)(nct + 8) * 4);
  tile_off_out.ensure((size_t)(nct + 8) * 8);
  DevBuf& scratch = q->str_scratch; scratch.ensure(scan_counts_scratch_bytes(nct));
  int32_t* dst_sizes = out_sizes;
  if (!dst_sizes) { out_sizes_tmp.ensure((size_t)std::max<int64_t>(cap, 1) * 4); dst_sizes = out_sizes_tmp.as<int32_t>(); }
  if (col.dict_n > 0) {
    // K9: the selected rows' codes, compacted by K3 (kept in q->dict_sel for the bytes pass), then sizes and byte totals per 1024 OUTPUT rows
    const int64_t n = std::min<int64_t>(cap, query_count(q, -1));
    const int64_t not_ = ceil_div(std::max<int64_t>(n, 1), kTileRows);
    q->dict_sel.ensure((size_t)std::max<int64_t>(n, 1) * 2 + 256);
    { LaunchTimer lt(ctx, "gather"); launch_gather(s, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), col.dict_codes.p, q->dict_sel.p, 2, q->t->nrows, n); }
    DevBuf& otb = q->tmp_c; otb.ensure((size_t)(not_ + 8) * 4);
    tile_off_out.ensure((size_t)(not_ + 8) * 8);
    scratch.ensure(scan_counts_scratch_bytes(not_));
    HIP_CHECK(hipMemsetAsync(otb.p, 0, (size_t)(not_ + 8) * 4, s));
    { LaunchTimer lt(ctx, "dict_expand_sizes"); launch_dict_expand_sizes(s, q->dict_sel.as<uint16_t>(), n, col.dict_len.as<int32_t>(), dst_sizes, otb.as<uint32_t>()); }
    launch_scan_counts(s, otb.as<uint32_t>(), tile_off_out.as<uint64_t>(), not_, scratch.as<uint64_t>());
    HIP_CHECK(hipMemcpyAsync(ctx->pinned_scalar + 1, tile_off_out.as<uint64_t>() + not_, 8, hipMemcpyDeviceToHost, s));
    stream_wait(q->t->ctx);
    return ctx->pinned_scalar[1];
  }
  if (string_captured(q, col)) {     // K5 kept the selected rows: their byte totals per tile are already there
    launch_scan_counts(s, q->cap_str_tb.as<uint32_t>(), tile_off_out.as<uint64_t>(), nct, scratch.as<uint64_t>());
    HIP_CHECK(hipMemcpyAsync(ctx->pinned_scalar + 1, tile_off_out.as<uint64_t>() + nct, 8, hipMemcpyDeviceToHost, s));
    stream_wait(q->t->ctx);
    return ctx->pinned_scalar[1];
  }
  { LaunchTimer lt(ctx, "str_gather_sizes");
    launch_str_gather_sizes(s, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), col.data.as<int32_t>(), dst_sizes, tb.as<uint32_t>(), q->t->nrows, cap); }
  launch_scan_counts(s, tb.as<uint32_t>(), tile_off_out.as<uint64_t>(), nct, scratch.as<uint64_t>());
  HIP_CHECK(hipMemcpyAsync(ctx->pinned_scalar + 1, tile_off_out.as<uint64_t>() + nct, 8, hipMemcpyDeviceToHost, s));
  stream_wait(q->t->ctx);
  return ctx->pinned_scalar[1];
}

int64_t query_string_bytes(dfdb_query* q, int i) {
  ensure_executed_checked(q);
  if (i < 0 || (size_t)i >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", i);
  const Node& e = *q->proj[(size_t)i].expr;
  if (dt_base(e.dtype) != DFDB_STRING) return 0;
  if (e.op != DFIR_COL) fail(DFDB_ERR_UNSUPPORTED, "computed String columns are outside the IR");
  const Column& col = need_resident(q->t, e.col);
  const int64_t cnt = query_count(q, -1);
  if (q->const_str_col == e.col && q->executed_stages == (int)q->stages.size()) return cnt * (int64_t)q->const_str.size();
  return string_out_offsets(q, col, q->str_sizes, nullptr, cnt, q->str_toff);
}

// one output column of the projection (ProjectionExecutor.eval_on_range for column p: projection.jl:128-154)
static void materialize_col(dfdb_query* q, int32_t p, dfdb_outcol& o, int64_t cnt) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const Node& e = *q->proj[(size_t)p].expr;
  o.dtype = e.dtype; o.count = cnt; o.nbytes = 0;
  const bool dev = o.memkind == DFDB_MEM_DEVICE;
  const int w = dt_width(e.dtype);
  if (cnt == 0) return;
  if (!o.data) fail(DFDB_ERR_ARGUMENT, "output column %d has no data buffer", p);
  if (e.op == DFIR_COL) {   // ColProjExec: buffer .= data[name][range] (projection.jl:130-133)
    const void* gsrc = nullptr;
    if (t->cols[(size_t)e.col].comp_only && !dt_nullable(e.dtype) && dt_base(e.dtype) != DFDB_STRING) {
      if (!t->cols[(size_t)e.col].resident) fail(DFDB_ERR_ARGUMENT, "column %s is not resident on the device (dfdb_table_load it first)", t->cols[(size_t)e.col].name.c_str());
      gsrc = gather_source(q, e.col);        // compressed-only: the blocks with survivors, decoded for this query (no whole-column decode)
    }
    const Column& col = gsrc ? t->cols[(size_t)e.col] : need_resident(t, e.col);
    if (!gsrc) gsrc = col.data.p;
    if (dt_base(e.dtype) == DFDB_STRING) {   // FlatStringsVector gather (FlatStringsVectors.jl:136-157)
      if (q->const_str_col == e.col && q->executed_stages == (int)q->stages.size()) {   // every selected row holds q->const_str (run_predicate)
        const int64_t plen = (int64_t)q->const_str.size(), total = cnt * plen;
        o.nbytes = total;
        if (total > o.bytes_cap) fail(DFDB_ERR_ARGUMENT, "output column %d needs %lld string bytes, capacity is %lld", p, (long long)total, (long long)o.bytes_cap);
        DevBuf &csz = q->str_sizes, &cby = q->str_bytes;
        int32_t* d_sizes = dev ? (int32_t*)o.data : (csz.ensure((size_t)cnt * 4), csz.as<int32_t>());
        uint8_t* d_bytes = dev ? o.bytes : (cby.ensure((size_t)total + 64), cby.as<uint8_t>());
        DevBuf& pb = q->tmp_a; pb.ensure((size_t)plen + 64);
        if (plen) { HIP_CHECK(hipMemcpyAsync(pb.p, q->const_str.data(), (size_t)plen, hipMemcpyHostToDevice, s)); stream_wait(ctx); }
        { LaunchTimer lt(ctx, "fill_const_strings"); launch_fill_const_strings(s, d_sizes, d_bytes, cnt, pb.as<uint8_t>(), (int32_t)plen); }
        if (!dev) {
          HIP_CHECK(hipMemcpyAsync(o.data, d_sizes, (size_t)cnt * 4, hipMemcpyDeviceToHost, s));
          if (total > 0) HIP_CHECK(hipMemcpyAsync(o.bytes, d_bytes, (size_t)total, hipMemcpyDeviceToHost, s));
          stream_wait(ctx);
        }
        return;
      }
      DevBuf &dsz = q->str_sizes, &toff = q->str_toff, &dbytes = q->str_bytes;   // reused across calls (hipFree would sync the device)
      int32_t* d_sizes = dev ? (int32_t*)o.data : nullptr;
      const int64_t total = string_out_offsets(q, col, dsz, d_sizes, cnt, toff);
      if (!d_sizes) d_sizes = dsz.as<int32_t>();
      o.nbytes = total;
      if (total > o.bytes_cap) fail(DFDB_ERR_ARGUMENT, "output column %d needs %lld string bytes, capacity is %lld", p, (long long)total, (long long)o.bytes_cap);
      uint8_t* d_bytes = dev ? o.bytes : nullptr;
      if (!dev) { dbytes.ensure((size_t)total + 64); d_bytes = dbytes.as<uint8_t>(); }
      if (col.dict_n > 0) {              // K9: the compacted codes of string_out_offsets -> bytes out of the dictionary
        if (total > 0) {
          LaunchTimer lt(ctx, "dict_expand_bytes");
          launch_dict_expand_bytes(s, q->dict_sel.as<uint16_t>(), cnt, col.dict_len.as<int32_t>(), col.dict_off.as<uint32_t>(), col.dict_bytes.as<uint8_t>(), toff.as<uint64_t>(), d_bytes, total);
        }
      } else if (string_captured(q, col)) {     // sizes and bytes: one contiguous copy per tile out of the match pass's capture
        LaunchTimer lt(ctx, "str_compact_captured");
        const StrCapture sc{q->cap_str_sizes.as<int32_t>(), q->cap_str_bytes.as<uint8_t>(), q->cap_str_tb.as<uint32_t>()};
        launch_str_compact_captured(s, sc, q->prefix.as<uint64_t>(), (const int64_t*)col.tile_off.p, toff.as<uint64_t>(), d_sizes, d_bytes, t->nrows, cnt, total);
      } else if (total > 0) {
        LaunchTimer lt(ctx, "str_gather_bytes");
        launch_str_gather_bytes(s, q->bitmap.as<uint64_t>(), col.data.as<int32_t>(), (const int64_t*)col.tile_off.p, col.bytes.as<uint8_t>(),
                                toff.as<uint64_t>(), d_bytes, t->nrows, total);
      }
      if (!dev) {
        HIP_CHECK(hipMemcpyAsync(o.data, d_sizes, (size_t)cnt * 4, hipMemcpyDeviceToHost, s));
        if (total > 0) HIP_CHECK(hipMemcpyAsync(o.bytes, d_bytes, (size_t)total, hipMemcpyDeviceToHost, s));
        stream_wait(ctx);
      }
      return;
    }
    DevBuf stage; void* dst = o.data;
    if (!dev) { stage.ensure((size_t)cnt * w); dst = stage.p; }
    if ((q->cap_col == e.col || q->cap_col2 == e.col) && w == 8 && q->executed_stages == (int)q->stages.size()) {   // the scan kept these values: contiguous copy per tile
      LaunchTimer lt(ctx, "compact_captured");
      launch_compact_captured(s, (q->cap_col == e.col ? q->cap_buf : q->cap_buf2).as<uint64_t>(), q->prefix.as<uint64_t>(), (uint64_t*)dst, t->nrows, cnt);
    } else {
      LaunchTimer lt(ctx, "gather");
      launch_gather(s, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), gsrc, dst, w, t->nrows, cnt);
    }
    if (!dev) HIP_CHECK(hipMemcpyAsync(o.data, dst, (size_t)cnt * w, hipMemcpyDeviceToHost, s));
    if (dt_nullable(e.dtype) && o.missing) {
      DevBuf ms; uint8_t* md = o.missing;
      if (!dev) { ms.ensure((size_t)cnt); md = ms.as<uint8_t>(); }
      launch_gather_bits(s, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), col.missing.as<uint64_t>(), md, t->nrows, cnt);
      if (!dev) { HIP_CHECK(hipMemcpyAsync(o.missing, md, (size_t)cnt, hipMemcpyDeviceToHost, s)); stream_wait(ctx); }
    }
    if (!dev) stream_wait(ctx);   // staging buffers die at scope exit; device outputs stay stream-ordered, no host wait
  } else {                  // BroadcastExecutor: computed column (projection.jl:128-129)
    if (dt_base(e.dtype) == DFDB_STRING) fail(DFDB_ERR_UNSUPPORTED, "computed String columns are outside the IR");
    DevBuf stage, mstage; void* dst = o.data;
    if (!dev) { stage.ensure((size_t)cnt * w); dst = stage.p; }
    uint8_t* mdst = nullptr;                               // Union{R,Missing} result: one flag byte per selected row
    if (dt_nullable(e.dtype) && o.missing) { mdst = o.missing; if (!dev) { mstage.ensure((size_t)cnt); mdst = mstage.as<uint8_t>(); } }
    // a transform of ONE plain column (rem / col * k + d / col / k: expr.cpp match_column_transform) rides on the gather of that column
    ScanTerm tf; const Node* tcol = nullptr;
    if (w == 8 && !dt_nullable(e.dtype) && match_column_transform(&e, tf, tcol) && tf.pre != 0 && !dt_nullable(tcol->dtype) && t->cols[(size_t)tcol->col].resident) {
      const Column& sc = t->cols[(size_t)tcol->col];
      if ((q->cap_col == tcol->col || q->cap_col2 == tcol->col) && dt_width(sc.dtype) == 8 && q->executed_stages == (int)q->stages.size()) {
        LaunchTimer lt(ctx, "compact_captured");                  // the scan kept the column's selected values: the transform rides on the copy
        launch_compact_captured_transform(s, (q->cap_col == tcol->col ? q->cap_buf : q->cap_buf2).as<uint64_t>(), q->prefix.as<uint64_t>(), dt_base(sc.dtype), tf, (uint64_t*)dst, t->nrows, cnt);
      } else {
        LaunchTimer lt(ctx, "gather");
        launch_gather_transform(s, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), gather_source(q, tcol->col), dt_base(sc.dtype), tf, dst, t->nrows, cnt);
      }
    } else
    run_interp_project(q, e, dst, cnt, mdst);
    if (!dev) {
      HIP_CHECK(hipMemcpyAsync(o.data, dst, (size_t)cnt * w, hipMemcpyDeviceToHost, s));
      if (mdst) HIP_CHECK(hipMemcpyAsync(o.missing, mdst, (size_t)cnt, hipMemcpyDeviceToHost, s));
      stream_wait(ctx);
    }
  }
}

void query_materialize(dfdb_query* q, dfdb_outcol* outs, int32_t ncols) {
  ensure_executed(q);
  if (ncols != (int32_t)q->proj.size()) fail(DFDB_ERR_ARGUMENT, "ArgumentError: view has %zu columns, %d outputs given", q->proj.size(), ncols);
  const int64_t cnt = query_count(q, -1);
  // a computed column that raises (DivideError / InexactError on a selected row): the reference evaluates block by block and, inside a block, the projection's
  // columns in order (projection.jl:149-154 under blocksiterator.jl:98-121) — the error it throws is the one of the first BLOCK that holds an erroring row, the
  // first such COLUMN in that block, the first such row of that column.  The columns are computed whole here, so their first erroring rows are collected and
  // the choice is made at the end.
  uint64_t pe[2], best_block = ~0ull; int best_kind = -1;
  const uint64_t bs = (uint64_t)std::max<int64_t>(q->t->block_size, 1);
  q->proj_err = pe;
  try {
    for (int32_t p = 0; p < ncols; p++) {
      pe[0] = pe[1] = ~0ull;
      materialize_col(q, p, outs[p], cnt);
      const uint64_t r = std::min(pe[0], pe[1]);
      if (r != ~0ull && r / bs < best_block) { best_block = r / bs; best_kind = pe[0] <= pe[1] ? 0 : 1; }
    }
  } catch (...) { q->proj_err = nullptr; throw; }
  q->proj_err = nullptr;
  if (best_kind == 0) fail(DFDB_ERR_DIVIDE, "DivideError: integer division error");
  if (best_kind == 1) fail(DFDB_ERR_ARGUMENT, "InexactError: conversion is not exact");
}

// add_column!(table, name, lazy_col) (src/tables/table.jl:96-124): the p-th column of the view materialised into a new
// RESIDENT column of dst without leaving the device.  dst may be the view's own table (then every row must be selected).
void launch_pack_flags(hipStream_t s, const uint8_t* flags, uint64_t* bits, int64_t n);
void table_add_from_query(dfdb_table* dst, const char* name, dfdb_query* q, int32_t p) {
  const bool ooc = query_out_of_core(q);      // the view's columns are not resident: the new column is made from the block stream (csrc/ooc.cpp), chunk by chunk
  if (!ooc) ensure_executed_checked(q);
  if (p < 0 || (size_t)p >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", p);
  if (dst->ctx != q->t->ctx) fail(DFDB_ERR_ARGUMENT, "the destination table lives on another context");
  for (auto& c : dst->cols) if (c.name == name) fail(DFDB_ERR_ARGUMENT, "ArgumentError: Duplicated column %s", name);
  const Node& e = *q->proj[(size_t)p].expr;
  const int64_t cnt = ooc ? ooc_count(q) : query_count(q, -1);
  if (dst->nrows >= 0 && dst->nrows != cnt)
    fail(DFDB_ERR_ARGUMENT, "ArgumentError: column has %lld rows but the table has %lld", (long long)cnt, (long long)dst->nrows);
  dfdb_ctx* ctx = dst->ctx; hipStream_t s = ctx->stream;
  Column c; c.name = name; c.dtype = e.dtype; c.id = 1; c.nrows = cnt;
  if (e.op == DFIR_COL) c.logical = q->t->cols[(size_t)e.col].logical;   // a projected Date / DateTime / Char column keeps its type
  for (auto& o : dst->cols) c.id = std::max(c.id, o.id + 1);
  dfdb_outcol o{}; o.memkind = DFDB_MEM_DEVICE;
  DevBuf flags;
  if (dt_base(e.dtype) == DFDB_STRING) {
    const int64_t total = ooc ? ooc_string_bytes(q, p) : query_string_bytes(q, p);
    c.data.ensure((size_t)cnt * 4 + 256);
    c.nbytes = total; c.bytes.ensure((size_t)total + 64);
    HIP_CHECK(hipMemsetAsync((char*)c.bytes.p + total, 0, 64, s));
    o.data = c.data.p; o.bytes = c.bytes.as<uint8_t>(); o.bytes_cap = total;
  } else {
    c.data.ensure((size_t)cnt * dt_width(e.dtype) + 256);
    o.data = c.data.p;
    if (dt_nullable(e.dtype)) { flags.ensure((size_t)cnt + 64); HIP_CHECK(hipMemsetAsync(flags.p, 0, (size_t)cnt + 64, s)); o.missing = flags.as<uint8_t>(); }
  }
  if (ooc) ooc_materialize_column(q, p, &o); else materialize_col(q, p, o, cnt);
  if (dt_base(e.dtype) == DFDB_STRING) set_string_tile_offsets(ctx, c);
  else if (dt_nullable(e.dtype)) {
    const size_t nw = (size_t)(round_up(cnt > 0 ? cnt : 1, kCTileRows) / 64 + 64);
    c.missing.ensure(nw * 8);
    HIP_CHECK(hipMemsetAsync(c.missing.p, 0, nw * 8, s));
    if (cnt) launch_pack_flags(s, flags.as<uint8_t>(), c.missing.as<uint64_t>(), cnt);
  }
  HIP_CHECK(hipStreamSynchronize(s));
  c.resident = true;
  if (dst->nrows < 0) dst->nrows = cnt;
  dst->cols.push_back(std::move(c));
}

// unique(col) (column.jl:102-126 driving Base.unique; docs/src/index.md:171-182,479-486): the current selection is narrowed to
// the rows that hold the FIRST occurrence of their value in projection column p (isequal semantics), so count / materialize
// afterwards return the distinct values in order of first appearance.  A later reset / execute restores the full selection.
void launch_dict_first_rows(hipStream_t s, const uint64_t* sel, const uint16_t* codes, int64_t nrows, uint64_t* first, int dict_n, int64_t tile0, int64_t tile1);
void launch_set_rows(hipStream_t s, const uint64_t* rows, int n, uint64_t* bitmap, uint32_t* tile_counts);
void launch_group_accumulate_codes(hipStream_t s, const uint64_t* sel, const uint16_t* codes, const uint32_t* rank_of_code, const void* valcol, int valdt, int op,
                                   int64_t nrows, uint64_t* cnt, uint64_t* val, int64_t ngroups, uint64_t val_init);
constexpr int64_t kGroupsInLds = 9216;      // groups a workgroup's LDS accumulators hold (k_unique.hip kGroupLdsBig): above it groupreduce goes by radix
// what unique leaves behind for groupreduce: the hash table {key, first row} (Strings: + where one holder's bytes are), or — integer keys of a small
// range — the first row per value; groupreduce turns the rows into group numbers in place
struct UniqueTables {
  DevBuf ent, aux, rep_off, rep_len, first, present;
  uint64_t cap = 0, salt = 0, lo = 0; uint32_t range = 0; bool is_str = false, dense = false;
  uint64_t span_lo = ~0ull, span_hi = 0;   // dense form: the first / last value index that is present (the table is laid out for the widest span it can hold)
  // in: the dense form may look at the HEAD of the column only (groupreduce over integer keys, round 5: its accumulate pass meets every row anyway and reports a key
  // that has no group — one outside the span laid out, or one that first turns up behind the head —, in which case everything runs again over all rows); out: it did
  bool allow_head = false, head_only = false;
  bool defer_verify = false;     // in: the caller's own pass over the rows compares every String with its slot's representative (groupreduce's accumulate pass)
  int salt_skip = 0;             // in: salts already found colliding
  // in: String keys with defer_verify — if the second insert chunk (16 M rows) met no string the first (1 M rows) had not, the rest of the rows are NOT
  // inserted: the caller's pass meets every row anyway and reports a string that is not in the table (`optimistic` comes back true; the caller then
  // runs everything again with allow_optimistic = false).  Ten brands over 5e8 rows: the insert pass 2.2 -> 0.1 ms.
  bool allow_optimistic = false, optimistic = false;
  // in: the caller can reduce by radix (group_radix: more groups than LDS accumulators hold) and wants to know EARLY whether that is the case; out: the number of
  // distinct keys the first chunk's rows promise when it is more than 9216 — and then nothing else was done: the selection is as it was, no table was finished
  bool group_probe = false; int64_t group_estimate = 0;
};
// K9: unique over a String column that has a dictionary — the first selected row of every code, no hash table.  Leaves what unique_impl leaves (the
// bitmap holds exactly the first occurrences, prefix scanned); rank_of_code (optional) maps a code to its group number in order of first appearance.
static int64_t dict_unique(dfdb_query* q, const Column& col, DevBuf* rank_of_code) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const int dn = col.dict_n;
  DevBuf& first = q->du_first; first.ensure((size_t)dn * 8 + 64);      // (scratch kept on the query: every hipFree synchronises the device)
  HIP_CHECK(hipMemsetAsync(first.p, 0xFF, (size_t)dn * 8, s));
  // the first 4 M rows first: a code that has a first row there can only keep it, and a dictionary's codes have usually all turned up by then (ten brands over
  // 5e8 rows: the pass over all the codes was 0.65 ms of the groupreduce's 2.8) — only if one is still without does the rest of the column get walked
  const int64_t nt = ceil_div(t->nrows, kTileRows), head = 4096;
  std::vector<uint64_t> fr((size_t)dn);
  auto walk = [&](int64_t t0, int64_t t1) {
    { LaunchTimer lt(ctx, "unique");
      launch_dict_first_rows(s, q->bitmap.as<uint64_t>(), col.dict_codes.as<uint16_t>(), t->nrows, first.as<uint64_t>(), dn, t0, t1); }
    HIP_CHECK(hipMemcpyAsync(fr.data(), first.p, (size_t)dn * 8, hipMemcpyDeviceToHost, s));
    stream_wait(ctx);
  };
  if (nt > 4 * head) {
    walk(0, head);
    bool all = true;
    for (int k = 0; k < dn; k++) all = all && fr[(size_t)k] != ~0ull;
    if (!all) walk(head, nt);
  } else walk(0, nt);
  std::vector<std::pair<uint64_t, uint32_t>> present;
  for (int k = 0; k < dn; k++) if (fr[(size_t)k] != ~0ull) present.emplace_back(fr[(size_t)k], (uint32_t)k);
  std::sort(present.begin(), present.end());
  const int64_t ng = (int64_t)present.size();
  std::vector<uint64_t> rows((size_t)std::max<int64_t>(ng, 1));
  std::vector<uint32_t> rank((size_t)dn, 0xffffffffu);
  for (int64_t g = 0; g < ng; g++) { rows[(size_t)g] = present[(size_t)g].first; rank[present[(size_t)g].second] = (uint32_t)g; }
  DevBuf& drows = q->du_rows; drows.ensure(rows.size() * 8 + 64);
  HIP_CHECK(hipMemcpyAsync(drows.p, rows.data(), rows.size() * 8, hipMemcpyHostToDevice, s));
  if (rank_of_code) { rank_of_code->ensure(rank.size() * 4 + 64); HIP_CHECK(hipMemcpyAsync(rank_of_code->p, rank.data(), rank.size() * 4, hipMemcpyHostToDevice, s)); }
  HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, padded_words(t->nrows) * 8, s));
  HIP_CHECK(hipMemsetAsync(q->tile_counts.p, 0, (size_t)ceil_div(t->nrows, kTileRows) * 4, s));
  launch_set_rows(s, drows.as<uint64_t>(), (int)ng, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>());
  scan_prefix(q);
  selection_changed(q); q->cap_col = -1; q->cap_col2 = -1; q->cap_str_col = -1; q->agg_col = -1; q->const_str_col = -1;
  stream_wait(ctx);                                        // rows / rank are pageable host memory
  return ng;
}
constexpr size_t kUniqueAuxBytes = 128;      // k_unique.hip: 0 first row of the key that cannot be stored, 1 first missing row, 2 claimed slots, 3 abort, 4 collision, 5-10 the dense form's (10: found-before snapshot)
static void unique_reset_aux(dfdb_ctx* ctx, DevBuf& aux) {
  hipStream_t s = ctx->stream;
  aux.ensure(kUniqueAuxBytes);
  HIP_CHECK(hipMemsetAsync(aux.p, 0, kUniqueAuxBytes, s));
  HIP_CHECK(hipMemsetAsync(aux.p, 0xFF, 16, s));                       // no such row yet
  HIP_CHECK(hipMemsetAsync((char*)aux.p + 64, 0xFF, 8, s));            // the running minimum (k_dense_minmax)
  HIP_CHECK(hipMemsetAsync((char*)aux.p + 88, 0xFF, 8, s));            // the smallest present value index (k_dense_count)
}
static uint64_t pow2_at_least(uint64_t n) { uint64_t c = 1024; while (c < n) c <<= 1; return c; }

// How many slots the table should have before the next chunk: `d` distinct keys met in the first `r` selected rows, `remaining` rows to come.
// If the rows so far were nearly all different nothing can be said (every remaining row may bring a key).  Otherwise the distinct count D of the whole
// selection is estimated as if keys were drawn uniformly — d = D (1 - exp(-r / D)), solved for D — and doubled: skewed data has more rare keys than that,
// and an estimate that turns out too small costs one more migration (or an aborted chunk), never a wrong result.
static uint64_t unique_capacity_wanted(uint64_t d, uint64_t r, uint64_t remaining, uint64_t cap, uint64_t capmax) {
  if (remaining == 0 || r == 0) return cap;
  double need;
  if ((double)d > 0.95 * (double)r) need = (double)d + (double)remaining;
  else {
    double lo = (double)(d ? d : 1), hi = 20.0 * (double)r;
    for (int it = 0; it < 64; it++) { const double D = 0.5 * (lo + hi); if (D * (1.0 - std::exp(-(double)r / D)) < (double)d) lo = D; else hi = D; }
    need = std::min((double)d + (double)remaining, 2.0 * hi + 1024.0);
  }
  const uint64_t want = pow2_at_least((uint64_t)(2.0 * need));
  return std::min(std::max(want, cap), capmax);
}

// integer keys whose selected values span less than the dense form's range (k_unique.hip, last section).  false: the span is too wide
static bool unique_dense(dfdb_query* q, const Column& col, UniqueTables& T) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const int dt = dt_base(col.dtype);
  const int64_t limit = std::min<int64_t>(unique_dense_max_range(), ctx_option(ctx, "unique_dense_range", unique_dense_max_range()));
  const uint64_t* miss = dt_nullable(col.dtype) ? col.missing.as<uint64_t>() : nullptr;
  const uint64_t flip = dt_issigned(dt) ? (1ull << 63) : 0ull;
  uint64_t* aux = nullptr;
  uint64_t distinct = 0;
  const int64_t nt_all = ceil_div(t->nrows, kTileRows), head_tiles = std::max<int64_t>(1, ctx_option(ctx, "dense_head_tiles", 4096));
  T.head_only = T.allow_head && nt_all >= 8 * head_tiles;               // (the sample below still looks at the whole column: the span is laid around what IT saw)
  auto prows = [&]() -> int64_t { return T.head_only ? head_tiles * kTileRows : t->nrows; };      // the rows the presence pass and the first-row launches walk
  auto minmax = [&](int64_t tile_step, uint64_t* mm) {                 // order-preserving images of the smallest / largest selected key (of every tile_step-th tile)
    unique_reset_aux(ctx, T.aux);
    { LaunchTimer lt(ctx, "unique_minmax"); launch_dense_minmax(s, q->bitmap.as<uint64_t>(), col.data.p, dt, miss, t->nrows, tile_step, T.aux.as<uint64_t>()); }
    HIP_CHECK(hipMemcpyAsync(mm, (char*)T.aux.p + 64, 16, hipMemcpyDeviceToHost, s));
    stream_wait(ctx);
  };
  auto presence = [&](uint64_t lo_im, uint64_t span) -> bool {         // false: a selected key lies outside [lo, lo + span]
    T.lo = lo_im ^ flip; T.range = (uint32_t)(span + 1);
    const size_t words = ((size_t)T.range + 31) / 32;
    T.first.ensure((size_t)T.range * 8 + 64); T.present.ensure(words * 4 + 64);
    HIP_CHECK(hipMemsetAsync(T.first.p, 0xFF, (size_t)T.range * 8, s));
    HIP_CHECK(hipMemsetAsync(T.present.p, 0, words * 4, s));
    unique_reset_aux(ctx, T.aux);
    aux = T.aux.as<uint64_t>();
    { LaunchTimer lt(ctx, "unique_presence");
      launch_dense_presence(s, q->bitmap.as<uint64_t>(), col.data.p, dt, miss, prows(), T.lo, T.range, T.present.as<uint32_t>(), aux); }
    uint64_t od[8] = {0, 0, 0, 0, 0, 0, 0, 0};                          // [0] a key outside?  [1] distinct values ... [6], [7] the span of the present values
    HIP_CHECK(hipMemcpyAsync(od, (char*)T.aux.p + 40, 64, hipMemcpyDeviceToHost, s));
    stream_wait(ctx);
    distinct = od[1];
    T.span_lo = od[6]; T.span_hi = od[7];
    return od[0] == 0;
  };
  const int64_t nt = ceil_div(t->nrows, kTileRows);
  bool placed = false;
  if (dt_width(dt) <= 2) {                                             // the type's own range fits
    const int bits = dt_width(dt) * 8;
    if ((1ull << bits) > (uint64_t)limit) return false;
    placed = presence(dt_issigned(dt) ? ((uint64_t)(-(int64_t)(1ull << (bits - 1))) ^ flip) : 0ull, (1ull << bits) - 1);
  } else {
    // where the keys lie is not known.  A SAMPLE (every nt/256-th tile) says where to expect them: the whole span the form can hold is laid around the sample's
    // range and the presence pass reports any key outside it — then, and for a small table, the exact range costs a pass of its own (1.3 ms per 1e9 rows)
    uint64_t mm[2];
    const int64_t sample_opt = ctx_option(ctx, "unique_dense_sample", 1);
    const int64_t sample_step = sample_opt == 2 ? std::max<int64_t>(2, nt / 256) : nt / 256;      // (2: a sample even of a small table, every other tile — tests of what follows a sample)
    if ((sample_step >= 16 || sample_opt == 2) && sample_opt != 0) {
      minmax(sample_step, mm);
      if (mm[0] <= mm[1]) {
        if (mm[1] - mm[0] >= (uint64_t)limit) return false;
        const uint64_t slack = ((uint64_t)limit - 1 - (mm[1] - mm[0])) / 2;
        uint64_t lo_im = mm[0] > slack ? mm[0] - slack : 0ull;
        if (lo_im > ~0ull - ((uint64_t)limit - 1)) lo_im = ~0ull - ((uint64_t)limit - 1);
        placed = presence(lo_im, (uint64_t)limit - 1);
      }
    }
    if (!placed) {
      T.head_only = false;                                              // (no sample, or a key outside what it suggested: the exact range, every row)
      minmax(1, mm);
      if (mm[0] > mm[1]) mm[0] = mm[1] = flip;                         // every selected key is missing
      if (mm[1] - mm[0] >= (uint64_t)limit) return false;
      placed = presence(mm[0], mm[1] - mm[0]);
    }
  }
  if (!placed) fail(DFDB_ERR_DEVICE, "unique: a key outside the range of its column");
  T.dense = true;
  { LaunchTimer lt(ctx, "unique_first");
    // launches of 1 M, 1 M, 2 M, 4 M rows, then four times the last: every row whose value has no first row yet costs an atomic, and rows that run side by side
    // cannot see each other's — 1e6 values spread over the column are all met within 20 M rows, and met in small steps they cost 1.5 M atomics instead of 4 M
    const bool sched_default = ctx_option(ctx, "unique_chunk_tiles", 0) <= 0;
    int64_t step = sched_default ? 1024 : ctx_option(ctx, "unique_chunk_tiles", 0);
    int launches = 0;
    const int64_t nt_walk = T.head_only ? head_tiles : nt;
    int64_t t0 = 0;
    auto walk = [&](int64_t a, int64_t b) {
      launch_dense_first(s, q->bitmap.as<uint64_t>(), col.data.p, dt, miss, a * kTileRows, std::min(prows(), b * kTileRows), T.lo, T.range, distinct, T.first.as<uint64_t>(), aux);
    };
    // FEW values: the first 1 M rows in three steps of 16, 112 and 896 tiles — rows that run side by side all see "no first row yet" and all send their atomicMin,
    // and with seven values those are a million atomics on seven words (2.1 ms: `tools/r5_fewgroups.py`); after sixteen tiles every value has its row and the
    // launches that follow return at once
    if (sched_default && distinct <= 4096 && nt_walk >= 1024) {
      for (int64_t pre : {(int64_t)16, (int64_t)112, (int64_t)896}) { walk(t0, t0 + pre); t0 += pre; }
      launches = 1;
    }
    for (; t0 < nt_walk; t0 += step, step *= (++launches < 2 ? 1 : (launches < 4 ? 2 : 4))) walk(t0, t0 + step);
  }
  HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, padded_words(t->nrows) * 8, s));
  HIP_CHECK(hipMemsetAsync(q->tile_counts.p, 0, (size_t)ceil_div(t->nrows, kTileRows) * 4, s));
  { LaunchTimer lt(ctx, "unique_mark"); launch_dense_scatter(s, T.first.as<uint64_t>(), T.range, aux, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>()); }
  uint64_t st[3];
  HIP_CHECK(hipMemcpyAsync(st, (char*)T.aux.p + 40, 24, hipMemcpyDeviceToHost, s));      // outside flag, distinct values, values whose first row is known
  stream_wait(ctx);
  if (st[0] != 0 || st[1] != st[2]) fail(DFDB_ERR_DEVICE, "unique: the dense form lost a key (outside %llu, distinct %llu, found %llu)", (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2]);
  return true;
}

// The radix-partitioned form of the general unique (k_radix.hip, round 6): when the hash table would outgrow the L2s — `d0` distinct values among the first `r0`
// selected rows say how many the whole selection holds, assuming they turn up evenly: d0 = D (1 - exp(-r0 / D)) — every selected row is written once as a
// {key image, row} record into one of P partitions (by the top bits of the key's hash) and each partition is reduced through a table in LDS.  Streams instead
// of one random line per row.  false = not taken (too few or too many distinct values, no room for the 12 bytes per selected row, a partition that outgrew
// its table): the caller goes on with the hash table; the selection is as it was.
// What the two radix forms (unique, groupreduce) share on the host: the pool of pages for `nsel` records in 2^kbits partitions (k_radix.hip), its control words —
// the sample's counts [P], the streams' running positions [P x share], the pool's counter, the hot keys' count, then `extra` bytes the caller lays out (64-byte aligned) — and the hot keys'
// list.  The temporaries go back to the buffer pool (not through hipFree, which drains the device: 1.5 ms of a 16-ms call once), the stream drained first; the
// records' scratch stays with the context between calls (hipMalloc / hipFree of 12 GB cost 3-4 ms; buffers above 1 GB never enter the pool; dfdb_ctx_destroy
// releases it) unless it is more than a quarter of the device's memory.
struct RadixRun {
  dfdb_ctx* ctx; int kbits, P, C; int64_t PS; RadixPool pool{}; DevBuf ctl, pt, hot; size_t tail = 0;
  RadixRun(dfdb_ctx* c, int kb) : ctx(c), kbits(kb), P(1 << kb), C((int)round_up(4 * std::max(1, c->prop.multiProcessorCount), radix_share())), PS((int64_t)(1 << kb) * radix_share()) {}
  ~RadixRun() {
    (void)hipStreamSynchronize(ctx->stream);
    { RecycleScope rs; ctl.release(); pt.release(); hot.release(); }
    if (ctx->radix_recs.bytes > ctx->prop.totalGlobalMem / 4) ctx->radix_recs.release();
  }
  char* extra() const { return (char*)ctl.p + tail; }
  // false: no room, or more records than a 32-bit place in the pool can name
  bool prepare(int64_t nsel, bool with_values, size_t extra_bytes) {
    if (radix_pool_pages(nsel, kbits) * 8192 >= (1ll << 32)) return false;
    pool.maxv = radix_pool_maxv(nsel, kbits);
    pool.dump_page = (uint32_t)(radix_pool_pages(nsel, kbits) - 1);
    pool.hot_cap = (uint32_t)C * (uint32_t)radix_hot_slots();
    const size_t ctl_words = (size_t)P + (size_t)PS + 16;                 // the sample's counts [P], the streams' running positions [PS], the pool's counter, the hot keys' count
    tail = (ctl_words * 4 + 63) / 64 * 64;
    try {
      ctl.ensure(tail + extra_bytes + 64); pt.ensure((size_t)PS * pool.maxv * 4 + 512);      // (+ 64 entries nobody owns: the table passes read 64 at a time)
      hot.ensure((size_t)pool.hot_cap * 24 + 64);
      ctx->radix_recs.ensure((size_t)radix_pool_record_bytes(nsel, kbits, with_values) + 256);
    } catch (const Error& e) { if (e.code != DFDB_ERR_NOMEM) throw; (void)hipGetLastError(); return false; }
    pool.front = ctl.as<uint32_t>() + P; pool.next_page = ctl.as<uint32_t>() + P + PS; pool.hot_n = ctl.as<uint32_t>() + P + PS + 1; pool.pt = pt.as<uint32_t>(); pool.hot = hot.as<uint64_t>();
    HIP_CHECK(hipMemsetAsync(ctl.p, 0, tail + extra_bytes, ctx->stream));
    HIP_CHECK(hipMemsetAsync(pt.p, 0xFF, (size_t)PS * pool.maxv * 4, ctx->stream));
    return true;
  }
  // unique: does some partition hold a large part of all the rows (a value that a third of the column has)?  Every 16th tile is counted (all of them in a small
  // table: 0.1 ms per 1e9 rows) and the counts come back.  1 = skewed, 0 = not, -1 = the sample could not be launched
  int skewed(const uint64_t* sel, const void* col, int dt, const uint64_t* miss, int64_t nrows, int times) {      // times: "large" = more than this many average partitions
    const int step = radix_rows_per_chunk(nrows, C) / 8192 >= 32 ? 16 : 1;
    { LaunchTimer lt(ctx, "radix_sample");
      if (!launch_radix_sample(ctx->stream, sel, col, dt, miss, nrows, kbits, C, step, ctl.as<uint32_t>())) return -1; }
    std::vector<uint32_t> cn((size_t)P);
    HIP_CHECK(hipMemcpyAsync(cn.data(), ctl.p, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
    stream_wait(ctx);
    uint64_t maxp = 0, total = 0;
    for (int p = 0; p < P; p++) { maxp = std::max<uint64_t>(maxp, cn[(size_t)p]); total += cn[(size_t)p]; }
    return maxp * (uint64_t)step > 65536 && maxp * (uint64_t)P > (uint64_t)times * total ? 1 : 0;
  }
};
// how many distinct values `cnt` selected rows hold when the first r0 of them held d0, assuming they turn up evenly: d0 = D (1 - exp(-r0 / D)), by bisection
static double estimate_distinct(uint64_t d0, uint64_t r0, int64_t cnt) {
  double D = (double)cnt;
  if (d0 < r0) {
    double lo = (double)d0, hi = (double)cnt;
    for (int it = 0; it < 60 && hi - lo > 1.0; it++) { const double mid = 0.5 * (lo + hi); if (mid * (1.0 - std::exp(-(double)r0 / mid)) < (double)d0) lo = mid; else hi = mid; }
    D = std::min((double)cnt, hi);
  }
  return D;
}
static bool unique_radix(dfdb_query* q, const Column& col, int64_t cnt, UniqueTables& T, uint64_t d0, uint64_t r0) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  // MEASURED, 1e9 Int64 / Float64 rows of 1e6 distinct values (profiles/r6_unique_radix.txt): sample 0.1 + partition 4.5 + LDS tables 3.2 ms = 8.8-9.3 ms end to end against the
  // hash table's 19.8 (the first build, with splitmix64, a counting pass and the partition re-hashed on the way out, was 18.4).  The partition pass waits for its STORES
  // (2.8 ms without them): 192-byte runs behind running positions that one XCD's workgroups share, so that whole lines leave that XCD's L2.  On by default; option = 0: the hash table.
  const int64_t mode = ctx_option(ctx, "unique_radix", 1);
  if (mode == 0 || t->nrows >= (1ll << 32) - 8192 || r0 == 0 || d0 == 0) return false;      // (rows and pool positions are 32-bit; all ones is "no row")
  if (mode < 2 && cnt < (4ll << 20)) return false;                       // (2: a test knob — any size; measured against the hash table down to 3 M selected rows of 1e9: tools/r6_radix_selective.py)
  const double D = estimate_distinct(d0, r0, cnt);
  if (mode < 2 && D < 131072.0) return false;                            // the hash table stays in the L2s: nothing to gain
  int kbits = 8;
  while (kbits < 10 && D / (double)(1 << kbits) > 2500.0) kbits++;       // (fewer partitions = longer runs per tile of the partition pass; more = emptier tables in the unique pass: 1e6 values -> 512)
  if (const char* e = getenv("DFDB_RADIX_KBITS")) kbits = std::min(10, std::max(6, atoi(e)));      // (an A/B switch for measurements)
  if (mode < 2 && D / (double)(1 << kbits) > 5500.0) return false;       // the partitions' tables (8192 slots) would overflow
  const int dt = dt_base(col.dtype);
  const uint64_t* miss = dt_nullable(col.dtype) ? col.missing.as<uint64_t>() : nullptr;
  const int64_t nt = ceil_div(t->nrows, kTileRows);
  const size_t nw = padded_words(t->nrows);
  struct Keep {                                                          // the selection, set aside (pooled like RadixRun's temporaries)
    dfdb_ctx* ctx; DevBuf sel, tc;
    ~Keep() { (void)hipStreamSynchronize(ctx->stream); RecycleScope rs; sel.release(); tc.release(); }
  } keep{ctx, {}, {}};
  DevBuf &sel_keep = keep.sel, &tc_keep = keep.tc;
  RadixRun run(ctx, kbits);
  if (!run.prepare(cnt, false, 0)) return false;
  try { sel_keep.ensure(nw * 8); tc_keep.ensure((size_t)nt * 4 + 64); } catch (const Error& e) { if (e.code != DFDB_ERR_NOMEM) throw; (void)hipGetLastError(); return false; }
  const RadixPool& pool = run.pool; const int C = run.C;
  DevBuf& recs = ctx->radix_recs;
  // SKEW: one workgroup reduces one partition, so a value that a large part of the rows hold would be one CU's work while 255 wait.  The sample says whether some
  // partition holds more than eight average ones; then the partition kernels that keep hot keys out of the records run (k_radix.hip, hot keys: 1.1 ms slower where
  // nothing is hot, which is why they are not the only ones)
  const int sk = run.skewed(q->bitmap.as<uint64_t>(), col.data.p, dt, miss, t->nrows, 8);
  if (sk < 0) return false;
  if (sk) prof_note(ctx, "unique_radix.skewed");
  { LaunchTimer lt(ctx, "radix_partition");
    if (!launch_radix_partition(s, q->bitmap.as<uint64_t>(), col.data.p, dt, miss, t->nrows, kbits, C, pool, recs.as<uint32_t>(), T.aux.as<uint64_t>(), nullptr, sk != 0)) return false; }
  // the selection is set aside (a partition that outgrows its table means: back to the hash table, over the same selection)
  HIP_CHECK(hipMemcpyAsync(sel_keep.p, q->bitmap.p, nw * 8, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(tc_keep.p, q->tile_counts.p, (size_t)nt * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, nw * 8, s));
  HIP_CHECK(hipMemsetAsync(q->tile_counts.p, 0, (size_t)nt * 4, s));
  bool ok;
  { LaunchTimer lt(ctx, "radix_unique");
    ok = launch_radix_unique(s, recs.as<uint32_t>(), pool, kbits, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), T.aux.as<uint64_t>(), ctx->prop.multiProcessorCount); }
  uint64_t aborted = 0;
  if (ok) { HIP_CHECK(hipMemcpyAsync(&aborted, (char*)T.aux.p + 24, 8, hipMemcpyDeviceToHost, s)); stream_wait(ctx); }
  if (!ok || aborted || ctx_option(ctx, "unique_radix", 1) == 3) {        // (3: a test knob — behave as if a partition had overflowed)
    HIP_CHECK(hipMemcpyAsync(q->bitmap.p, sel_keep.p, nw * 8, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(q->tile_counts.p, tc_keep.p, (size_t)nt * 4, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemsetAsync((char*)T.aux.p + 24, 0, 8, s));
    stream_wait(ctx);
    prof_note(ctx, "unique_radix.fell_back");
    return false;
  }
  prof_note(ctx, "unique_radix.taken");
  stream_wait(ctx);                                                      // (the temporaries die with this frame)
  return true;
}

// the general form: an open-addressing table of {key, first row} sized by the distinct values as they turn up (k_unique.hip)
static void unique_hashed(dfdb_query* q, const Column& col, int64_t cnt, UniqueTables& T) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const bool is_str = T.is_str;
  const int dt = dt_base(col.dtype);
  const uint64_t* miss = (!is_str && dt_nullable(col.dtype)) ? col.missing.as<uint64_t>() : nullptr;
  const int64_t nt = ceil_div(t->nrows, kTileRows);
  const uint64_t capmax = pow2_at_least((uint64_t)cnt * 2);
  const uint64_t cap0 = std::min(capmax, pow2_at_least(1ull << std::min<int64_t>(40, std::max<int64_t>(10, ctx_option(ctx, "unique_cap0_log2", 21)))));
  const int64_t c0 = ctx_option(ctx, "unique_chunk_tiles", 0) > 0 ? ctx_option(ctx, "unique_chunk_tiles", 0) : 1024;
  const int64_t bounds[3] = {std::min(nt, c0), std::min(nt, c0 * 17), nt};
  auto alloc = [&](UniqueTables& U, uint64_t cap) {
    U.cap = cap;
    U.ent.ensure(cap * sizeof(UniqueEntry));
    HIP_CHECK(hipMemsetAsync(U.ent.p, 0xFF, cap * sizeof(UniqueEntry), s));
    if (is_str) { U.rep_off.ensure(cap * 8); U.rep_len.ensure(cap * 4); }
  };
  auto insert = [&](int64_t t0, int64_t t1) {
    LaunchTimer lt(ctx, "unique_insert");
    if (is_str) launch_unique_str(s, 0, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), col.data.as<int32_t>(), (const int64_t*)col.tile_off.p, col.bytes.as<uint8_t>(),
                                  t->nrows, t0, t1, T.ent.as<UniqueEntry>(), T.rep_off.as<uint64_t>(), T.rep_len.as<uint32_t>(), T.cap - 1, T.aux.as<uint64_t>(), T.salt);
    else launch_unique_insert(s, q->bitmap.as<uint64_t>(), col.data.p, dt, miss, t0 * kTileRows, std::min(t->nrows, t1 * kTileRows), T.ent.as<UniqueEntry>(), T.cap - 1, T.aux.as<uint64_t>());
  };
  uint64_t st[2] = {0, 0};                                             // claimed slots, abort flag
  auto read_state = [&](uint64_t* selected_before, int64_t tile) {
    HIP_CHECK(hipMemcpyAsync(st, (char*)T.aux.p + 16, 16, hipMemcpyDeviceToHost, s));
    if (selected_before) HIP_CHECK(hipMemcpyAsync(selected_before, q->prefix.as<uint64_t>() + tile, 8, hipMemcpyDeviceToHost, s));
    stream_wait(ctx);
  };
  auto grow = [&](uint64_t cap) {                                      // the entries move to a table of `cap` slots
    LaunchTimer lt(ctx, "unique_migrate");
    UniqueTables N; N.is_str = is_str;
    alloc(N, cap);
    HIP_CHECK(hipMemsetAsync((char*)T.aux.p + 16, 0, 16, s));
    launch_unique_migrate(s, T.ent.as<UniqueEntry>(), is_str ? T.rep_off.as<uint64_t>() : nullptr, is_str ? T.rep_len.as<uint32_t>() : nullptr, T.cap, N.ent.as<UniqueEntry>(),
                          N.rep_off.as<uint64_t>(), N.rep_len.as<uint32_t>(), cap - 1, T.aux.as<uint64_t>());
    read_state(nullptr, 0);
    if (st[1]) fail(DFDB_ERR_DEVICE, "unique: %llu entries do not fit a table of %llu slots", (unsigned long long)st[0], (unsigned long long)cap);
    T.ent = std::move(N.ent); T.rep_off = std::move(N.rep_off); T.rep_len = std::move(N.rep_len); T.cap = cap;
  };
  T.salt = 0x51ED270B27B4F3CFull;
  for (int k = 0; k < T.salt_skip; k++) T.salt = splitmix64_host(T.salt);
  for (int tries = T.salt_skip;; tries++, T.salt = splitmix64_host(T.salt)) {
    if (!T.ent.p || T.cap != cap0) { T.ent.release(); alloc(T, cap0); } else HIP_CHECK(hipMemsetAsync(T.ent.p, 0xFF, T.cap * sizeof(UniqueEntry), s));
    unique_reset_aux(ctx, T.aux);
    int64_t t0 = 0;
    uint64_t claims_c0 = 0, rows_c0 = 0;
    T.optimistic = false;
    // (a column of FEW keys: the rows of a chunk run side by side, all see an empty slot or a larger first row, and all send their atomic — 1 M rows on seven
    // slots took 1.1 ms.  Sixteen tiles first: after them the chunk's rows find their key with a smaller row and send nothing.  Inserts are idempotent.)
    if (bounds[0] > 64) insert(0, 16);
    for (int c = 0; c < 3; c++) {
      const int64_t t1 = bounds[c];
      if (t1 <= t0) continue;
      uint64_t r = 0;
      // (numeric keys: only for a caller whose own pass over the rows can report a key without a slot — groupreduce's accumulate pass, defer_verify —, and only when
      // the distinct keys are few beside the rows, so that the first rows come straight out of the table below)
      // (String keys too.  After an optimistic break the mark step below must take the first rows straight out of the table — launch_unique_scatter,
      // `st[0] * 8 <= cnt` — because the per-row mark pass looks every selected row up with an unbounded probe, and a row of the un-inserted tail whose
      // string the table does not hold would never leave that loop: a GPU hang, which no host-side guard can catch.)
      if (c == 2 && (is_str || T.defer_verify) && T.allow_optimistic && claims_c0 != ~0ull && st[0] * 8 <= (uint64_t)cnt) { T.optimistic = true; break; }
      for (;;) {
        insert(t0, t1);
        read_state(&r, t1);
        if (!st[1]) break;
        if (T.cap >= capmax) fail(DFDB_ERR_DEVICE, "unique: probe sequences of a half-empty table got too long (%llu slots, %llu keys)", (unsigned long long)T.cap, (unsigned long long)st[0]);
        grow(std::min(capmax, T.cap * 4));                             // (that clears the flag) and the chunk again: inserts are idempotent
      }
      // (optimistic: chunk 1 must have fed at least 64 K selected rows and claimed nothing that chunk 0 had not)
      if (c == 0) { claims_c0 = st[0]; rows_c0 = r; }
      if (c == 0 && !is_str && T.group_probe && r > 0) {               // groupreduce asks: more groups than LDS accumulators hold?  Then it reduces by radix, first rows included
        const double D = estimate_distinct(st[0], r, cnt);
        if (D > (double)kGroupsInLds) { T.group_estimate = (int64_t)D; return; }
      }
      if (c == 0 && !is_str && !T.defer_verify && unique_radix(q, col, cnt, T, st[0], r)) return;      // the radix-partitioned form took it: q's bitmap holds the first occurrences
      else if (c == 1 && !(st[0] == claims_c0 && r >= rows_c0 + 65536)) claims_c0 = ~0ull;
      if (c == 0 && bounds[1] <= bounds[0]) claims_c0 = ~0ull;
      const uint64_t want = unique_capacity_wanted(st[0], r, (uint64_t)cnt - std::min<uint64_t>(r, (uint64_t)cnt), T.cap, capmax);
      if (want > T.cap) grow(want);
      t0 = t1;
    }
    if (!is_str || T.defer_verify) break;
    // the pass that compares every selected row with its slot's representative.  After an optimistic insert it also meets the rows that were never inserted: a string
    // the table does not hold raises the abort word and everything runs again with every row inserted (as groupreduce's accumulate pass does it)
    if (T.optimistic) HIP_CHECK(hipMemsetAsync((char*)T.aux.p + 24, 0, 8, s));
    launch_unique_str(s, T.optimistic ? 3 : 1, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), col.data.as<int32_t>(), (const int64_t*)col.tile_off.p, col.bytes.as<uint8_t>(),
                      t->nrows, 0, nt, T.ent.as<UniqueEntry>(), T.rep_off.as<uint64_t>(), T.rep_len.as<uint32_t>(), T.cap - 1, T.aux.as<uint64_t>(), T.salt);
    int hit = 0; uint64_t unknown = 0;
    HIP_CHECK(hipMemcpyAsync(&hit, (char*)T.aux.p + 32, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&unknown, (char*)T.aux.p + 24, 8, hipMemcpyDeviceToHost, s));
    stream_wait(ctx);
    if (T.optimistic && (unknown != 0 || ctx_option(ctx, "groupreduce_optimistic", 1) == 2)) { T.allow_optimistic = false; continue; }   // (2: the test knob, as in groupreduce)
    if (!hit) break;                                                   // no two different strings shared a key: the table is exact
    if (tries >= 8) fail(DFDB_ERR_DEVICE, "unique: hash collisions under 8 different salts");
  }
  LaunchTimer lt(ctx, "unique_mark");
  if (st[0] * 8 <= (uint64_t)cnt) {                                    // few distinct values beside the rows: the first rows straight out of the table
    HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, padded_words(t->nrows) * 8, s));
    HIP_CHECK(hipMemsetAsync(q->tile_counts.p, 0, (size_t)nt * 4, s));
    launch_unique_scatter(s, T.ent.as<UniqueEntry>(), T.cap, T.aux.as<uint64_t>(), q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>());
  } else if (is_str) {
    launch_unique_str(s, 2, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), col.data.as<int32_t>(), (const int64_t*)col.tile_off.p, col.bytes.as<uint8_t>(),
                      t->nrows, 0, nt, T.ent.as<UniqueEntry>(), T.rep_off.as<uint64_t>(), T.rep_len.as<uint32_t>(), T.cap - 1, T.aux.as<uint64_t>(), T.salt);
  } else {
    launch_unique_mark(s, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), col.data.p, dt, miss, t->nrows, T.ent.as<UniqueEntry>(), T.cap - 1, T.aux.as<uint64_t>());
  }
}

static void unique_impl(dfdb_query* q, int32_t p, UniqueTables* keep) {
  ensure_executed_checked(q);
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx;
  if (p < 0 || (size_t)p >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", p);
  const Node& e = *q->proj[(size_t)p].expr;
  if (e.op != DFIR_COL) fail(DFDB_ERR_UNSUPPORTED, "unique of a computed column: materialise it as a column first (dfdb_table_add_from_query)");
  const Column& col = need_resident(t, e.col);
  const int64_t cnt = query_count(q, -1);
  if (cnt == 0 || t->nrows == 0) return;
  if (!keep && col.dict_n > 0) { dict_unique(q, col, nullptr); return; }
  UniqueTables local;
  UniqueTables& T = keep ? *keep : local;
  T.is_str = dt_base(col.dtype) == DFDB_STRING; T.dense = false;
  if (!keep) local.allow_optimistic = ctx_option(ctx, "groupreduce_optimistic", 1) != 0;      // (unique's own compare pass meets every row: the same bargain)
  {
    LaunchTimer lt(ctx, "unique");
    const bool dense = !T.is_str && unique_dense_dtype(dt_base(col.dtype)) && ctx_option(ctx, "unique_dense", 1) != 0 && unique_dense(q, col, T);
    if (!dense) unique_hashed(q, col, cnt, T);
  }
  scan_prefix(q);
  selection_changed(q); q->cap_col = -1; q->cap_col2 = -1; q->cap_str_col = -1; q->agg_col = -1;
  stream_wait(ctx);                                        // the tables die here (or stay with the caller: groupreduce looks rows up in them)
  if (!keep) { RecycleScope rs; local = UniqueTables(); }
}
void query_unique(dfdb_query* q, int32_t p) { unique_impl(q, p, nullptr); }

// groupreduce over MORE groups than a workgroup's LDS accumulators hold, by radix (k_radix.hip, round 6): with the first occurrences already in q's bitmap (unique
// made them: `ng` groups, q->prefix scanned) every selected row of `sel` is written as a {key, row, value} record into one of P partitions, each partition is reduced
// through a table in LDS, and a key's result goes to the place its first row's rank names.  false = not taken (nothing was written to the outputs): too many groups
// for the tables, a skewed column, no room — the caller's accumulate pass (global atomics) runs.  85-135 ms -> see profiles/r6_groupreduce_radix.txt.
// mark: q's bitmap does NOT hold every group's first row yet (unique looked at the head of the column only): the table pass marks them itself — the bitmap is
// cleared first, scanned afterwards, and *ng_io becomes the number of groups; on false the bitmap is whatever the pass left (the caller restores the selection).
// (keys: the key column's values, their base dtype and missing bits — or a dictionary's 16-bit codes)
static bool group_radix(dfdb_query* q, const void* keys, int dt, const uint64_t* kmiss, const Column* vc, int op, int64_t nsel, int64_t* ng_io, const uint64_t* sel, bool mark) {
  int64_t ng = mark ? *ng_io + *ng_io / 4 + 1024 : *ng_io;                // (marking: an estimate from the head — the tables are sized with room to spare)
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const int64_t mode = ctx_option(ctx, "unique_radix", 1);
  if (mode == 0 || t->nrows >= (1ll << 32) - 8192 || dt == DFDB_STRING) return false;
  int kbits = 9;                                                         // (512 partitions at least: one workgroup reduces one partition, and there are 256 CUs)
  while (kbits < 10 && ng / (1ll << kbits) > 1200) kbits++;              // (a 4096-slot table at 30 % load)
  if ((mark ? *ng_io : ng) / (1ll << kbits) > 2400) return false;        // (59 % load: the claims stop a table at 7/8, and that raises the abort word)
  const int P = 1 << kbits;
  struct Res { dfdb_ctx* ctx; DevBuf res; ~Res() { (void)hipStreamSynchronize(ctx->stream); RecycleScope rs; res.release(); } } tmp{ctx, {}};
  RadixRun run(ctx, kbits);
  if (!run.prepare(nsel, true, 128)) return false;
  try { tmp.res.ensure((size_t)(mark ? (int64_t)P * radix_group_slots() : ng) * 16 + 256); } catch (const Error& e) { if (e.code != DFDB_ERR_NOMEM) throw; (void)hipGetLastError(); return false; }
  const RadixPool& pool = run.pool; const int C = run.C;
  DevBuf& recs = ctx->radix_recs;
  uint64_t* aux = (uint64_t*)run.extra();                                // [0] the unstorable key's first row, [1] the missing key's, [3] abort; [4..7] gspec; [8] nres
  RadixGroup g{};
  g.valcol = vc ? vc->data.p : nullptr; g.valdt = vc ? dt_base(vc->dtype) : 0;
  g.gop = op == DFDB_AGG_SUM ? (q->gr_kind == 2 ? 2 : 1) : (op == DFDB_AGG_MIN ? 3 : (op == DFDB_AGG_MAX ? 4 : 0));
  if (!vc) g.gop = 0;
  g.vkind = q->gr_kind; g.results = tmp.res.p; g.gspec = aux + 4; g.nres = (uint32_t*)(aux + 8);
  HIP_CHECK(hipMemsetAsync(aux, 0xFF, 16, s));                           // aux[0], aux[1] = none
  if (g.gop == 3) { HIP_CHECK(hipMemsetAsync(aux + 5, 0xFF, 8, s)); HIP_CHECK(hipMemsetAsync(aux + 7, 0xFF, 8, s)); }      // gspec[1], [3]: a minimum starts at all ones
  // a key that a large part of the rows hold is reduced by the partition pass itself (k_radix.hip, hot keys: the form this replaces sent every one of its rows
  // through a global atomic on ONE address — 3.6 s per 1e9 rows with a key that 30 % of them hold); the kernels that do so are 0.8-1.1 ms slower where nothing is
  // hot, so the sample picks: some partition above THREE average ones, and they run
  const int sk = run.skewed(sel, keys, dt, kmiss, t->nrows, 3);
  if (sk < 0) return false;
  if (sk) prof_note(ctx, "group_radix.skewed");
  { LaunchTimer lt(ctx, "radix_partition");
    if (!launch_radix_partition(s, sel, keys, dt, kmiss, t->nrows, kbits, C, pool, recs.as<uint32_t>(), aux, &g, sk != 0)) return false; }
  if (mark) {
    HIP_CHECK(hipMemsetAsync(q->bitmap.p, 0, padded_words(t->nrows) * 8, s));
    HIP_CHECK(hipMemsetAsync(q->tile_counts.p, 0, (size_t)ceil_div(t->nrows, kTileRows) * 4, s));
  }
  { LaunchTimer lt(ctx, "radix_group");
    if (!launch_radix_group(s, recs.as<uint32_t>(), pool, kbits, mark, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), aux, g, ctx->prop.multiProcessorCount)) return false; }
  uint64_t tailw[9] = {};
  HIP_CHECK(hipMemcpyAsync(tailw, aux, 72, hipMemcpyDeviceToHost, s));
  stream_wait(ctx);
  const uint32_t nres = (uint32_t)tailw[8];
  if (tailw[3] != 0) { prof_note(ctx, "group_radix.fell_back"); return false; }
  if (mark) {
    scan_prefix(q);
    selection_changed(q);
    ng = query_count(q, -1);
    *ng_io = ng;
  }
  if ((int64_t)nres + (tailw[0] != ~0ull ? 1 : 0) + (tailw[1] != ~0ull ? 1 : 0) != ng) { prof_note(ctx, "group_radix.mismatch"); return false; }     // (cannot happen: the table pass saw every key unique saw)
  q->gr_cnt.ensure((size_t)ng * 8 + 64); q->gr_val.ensure((size_t)ng * 8 + 64);
  HIP_CHECK(hipMemsetAsync(q->gr_cnt.p, 0, (size_t)ng * 8 + 64, s));
  HIP_CHECK(hipMemsetAsync(q->gr_val.p, op == DFDB_AGG_MIN ? 0xFF : 0, (size_t)ng * 8 + 64, s));
  launch_radix_group_finish(s, g, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), aux, q->gr_cnt.as<uint64_t>(), q->gr_val.as<uint64_t>());
  prof_note(ctx, "group_radix.taken");
  return true;                                                           // (the temporaries' destructor drains the stream)
}

// groupreduce(view, (:key,); out = :val => Stat()) (src/tables/aggregate.jl:1-36; unfinished in the reference: it numbers the groups in order of
// first appearance of the key and prints the map).  Completed to that intent: one group per distinct key (isequal), groups in order of first
// appearance, count and one reduced value per group.  Device side: unique's table + k_group_ids + k_group_accumulate (k_unique.hip).
void launch_group_ids(hipStream_t s, UniqueEntry* ent, uint64_t cap, uint64_t* special, const uint64_t* ubits, const uint64_t* uprefix);
int launch_group_accumulate(hipStream_t s, const uint64_t* sel, const void* keycol, int keydt, const uint64_t* missing, const void* valcol, int valdt, int op,
                            int64_t nrows, const UniqueEntry* ent, uint64_t mask, const uint64_t* special, uint64_t* cnt, uint64_t* val,
                            int64_t ngroups, uint64_t val_init, uint64_t* unknown_flag = nullptr, const void* gkeys = nullptr);
void launch_group_accumulate_str(hipStream_t s, const uint64_t* sel, const int32_t* sizes, const int64_t* tile_off, const uint8_t* bytes, const void* valcol, int valdt,
                                 int op, int64_t nrows, const UniqueEntry* ent, const uint64_t* rep_off, const uint32_t* rep_len, uint64_t mask, uint64_t* special, uint64_t salt,
                                 uint64_t* cnt, uint64_t* val, int64_t ngroups, uint64_t val_init);
void launch_group_finish(hipStream_t s, uint64_t* val, int64_t ng, int kind, int op);

void query_groupreduce(dfdb_query* q, int32_t key_p, int32_t val_p, int32_t op, int64_t* ngroups, int64_t* key_bytes) {
  ensure_executed_checked(q);
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  if (key_p < 0 || (size_t)key_p >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", key_p);
  const Node& ke = *q->proj[(size_t)key_p].expr;
  if (ke.op != DFIR_COL) fail(DFDB_ERR_UNSUPPORTED, "groupreduce by a computed column: materialise it as a column first (dfdb_table_add_from_query)");
  if (op != DFDB_AGG_COUNT && op != DFDB_AGG_SUM && op != DFDB_AGG_MIN && op != DFDB_AGG_MAX) fail(DFDB_ERR_ARGUMENT, "unknown statistic %d", op);
  const Column* vc = nullptr;
  if (op != DFDB_AGG_COUNT) {
    if (val_p < 0 || (size_t)val_p >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", val_p);
    const Node& ve = *q->proj[(size_t)val_p].expr;
    if (ve.op != DFIR_COL || !dt_isnum(ve.dtype) || dt_nullable(ve.dtype)) fail(DFDB_ERR_UNSUPPORTED, "groupreduce over %s: a plain numeric column is needed", dt_name(ve.dtype).c_str());
    vc = &need_resident(t, ve.col);
  }
  const Column& kc = need_resident(t, ke.col);
  q->gr_n = 0; q->gr_key = key_p; q->gr_op = op; q->gr_kind = 0;
  if (vc) { const int b = dt_base(vc->dtype); q->gr_kind = dt_isfloat(b) ? 2 : (dt_issigned(b) ? 0 : 1); }
  if (ngroups) *ngroups = 0;
  if (key_bytes) *key_bytes = 0;
  const int64_t nsel = query_count(q, -1);
  if (nsel == 0 || t->nrows == 0) { q->gr_state = 1; return; }
  // the full selection is kept aside: unique narrows q's bitmap to the first occurrences (= the groups, in order)
  const size_t nw = padded_words(t->nrows);
  q->gr_sel.ensure(nw * 8);
  HIP_CHECK(hipMemcpyAsync(q->gr_sel.p, q->bitmap.p, nw * 8, hipMemcpyDeviceToDevice, s));
  if (kc.dict_n > 0) {                                       // K9: group by the dictionary codes
    DevBuf& rank = q->du_rank;
    int64_t ng = dict_unique(q, kc, &rank);
    // more codes in use than LDS accumulators hold: the codes are the keys of the radix form (the first occurrences are in the bitmap already)
    if (ng > kGroupsInLds && group_radix(q, kc.dict_codes.p, DFDB_U16, nullptr, vc, op, nsel, &ng, q->gr_sel.as<uint64_t>(), false)) {
      launch_group_finish(s, q->gr_val.as<uint64_t>(), ng, q->gr_kind, op);
      stream_wait(ctx);
      q->gr_n = ng; q->gr_state = 2;
      if (ngroups) *ngroups = ng;
      if (key_bytes) *key_bytes = query_string_bytes(q, key_p);
      return;
    }
    q->gr_cnt.ensure((size_t)ng * 8 + 64); q->gr_val.ensure((size_t)ng * 8 + 64);
    const uint64_t init = op == DFDB_AGG_MIN ? ~0ull : 0ull;
    HIP_CHECK(hipMemsetAsync(q->gr_cnt.p, 0, (size_t)ng * 8 + 64, s));
    HIP_CHECK(hipMemsetAsync(q->gr_val.p, op == DFDB_AGG_MIN ? 0xFF : 0, (size_t)ng * 8 + 64, s));
    { LaunchTimer lt(ctx, "group_accumulate");
      launch_group_accumulate_codes(s, q->gr_sel.as<uint64_t>(), kc.dict_codes.as<uint16_t>(), rank.as<uint32_t>(), vc ? vc->data.p : nullptr, vc ? dt_base(vc->dtype) : 0, op,
                                    t->nrows, q->gr_cnt.as<uint64_t>(), q->gr_val.as<uint64_t>(), ng, init); }
    launch_group_finish(s, q->gr_val.as<uint64_t>(), ng, q->gr_kind, op);
    stream_wait(ctx);
    q->gr_n = ng; q->gr_state = 2;
    if (ngroups) *ngroups = ng;
    if (key_bytes) *key_bytes = query_string_bytes(q, key_p);
    return;
  }
  UniqueTables T;
  int64_t ng = 0;
  bool pessimistic = false, whole_dense = false, radix_failed = false;
  // integer keys, dense form: the table of first rows / group numbers is made from the HEAD of the column (4 M rows) when the accumulate pass can be the one with
  // the table in LDS — that pass meets every row anyway and raises a flag for a key that has no group, after which everything runs again over every row
  // (the presence pass over the whole key column was 1.3 of the 5.0 ms of 1e9 rows by 5000 keys)
  const bool head_ok = ctx_option(ctx, "groupreduce_optimistic", 1) != 0 && (dt_base(kc.dtype) == DFDB_I64 || dt_base(kc.dtype) == DFDB_U64) &&
                       (!vc || dt_width(vc->dtype) == 8);
  for (;;) {
    // String keys: the pass that compares every row with its slot's representative (unique's verify pass) is folded into the accumulate pass below, which
    // hashes every row and finds its slot anyway; should two different strings share a key the selection is put back and everything runs again under the next salt
    T.defer_verify = true;
    T.allow_optimistic = !pessimistic && ctx_option(ctx, "groupreduce_optimistic", 1) != 0;
    T.allow_head = head_ok && !whole_dense;
    T.group_probe = !radix_failed && ctx_option(ctx, "unique_radix", 1) != 0 && dt_base(kc.dtype) != DFDB_STRING;
    T.group_estimate = 0;
    unique_impl(q, key_p, &T);
    if (T.group_estimate > 0) {                                  // the first chunk of rows promises more groups than any accumulate pass's LDS holds: by radix, first rows and all
      ng = T.group_estimate;
      if (group_radix(q, kc.data.p, dt_base(kc.dtype), dt_nullable(kc.dtype) ? kc.missing.as<uint64_t>() : nullptr, vc, op, nsel, &ng, q->gr_sel.as<uint64_t>(), true)) break;
      radix_failed = true;                                       // (skewed, too many groups, no room: the bitmap may have been cleared — everything again, the old way)
      launch_missing_mask(s, q->gr_sel.as<uint64_t>(), false, false, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), t->nrows);
      scan_prefix(q);
      selection_changed(q);
      continue;
    }
    const bool head_table = T.dense && T.head_only;
    if (head_table) HIP_CHECK(hipMemsetAsync((char*)T.aux.p + 24, 0, 8, s));        // (the word the accumulate pass raises: aux[3])
    if (T.optimistic) HIP_CHECK(hipMemsetAsync((char*)T.aux.p + 24, 0, 8, s));      // the accumulate pass raises this word when it meets a string the table does not hold
    ng = query_count(q, -1);
    uint64_t* special = T.aux.as<uint64_t>();
    if (T.dense) launch_dense_group_ids(s, T.first.as<uint64_t>(), T.range, special, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>());
    else launch_group_ids(s, T.ent.as<UniqueEntry>(), T.cap, special, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>());
    q->gr_cnt.ensure((size_t)ng * 8 + 64); q->gr_val.ensure((size_t)ng * 8 + 64);
    const uint64_t init = op == DFDB_AGG_MIN ? ~0ull : 0ull;
    HIP_CHECK(hipMemsetAsync(q->gr_cnt.p, 0, (size_t)ng * 8 + 64, s));
    HIP_CHECK(hipMemsetAsync(q->gr_val.p, op == DFDB_AGG_MIN ? 0xFF : 0, (size_t)ng * 8 + 64, s));
    // more groups than the LDS accumulators of any accumulate pass hold: by radix (group_radix) — which needs EVERY group's first row in the bitmap: a table made
    // from the head of the column / a prefix of the rows is made again from all of them first
    if (!radix_failed && !T.is_str && ng > kGroupsInLds && ctx_option(ctx, "unique_radix", 1) != 0 && ng <= 2400 * 1024) {
      const bool partial = head_table || T.optimistic;          // unique looked at the head of the column / a prefix of the rows: not every group's first row is marked
      if (group_radix(q, kc.data.p, dt_base(kc.dtype), dt_nullable(kc.dtype) ? kc.missing.as<uint64_t>() : nullptr, vc, op, nsel, &ng, q->gr_sel.as<uint64_t>(), partial)) break;
      radix_failed = true;
      if (partial) {                                             // not taken, and the bitmap may have been cleared: everything again, every row looked at
        whole_dense = true; pessimistic = true;
        launch_missing_mask(s, q->gr_sel.as<uint64_t>(), false, false, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), t->nrows);     // the full selection again
        scan_prefix(q);
        selection_changed(q);
        continue;
      }
    }
    int dense_lds = 0;
    { LaunchTimer lt(ctx, "group_accumulate");
      const uint64_t* kmiss = dt_nullable(kc.dtype) ? kc.missing.as<uint64_t>() : nullptr;
      if (T.dense)
        dense_lds = launch_group_accumulate_dense(s, q->gr_sel.as<uint64_t>(), kc.data.p, dt_base(kc.dtype), kmiss, vc ? vc->data.p : nullptr, vc ? dt_base(vc->dtype) : 0, op, t->nrows, T.lo,
                                      T.range, T.span_lo, T.span_hi, T.first.as<uint64_t>(), special, q->gr_cnt.as<uint64_t>(), q->gr_val.as<uint64_t>(), ng, init,
                                      head_table ? T.aux.as<uint64_t>() + 3 : nullptr);
      else if (T.is_str)
        launch_group_accumulate_str(s, q->gr_sel.as<uint64_t>(), kc.data.as<int32_t>(), (const int64_t*)kc.tile_off.p, kc.bytes.as<uint8_t>(), vc ? vc->data.p : nullptr,
                                    vc ? dt_base(vc->dtype) : 0, op, t->nrows, T.ent.as<UniqueEntry>(), T.rep_off.as<uint64_t>(), T.rep_len.as<uint32_t>(), T.cap - 1, special, T.salt,
                                    q->gr_cnt.as<uint64_t>(), q->gr_val.as<uint64_t>(), ng, init);
      else {
        // few groups of an 8-byte key: their keys (the key column at the first rows q's bitmap now holds, i.e. in group order) for the accumulate pass's LDS table
        const void* gkeys = nullptr;
        if (ng <= kGroupsInLds && dt_width(kc.dtype) == 8) {
          q->gr_keys.ensure((size_t)ng * 8 + 64);
          launch_gather(s, q->bitmap.as<uint64_t>(), q->prefix.as<uint64_t>(), kc.data.p, q->gr_keys.p, 8, t->nrows, ng);
          gkeys = q->gr_keys.p;
        }
        if (launch_group_accumulate(s, q->gr_sel.as<uint64_t>(), kc.data.p, dt_base(kc.dtype), kmiss, vc ? vc->data.p : nullptr,
                                    vc ? dt_base(vc->dtype) : 0, op, t->nrows, T.ent.as<UniqueEntry>(), T.cap - 1, special,
                                    q->gr_cnt.as<uint64_t>(), q->gr_val.as<uint64_t>(), ng, init, T.optimistic ? T.aux.as<uint64_t>() + 3 : nullptr, gkeys) > 0)
          prof_note(ctx, "group_accumulate.hash_lds");
      } }
    if (dense_lds > 0) prof_note(ctx, "group_accumulate.dense_lds");
    if (head_table) {
      uint64_t unknown = dense_lds < 0 ? 1ull : 0ull;                  // (-1: the LDS form could not take the job and nothing ran)
      if (!unknown) {
        HIP_CHECK(hipMemcpyAsync(&unknown, (char*)T.aux.p + 24, 8, hipMemcpyDeviceToHost, s));
        stream_wait(ctx);
      }
      if (unknown || ctx_option(ctx, "groupreduce_optimistic", 1) == 2) {          // (2: the test knob — as if a key had been missing)
        whole_dense = true;
        prof_note(ctx, "group_accumulate.head_redo");
        launch_missing_mask(s, q->gr_sel.as<uint64_t>(), false, false, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), t->nrows);     // the full selection again
        scan_prefix(q);
        selection_changed(q);
        continue;
      }
      prof_note(ctx, "group_accumulate.head_table");
    }
    if (!T.is_str && !T.optimistic) break;
    int hit = 0; uint64_t unknown = 0;
    HIP_CHECK(hipMemcpyAsync(&hit, (char*)T.aux.p + 32, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&unknown, (char*)T.aux.p + 24, 8, hipMemcpyDeviceToHost, s));
    stream_wait(ctx);
    const bool redo_all = T.optimistic && (unknown != 0 || ctx_option(ctx, "groupreduce_optimistic", 1) == 2);      // (2: a test knob — behave as if a string had been missing)
    if (!T.is_str && !redo_all) break;                       // (numeric keys: nothing to collide)
    if (!hit && !redo_all && ctx_option(ctx, "unique_test_collide", 0) <= T.salt_skip) break;
    if (redo_all) pessimistic = true;                         // a string first met after the inserted prefix: everything again, every row inserted
    else if (++T.salt_skip > 8) fail(DFDB_ERR_DEVICE, "unique: hash collisions under 8 different salts");
    launch_missing_mask(s, q->gr_sel.as<uint64_t>(), false, false, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), t->nrows);     // the full selection again
    scan_prefix(q);
    selection_changed(q);
  }
  launch_group_finish(s, q->gr_val.as<uint64_t>(), ng, q->gr_kind, op);
  stream_wait(ctx);                                        // the tables die here
  { RecycleScope rs; T = UniqueTables(); }
  q->gr_n = ng; q->gr_state = 2;
  if (ngroups) *ngroups = ng;
  if (key_bytes && dt_base(kc.dtype) == DFDB_STRING) *key_bytes = query_string_bytes(q, key_p);
}

// the groups' keys (gathered over the first occurrences, i.e. in order of first appearance), counts and values -> caller buffers (host);
// restores the query's full selection afterwards
void query_groupreduce_fetch(dfdb_query* q, dfdb_outcol* keys, int64_t* counts, int64_t* vals_i, double* vals_f) {
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  if (q->gr_state == 0) fail(DFDB_ERR_ARGUMENT, "ArgumentError: dfdb_query_groupreduce has not been called (or the query was executed, reset or changed since)");
  if (q->bitmap_rows != t->nrows) { q->gr_state = 0; fail(DFDB_ERR_ARGUMENT, "ArgumentError: the table changed between dfdb_query_groupreduce and its fetch"); }
  const int64_t ng = q->gr_n;
  if (ng > 0) {
    if (keys) { keys->memkind = keys->memkind == DFDB_MEM_DEVICE ? DFDB_MEM_DEVICE : DFDB_MEM_HOST; materialize_col(q, q->gr_key, *keys, ng); }
    std::vector<uint64_t> c((size_t)ng), v((size_t)ng);
    HIP_CHECK(hipMemcpyAsync(c.data(), q->gr_cnt.p, (size_t)ng * 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(v.data(), q->gr_val.p, (size_t)ng * 8, hipMemcpyDeviceToHost, s));
    stream_wait(ctx);
    for (int64_t g = 0; g < ng; g++) {
      if (counts) counts[g] = (int64_t)c[(size_t)g];
      const uint64_t b = v[(size_t)g];
      double d; memcpy(&d, &b, 8);
      if (q->gr_kind == 2) { if (vals_f) vals_f[g] = d; if (vals_i) vals_i[g] = (int64_t)d; }
      else { if (vals_i) vals_i[g] = (int64_t)b; if (vals_f) vals_f[g] = q->gr_kind == 1 ? (double)b : (double)(int64_t)b; }
    }
  } else if (keys) { keys->count = 0; keys->nbytes = 0; }
  if (q->gr_state == 2) {                                  // back to the full selection: bitmap + tile counts + prefix
    launch_missing_mask(s, q->gr_sel.as<uint64_t>(), false, false, q->bitmap.as<uint64_t>(), q->tile_counts.as<uint32_t>(), t->nrows);
    scan_prefix(q);
    selection_changed(q);
  }
  q->gr_state = 0;
}



// the device half of an aggregate: leaves {value, selected count} (16 bytes) of sum / min / max over projection column i in
// q->red_result on the engine stream and returns the accumulator dtype (DFDB_I64 / DFDB_U64 / DFDB_F64).  No host wait: the group
// layer (group.cpp) hands the 16 bytes to the RCCL all-reduce as they are; an empty selection leaves the identity of `op`.
int query_aggregate_device(dfdb_query* q, int32_t op, int32_t i) {
  ensure_executed(q);
  dfdb_table* t = q->t; dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  if (i < 0 || (size_t)i >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", i);
  const Node& e = *q->proj[(size_t)i].expr;
  if (!dt_isnum(e.dtype) || dt_nullable(e.dtype)) fail(DFDB_ERR_UNSUPPORTED, "aggregate over %s is not supported", dt_name(e.dtype).c_str());
  int dt = dt_base(e.dtype);
  q->red_scratch.ensure(reduce_scratch_bytes()); q->red_result.ensure(64);
  if (op == q->agg_op && e.op == DFIR_COL && q->agg_col == e.col && q->executed_stages == (int)q->stages.size()) {
    // the scan already reduced the selected values of this column per tile (k_scan_terms EXTRA = 2 / 3 / 4): reduce the partials
    const int64_t nt = ceil_div(t->nrows, kTileRows);
    if (q->agg_ones_tiles != nt) {
      q->agg_ones.ensure(padded_words(nt) * 8);
      HIP_CHECK(hipMemsetAsync(q->agg_ones.p, 0xff, (size_t)(nt / 64) * 8, s));
      const uint64_t tail = (nt & 63) ? ((1ull << (nt & 63)) - 1ull) : 0ull;
      HIP_CHECK(hipMemcpyAsync((uint64_t*)q->agg_ones.p + nt / 64, &tail, 8, hipMemcpyHostToDevice, s));
      stream_wait(ctx);
      q->agg_ones_tiles = nt;
    }
    dt = q->agg_dtype == DFDB_F64 ? DFDB_F64 : (q->agg_dtype == DFDB_U64 ? DFDB_U64 : DFDB_I64);
    { LaunchTimer lt(ctx, "reduce_partials"); launch_reduce(s, q->agg_ones.as<uint64_t>(), q->agg_partials.p, dt, op, nt, q->red_scratch.p, q->red_result.p); }
    // the count slot of the partial reduce counts TILES: the selected rows are the scan total
    HIP_CHECK(hipMemcpyAsync((uint64_t*)q->red_result.p + 1, q->prefix.as<uint64_t>() + nt, 8, hipMemcpyDeviceToDevice, s));
  } else if (e.op == DFIR_COL) {
    { LaunchTimer lt(ctx, "reduce"); launch_reduce(s, q->bitmap.as<uint64_t>(), need_resident(t, e.col).data.p, dt, op, t->nrows, q->red_scratch.p, q->red_result.p); }
  } else {   // computed column: materialise the selected values, then reduce them all
    const int64_t cnt = query_count(q, -1);
    DevBuf &full = q->tmp_b, &ones = q->tmp_c;
    full.ensure((size_t)std::max<int64_t>(cnt, 1) * dt_width(dt) + 256);
    if (cnt) run_interp_project(q, e, full.p, cnt, nullptr);
    const size_t nw = padded_words(cnt);
    ones.ensure(nw * 8);
    HIP_CHECK(hipMemsetAsync(ones.p, 0xff, (size_t)(cnt / 64) * 8, s));
    uint64_t tail = (cnt & 63) ? ((1ull << (cnt & 63)) - 1ull) : 0ull;
    HIP_CHECK(hipMemcpyAsync((uint64_t*)ones.p + cnt / 64, &tail, 8, hipMemcpyHostToDevice, s));
    stream_wait(q->t->ctx);
    { LaunchTimer lt(ctx, "reduce"); launch_reduce(s, ones.as<uint64_t>(), full.p, dt, op, cnt, q->red_scratch.p, q->red_result.p); }
  }
  return dt_isfloat(dt) ? DFDB_F64 : (dt == DFDB_U64 ? DFDB_U64 : DFDB_I64);
}

void query_aggregate(dfdb_query* q, int32_t op, int32_t i, int64_t* out_i, double* out_f) {
  if (op == DFDB_AGG_COUNT) { const int64_t n = query_count(q, -1); if (out_i) *out_i = n; if (out_f) *out_f = (double)n; return; }
  ensure_executed_checked(q);                       // (a host-facing result: a decode_on_scan execution answers for its decode first)
  const int dt = query_aggregate_device(q, op, i);
  dfdb_ctx* ctx = q->t->ctx;
  HIP_CHECK(hipMemcpyAsync(ctx->pinned_scalar, q->red_result.p, 16, hipMemcpyDeviceToHost, ctx->stream));
  stream_wait(ctx);
  if (ctx->pinned_scalar[1] == 0 && op != DFDB_AGG_SUM) fail(DFDB_ERR_ARGUMENT, "ArgumentError: reducing over an empty collection is not allowed");
  if (dt == DFDB_F64) { double d; memcpy(&d, &ctx->pinned_scalar[0], 8); if (out_f) *out_f = d; if (out_i) *out_i = (int64_t)d; }
  else { const int64_t v = ctx->pinned_scalar[0]; if (out_i) *out_i = v; if (out_f) *out_f = dt == DFDB_U64 ? (double)(uint64_t)v : (double)v; }
}

}  // namespace dfdb
