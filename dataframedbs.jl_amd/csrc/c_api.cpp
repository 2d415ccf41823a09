// c_api.cpp — the extern "C" boundary of libdfdb_hip.so (include/dfdb.h).  Every entry point converts
// engine exceptions to status codes; nothing else crosses the ABI.
#include "engine.hpp"
#include "ooc.hpp"
#include <mutex>
#include <cstdio>

using namespace dfdb;

static thread_local std::string g_last_error;
namespace dfdb { void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; } }   // group.cpp reports through the same thread-local text

template <class F>
static int32_t guard(F&& f) noexcept {
  try { f(); return DFDB_OK; }
  catch (const Error& e) { g_last_error = e.what(); return e.code; }
  catch (const std::bad_alloc&) { g_last_error = "out of host memory"; return DFDB_ERR_NOMEM; }
  catch (const std::exception& e) { g_last_error = e.what(); return DFDB_ERR_DEVICE; }
  catch (...) { g_last_error = "unknown error"; return DFDB_ERR_DEVICE; }
}
#define NEED(p) do { if (!(p)) fail(DFDB_ERR_ARGUMENT, "null argument: " #p); } while (0)
#define NEEDQ(q) do { NEED(q); if (!(q)->t) fail(DFDB_ERR_ARGUMENT, "the table of this query was closed"); } while (0)
// NEEDQ for an entry point that may have to decode a compressed-only column whole (keep_compressed = 2; table.cpp column_data): such a decode lives until this call returns
#define NEEDQT(q) NEEDQ(q); ::dfdb::TransientScope transient_scope_((q)->t)

namespace dfdb {
static hipEvent_t prof_event(dfdb_ctx* ctx) {
  if (!ctx->prof_pool.empty()) { hipEvent_t e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
// ---- DevPool / RecycleScope (common.hpp)
namespace {
struct PoolEntry { void* p; size_t cls; uint64_t age; };
struct PoolState { std::mutex m; std::vector<PoolEntry> free[64]; size_t bytes[64] = {}; uint64_t clock = 0; };
PoolState& pool_state() { static PoolState* s = new PoolState(); return *s; }    // (never destroyed: buffers may be released during static destruction)
int pool_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return (d >= 0 && d < 64) ? d : 0; }
thread_local bool tls_recycle = false;
}  // namespace
size_t DevPool::size_class(size_t n) {
  if (n <= 4096) return 4096;
  if (n > ((size_t)1 << 30)) return (n + 255) / 256 * 256;                   // columns and other giants: their own size, and (being no class) back to the driver when freed
  int k = 63 - __builtin_clzll((unsigned long long)(n - 1));                 // 2^k <= n - 1 < 2^(k+1)
  const size_t step = (size_t)1 << (k - 3);                                  // eight classes per octave
  return (n + step - 1) / step * step;
}
void* DevPool::take(size_t cls) {
  PoolState& S = pool_state();
  const int d = pool_device();
  std::lock_guard<std::mutex> g(S.m);
  auto& v = S.free[d];
  for (size_t i = v.size(); i-- > 0;)
    if (v[i].cls == cls) { void* p = v[i].p; v.erase(v.begin() + (long)i); S.bytes[d] -= cls; return p; }
  return nullptr;
}
void DevPool::give(void* p, size_t cls) {
  if (cls > ((size_t)1 << 30)) { (void)hipFree(p); return; }
  // DFDB_POOL_POISON=1 (tests): a buffer comes back filled with 0xA5 — whoever reads memory it has not written gets garbage every time, not once in a while
  static const bool poison = [] { const char* e = getenv("DFDB_POOL_POISON"); return e && e[0] == '1'; }();
  if (poison) { (void)hipDeviceSynchronize(); (void)hipMemset(p, 0xA5, cls); (void)hipDeviceSynchronize(); }      // (the fill has landed before anyone can take the buffer)
  PoolState& S = pool_state();
  // filed under the device that OWNS the buffer, not the one that happens to be current on the calling thread: a thread that holds contexts on two GPUs may
  // free a device-1 query while device 0 is current, and the buffer must not be handed to a device-0 allocation later
  int d = pool_device();
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) == hipSuccess) { if (at.device >= 0 && at.device < 64) d = at.device; } else (void)hipGetLastError();
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(S.m);
    auto& v = S.free[d];
    v.push_back({p, cls, ++S.clock}); S.bytes[d] += cls;
    while ((S.bytes[d] > kPoolBytes || v.size() > 512) && !v.empty()) { drop.push_back(v.front().p); S.bytes[d] -= v.front().cls; v.erase(v.begin()); }
  }
  for (void* q : drop) (void)hipFree(q);
}
void DevPool::flush() {
  PoolState& S = pool_state();
  const int d = pool_device();
  std::vector<void*> drop;
  { std::lock_guard<std::mutex> g(S.m); for (auto& e : S.free[d]) drop.push_back(e.p); S.free[d].clear(); S.bytes[d] = 0; }
  for (void* q : drop) (void)hipFree(q);
}
RecycleScope::RecycleScope() : prev(tls_recycle) { tls_recycle = true; }
RecycleScope::~RecycleScope() { tls_recycle = prev; }
bool RecycleScope::active() {
  static const bool enabled = [] { const char* e = getenv("DFDB_POOL"); return !(e && e[0] == '0'); }();      // DFDB_POOL=0: every buffer back to the driver at once
  return enabled && tls_recycle;
}

LaunchTimer::LaunchTimer(dfdb_ctx* c, const char* n, hipStream_t on) : ctx(c), name(n), stream(on ? on : c->stream) {
  if (ctx->profiling) { e0 = prof_event(ctx); (void)hipEventRecord(e0, stream); }
}
LaunchTimer::~LaunchTimer() {
  if (!e0) return;
  hipEvent_t e1 = prof_event(ctx);
  (void)hipEventRecord(e1, stream);
  ctx->prof_pending.push_back({name, e0, e1});
  if (ctx->prof_pending.size() >= 8192) profile_resolve(ctx);
}
void profile_resolve(dfdb_ctx* ctx) {
  for (auto& p : ctx->prof_pending) {
    (void)hipEventSynchronize(p.e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, p.e0, p.e1);
    auto& e = ctx->prof[p.name]; e.launches++; e.ms += ms;
    ctx->prof_pool.push_back(p.e0); ctx->prof_pool.push_back(p.e1);
  }
  ctx->prof_pending.clear();
}
void stream_wait(dfdb_ctx* ctx) {
  HIP_CHECK(hipEventRecord(ctx->sync_ev, ctx->stream));
  for (;;) {
    const hipError_t e = hipEventQuery(ctx->sync_ev);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) fail(DFDB_ERR_DEVICE, "hipEventQuery failed: %s", hipGetErrorString(e));
    __builtin_ia32_pause();
  }
}
int64_t ctx_option(const dfdb_ctx* ctx, const char* key, int64_t dflt) {
  auto it = ctx->options.find(key);
  return it == ctx->options.end() ? dflt : it->second;
}
}  // namespace dfdb

extern "C" {

int32_t dfdb_version(void) { return DFDB_ABI_VERSION; }
int32_t dfdb_shutdown(void) { return guard([&] { jit_shutdown(); }); }
int32_t dfdb_jit_cache_dir(char* buf, size_t cap) {
  return guard([&] {
    NEED(buf);
    if (!cap) fail(DFDB_ERR_ARGUMENT, "ArgumentError: dfdb_jit_cache_dir needs a buffer");
    std::string why;
    const std::string d = jit_cache_dir(&why);
    snprintf(buf, cap, "%s", d.c_str());
    if (d.empty()) set_last_error(("the run-time kernel cache on disk is off: " + why).c_str());   // (status stays 0: no cache is a state, not a failure)
  });
}
int32_t dfdb_device_count(int32_t* n) {
  return guard([&] { NEED(n); int nd = 0; if (hipGetDeviceCount(&nd) != hipSuccess) { (void)hipGetLastError(); nd = 0; } *n = nd; });
}

int32_t dfdb_last_error(char* buf, size_t cap) {
  if (buf && cap) { snprintf(buf, cap, "%s", g_last_error.c_str()); }
  return (int32_t)g_last_error.size();
}

// ------------------------------------------------------------------ context
int32_t dfdb_ctx_create(int32_t device_id, void* hip_stream, dfdb_ctx** out) {
  return guard([&] {
    NEED(out);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) fail(DFDB_ERR_DEVICE, "no HIP device visible (%s): libdfdb_hip has no CPU fallback", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) fail(DFDB_ERR_ARGUMENT, "device %d out of range (have %d)", device_id, ndev);
    auto c = std::make_unique<dfdb_ctx>();
    c->device = device_id;
    HIP_CHECK(hipSetDevice(device_id));
    HIP_CHECK(hipGetDeviceProperties(&c->prop, device_id));
    if (std::string(c->prop.gcnArchName).rfind("gfx950", 0) != 0)
      fail(DFDB_ERR_DEVICE, "libdfdb_hip is built for gfx950 (MI355X); device %d is %s", device_id, c->prop.gcnArchName);
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    else { HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    HIP_CHECK(hipEventCreate(&c->ev0)); HIP_CHECK(hipEventCreate(&c->ev1));
    HIP_CHECK(hipEventCreateWithFlags(&c->sync_ev, hipEventDisableTiming));
    HIP_CHECK(hipHostMalloc((void**)&c->pinned_scalar, 64, hipHostMallocDefault));
    *out = c.release();
  });
}
int32_t dfdb_ctx_destroy(dfdb_ctx* ctx) {
  return guard([&] {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    stream_drop_parked(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(ctx->ev0); (void)hipEventDestroy(ctx->ev1);
    profile_resolve(ctx);
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
    if (ctx->pinned_scalar) (void)hipHostFree(ctx->pinned_scalar);
    for (int i = 0; i < 2; i++) { if (ctx->pin_ring[i]) (void)hipHostFree(ctx->pin_ring[i]); if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
  });
}
int32_t dfdb_table_build_dictionary(dfdb_table* t, int32_t ordinal, int64_t max_entries, int64_t* entries) {
  return guard([&] { NEED(t); const int64_t n = table_build_dictionary(t, ordinal, max_entries); if (entries) *entries = n; });
}
int32_t dfdb_stream_open(dfdb_query* q, int64_t chunk_blocks, dfdb_stream** out) { return guard([&] { NEEDQT(q); NEED(out); stream_open(q, chunk_blocks, out); }); }
int32_t dfdb_stream_next(dfdb_stream* s, dfdb_query** chunk, int64_t* chunk_rows, int64_t* first_row) {
  return guard([&] { NEED(s); NEED(chunk); *chunk = nullptr; *chunk = stream_next(s, chunk_rows, first_row); });
}
int32_t dfdb_stream_stats(dfdb_stream* s, dfdb_sizestats* stats) { return guard([&] { NEED(s); NEED(stats); stream_stats(s, stats); }); }
int32_t dfdb_stream_read_stats(dfdb_stream* s, int32_t ordinal, dfdb_sizestats* stats) { return guard([&] { NEED(s); NEED(stats); stream_read_stats(s, ordinal, stats); }); }
int32_t dfdb_stream_close(dfdb_stream* s) { return guard([&] { stream_close(s); }); }

int32_t dfdb_ctx_synchronize(dfdb_ctx* ctx) { return guard([&] { NEED(ctx); HIP_CHECK(hipStreamSynchronize(ctx->stream)); }); }
int32_t dfdb_ctx_device_info(dfdb_ctx* ctx, dfdb_device_info* out) {
  return guard([&] {
    NEED(ctx); NEED(out);
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    out->compute_units = ctx->prop.multiProcessorCount;
    out->wavefront_size = ctx->prop.warpSize;
    out->hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    // The engine only runs on gfx950 (dfdb_ctx_create refuses anything else), and every gfx950 part it targets is an MI350-series OAM with
    // eight HBM3E stacks: 8192 bits x 8 Gb/s per pin = 8.0 TB/s (MI355X_MICROARCH.md).  hipDeviceProp's memoryClockRate x memoryBusWidth gives
    // half of that on this driver (it reports the command clock, and HBM3E moves four bits per pin per such cycle, not two), so the
    // formula is only the fallback for a part whose properties say more than the table does.
    const double by_props = 2.0 * (double)ctx->prop.memoryClockRate * 1e3 * ((double)ctx->prop.memoryBusWidth / 8.0) / 1e9;
    out->peak_hbm_gbps = by_props > 8000.0 ? by_props : 8000.0;
  });
}
int32_t dfdb_ctx_set_option(dfdb_ctx* ctx, const char* key, int64_t value) { return guard([&] { NEED(ctx); NEED(key); ctx->options[key] = value; }); }
int32_t dfdb_ctx_timer_start(dfdb_ctx* ctx) { return guard([&] { NEED(ctx); HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream)); }); }
int32_t dfdb_ctx_timer_stop(dfdb_ctx* ctx, double* elapsed_ms) {
  return guard([&] {
    NEED(ctx); NEED(elapsed_ms);
    HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = ms;
  });
}
int32_t dfdb_ctx_profile_enable(dfdb_ctx* ctx, int32_t on) { return guard([&] { NEED(ctx); profile_resolve(ctx); ctx->profiling = on != 0; if (on) ctx->prof.clear(); }); }
int32_t dfdb_ctx_profile_get(dfdb_ctx* ctx, const char* kernel, int64_t* launches, double* total_ms) {
  return guard([&] {
    NEED(ctx); NEED(kernel);
    if (!strncmp(kernel, "jit.", 4)) {               // the run-time compiler's process-wide counters: shapes compiled by hipRTC, read from the disk cache, failed
      int64_t c = 0, f = 0, p = 0, d = 0;
      jit_stats(&c, &f, &p, &d);
      const std::string k = kernel + 4;
      if (launches) *launches = k == "compiled" ? c : k == "from_disk" ? d : k == "failed" ? f : k == "pending" ? p : 0;
      if (total_ms) *total_ms = 0.0;
      return;
    }
    profile_resolve(ctx);
    auto it = ctx->prof.find(kernel);
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
  });
}

// ------------------------------------------------------------------ tables
int32_t dfdb_table_open(dfdb_ctx* ctx, const char* path, dfdb_table** out) { return guard([&] { NEED(ctx); NEED(path); NEED(out); table_open(ctx, path, out); }); }
int32_t dfdb_table_new(dfdb_ctx* ctx, int64_t block_size, dfdb_table** out) {
  return guard([&] {
    NEED(ctx); NEED(out);
    auto t = std::make_unique<dfdb_table>();
    t->ctx = ctx; t->block_size = block_size > 0 ? block_size : 65536;
    *out = t.release();
  });
}
int32_t dfdb_table_close(dfdb_table* t) {
  return guard([&] {
    if (!t) return;
    (void)hipStreamSynchronize(t->ctx->stream);
    for (dfdb_query* q : t->queries) q->t = nullptr;   // orphan live queries: they may be freed later, never used
    delete t;
  });
}
int32_t dfdb_table_ncols(dfdb_table* t, int32_t* n) { return guard([&] { NEED(t); NEED(n); *n = (int32_t)t->cols.size(); }); }
int32_t dfdb_table_nrows(dfdb_table* t, int64_t* n) { return guard([&] { NEED(t); NEED(n); *n = t->nrows < 0 ? 0 : t->nrows; }); }
int32_t dfdb_table_block_size(dfdb_table* t, int64_t* bs) { return guard([&] { NEED(t); NEED(bs); *bs = t->block_size; }); }
int32_t dfdb_table_colinfo(dfdb_table* t, int32_t ordinal, dfdb_colinfo* out) {
  return guard([&] {
    NEED(t); NEED(out);
    if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: column ordinal %d", ordinal);
    const Column& c = t->cols[(size_t)ordinal];
    memset(out, 0, sizeof *out);
    out->id = c.id; snprintf(out->name, sizeof out->name, "%s", c.name.c_str()); out->dtype = c.dtype; out->resident = c.resident ? 1 : 0;
    snprintf(out->logical, sizeof out->logical, "%s", c.logical.c_str());
  });
}
int32_t dfdb_table_find_column(dfdb_table* t, const char* name, int32_t* ordinal) {
  return guard([&] {
    NEED(t); NEED(name); NEED(ordinal);
    for (size_t i = 0; i < t->cols.size(); i++) if (t->cols[i].name == name) { *ordinal = (int32_t)i; return; }
    fail(DFDB_ERR_KEY, "KeyError: key :%s not found", name);   // getmeta: table.jl:52-56
  });
}
int32_t dfdb_table_load(dfdb_table* t, const int32_t* ordinals, int32_t ncols, int64_t block_first, int64_t block_last, dfdb_sizestats* stats) {
  return guard([&] { NEED(t); table_load(t, ordinals, ncols, block_first, block_last, stats); });
}
int32_t dfdb_table_load_image(dfdb_table* t, int32_t ordinal, const uint8_t* image, size_t nbytes, int64_t block_first, int64_t block_last,
                              dfdb_sizestats* stats) {
  return guard([&] { NEED(t); NEED(image); table_load_image(t, ordinal, image, nbytes, block_first, block_last, stats); });
}
int32_t dfdb_table_add_column(dfdb_table* t, const char* name, int32_t dtype, int64_t nrows, const void* data, const uint8_t* bytes,
                              int64_t nbytes, const uint8_t* missing) {
  return guard([&] { NEED(t); NEED(name); if (nrows > 0) NEED(data); table_add_column(t, name, dtype, nrows, data, bytes, nbytes, missing); });
}
int32_t dfdb_table_add_generated(dfdb_table* t, const char* name, int32_t generator, uint64_t seed, int64_t row_first, int64_t nrows) {
  return guard([&] { NEED(t); NEED(name); table_add_generated(t, name, generator, seed, row_first, nrows); });
}
int32_t dfdb_table_add_from_query(dfdb_table* dst, const char* name, dfdb_query* q, int32_t proj_col) {
  return guard([&] { NEED(dst); NEED(name); NEEDQT(q); table_add_from_query(dst, name, q, proj_col); });
}
int32_t dfdb_table_save(dfdb_table* t, const char* path, dfdb_sizestats* stats) { return guard([&] { NEED(t); NEED(path); TransientScope ts(t); table_save(t, path, stats); }); }
int32_t dfdb_table_save_column(dfdb_table* t, int32_t ordinal, const char* file, dfdb_sizestats* stats) {
  return guard([&] { NEED(t); NEED(file); TransientScope ts(t); table_save_column(t, ordinal, file, stats); });
}
int32_t dfdb_table_compress_column(dfdb_table* t, int32_t ordinal, int32_t mode, dfdb_sizestats* stats) {
  return guard([&] { NEED(t); HIP_CHECK(hipSetDevice(t->ctx->device)); table_compress_column(t, ordinal, mode, stats); });
}
int32_t dfdb_table_resident_bytes(dfdb_table* t, int32_t ordinal, int64_t* decoded, int64_t* compressed) {
  return guard([&] { NEED(t); table_resident_bytes(t, ordinal, decoded, compressed); });
}
int32_t dfdb_table_column_stats(dfdb_table* t, int32_t ordinal, dfdb_sizestats* stats) { return guard([&] { NEED(t); NEED(stats); table_column_stats(t, ordinal, stats); }); }
int32_t dfdb_table_decode_resident(dfdb_table* t, int32_t ordinal) { return guard([&] { NEED(t); table_decode_resident(t, ordinal); }); }
int32_t dfdb_table_decode_status(dfdb_table* t, int32_t ordinal, int64_t* bad_blocks) {
  return guard([&] {
    NEED(t);
    const int64_t bad = table_decode_status(t, ordinal);
    if (bad_blocks) *bad_blocks = bad;
    else if (bad > 0) fail(DFDB_ERR_FORMAT, "decompression error: %lld resident block(s) did not decode to their stored size", (long long)bad);
  });
}
int32_t dfdb_table_read_probe(dfdb_table* t, int32_t ordinal, int32_t repeats, double* best_ms, double* avg_ms) {
  return guard([&] {
    NEED(t);
    if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: column ordinal %d", ordinal);
    const Column& c = t->cols[(size_t)ordinal];
    if (!c.resident || dt_width(c.dtype) != 8 || dt_base(c.dtype) == DFDB_STRING) fail(DFDB_ERR_ARGUMENT, "ArgumentError: the read probe takes a resident 8-byte column");
    if (c.comp_only || !c.data.p) fail(DFDB_ERR_ARGUMENT, "ArgumentError: column %s is compressed-only (keep_compressed = 2): there is no decoded array to probe", c.name.c_str());
    dfdb_ctx* ctx = t->ctx;
    HIP_CHECK(hipSetDevice(ctx->device));
    if (repeats < 1) repeats = 1;
    if (repeats > 64) repeats = 64;
    DevBuf sink; sink.ensure(256);
    struct Events { std::vector<hipEvent_t> v; ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } evs;   // (released on every path: HIP_CHECK throws)
    evs.v.assign((size_t)repeats + 1, nullptr);
    std::vector<hipEvent_t>& ev = evs.v;
    for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
    launch_read_probe(ctx->stream, c.data.p, c.nrows, sink.as<uint64_t>());      // (untimed: the first launch of a kernel loads its code object)
    HIP_CHECK(hipEventRecord(ev[0], ctx->stream));
    for (int r = 0; r < repeats; r++) {
      LaunchTimer lt(ctx, "read_probe");
      launch_read_probe(ctx->stream, c.data.p, c.nrows, sink.as<uint64_t>());
      HIP_CHECK(hipEventRecord(ev[(size_t)r + 1], ctx->stream));
    }
    HIP_CHECK(hipEventSynchronize(ev[(size_t)repeats]));
    double best = 0, sum = 0;
    for (int r = 0; r < repeats; r++) {
      float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ev[(size_t)r], ev[(size_t)r + 1]));
      sum += ms; if (r == 0 || ms < best) best = ms;
    }
    if (best_ms) *best_ms = best;
    if (avg_ms) *avg_ms = sum / repeats;
  });
}
int32_t dfdb_table_set_logical_type(dfdb_table* t, int32_t ordinal, const char* logical) {
  return guard([&] {
    NEED(t); NEED(logical);
    if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: column ordinal %d", ordinal);
    Column& c = t->cols[(size_t)ordinal];
    std::string lg;
    const int32_t want = *logical ? dt_parse_ex(logical, &lg) : c.dtype;
    if (*logical && (lg.empty() || dt_base(want) != dt_base(c.dtype))) fail(DFDB_ERR_ARGUMENT, "ArgumentError: %s is not the representation of a %s column", dt_name(c.dtype).c_str(), logical);
    c.logical = lg;
  });
}
int32_t dfdb_table_set_row_base(dfdb_table* t, int64_t row_base) { return guard([&] { NEED(t); t->row_base = row_base; }); }

// ------------------------------------------------------------------ queries
int32_t dfdb_query_new(dfdb_table* t, dfdb_query** out) {
  return guard([&] {
    NEED(t); NEED(out);
    auto q = std::make_unique<dfdb_query>();
    q->t = t;
    for (size_t i = 0; i < t->cols.size(); i++) {   // full_table_projection: view.jl:43-48
      auto n = std::make_unique<Node>(); n->op = DFIR_COL; n->col = (int)i; n->dtype = t->cols[i].dtype;
      q->proj.push_back(ProjCol{t->cols[i].name, std::move(n)});
    }
    t->queries.push_back(q.get());
    *out = q.release();
  });
}
int32_t dfdb_query_free(dfdb_query* q) {
  return guard([&] {
    if (!q) return;
    if (q->stream_owned) fail(DFDB_ERR_ARGUMENT, "ArgumentError: a chunk query belongs to its stream (it dies at the next dfdb_stream_next / dfdb_stream_close)");
    if (q->t) {
      (void)hipStreamSynchronize(q->t->ctx->stream);
      query_return_mask(q);
      auto& v = q->t->queries;
      for (size_t i = 0; i < v.size(); i++) if (v[i] == q) { v[i] = v.back(); v.pop_back(); break; }
    } else (void)hipDeviceSynchronize();
    RecycleScope rs;                                   // (the stream is drained: the query's buffers go to the pool, not through hipFree)
    delete q;
  });
}
int32_t dfdb_query_add_range(dfdb_query* q, int64_t start, int64_t step, int64_t stop) {
  return guard([&] { NEEDQ(q); ooc_reset(q); Stage s; s.kind = ST_RANGE; s.start = start; s.step = step; s.stop = stop; query_add_stage(q, std::move(s)); });
}
int32_t dfdb_query_add_indices(dfdb_query* q, const int64_t* idx, int64_t n) {
  return guard([&] {
    NEEDQ(q); if (n > 0) NEED(idx);
    ooc_reset(q);
    if (n < 0) fail(DFDB_ERR_ARGUMENT, "negative index count");
    Stage s; s.kind = ST_INDICES; s.idx.assign(idx, idx + n);
    query_add_stage(q, std::move(s));
  });
}
int32_t dfdb_query_add_integer(dfdb_query* q, int64_t i) {
  return guard([&] { NEEDQ(q); ooc_reset(q); Stage s; s.kind = ST_INTEGER; s.idx = {i}; query_add_stage(q, std::move(s)); });
}
int32_t dfdb_query_add_predicate(dfdb_query* q, const uint8_t* ir, size_t len) {
  return guard([&] {
    NEEDQ(q); NEED(ir);
    ooc_reset(q);
    Stage s; s.kind = ST_PRED; s.pred = parse_ir(*q->t, ir, len);
    if (s.pred->dtype != DFDB_BOOL) fail(DFDB_ERR_ARGUMENT, "ArgumentError: Function for selection must have Bool result type");   // selection.jl:52-55
    query_add_stage(q, std::move(s));
  });
}
int32_t dfdb_query_nstages(dfdb_query* q, int32_t* n) { return guard([&] { NEEDQ(q); NEED(n); *n = (int32_t)q->stages.size(); }); }
int32_t dfdb_query_set_projection(dfdb_query* q, int32_t n, const char* const* names, const uint8_t* const* irs, const size_t* lens) {
  return guard([&] {
    NEEDQ(q); if (n > 0) { NEED(names); NEED(irs); NEED(lens); }
    std::vector<ProjCol> np;
    for (int32_t i = 0; i < n; i++) {
      for (int32_t j = 0; j < i; j++) if (std::string(names[j]) == names[i]) fail(DFDB_ERR_ARGUMENT, "ArgumentError: Duplicated column %s", names[i]);   // projection.jl:25-28
      np.push_back(ProjCol{names[i], parse_ir(*q->t, irs[i], lens[i])});
    }
    q->proj = std::move(np);
    ooc_reset(q);
  });
}
int32_t dfdb_query_ncols(dfdb_query* q, int32_t* n) { return guard([&] { NEEDQ(q); NEED(n); *n = (int32_t)q->proj.size(); }); }
int32_t dfdb_query_coltype(dfdb_query* q, int32_t i, int32_t* dtype) {
  return guard([&] {
    NEEDQ(q); NEED(dtype);
    if (i < 0 || (size_t)i >= q->proj.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: projection column %d", i);
    *dtype = q->proj[(size_t)i].expr->dtype;
  });
}
int32_t dfdb_expr_result_type(dfdb_table* t, const uint8_t* ir, size_t len, int32_t* dtype) {
  return guard([&] { NEED(t); NEED(ir); NEED(dtype); *dtype = parse_ir(*t, ir, len)->dtype; });
}
int32_t dfdb_query_set_stage_base(dfdb_query* q, int32_t stage, int64_t survivors_before) {
  return guard([&] {
    NEEDQ(q);
    if (stage < 0 || (size_t)stage >= q->stages.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: stage %d", stage);
    q->stages[(size_t)stage].stage_base = survivors_before; q->executed_stages = -1; q->count = -1; ooc_reset(q);
  });
}
int32_t dfdb_query_count_prefix(dfdb_query* q, int32_t nstages, int64_t* n) { return guard([&] { NEEDQT(q); NEED(n); *n = query_count(q, nstages); q->executed_stages = -1; }); }

// ------------------------------------------------------------------ execution
// (out of core there is nothing to leave in HBM: the consumers stream when they are asked)
int32_t dfdb_query_execute(dfdb_query* q) { return guard([&] { NEEDQT(q); if (query_out_of_core(q)) { ooc_reset(q); return; } query_execute(q, -1); }); }
int32_t dfdb_query_unique(dfdb_query* q, int32_t proj_col) { return guard([&] { NEEDQT(q); if (query_out_of_core(q)) { ooc_unique(q, proj_col); return; } query_unique(q, proj_col); }); }
int32_t dfdb_query_groupreduce(dfdb_query* q, int32_t key_col, int32_t val_col, int32_t stat, int64_t* ngroups, int64_t* key_string_bytes) {
  return guard([&] { NEEDQT(q); if (query_out_of_core(q)) { ooc_groupreduce(q, key_col, val_col, stat, ngroups, key_string_bytes); return; } query_groupreduce(q, key_col, val_col, stat, ngroups, key_string_bytes); });
}
int32_t dfdb_query_groupreduce_fetch(dfdb_query* q, dfdb_outcol* keys, int64_t* counts, int64_t* values_i, double* values_f) {
  return guard([&] { NEEDQT(q); if (q->ooc && q->ooc->gr_pending) { ooc_groupreduce_fetch(q, keys, counts, values_i, values_f); return; } query_groupreduce_fetch(q, keys, counts, values_i, values_f); });
}
int32_t dfdb_query_hint_aggregate(dfdb_query* q, int32_t op, int32_t proj_col) {
  return guard([&] { NEEDQ(q); q->hint_agg_op = op; q->hint_agg_proj = proj_col; });   // (affects the NEXT execution only: an executed query keeps its results)
}
int32_t dfdb_query_hint_materialize(dfdb_query* q, int32_t on) {
  return guard([&] { NEEDQ(q); if (q->hint_materialize != (on != 0)) { q->hint_materialize = on != 0; q->executed_stages = -1; q->count = -1; q->prefix_valid = false; } });
}
int32_t dfdb_query_reset(dfdb_query* q) { return guard([&] { NEEDQ(q); ooc_reset(q); q->executed_stages = -1; q->count = -1; q->prefix_valid = false; q->gr_state = 0; }); }
int32_t dfdb_count(dfdb_query* q, int64_t* n) { return guard([&] { NEEDQT(q); NEED(n); *n = query_out_of_core(q) ? ooc_count(q) : query_count(q, -1); }); }
int32_t dfdb_count_to(dfdb_query* q, int64_t* out, int32_t memkind) {
  return guard([&] {
    NEEDQT(q); NEED(out);
    if (query_out_of_core(q)) {
      const int64_t n = ooc_count(q);
      if (memkind != DFDB_MEM_DEVICE) { *out = n; return; }
      HIP_CHECK(hipMemcpyAsync(out, &n, 8, hipMemcpyHostToDevice, q->t->ctx->stream)); HIP_CHECK(hipStreamSynchronize(q->t->ctx->stream));
      return;
    }
    if (memkind != DFDB_MEM_DEVICE) { *out = query_count(q, -1); return; }
    if (q->executed_stages != (int)q->stages.size() || q->bitmap_rows != q->t->nrows) query_execute(q, -1);
    const int64_t ntiles = ceil_div(q->t->nrows, kTileRows);
    HIP_CHECK(hipMemcpyAsync(out, q->prefix.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToDevice, q->t->ctx->stream));
  });
}
int32_t dfdb_select_bitmap(dfdb_query* q, uint64_t* out, int32_t memkind) {
  return guard([&] {
    NEEDQT(q); NEED(out);
    if (query_out_of_core(q)) fail(DFDB_ERR_UNSUPPORTED, "the selection bitmap of a view whose columns are not resident exists one chunk at a time: take it from the chunks of dfdb_stream_next");
    query_select_bitmap(q, out, memkind);
  });
}
int32_t dfdb_select_indices(dfdb_query* q, int64_t* out, int64_t cap, int32_t memkind, int64_t* n) {
  return guard([&] { NEEDQT(q); if (cap > 0) NEED(out); if (query_out_of_core(q)) { ooc_select_indices(q, out, cap, memkind, n); return; } query_select_indices(q, out, cap, memkind, n); });
}
int32_t dfdb_result_string_bytes(dfdb_query* q, int32_t i, int64_t* nbytes) { return guard([&] { NEEDQT(q); NEED(nbytes); *nbytes = query_out_of_core(q) ? ooc_string_bytes(q, i) : query_string_bytes(q, i); }); }
int32_t dfdb_materialize(dfdb_query* q, dfdb_outcol* outs, int32_t ncols) { return guard([&] { NEEDQT(q); if (ncols > 0) NEED(outs); if (query_out_of_core(q)) { ooc_materialize(q, outs, ncols); return; } query_materialize(q, outs, ncols); }); }
int32_t dfdb_aggregate(dfdb_query* q, int32_t op, int32_t i, int64_t* out_i, double* out_f) { return guard([&] { NEEDQT(q); if (query_out_of_core(q)) { ooc_aggregate(q, op, i, out_i, out_f); return; } query_aggregate(q, op, i, out_i, out_f); }); }


/* ---- only what the view needs, resident only if it fits (view.jl:183-190, blocksiterator.jl:20-33) ---- */
int32_t dfdb_query_prepare(dfdb_query* q, int32_t* how) {
  return guard([&] { NEEDQ(q); const int32_t h = query_prepare(q); if (how) *how = h; });
}
int32_t dfdb_query_read_stats(dfdb_query* q, dfdb_sizestats* stats) {
  return guard([&] { NEEDQ(q); NEED(stats); *stats = q->ooc ? q->ooc->read : dfdb_sizestats{0, 0, 0}; });
}
int32_t dfdb_table_unload(dfdb_table* t, const int32_t* ordinals, int32_t ncols) {
  return guard([&] {
    NEED(t); if (ncols > 0) NEED(ordinals);
    HIP_CHECK(hipSetDevice(t->ctx->device));
    HIP_CHECK(hipStreamSynchronize(t->ctx->stream));
    for (dfdb_query* q : t->queries) { query_return_mask(q); q->executed_stages = -1; q->count = -1; q->prefix_valid = false; q->gr_state = 0; q->arenas.clear(); ooc_reset(q); }
    std::vector<int32_t> all;
    if (!ordinals) { for (size_t i = 0; i < t->cols.size(); i++) all.push_back((int32_t)i); ordinals = all.data(); ncols = (int32_t)all.size(); }
    for (int32_t k = 0; k < ncols; k++) {
      const int32_t o = ordinals[k];
      if (o < 0 || (size_t)o >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", o);
      Column& c = t->cols[(size_t)o];
      if (c.file.empty()) fail(DFDB_ERR_ARGUMENT, "ArgumentError: column %s has no backing file: unloading it would lose it", c.name.c_str());
      c.data.release(); c.bytes.release(); c.tile_off.release(); c.missing.release(); c.comp.release(); c.comp_blocks.release(); c.comp_status.release(); c.comp_index.release();
      c.mask_pref.release(); c.mask_calibrated = false; c.mask_lent = false;
      c.dict_codes.release(); c.dict_len.release(); c.dict_off.release(); c.dict_bytes.release(); c.dict_host.clear(); c.dict_n = 0;
      c.comp_blocks_host.clear(); c.comp_nblocks = 0; c.comp_index_state = 0; c.comp_only = false; c.transient = false; c.resident = false; c.nrows = 0; c.nbytes = 0;
    }
    bool any = false;
    for (const Column& c : t->cols) any = any || c.resident;
    if (!any) { t->nrows = -1; t->block_first = 0; }
  });
}
}  // extern "C"

namespace dfdb {
// a second context on the same device with its own stream and the same options (stream.cpp: loader / consumer slots)
int32_t ctx_create_like(const dfdb_ctx* like, dfdb_ctx** out) {
  const int32_t rc = dfdb_ctx_create(like->device, nullptr, out);
  if (rc == 0) (*out)->options = like->options;
  return rc;
}
void ctx_destroy(dfdb_ctx* c) { (void)dfdb_ctx_destroy(c); }
// "0-63,128-191" -> cpu_set_t
static bool parse_cpulist(const std::string& txt, cpu_set_t* out) {
  CPU_ZERO(out);
  bool any = false;
  size_t i = 0;
  while (i < txt.size()) {
    while (i < txt.size() && !isdigit((unsigned char)txt[i])) i++;
    if (i >= txt.size()) break;
    long a = 0; while (i < txt.size() && isdigit((unsigned char)txt[i])) a = a * 10 + (txt[i++] - '0');
    long b = a;
    if (i < txt.size() && txt[i] == '-') { i++; b = 0; while (i < txt.size() && isdigit((unsigned char)txt[i])) b = b * 10 + (txt[i++] - '0'); }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, out); any = true; }
  }
  return any;
}
static std::string slurp_text(const std::string& path) {
  std::string r;
  if (FILE* f = fopen(path.c_str(), "r")) { char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) r.append(buf, n); fclose(f); }
  return r;
}
// (returns the state instead of writing it: NodeBind publishes it last, under its lock — the loaders of a stream all bind at once, and one that read a
// half-made answer stayed unbound for its whole life: its preads crossed the socket interconnect, 1.3-1.5 ms per 64-MB piece where a bound thread takes 0.8-0.9)
static int lookup_node(dfdb_ctx* ctx) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, ctx->device) != hipSuccess) { (void)hipGetLastError(); return -1; }
  std::string id(bus);
  for (char& ch : id) ch = (char)tolower((unsigned char)ch);
  const std::string nn = slurp_text("/sys/bus/pci/devices/" + id + "/numa_node");
  if (nn.empty()) return -1;
  const int node = atoi(nn.c_str());
  if (node < 0) return -1;
  if (!parse_cpulist(slurp_text("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist"), &ctx->node_cpus)) return -1;
  return 1;
}
NodeBind::NodeBind(dfdb_ctx* ctx) {
  if (!ctx || ctx_option(ctx, "numa_bind", 1) == 0) return;
  {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (ctx->node_state == 0) ctx->node_state = lookup_node(ctx);
    if (ctx->node_state != 1) { if (getenv("DFDB_STREAM_DEBUG")) fprintf(stderr, "[numa] the device's node is unknown: nothing bound\n"); return; }
  }
  if (sched_getaffinity(0, sizeof old, &old) != 0) return;
  cpu_set_t want;
  CPU_AND(&want, &old, &ctx->node_cpus);
  if (CPU_COUNT(&want) == 0 || CPU_EQUAL(&want, &old)) { if (getenv("DFDB_STREAM_DEBUG")) fprintf(stderr, "[numa] thread on cpu %d: %d of its %d CPUs are on the device's node, nothing to narrow\n", sched_getcpu(), CPU_COUNT(&want), CPU_COUNT(&old)); return; }
  if (sched_setaffinity(0, sizeof want, &want) == 0) active = true;
  if (getenv("DFDB_STREAM_DEBUG")) fprintf(stderr, "[numa] thread on cpu %d: %d of its %d CPUs are on the device's node, bound: %d\n", sched_getcpu(), CPU_COUNT(&want), CPU_COUNT(&old), (int)active);
}
NodeBind::~NodeBind() { if (active) (void)sched_setaffinity(0, sizeof old, &old); }

void ensure_pin_ring(dfdb_ctx* ctx, size_t bytes) {
  if (ctx->pin_ring_cap >= bytes) return;
  for (int i = 0; i < 2; i++) {
    if (ctx->pin_ring[i]) { (void)hipHostFree(ctx->pin_ring[i]); ctx->pin_ring[i] = nullptr; }
    HIP_CHECK(hipHostMalloc((void**)&ctx->pin_ring[i], bytes, hipHostMallocDefault));
    if (!ctx->pin_ev[i]) HIP_CHECK(hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
  }
  ctx->pin_ring_cap = bytes;
}
}  // namespace dfdb
