// common.hpp — shared host-side declarations of libdfdb_hip.so (MI355X / gfx950 only).
#pragma once
#include <sched.h>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/dfdb.h"

namespace dfdb {

// ---- errors: thrown inside the engine, turned into status codes at the C ABI --------------------
struct Error : std::runtime_error {
  int code;
  uint64_t row = ~0ull;   // DivideError / InexactError of a predicate: the GLOBAL 0-based table row that raised it (the shards of a group agree on the lowest)
  Error(int c, const std::string& m, uint64_t r = ~0ull) : std::runtime_error(m), code(c), row(r) {}
};
[[noreturn]] void fail(int code, const char* fmt, ...);

#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) ::dfdb::fail(DFDB_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// ---- dtypes -------------------------------------------------------------------------------------
inline int dt_base(int32_t dt) { return dt & DFDB_DTYPE_MASK; }
inline bool dt_nullable(int32_t dt) { return (dt & DFDB_NULLABLE) != 0; }
inline bool dt_isint(int32_t dt) { int b = dt_base(dt); return b >= DFDB_I8 && b <= DFDB_U64; }
inline bool dt_issigned(int32_t dt) { int b = dt_base(dt); return b >= DFDB_I8 && b <= DFDB_I64; }
inline bool dt_isfloat(int32_t dt) { int b = dt_base(dt); return b == DFDB_F32 || b == DFDB_F64; }
inline bool dt_isnum(int32_t dt) { return dt_isint(dt) || dt_isfloat(dt) || dt_base(dt) == DFDB_BOOL; }
int dt_width(int32_t dt);
std::string dt_name(int32_t dt);
int32_t dt_parse(const std::string& s);
// + the Julia bits types whose blocks are plain integers (Date / DateTime / Time -> Int64, Char -> UInt32): the storage dtype,
// and the type string itself in *logical ("" for an ordinary dtype)
int32_t dt_parse_ex(const std::string& s, std::string* logical);
std::string dt_type_string(int32_t dt, const std::string& logical);

// ---- geometry -----------------------------------------------------------------------------------
// A "tile" is the unit one wavefront scans: 1024 rows = 16 bitmap words = one 128-B line of bitmap.
// A "ctile" (compaction tile) is 4096 rows = 64 bitmap words = one word per lane.
constexpr int64_t kTileRows = 1024;
constexpr int64_t kTileWords = 16;
constexpr int64_t kCTileRows = 4096;
constexpr int64_t kStrTileRows = 1024;  // string byte offsets are kept per 1024 rows

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ---- device buffer ------------------------------------------------------------------------------
// Device buffers nothing in flight can touch any more, kept for the next allocation of their size class instead of going back to the driver (round 4: hipFree
// drains the whole device and a fresh query over a 1e9-row table spent 1-2 ms in hipMalloc / hipFree — a fifth of a groupreduce).  A buffer enters ONLY through
// DevBuf::release() inside a RecycleScope, which an owner opens after it has drained the stream its buffers were used on (dfdb_query_free, the end of unique /
// groupreduce); everything else still goes through hipFree.  Sizes are rounded up to m * 2^k, m = 8 .. 15 (at most 12.5 % over), so a later request of about
// the same size finds the buffer (requests above 1 GB — columns — are neither rounded nor kept); per device at most kPoolBytes stay (oldest out first) and a failed hipMalloc empties the pool and tries again.
// Environment: DFDB_POOL=0 turns the pool off; DFDB_POOL_POISON=1 fills every buffer with 0xA5 (between two device synchronisations) as it enters — the GPU suite
// passes with it, i.e. nothing reads a buffer it has not written.
struct DevPool {
  static constexpr size_t kPoolBytes = 8ull << 30;
  static size_t size_class(size_t n);
  static void* take(size_t cls);              // nullptr: none of that class on the current device
  static void give(void* p, size_t cls);
  static void flush();                        // the current device's buffers back to the driver
};
struct RecycleScope {
  bool prev;
  RecycleScope();
  ~RecycleScope();
  static bool active();
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
  ~DevBuf() { release(); }
  void release() {
    if (p) { if (RecycleScope::active() && DevPool::size_class(bytes) == bytes) DevPool::give(p, bytes); else (void)hipFree(p); }
    p = nullptr; bytes = 0;
  }
  void ensure(size_t n) {  // grow-only
    if (n <= bytes && p) return;
    release();
    if (n == 0) n = 256;
    const size_t cls = DevPool::size_class(n);
    p = DevPool::take(cls);
    if (!p) {
      hipError_t e = hipMalloc(&p, cls);
      if (e != hipSuccess) { (void)hipGetLastError(); DevPool::flush(); e = hipMalloc(&p, cls); }
      if (e != hipSuccess) { p = nullptr; fail(DFDB_ERR_NOMEM, "hipMalloc(%zu) failed: %s", cls, hipGetErrorString(e)); }
    }
    bytes = cls;
  }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace dfdb

// ---- context ------------------------------------------------------------------------------------
struct dfdb_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // dfdb_ctx_timer_*
  bool profiling = false;
  struct ProfEntry { int64_t launches = 0; double ms = 0; };
  std::map<std::string, ProfEntry> prof;
  // per-launch profiling without host synchronisation: event pairs are recorded on the launch stream and resolved when the
  // numbers are read (dfdb_ctx_profile_get / disable), so a profiled step runs at the speed of an unprofiled one
  struct ProfPending { const char* name; hipEvent_t e0, e1; };
  std::vector<ProfPending> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  hipDeviceProp_t prop{};
  int64_t* pinned_scalar = nullptr;          // 64 B of pinned host memory for small readbacks
  hipEvent_t sync_ev = nullptr;              // stream_wait()
  std::map<std::string, int64_t> options;    // dfdb_ctx_set_option
  dfdb::DevBuf radix_recs;                   // unique by radix partition (k_radix.hip): the 12-byte {key image, row} records' scratch, kept between calls
  dfdb::DevBuf hist;                         // K7 HIST forms: a ticket word + one 64-KB history ring per resident wave (k_decode.hip), made on first use
  int hist_waves = 0;
  // two pinned bounce buffers for dfdb_table_load: the file is read piece by piece into one while the other is in flight to HBM
  uint8_t* pin_ring[2] = {nullptr, nullptr};
  size_t pin_ring_cap = 0;
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  // a closed dfdb_stream parked for the next dfdb_stream_open on this context (slot contexts, pinned buffers, device buffers and loader
  // threads are what opening and closing a stream costs: ~40 ms); `alive` lets a stream that outlives its context notice (stream.cpp)
  void* parked_stream = nullptr;
  std::shared_ptr<int> alive = std::make_shared<int>(0);
  // CPUs of the NUMA node this GPU hangs off (NodeBind below): 0 = not looked up yet, 1 = node_cpus is valid, -1 = unknown / not applicable
  int node_state = 0;
  cpu_set_t node_cpus;
};

namespace dfdb {
// RAII per-launch profiler: when ctx->profiling, brackets a launch with events on the engine stream
struct LaunchTimer {
  dfdb_ctx* ctx; const char* name; hipEvent_t e0 = nullptr; hipStream_t stream;
  LaunchTimer(dfdb_ctx* c, const char* n, hipStream_t on = nullptr);   // on: the stream the launch goes to (default: the engine stream)
  ~LaunchTimer();
};
}  // namespace dfdb

namespace dfdb {
// low-latency wait for everything enqueued on the engine stream: event + spin on hipEventQuery
// (hipStreamSynchronize's blocking wait costs ~0.5 ms per call on this stack; a step must not pay that)
void stream_wait(dfdb_ctx* ctx);
// runtime tuning knobs (dfdb_ctx_set_option)
int64_t ctx_option(const dfdb_ctx* ctx, const char* key, int64_t dflt);
void profile_resolve(dfdb_ctx* ctx);   // fold the pending event pairs into ctx->prof
// while profiling is on: count one launch of a named VARIANT (no time; "family.variant" beside the family's timed entry), so that a caller can see
// which form of a kernel its context's options selected (dfdb_ctx_profile_get)
inline void prof_note(dfdb_ctx* ctx, const char* name) { if (ctx->profiling) ctx->prof[name].launches++; }
constexpr int kCompactStoreDefault = 3;   // K2 form (ctx option "compact_store"): wide, nontemporal 16-byte stores — see k_compact.hip / kernels.hpp
// The host threads that move file bytes (pread into pinned memory, the writer's fills of a file mapping) belong on the CPUs of the NUMA node the GPU
// hangs off: page cache -> pinned buffer -> DMA crosses the socket interconnect otherwise.  Measured on a two-socket box, block-streamed count of
// 2e9 rows: 47 GB/s of file bytes bound to the GPU's node, 40 unbound, 35 bound to the other node.  RAII: narrows the CALLING thread's affinity to
// (its current mask AND the node's CPUs) — threads it starts inherit that, pinned memory it allocates is first touched there — and puts the old
// mask back.  No-op when the node is unknown, the intersection is empty, or ctx option "numa_bind" = 0.
struct NodeBind {
  cpu_set_t old; bool active = false;
  explicit NodeBind(dfdb_ctx* ctx);
  ~NodeBind();
  NodeBind(const NodeBind&) = delete;
  NodeBind& operator=(const NodeBind&) = delete;
};
// the context's two pinned bounce buffers (file <-> HBM pipelines of dfdb_table_load / dfdb_table_save), at least `bytes` each
void ensure_pin_ring(dfdb_ctx* ctx, size_t bytes);
}  // namespace dfdb
