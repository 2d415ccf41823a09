// group.cpp — one table block-range sharded over the GPUs of a node, behind the C ABI (SURVEY.md §8e, §8b backend HIP_N).
//
// The reference is single-process and walks blocks serially (src/io/blocksiterator.jl:98-145); its blocks are independent
// 65 536-row units that every column shares (check_column_head, src/io/filesystem.jl:47-54).  Here rank g of G owns the contiguous
// block range [g*ceil(nb/G), (g+1)*ceil(nb/G)) of EVERY column, evaluates the view over its shard with the ordinary single-GPU
// engine, and the shards meet only in
//   * nrow(v) / length(col)  (src/tables/view.jl:192-206)            -> one RCCL all-reduce of an Int64
//   * sum / minimum / maximum (Base.iterate(::DFColumn), column.jl:102-126) -> one RCCL all-reduce of {value, count}
//   * a range stage AFTER a predicate stage numbers the global survivor stream (selection.jl:94-111, RangeToProcess.offset :68-75)
//     -> all-gather of one Int64 per rank + exclusive scan = the survivors on lower ranks (dfdb_query_set_stage_base)
// No column data ever crosses xGMI; results stay sharded on the devices or are written to the caller's host buffers in rank order
// (= table order).
//
// Two ways to form a group:
//   dfdb_group_create(devices, n)            one process drives n GPUs: a host worker thread per GPU issues that shard's launches,
//                                            ncclCommInitAll connects them (what a Julia session calling the drop-in gets)
//   dfdb_group_create_rank(dev, id, r, G)    one process per GPU (bench.py under torch.distributed.run): ncclCommInitRank with an id
//                                            made by dfdb_group_unique_id on rank 0 and handed round by the launcher
//   dfdb_group_create_rank_callbacks(...)    one process per GPU whose host brings its own collectives (MPI, gloo): DFDB_EXCHANGE_CALLBACK
// RCCL is loaded lazily (dlopen "librccl.so.1"): a single-GPU user never touches it, and a process that already holds RCCL (PyTorch)
// shares that copy.  DFDB_EXCHANGE_HOST does the same exchanges through host memory; it exists for single-process groups whose
// "ranks" share one physical GPU (functional tests on a 1-GPU box: RCCL refuses duplicate devices).
//
// Round 3: (1) a shard whose half of a collective call fails still takes part in the exchange — every exchange carries a fault key (MIN) and
// all ranks raise the same error, the one of the lowest table row (for_shards_deferred / settle_fault); whether an exchange is needed is decided
// by flags every rank changes alike, never by a shard's local state; (2) {value, count} of an aggregate travel in ONE exchange; (3) unique /
// groupreduce over the whole table: per-shard device reduction, packed records all-gathered, merged by key in rank order (group_reduce_all);
// (4) materialize with the result left sharded on the devices.
#include "engine.hpp"
#include "ooc.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace dfdb {
int query_aggregate_device(dfdb_query* q, int32_t op, int32_t i);   // query.cpp

// ---------------------------------------------------------------- RCCL, resolved at run time
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) return;
    auto sym = [&](const char* n) { return dlsym(r.h, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.AllGather || !r.GroupStart || !r.GroupEnd)
    fail(DFDB_ERR_DEVICE, "RCCL (librccl.so.1) could not be loaded: %s", r.h ? "missing symbols" : dlerror());
  return r;
}
#define RCCL_CHECK(expr)                                                                                          \
  do {                                                                                                            \
    ncclResult_t _r = (expr);                                                                                     \
    if (_r != ncclSuccess) ::dfdb::fail(DFDB_ERR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(_r) : "rccl error"); \
  } while (0)

// ncclGroupStart ... ncclGroupEnd around `body`, whatever happens in between.  A collective that fails inside the bracket used to throw with the group OPEN
// (RCCL_CHECK between the two calls): the thread's next collective would then have nested inside the abandoned one.  Here GroupEnd always runs, and a failure
// of anything inside the bracket (or of the bracket's own calls) marks the communicator dead: what the other ranks have enqueued by then is unknowable, so every
// later collective of the group fails at once with ONE clear error instead of hanging or nesting.  `r` is a parameter so that a CPU self-test can drive
// the bracket with a stub table (dfdb_selftest "rccl_bracket"); nothing in it touches a device.
struct CommState { bool dead = false; std::string why; };
template <class F>
static void rccl_bracket(Rccl& r, CommState& cs, F&& body) {
  if (cs.dead) fail(DFDB_ERR_DEVICE, "this group's RCCL communicator failed earlier (%s): destroy the group and create a new one", cs.why.c_str());
  auto name_of = [&](ncclResult_t rc) { return std::string(r.GetErrorString ? r.GetErrorString(rc) : "rccl error"); };
  ncclResult_t rc = r.GroupStart();
  if (rc != ncclSuccess) { cs.dead = true; cs.why = "ncclGroupStart: " + name_of(rc); fail(DFDB_ERR_DEVICE, "ncclGroupStart failed: %s", name_of(rc).c_str()); }
  try { body(); }
  catch (const std::exception& e) { (void)r.GroupEnd(); cs.dead = true; cs.why = e.what(); throw; }
  catch (...) { (void)r.GroupEnd(); cs.dead = true; cs.why = "unknown error inside a collective bracket"; throw; }
  rc = r.GroupEnd();
  if (rc != ncclSuccess) { cs.dead = true; cs.why = "ncclGroupEnd: " + name_of(rc); fail(DFDB_ERR_DEVICE, "ncclGroupEnd failed: %s", name_of(rc).c_str()); }
}
// the same check for a call inside a bracket, against the table the bracket was given
#define RCCL_IN(r, expr)                                                                                          \
  do {                                                                                                            \
    ncclResult_t _r = (expr);                                                                                     \
    if (_r != ncclSuccess) ::dfdb::fail(DFDB_ERR_DEVICE, "%s failed: %s", #expr, (r).GetErrorString ? (r).GetErrorString(_r) : "rccl error"); \
  } while (0)

// ---------------------------------------------------------------- one host thread per local shard
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, quit = false;
  int device = 0;
  void start(int dev) {
    device = dev;
    th = std::thread([this] {
      (void)hipSetDevice(device);
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        auto j = std::move(job); has_job = false;
        lk.unlock(); j(); lk.lock();
        done = true; cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> j) { std::lock_guard<std::mutex> lk(m); job = std::move(j); has_job = true; done = false; cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return done; }); }
  void stop() { { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); } if (th.joinable()) th.join(); }
};
}  // namespace dfdb

using namespace dfdb;

constexpr int kXSlots = 16;   // 8-byte exchange slots per shard before the all-gather area
constexpr int kFaultSlot = 3; // the fault key every exchange carries along (below)
constexpr int kCountPin = 12; // pinned landing place (kFaultSlot + 1 slots, host side only) of a query's own {count, .., fault key} copy

struct dfdb_group {
  int world = 1, first_rank = 0, exchange = DFDB_EXCHANGE_HOST;
  std::vector<dfdb_ctx*> ctx;            // one per local shard; owned
  std::vector<ncclComm_t> comm;          // RCCL only
  std::vector<std::unique_ptr<Worker>> workers;   // single-process groups with more than one shard
  std::vector<DevBuf> xbuf;              // per shard: kXSlots reduce slots + world all-gather slots (device)
  std::vector<int64_t*> xpin;            // the same, pinned host memory
  // A failure of the per-shard half of a collective operation (a DivideError only one shard's rows reach, an OOM, a column that is not
  // resident) must not keep its rank out of the exchange that follows: with one process per GPU the other ranks would wait in the
  // all-reduce for ever.  The failure is remembered here instead, the rank takes part in the exchange, and every exchange carries one
  // extra 8-byte slot reduced with MIN — the fault key, ~0 = none, else ((global row + 1) << 8 | status) for an error that knows its row
  // and the bare status otherwise — so that all ranks learn of the failure at the same point and raise the SAME error: the one of the
  // lowest table row, which is the one the reference's serial block iteration would have met first.
  int fault_code = 0; std::string fault_msg; uint64_t fault_key = ~0ull;
  CommState comm_state;                  // RCCL: dead once anything failed inside a collective bracket (rccl_bracket)
  dfdb_exchange_fns fns{nullptr, nullptr, nullptr};   // DFDB_EXCHANGE_CALLBACK: the caller's collectives (host memory, blocking)
  int nlocal() const { return (int)ctx.size(); }
};
struct dfdb_gtable {
  dfdb_group* g = nullptr;
  std::vector<dfdb_table*> shard;        // owned
  int64_t total_rows = -1;               // rows of the whole table (all ranks); -1 until something is resident
  std::vector<dfdb_gquery*> queries;
};
// (GroupMerged — unique / groupreduce over every shard: one record per distinct key, merged in rank order = table order, kept until fetched — lives in ooc.hpp)
struct dfdb_gquery {
  GroupMerged merged;
  dfdb_gtable* gt = nullptr;
  std::vector<dfdb_query*> shard;        // owned
  bool planned = false;                  // stage bases are set for the current stage list
  bool count_enqueued = false;           // the reduced count and the fault key of ITS exchange sit in `cres` (device, local shard 0)
  int64_t count = -1;                    // host copy of the global count
  // {global count, .., fault key} (slots 0 .. kFaultSlot) of the exchange group_count_enqueue made for THIS query.  The group's exchange slots are shared by every
  // collective of the group: a barrier, an allreduce_f64, another query's count or a failed aggregate between an enqueue-only
  // dfdb_group_count(gq, NULL) and the call that reads the count rewrite slot 0 and the fault slot.  The pair is therefore copied out of the
  // slots on the engine stream right behind the exchange, and group_count reads this copy: a count answers for its own exchange only.
  DevBuf cres;
};

namespace dfdb {

// run fn(local shard) on every shard's own thread (or inline for a one-shard group); the first failure is rethrown on the caller
static void for_shards(dfdb_group* g, const std::function<void(int)>& fn) {
  const int n = g->nlocal();
  if (g->workers.empty()) {
    for (int l = 0; l < n; l++) { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); fn(l); }
    return;
  }
  std::vector<std::unique_ptr<Error>> errs((size_t)n);
  for (int l = 0; l < n; l++)
    g->workers[(size_t)l]->submit([&, l] {
      try { fn(l); }
      catch (const Error& e) { errs[(size_t)l] = std::make_unique<Error>(e.code, e.what()); }
      catch (const std::exception& e) { errs[(size_t)l] = std::make_unique<Error>(DFDB_ERR_DEVICE, e.what()); }
    });
  for (int l = 0; l < n; l++) g->workers[(size_t)l]->wait();
  for (int l = 0; l < n; l++) if (errs[(size_t)l]) throw Error(errs[(size_t)l]->code, errs[(size_t)l]->what());
}

// every collective entry point starts with a clean slate: a note left behind by an operation that died on its way to the exchange (a failed RCCL
// call, a caller's collective that returned an error) must not be taken for this operation's
static void fresh(dfdb_group* g) { g->fault_key = ~0ull; g->fault_code = 0; g->fault_msg.clear(); }
static uint64_t fault_key_of(const Error& e) {
  const uint64_t code = (uint64_t)(e.code > 0 && e.code < 256 ? e.code : DFDB_ERR_DEVICE);
  return e.row != ~0ull ? (((e.row + 1) << 8) | code) : code;
}
static void note_fault(dfdb_group* g, const Error& e) {
  const uint64_t k = fault_key_of(e);
  if (k < g->fault_key) { g->fault_key = k; g->fault_code = e.code; g->fault_msg = e.what(); }
}
// the per-shard half of a collective operation: like for_shards, but a failure is noted (the smallest key wins) instead of thrown
static void for_shards_deferred(dfdb_group* g, const std::function<void(int)>& fn) {
  const int n = g->nlocal();
  if (g->workers.empty()) {
    for (int l = 0; l < n; l++) {
      try { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); fn(l); }
      catch (const Error& e) { note_fault(g, e); }
      catch (const std::exception& e) { note_fault(g, Error(DFDB_ERR_DEVICE, e.what())); }
    }
    return;
  }
  std::vector<std::unique_ptr<Error>> errs((size_t)n);
  for (int l = 0; l < n; l++)
    g->workers[(size_t)l]->submit([&, l] {
      try { fn(l); }
      catch (const Error& e) { errs[(size_t)l] = std::make_unique<Error>(e.code, e.what(), e.row); }
      catch (const std::exception& e) { errs[(size_t)l] = std::make_unique<Error>(DFDB_ERR_DEVICE, e.what()); }
    });
  for (int l = 0; l < n; l++) g->workers[(size_t)l]->wait();
  for (int l = 0; l < n; l++) if (errs[(size_t)l]) note_fault(g, *errs[(size_t)l]);
}
// what the ranks agreed on: raise it (with the local text when the local failure is the agreed one) and forget the local note
static void settle_fault(dfdb_group* g, uint64_t agreed) {
  const uint64_t mine = g->fault_key; const int code = g->fault_code; const std::string msg = std::move(g->fault_msg);
  g->fault_key = ~0ull; g->fault_code = 0; g->fault_msg.clear();
  if (agreed == ~0ull) {
    if (mine != ~0ull) throw Error(code, msg);            // (no exchange carried it: a group of one process)
    return;
  }
  const int acode = (int)(agreed & 0xffu);
  const uint64_t arow = (agreed >> 8) ? (agreed >> 8) - 1 : ~0ull;
  if (agreed == mine) throw Error(code, msg, arow);
  if (arow != ~0ull && acode == DFDB_ERR_DIVIDE) throw Error(acode, "DivideError: integer division error", arow);
  if (arow != ~0ull && acode == DFDB_ERR_ARGUMENT) throw Error(acode, "InexactError: conversion is not exact", arow);
  throw Error(acode, "another shard of the group failed with status " + std::to_string(acode) + " (its own rank holds the message)");
}

// a one-rank RCCL group still issues its collectives (they are copies): the same code runs at every world size
static bool exchanges(const dfdb_group* g) { return g->world > 1 || g->exchange == DFDB_EXCHANGE_RCCL; }

static void group_alloc_exchange(dfdb_group* g) {
  const int n = g->nlocal();
  g->xbuf.resize((size_t)n); g->xpin.assign((size_t)n, nullptr);
  const size_t bytes = (size_t)(kXSlots + g->world + 8) * 8;
  for (int l = 0; l < n; l++) {
    HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device));
    g->xbuf[(size_t)l].ensure(bytes);
    HIP_CHECK(hipMemset(g->xbuf[(size_t)l].p, 0, bytes));
    HIP_CHECK(hipHostMalloc((void**)&g->xpin[(size_t)l], bytes, hipHostMallocDefault));
    memset(g->xpin[(size_t)l], 0, bytes);
  }
}

// ---- exchanges.  Every shard's operands are already in its xbuf (written on its engine stream); results come back there.
static ncclDataType_t nccl_type(int dt) { return dt == DFDB_F64 ? ncclFloat64 : (dt == DFDB_U64 ? ncclUint64 : ncclInt64); }
static ncclRedOp_t nccl_op(int op) { return op == DFDB_AGG_MIN ? ncclMin : (op == DFDB_AGG_MAX ? ncclMax : ncclSum); }

// (fold_f64 / fold_bits: ooc.cpp — Julia's NaN and signed-zero rules, wrapping Int sums; shared with the block-streamed merges)

// stream-ordered host value -> slot of shard l
static void put_slot(dfdb_group* g, int l, int slot, int64_t v) {
  g->xpin[(size_t)l][slot] = v;
  HIP_CHECK(hipMemcpyAsync(g->xbuf[(size_t)l].as<uint64_t>() + slot, g->xpin[(size_t)l] + slot, 8, hipMemcpyHostToDevice, g->ctx[(size_t)l]->stream));
}
// A shard whose required columns are NOT resident (dfdb_group_query_prepare left them on disk: they do not fit) answers by streaming ITS block window of the
// column files (ooc.cpp; dfdb_table::win_first / win_last): the same per-shard values, taken from the stream instead of from HBM.
static int64_t shard_count(dfdb_query* q) { return query_out_of_core(q) ? ooc_count(q) : query_count(q, -1); }
static int64_t shard_string_bytes(dfdb_query* q, int32_t i) { return query_out_of_core(q) ? ooc_string_bytes(q, i) : query_string_bytes(q, i); }
static void shard_materialize(dfdb_query* q, dfdb_outcol* outs, int32_t ncols) { if (query_out_of_core(q)) ooc_materialize(q, outs, ncols); else query_materialize(q, outs, ncols); }
static void shard_select_indices(dfdb_query* q, int64_t* out, int64_t cap, int32_t memkind, int64_t* n) {
  if (query_out_of_core(q)) ooc_select_indices(q, out, cap, memkind, n); else query_select_indices(q, out, cap, memkind, n);
}
// every local shard's fault slot := the local fault key (what this process knows so far)
static void post_fault(dfdb_group* g) {
  for (int l = 0; l < g->nlocal(); l++) { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); put_slot(g, l, kFaultSlot, (int64_t)g->fault_key); }
}

// One exchange: every spec's slots [slot, slot+n) of every shard := their reduction over all ranks, in place, on the engine streams (no host wait
// with RCCL), all inside ONE ncclGroupStart/End — {value, count} of an aggregate are one launch — plus the fault slot (MIN).
struct XSpec { int slot, n, dt, op; };
static void exchange_reduce(dfdb_group* g, std::initializer_list<XSpec> specs_in) {
  const int nl = g->nlocal();
  std::vector<XSpec> specs(specs_in);
  post_fault(g);
  specs.push_back(XSpec{kFaultSlot, 1, DFDB_U64, DFDB_AGG_MIN});
  if (!exchanges(g)) return;
  if (g->exchange == DFDB_EXCHANGE_RCCL) {
    Rccl& r = rccl();
    rccl_bracket(r, g->comm_state, [&] {
      for (int l = 0; l < nl; l++)
        for (const XSpec& x : specs) {
          uint64_t* p = g->xbuf[(size_t)l].as<uint64_t>() + x.slot;
          RCCL_IN(r, r.AllReduce(p, p, (size_t)x.n, nccl_type(x.dt), nccl_op(x.op), g->comm[(size_t)l], g->ctx[(size_t)l]->stream));
        }
    });
    return;
  }
  if (g->exchange == DFDB_EXCHANGE_CALLBACK) {       // one shard per process: operands to pinned memory, the caller's all-reduce, results back
    for (const XSpec& x : specs)
      HIP_CHECK(hipMemcpyAsync(g->xpin[0] + x.slot, g->xbuf[0].as<uint64_t>() + x.slot, (size_t)x.n * 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    stream_wait(g->ctx[0]);
    for (const XSpec& x : specs) {
      const int32_t rc = g->fns.allreduce(g->fns.user, g->xpin[0] + x.slot, x.n, x.dt, x.op);
      if (rc != 0) fail(DFDB_ERR_DEVICE, "the caller's allreduce failed with %d", rc);
      HIP_CHECK(hipMemcpyAsync(g->xbuf[0].as<uint64_t>() + x.slot, g->xpin[0] + x.slot, (size_t)x.n * 8, hipMemcpyHostToDevice, g->ctx[0]->stream));
    }
    return;
  }
  // host exchange (single-process groups only): read every shard's operands, fold in rank order, write the result back
  for (const XSpec& x : specs) {
    for (int l = 0; l < nl; l++) {
      HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device));
      HIP_CHECK(hipMemcpyAsync(g->xpin[(size_t)l] + x.slot, g->xbuf[(size_t)l].as<uint64_t>() + x.slot, (size_t)x.n * 8, hipMemcpyDeviceToHost, g->ctx[(size_t)l]->stream));
    }
  }
  for (int l = 0; l < nl; l++) { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); stream_wait(g->ctx[(size_t)l]); }
  for (const XSpec& x : specs) {
    for (int k = 0; k < x.n; k++) {
      uint64_t acc = (uint64_t)g->xpin[0][x.slot + k];
      for (int l = 1; l < nl; l++) acc = fold_bits(acc, (uint64_t)g->xpin[(size_t)l][x.slot + k], x.dt, x.op);
      for (int l = 0; l < nl; l++) g->xpin[(size_t)l][x.slot + k] = (int64_t)acc;
    }
    for (int l = 0; l < nl; l++) {
      HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device));
      HIP_CHECK(hipMemcpyAsync(g->xbuf[(size_t)l].as<uint64_t>() + x.slot, g->xpin[(size_t)l] + x.slot, (size_t)x.n * 8, hipMemcpyHostToDevice, g->ctx[(size_t)l]->stream));
    }
  }
}

// slots [slot, slot+n) of local shard 0 -> host, together with the fault slot of the exchange that filled them (waits); raises the agreed fault
static int64_t get_slot0(dfdb_group* g, int slot, int n, int64_t* out) {
  HIP_CHECK(hipSetDevice(g->ctx[0]->device));
  HIP_CHECK(hipMemcpyAsync(g->xpin[0] + slot, g->xbuf[0].as<uint64_t>() + slot, (size_t)n * 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
  if (slot > kFaultSlot || slot + n <= kFaultSlot)
    HIP_CHECK(hipMemcpyAsync(g->xpin[0] + kFaultSlot, g->xbuf[0].as<uint64_t>() + kFaultSlot, 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
  stream_wait(g->ctx[0]);
  settle_fault(g, (uint64_t)g->xpin[0][kFaultSlot]);
  for (int k = 0; k < n; k++) out[k] = g->xpin[0][slot + k];
  return out[0];
}

// slot `slot` of every rank, gathered in rank order -> host vector of `world` values (waits; raises the agreed fault)
static std::vector<int64_t> exchange_gather(dfdb_group* g, int slot) {
  const int nl = g->nlocal();
  std::vector<int64_t> all((size_t)g->world, 0);
  post_fault(g);
  if (g->exchange == DFDB_EXCHANGE_RCCL) {
    Rccl& r = rccl();
    rccl_bracket(r, g->comm_state, [&] {
      for (int l = 0; l < nl; l++) {
        uint64_t* p = g->xbuf[(size_t)l].as<uint64_t>();
        RCCL_IN(r, r.AllGather(p + slot, p + kXSlots, 1, ncclInt64, g->comm[(size_t)l], g->ctx[(size_t)l]->stream));
        RCCL_IN(r, r.AllReduce(p + kFaultSlot, p + kFaultSlot, 1, ncclUint64, ncclMin, g->comm[(size_t)l], g->ctx[(size_t)l]->stream));
      }
    });
    HIP_CHECK(hipSetDevice(g->ctx[0]->device));
    HIP_CHECK(hipMemcpyAsync(g->xpin[0] + kXSlots, g->xbuf[0].as<uint64_t>() + kXSlots, (size_t)g->world * 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    HIP_CHECK(hipMemcpyAsync(g->xpin[0] + kFaultSlot, g->xbuf[0].as<uint64_t>() + kFaultSlot, 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    stream_wait(g->ctx[0]);
    settle_fault(g, (uint64_t)g->xpin[0][kFaultSlot]);
    for (int k = 0; k < g->world; k++) all[(size_t)k] = g->xpin[0][kXSlots + k];
    return all;
  }
  if (g->exchange == DFDB_EXCHANGE_CALLBACK) {
    HIP_CHECK(hipMemcpyAsync(g->xpin[0] + slot, g->xbuf[0].as<uint64_t>() + slot, 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    stream_wait(g->ctx[0]);
    int32_t rc = g->fns.allgather(g->fns.user, g->xpin[0] + slot, all.data(), 8);
    if (rc != 0) fail(DFDB_ERR_DEVICE, "the caller's allgather failed with %d", rc);
    uint64_t key = g->fault_key;
    rc = g->fns.allreduce(g->fns.user, &key, 1, DFDB_U64, DFDB_AGG_MIN);
    if (rc != 0) fail(DFDB_ERR_DEVICE, "the caller's allreduce failed with %d", rc);
    settle_fault(g, key);
    return all;
  }
  for (int l = 0; l < nl; l++) {
    HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device));
    HIP_CHECK(hipMemcpyAsync(g->xpin[(size_t)l] + slot, g->xbuf[(size_t)l].as<uint64_t>() + slot, 8, hipMemcpyDeviceToHost, g->ctx[(size_t)l]->stream));
  }
  for (int l = 0; l < nl; l++) { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); stream_wait(g->ctx[(size_t)l]); all[(size_t)l] = g->xpin[(size_t)l][slot]; }
  settle_fault(g, g->fault_key);                           // one process: what it noted is what there is
  return all;
}

// blocks [first, last) owned by `rank`: g * ceil(nb/G) .. (g+1) * ceil(nb/G), clipped (the rule of dfdb/sharding.py and DESIGN.md §8)
static void block_range(int64_t nblocks, int rank, int world, int64_t& first, int64_t& last) {
  const int64_t per = world > 0 ? ceil_div(nblocks, world) : nblocks;
  first = std::min<int64_t>((int64_t)rank * per, nblocks);
  last = std::min<int64_t>(first + per, nblocks);
}

static void group_finish_create(std::unique_ptr<dfdb_group>& g) {
  group_alloc_exchange(g.get());
  if (g->nlocal() > 1) {
    for (int l = 0; l < g->nlocal(); l++) { g->workers.push_back(std::make_unique<Worker>()); g->workers.back()->start(g->ctx[(size_t)l]->device); }
  }
}

static void group_destroy(dfdb_group* g) {
  if (!g) return;
  for (auto& w : g->workers) w->stop();
  for (size_t l = 0; l < g->ctx.size(); l++) {
    (void)hipSetDevice(g->ctx[l]->device);
    (void)hipStreamSynchronize(g->ctx[l]->stream);
    if (l < g->comm.size() && g->comm[l]) (void)rccl().CommDestroy(g->comm[l]);
    if (l < g->xpin.size() && g->xpin[l]) (void)hipHostFree(g->xpin[l]);
    if (l < g->xbuf.size()) g->xbuf[l].release();
    ctx_destroy(g->ctx[l]);
  }
  delete g;
}

// ---- stage bases: for every range-like stage that follows another stage, the survivors of the stages before it that live on lower
// ranks (all-gather + exclusive scan; left to right because a later base depends on the earlier ones being set)
static void plan_stage_bases(dfdb_gquery* gq) {
  if (gq->planned) return;
  dfdb_group* g = gq->gt->g;
  fresh(g);                              // (planning is the first collective step of whatever operation asked for it)
  const size_t ns = gq->shard[0]->stages.size();
  if (exchanges(g))
    for (size_t k = 1; k < ns; k++) {
      if (gq->shard[0]->stages[k].kind == ST_PRED) continue;
      for_shards_deferred(g, [&](int l) {
        dfdb_query* q = gq->shard[(size_t)l];
        if (query_out_of_core(q)) { put_slot(g, l, 1, ooc_count_prefix(q, (int)k)); return; }
        // planning raises nothing: whether a DivideError / InexactError of a predicate is reached is decided by the full execution, once every
        // stage knows the survivors on the lower ranks (query.cpp: error_is_reached); the erroring rows count as not selected meanwhile
        q->err_checking = true;
        try { query_execute(q, (int)k); } catch (...) { q->err_checking = false; throw; }
        q->err_checking = false;
        const int64_t ntiles = ceil_div(q->t->nrows, kTileRows);
        HIP_CHECK(hipMemcpyAsync(g->xbuf[(size_t)l].as<uint64_t>() + 1, q->prefix.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToDevice, q->t->ctx->stream));
        q->executed_stages = -1;   // a partial evaluation is not the view's result
      });
      const std::vector<int64_t> counts = exchange_gather(g, 1);
      int64_t base = 0;
      for (int r = 0; r < g->first_rank; r++) base += counts[(size_t)r];
      for (int l = 0; l < g->nlocal(); l++) {
        dfdb_query* q = gq->shard[(size_t)l];
        q->stages[k].stage_base = base; q->executed_stages = -1; q->count = -1; ooc_reset(q);
        base += counts[(size_t)(g->first_rank + l)];
      }
    }
  gq->planned = true;
}

static void gq_invalidate(dfdb_gquery* gq) { gq->planned = false; gq->count_enqueued = false; gq->count = -1; }

static bool shard_needs_exec(const dfdb_query* q) { return q->executed_stages != (int)q->stages.size() || q->bitmap_rows != q->t->nrows; }

// evaluate the view on every shard and leave the GLOBAL count in slot 0 of every shard's exchange buffer (no host wait with RCCL)
static void group_count_enqueue(dfdb_gquery* gq, bool wait) {
  dfdb_group* g = gq->gt->g;
  plan_stage_bases(gq);
  for_shards_deferred(g, [&](int l) {
    dfdb_query* q = gq->shard[(size_t)l];
    if (query_out_of_core(q)) { put_slot(g, l, 0, ooc_count(q)); return; }
    if (shard_needs_exec(q)) query_execute(q, -1);
    const int64_t ntiles = ceil_div(q->t->nrows, kTileRows);
    HIP_CHECK(hipMemcpyAsync(g->xbuf[(size_t)l].as<uint64_t>(), q->prefix.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToDevice, q->t->ctx->stream));
  });
  exchange_reduce(g, {XSpec{0, 1, DFDB_I64, DFDB_AGG_SUM}});      // a shard that failed is in the exchange all the same: its fault key travels with the count
  // the pair this exchange produced, kept apart from the shared slots (dfdb_gquery::cres)
  HIP_CHECK(hipSetDevice(g->ctx[0]->device));
  gq->cres.ensure((kFaultSlot + 1) * 8);                          // slots 0 .. kFaultSlot in one copy: the count is slot 0, the key the last
  HIP_CHECK(hipMemcpyAsync(gq->cres.p, g->xbuf[0].p, (kFaultSlot + 1) * 8, hipMemcpyDeviceToDevice, g->ctx[0]->stream));
  gq->count_enqueued = true; gq->count = -1;
  // enqueue-only callers (dfdb_group_count(gq, NULL)) never read the result back: a LOCAL failure is theirs to hear now; the other ranks — and
  // this one again — meet the agreed key in `cres` at their next read (group_count).  count_enqueued stays set on the failing rank too: clearing
  // it here would make this rank alone enqueue a new exchange at the next dfdb_group_count(gq, &n) while the healthy ranks only read; the read
  // raises on every rank alike (the key in `cres` includes this rank's) and THAT clears the flag everywhere.
  if (!wait && g->fault_key != ~0ull) { const Error e(g->fault_code, g->fault_msg); g->fault_key = ~0ull; g->fault_code = 0; g->fault_msg.clear(); throw e; }
}

static int64_t group_count(dfdb_gquery* gq) {
  if (gq->count >= 0) return gq->count;
  // Whether a new exchange is needed must be decided ALIKE on every rank: by the group query's own flag, which only calls that every rank makes
  // (new stages, reset, hints, table loads) clear — never by a shard's local state.  A shard whose execution failed "needs execution" on its rank only;
  // re-enqueueing there while the healthy ranks just read their slots would leave it alone in a collective.
  if (!gq->count_enqueued) group_count_enqueue(gq, true);
  int64_t n = 0;
  // (a fault the ranks agreed on invalidates the exchange for all of them alike: the next call enqueues again on every rank)
  dfdb_group* g = gq->gt->g;
  try {
    HIP_CHECK(hipSetDevice(g->ctx[0]->device));
    HIP_CHECK(hipMemcpyAsync(g->xpin[0] + kCountPin, gq->cres.p, (kFaultSlot + 1) * 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    stream_wait(g->ctx[0]);
    settle_fault(g, (uint64_t)g->xpin[0][kCountPin + kFaultSlot]);
    n = g->xpin[0][kCountPin];
  } catch (...) { gq->count_enqueued = false; throw; }
  gq->count = n;
  return n;
}

}  // namespace dfdb

// =================================================================================================== C ABI
namespace dfdb { void set_last_error(const char* msg); }   // c_api.cpp: the thread-local text dfdb_last_error returns

template <class F>
static int32_t gguard(F&& f) noexcept {
  try { f(); return DFDB_OK; }
  catch (const Error& e) { set_last_error(e.what()); return e.code; }
  catch (const std::bad_alloc&) { set_last_error("out of host memory"); return DFDB_ERR_NOMEM; }
  catch (const std::exception& e) { set_last_error(e.what()); return DFDB_ERR_DEVICE; }
  catch (...) { set_last_error("unknown error"); return DFDB_ERR_DEVICE; }
}
#define GNEED(p) do { if (!(p)) fail(DFDB_ERR_ARGUMENT, "null argument: " #p); } while (0)
// (the whole-column decodes of compressed-only columns a group entry point makes on its shards live until that entry point returns, like NEEDQT's in c_api.cpp:
// without this they stayed in HBM — hidden from dfdb_table_resident_bytes — until some plain query call on the shard's table)
struct GroupTransientScope {
  dfdb_gquery* gq;
  explicit GroupTransientScope(dfdb_gquery* q) : gq(q) {}
  ~GroupTransientScope() {
    if (!gq || !gq->gt) return;
    for (dfdb_table* t : gq->gt->shard) if (t) { (void)hipSetDevice(t->ctx->device); dfdb::table_drop_transient(t); }
  }
};
#define GNEEDQ(gq) GNEED(gq); if (!(gq)->gt) fail(DFDB_ERR_ARGUMENT, "the table of this query was closed"); GroupTransientScope group_transient_scope_(gq)

extern "C" {

int32_t dfdb_group_create(const int32_t* device_ids, int32_t n, int32_t exchange, dfdb_group** out) {
  return gguard([&] {
    GNEED(out); if (n > 0) GNEED(device_ids);
    if (n <= 0 || n > 64) fail(DFDB_ERR_ARGUMENT, "a group needs 1..64 devices, got %d", n);
    bool distinct = true;
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (device_ids[i] == device_ids[j]) distinct = false;
    if (exchange == DFDB_EXCHANGE_AUTO) exchange = distinct ? DFDB_EXCHANGE_RCCL : DFDB_EXCHANGE_HOST;
    if (exchange != DFDB_EXCHANGE_RCCL && exchange != DFDB_EXCHANGE_HOST) fail(DFDB_ERR_ARGUMENT, "unknown exchange %d", exchange);
    if (exchange == DFDB_EXCHANGE_RCCL && !distinct) fail(DFDB_ERR_ARGUMENT, "RCCL needs distinct devices: use DFDB_EXCHANGE_HOST when shards share a GPU");
    std::unique_ptr<dfdb_group> g(new dfdb_group);
    g->world = n; g->first_rank = 0; g->exchange = exchange;
    try {
      for (int i = 0; i < n; i++) {
        dfdb_ctx* c = nullptr;
        const int32_t rc = dfdb_ctx_create(device_ids[i], nullptr, &c);
        if (rc != DFDB_OK) { char buf[512]; dfdb_last_error(buf, sizeof buf); fail(rc, "%s", buf); }
        g->ctx.push_back(c);
      }
      if (exchange == DFDB_EXCHANGE_RCCL) {
        g->comm.assign((size_t)n, nullptr);
        std::vector<int> devs(device_ids, device_ids + n);
        RCCL_CHECK(rccl().CommInitAll(g->comm.data(), n, devs.data()));
      }
      group_finish_create(g);
    } catch (...) { group_destroy(g.release()); throw; }
    *out = g.release();
  });
}

int32_t dfdb_group_unique_id(uint8_t id[DFDB_GROUP_ID_BYTES]) {
  return gguard([&] {
    GNEED(id);
    static_assert(sizeof(ncclUniqueId) <= DFDB_GROUP_ID_BYTES, "ncclUniqueId grew");
    ncclUniqueId u; memset(&u, 0, sizeof u);
    RCCL_CHECK(rccl().GetUniqueId(&u));
    memset(id, 0, DFDB_GROUP_ID_BYTES); memcpy(id, &u, sizeof u);
  });
}

int32_t dfdb_group_create_rank(int32_t device_id, void* hip_stream, const uint8_t id[DFDB_GROUP_ID_BYTES], int32_t rank, int32_t world, dfdb_group** out) {
  return gguard([&] {
    GNEED(out);
    if (world < 1 || rank < 0 || rank >= world) fail(DFDB_ERR_ARGUMENT, "rank %d of %d", rank, world);
    std::unique_ptr<dfdb_group> g(new dfdb_group);
    g->world = world; g->first_rank = rank; g->exchange = DFDB_EXCHANGE_RCCL;
    try {
      dfdb_ctx* c = nullptr;
      const int32_t rc = dfdb_ctx_create(device_id, hip_stream, &c);
      if (rc != DFDB_OK) { char buf[512]; dfdb_last_error(buf, sizeof buf); fail(rc, "%s", buf); }
      g->ctx.push_back(c);
      g->comm.assign(1, nullptr);
      ncclUniqueId u; memset(&u, 0, sizeof u);
      if (id) memcpy(&u, id, sizeof u);
      else if (world == 1) RCCL_CHECK(rccl().GetUniqueId(&u));      // a one-rank group needs nobody else's id
      else fail(DFDB_ERR_ARGUMENT, "null argument: id (dfdb_group_unique_id on rank 0, handed to every rank)");
      HIP_CHECK(hipSetDevice(device_id));
      RCCL_CHECK(rccl().CommInitRank(&g->comm[0], world, u, rank));
      group_finish_create(g);
    } catch (...) { group_destroy(g.release()); throw; }
    *out = g.release();
  });
}

int32_t dfdb_group_create_rank_callbacks(int32_t device_id, void* hip_stream, int32_t rank, int32_t world, const dfdb_exchange_fns* fns, dfdb_group** out) {
  return gguard([&] {
    GNEED(out); GNEED(fns);
    if (world < 1 || rank < 0 || rank >= world) fail(DFDB_ERR_ARGUMENT, "rank %d of %d", rank, world);
    if (!fns->allreduce || !fns->allgather) fail(DFDB_ERR_ARGUMENT, "null argument: both allreduce and allgather are needed");
    std::unique_ptr<dfdb_group> g(new dfdb_group);
    g->world = world; g->first_rank = rank; g->exchange = DFDB_EXCHANGE_CALLBACK; g->fns = *fns;
    try {
      dfdb_ctx* c = nullptr;
      const int32_t rc = dfdb_ctx_create(device_id, hip_stream, &c);
      if (rc != DFDB_OK) { char buf[512]; dfdb_last_error(buf, sizeof buf); fail(rc, "%s", buf); }
      g->ctx.push_back(c);
      group_finish_create(g);
    } catch (...) { group_destroy(g.release()); throw; }
    *out = g.release();
  });
}

int32_t dfdb_group_destroy(dfdb_group* g) { return gguard([&] { group_destroy(g); }); }

int32_t dfdb_group_info(dfdb_group* g, int32_t* world, int32_t* nlocal, int32_t* first_rank, int32_t* exchange) {
  return gguard([&] { GNEED(g); if (world) *world = g->world; if (nlocal) *nlocal = g->nlocal(); if (first_rank) *first_rank = g->first_rank; if (exchange) *exchange = g->exchange; });
}
int32_t dfdb_group_ctx(dfdb_group* g, int32_t local, dfdb_ctx** ctx) {
  return gguard([&] { GNEED(g); GNEED(ctx); if (local < 0 || local >= g->nlocal()) fail(DFDB_ERR_BOUNDS, "BoundsError: local shard %d", local); *ctx = g->ctx[(size_t)local]; });
}
int32_t dfdb_group_synchronize(dfdb_group* g) {
  return gguard([&] { GNEED(g); for (dfdb_ctx* c : g->ctx) { HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipStreamSynchronize(c->stream)); } });
}
int32_t dfdb_group_set_option(dfdb_group* g, const char* key, int64_t value) {
  return gguard([&] { GNEED(g); GNEED(key); for (dfdb_ctx* c : g->ctx) c->options[key] = value; });
}
int32_t dfdb_group_barrier(dfdb_group* g) {   // every rank's engine stream has drained, on every rank
  return gguard([&] {
    GNEED(g);
    fresh(g);
    for (int l = 0; l < g->nlocal(); l++) { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); put_slot(g, l, 2, 1); }
    exchange_reduce(g, {XSpec{2, 1, DFDB_I64, DFDB_AGG_SUM}});
    for (int l = 0; l < g->nlocal(); l++) { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); HIP_CHECK(hipStreamSynchronize(g->ctx[(size_t)l]->stream)); }
  });
}
/* all-reduce of a few caller scalars over the ranks (e.g. the MAX of a per-rank wall time): vals[nlocal][n] in, reduced in place */
int32_t dfdb_group_allreduce_f64(dfdb_group* g, double* vals, int32_t n, int32_t op) {
  return gguard([&] {
    GNEED(g); GNEED(vals);
    if (n < 1 || n > kXSlots - 4) fail(DFDB_ERR_ARGUMENT, "1..%d values", kXSlots - 4);
    fresh(g);
    for (int l = 0; l < g->nlocal(); l++) {
      HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device));
      for (int k = 0; k < n; k++) { int64_t b; memcpy(&b, &vals[(size_t)l * n + k], 8); put_slot(g, l, 4 + k, b); }
    }
    if (op != DFDB_AGG_SUM && (g->exchange == DFDB_EXCHANGE_RCCL || g->exchange == DFDB_EXCHANGE_CALLBACK) && exchanges(g)) {
      // Float64 MIN / MAX never reach ncclMin / ncclMax or the caller's allreduce (dfdb_exchange_fns promises that: a callback written to the
      // header may only know integer min / max, and Julia's min / max propagate NaN): gather every rank's value and fold on the host, as
      // dfdb_group_aggregate does
      for (int k = 0; k < n; k++) {
        const std::vector<int64_t> all = exchange_gather(g, 4 + k);
        uint64_t acc = (uint64_t)all[0];
        for (int r = 1; r < g->world; r++) acc = fold_bits(acc, (uint64_t)all[(size_t)r], DFDB_F64, op);
        for (int l = 0; l < g->nlocal(); l++) memcpy(&vals[(size_t)l * n + k], &acc, 8);
      }
      return;
    }
    exchange_reduce(g, {XSpec{4, n, DFDB_F64, op}});
    for (int l = 0; l < g->nlocal(); l++) {
      HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device));
      HIP_CHECK(hipMemcpyAsync(g->xpin[(size_t)l] + 4, g->xbuf[(size_t)l].as<uint64_t>() + 4, (size_t)n * 8, hipMemcpyDeviceToHost, g->ctx[(size_t)l]->stream));
      stream_wait(g->ctx[(size_t)l]);
      for (int k = 0; k < n; k++) memcpy(&vals[(size_t)l * n + k], &g->xpin[(size_t)l][4 + k], 8);
    }
  });
}

// ------------------------------------------------------------------ tables
static void gtable_changed(dfdb_gtable* gt) {   // rows came or went on every rank alike: the table's queries start over, in lockstep
  for (dfdb_gquery* gq : gt->queries) { gq->planned = false; gq->count_enqueued = false; gq->count = -1; gq->merged = GroupMerged{}; }
}
static dfdb_gtable* new_gtable(dfdb_group* g) { auto* gt = new dfdb_gtable; gt->g = g; gt->shard.assign((size_t)g->nlocal(), nullptr); return gt; }
static void gtable_free(dfdb_gtable* gt) {
  if (!gt) return;
  for (dfdb_gquery* gq : gt->queries) gq->gt = nullptr;
  for (size_t l = 0; l < gt->shard.size(); l++) if (gt->shard[l]) { (void)hipSetDevice(gt->g->ctx[l]->device); (void)dfdb_table_close(gt->shard[l]); }
  delete gt;
}
static void rethrow_rc(int32_t rc) { if (rc != DFDB_OK) { char buf[1024]; dfdb_last_error(buf, sizeof buf); fail(rc, "%s", buf); } }

int32_t dfdb_group_table_open(dfdb_group* g, const char* path, dfdb_gtable** out) {
  return gguard([&] {
    GNEED(g); GNEED(path); GNEED(out);
    std::unique_ptr<dfdb_gtable, void (*)(dfdb_gtable*)> gt(new_gtable(g), gtable_free);
    for (int l = 0; l < g->nlocal(); l++) { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); table_open(g->ctx[(size_t)l], path, &gt->shard[(size_t)l]); }
    // every shard's block window, from the headers of the first column (20 bytes per block: skip_block, BlockStreams.jl:74-78): a shard whose columns are never
    // loaded streams ITS range of the files, not the table (ooc.cpp)
    dfdb_table* t0 = gt->shard[0];
    if (!t0->cols.empty()) {
      dfdb_sizestats hs{0, 0, 0};
      table_column_stats(t0, 0, &hs);
      const int64_t nblocks = ceil_div(hs.rows, t0->block_size);
      for (int l = 0; l < g->nlocal(); l++) {
        int64_t b0, b1; block_range(nblocks, g->first_rank + l, g->world, b0, b1);
        gt->shard[(size_t)l]->win_first = b0; gt->shard[(size_t)l]->win_last = b1;
      }
      gt->total_rows = hs.rows;
    }
    *out = gt.release();
  });
}
int32_t dfdb_group_table_new(dfdb_group* g, int64_t block_size, dfdb_gtable** out) {
  return gguard([&] {
    GNEED(g); GNEED(out);
    std::unique_ptr<dfdb_gtable, void (*)(dfdb_gtable*)> gt(new_gtable(g), gtable_free);
    for (int l = 0; l < g->nlocal(); l++) rethrow_rc(dfdb_table_new(g->ctx[(size_t)l], block_size, &gt->shard[(size_t)l]));
    *out = gt.release();
  });
}
int32_t dfdb_group_table_close(dfdb_gtable* gt) { return gguard([&] { gtable_free(gt); }); }
/* dfdb_table_unload on every shard: the listed columns (NULL = all) leave HBM; the files stay, and the group's entry points stream them from then on */
int32_t dfdb_group_table_unload(dfdb_gtable* gt, const int32_t* ordinals, int32_t ncols) {
  return gguard([&] {
    GNEED(gt);
    dfdb_group* g = gt->g;
    for_shards(g, [&](int l) { rethrow_rc(dfdb_table_unload(gt->shard[(size_t)l], ordinals, ncols)); });
    gtable_changed(gt);
  });
}
int32_t dfdb_group_table_shard(dfdb_gtable* gt, int32_t local, dfdb_table** t) {
  return gguard([&] { GNEED(gt); GNEED(t); if (local < 0 || (size_t)local >= gt->shard.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: local shard %d", local); *t = gt->shard[(size_t)local]; });
}
int32_t dfdb_group_table_nrows(dfdb_gtable* gt, int64_t* total) { return gguard([&] { GNEED(gt); GNEED(total); *total = gt->total_rows < 0 ? 0 : gt->total_rows; }); }

/* every shard loads its block range of the listed columns (NULL = all): dfdb_table_load(block_range(nblocks, rank, world)) */
int32_t dfdb_group_table_load(dfdb_gtable* gt, const int32_t* ordinals, int32_t ncols, dfdb_sizestats* stats) {
  return gguard([&] {
    GNEED(gt);
    dfdb_group* g = gt->g;
    dfdb_table* t0 = gt->shard[0];
    if (t0->cols.empty()) fail(DFDB_ERR_ARGUMENT, "the table has no columns");
    // the block count comes from the headers of one column (every column shares the block boundaries)
    const int32_t probe = ordinals && ncols > 0 ? ordinals[0] : 0;
    dfdb_sizestats hs{0, 0, 0};
    table_column_stats(t0, probe, &hs);
    const int64_t nblocks = ceil_div(hs.rows, t0->block_size);
    std::vector<dfdb_sizestats> st((size_t)g->nlocal(), dfdb_sizestats{0, 0, 0});
    for_shards(g, [&](int l) {
      int64_t b0, b1; block_range(nblocks, g->first_rank + l, g->world, b0, b1);
      table_load(gt->shard[(size_t)l], ordinals, ncols, b0, b1, &st[(size_t)l]);
      gt->shard[(size_t)l]->row_base = b0 * t0->block_size;
    });
    gt->total_rows = hs.rows;
    gtable_changed(gt);
    if (stats) { *stats = dfdb_sizestats{0, 0, 0}; for (auto& s : st) { stats->rows += s.rows; stats->compressed += s.compressed; stats->uncompressed += s.uncompressed; } }
  });
}

/* synthetic column of nrows_total rows: every shard generates the rows of its block range on its own device */
int32_t dfdb_group_table_add_generated(dfdb_gtable* gt, const char* name, int32_t generator, uint64_t seed, int64_t nrows_total) {
  return gguard([&] {
    GNEED(gt); GNEED(name);
    if (nrows_total < 0) fail(DFDB_ERR_ARGUMENT, "negative row count");
    if (gt->total_rows >= 0 && gt->total_rows != nrows_total) fail(DFDB_ERR_ARGUMENT, "column has %lld rows but the table has %lld", (long long)nrows_total, (long long)gt->total_rows);
    dfdb_group* g = gt->g;
    const int64_t bs = gt->shard[0]->block_size, nblocks = ceil_div(nrows_total, bs);
    for_shards(g, [&](int l) {
      int64_t b0, b1; block_range(nblocks, g->first_rank + l, g->world, b0, b1);
      const int64_t r0 = std::min(b0 * bs, nrows_total), r1 = std::min(b1 * bs, nrows_total);
      table_add_generated(gt->shard[(size_t)l], name, generator, seed, r0, r1 - r0);
      gt->shard[(size_t)l]->row_base = r0; gt->shard[(size_t)l]->block_first = b0;
    });
    gt->total_rows = nrows_total;
    gtable_changed(gt);
  });
}

/* caller-supplied decoded column of the WHOLE table (host memory, layout of dfdb_table_add_column): every shard uploads the rows
 * of its block range */
int32_t dfdb_group_table_add_column(dfdb_gtable* gt, const char* name, int32_t dtype, int64_t nrows_total, const void* data, const uint8_t* bytes,
                                    int64_t nbytes, const uint8_t* missing) {
  return gguard([&] {
    GNEED(gt); GNEED(name); if (nrows_total > 0) GNEED(data);
    if (nrows_total < 0) fail(DFDB_ERR_ARGUMENT, "negative row count");
    if (gt->total_rows >= 0 && gt->total_rows != nrows_total) fail(DFDB_ERR_ARGUMENT, "column has %lld rows but the table has %lld", (long long)nrows_total, (long long)gt->total_rows);
    dfdb_group* g = gt->g;
    const int64_t bs = gt->shard[0]->block_size, nblocks = ceil_div(nrows_total, bs);
    const bool is_str = dt_base(dtype) == DFDB_STRING;
    const int w = is_str ? 4 : dt_width(dtype);
    // byte offset of every local shard's first row inside the arena (String columns)
    std::vector<int64_t> r0s((size_t)g->nlocal()), r1s((size_t)g->nlocal()), boff((size_t)g->nlocal() + 1, 0);
    for (int l = 0; l < g->nlocal(); l++) {
      int64_t b0, b1; block_range(nblocks, g->first_rank + l, g->world, b0, b1);
      r0s[(size_t)l] = std::min(b0 * bs, nrows_total); r1s[(size_t)l] = std::min(b1 * bs, nrows_total);
    }
    if (is_str) {
      const int32_t* sz = (const int32_t*)data;
      int64_t acc = 0, r = 0;
      for (int l = 0; l < g->nlocal(); l++) {
        for (; r < r0s[(size_t)l]; r++) acc += sz[r] > 0 ? sz[r] : 0;
        boff[(size_t)l] = acc;
      }
      for (; r < r1s[(size_t)g->nlocal() - 1]; r++) acc += sz[r] > 0 ? sz[r] : 0;
      boff[(size_t)g->nlocal()] = acc;
      if (acc > nbytes) fail(DFDB_ERR_ARGUMENT, "string sizes sum to %lld bytes but the arena holds %lld", (long long)acc, (long long)nbytes);
    }
    for_shards(g, [&](int l) {
      const int64_t r0 = r0s[(size_t)l], n = r1s[(size_t)l] - r0;
      const int64_t nb = is_str ? boff[(size_t)l + 1] - boff[(size_t)l] : 0;
      table_add_column(gt->shard[(size_t)l], name, dtype, n, (const char*)data + r0 * w, is_str && bytes ? bytes + boff[(size_t)l] : bytes, nb, missing ? missing + r0 : nullptr);
      gt->shard[(size_t)l]->row_base = r0; gt->shard[(size_t)l]->block_first = r0 / bs;
    });
    gt->total_rows = nrows_total;
    gtable_changed(gt);
  });
}

// ------------------------------------------------------------------ queries
static void gquery_free(dfdb_gquery* gq) {
  if (!gq) return;
  if (gq->gt) {
    auto& v = gq->gt->queries;
    for (size_t i = 0; i < v.size(); i++) if (v[i] == gq) { v[i] = v.back(); v.pop_back(); break; }
    for (size_t l = 0; l < gq->shard.size(); l++) if (gq->shard[l]) { (void)hipSetDevice(gq->gt->g->ctx[l]->device); (void)dfdb_query_free(gq->shard[l]); }
  } else for (dfdb_query* q : gq->shard) if (q) (void)dfdb_query_free(q);
  delete gq;
}

int32_t dfdb_group_query_new(dfdb_gtable* gt, dfdb_gquery** out) {
  return gguard([&] {
    GNEED(gt); GNEED(out);
    std::unique_ptr<dfdb_gquery, void (*)(dfdb_gquery*)> gq(new dfdb_gquery, gquery_free);
    gq->gt = gt; gq->shard.assign(gt->shard.size(), nullptr);
    gt->queries.push_back(gq.get());
    for (size_t l = 0; l < gt->shard.size(); l++) rethrow_rc(dfdb_query_new(gt->shard[l], &gq->shard[l]));
    *out = gq.release();
  });
}
int32_t dfdb_group_query_free(dfdb_gquery* gq) { return gguard([&] { gquery_free(gq); }); }
/* dfdb_query_prepare for a sharded table: only the columns the view needs are opened (view.jl:183-190, blocksiterator.jl:20-33).  Every shard loads ITS block
 * range of exactly those columns when its share fits (*how = 1; 0: they were resident already); otherwise the shards keep nothing and every group entry point
 * streams each shard's block range from the files (*how = 3; per-rank values meet in the same exchanges).  The decision is made ALIKE on every rank — from the
 * files' headers, group option "hbm_budget_mb" (0: 80 % of the device's HBM) and what the shards hold —, never from a rank's momentary free memory. */
int32_t dfdb_group_query_prepare(dfdb_gquery* gq, int32_t* how) {
  return gguard([&] {
    GNEEDQ(gq);
    dfdb_gtable* gt = gq->gt; dfdb_group* g = gt->g;
    dfdb_table* t0 = gt->shard[0];
    if (how) *how = 0;
    if (t0->path.empty() || t0->cols.empty()) return;
    dfdb_query* q0 = gq->shard[0];
    std::vector<int> req;
    for (const Stage& st : q0->stages) if (st.kind == ST_PRED) required_columns(*st.pred, req);
    for (const ProjCol& pc : q0->proj) required_columns(*pc.expr, req);
    if (req.empty()) req.push_back(0);
    std::vector<int32_t> need;
    for (int o : req) if (!t0->cols[(size_t)o].resident) need.push_back(o);
    if (need.empty()) return;
    int64_t dec = 0, comp_max = 0; dfdb_sizestats hs{0, 0, 0};
    for (int32_t o : need) { dfdb_sizestats st{0, 0, 0}; table_column_stats(t0, o, &st); dec += st.uncompressed; comp_max = std::max(comp_max, st.compressed); hs = st; }
    const int64_t nblocks = ceil_div(hs.rows, t0->block_size);
    int64_t budget = ctx_option(g->ctx[0], "hbm_budget_mb", 0) << 20;
    int64_t held = 0;
    { int64_t d = 0, k = 0; table_resident_bytes(t0, -1, &d, &k); held = d + k; }
    if (budget <= 0) budget = (int64_t)((double)g->ctx[0]->prop.totalGlobalMem * 0.8);
    const int64_t share = ceil_div(dec + comp_max, (int64_t)g->world) + (64 << 20);
    if (share <= budget - held) {
      std::vector<dfdb_sizestats> st((size_t)g->nlocal(), dfdb_sizestats{0, 0, 0});
      bool nomem = false;
      try {
        for_shards(g, [&](int l) {
          int64_t b0, b1; block_range(nblocks, g->first_rank + l, g->world, b0, b1);
          dfdb_table* t = gt->shard[(size_t)l];
          table_load(t, need.data(), (int32_t)need.size(), b0, b1, &st[(size_t)l]);
          t->row_base = b0 * t0->block_size;
        });
      } catch (const Error& e) { if (e.code != DFDB_ERR_NOMEM) throw; nomem = true; }
      gt->total_rows = hs.rows;
      if (!nomem) { gtable_changed(gt); if (how) *how = 1; return; }
      // HBM ran out inside a load after all (other tables, fragmentation): the view's columns leave every local shard again and the shards stream.  (Ranks of other
      // processes may have loaded theirs: every shard answers for itself, from HBM or from its files, and the exchanges are the same either way.)
      for_shards(g, [&](int l) { (void)hipGetLastError(); rethrow_rc(dfdb_table_unload(gt->shard[(size_t)l], need.data(), (int32_t)need.size())); });
      gtable_changed(gt);
    }
    if (how) *how = 3;                                     // (the shards' block windows were set when the table was opened)
  });
}
int32_t dfdb_group_query_shard(dfdb_gquery* gq, int32_t local, dfdb_query** q) {
  return gguard([&] { GNEEDQ(gq); GNEED(q); if (local < 0 || (size_t)local >= gq->shard.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: local shard %d", local); *q = gq->shard[(size_t)local]; });
}
// the composition rules (selection.jl:37-60) are applied per shard by the single-GPU entry points; a stage the first shard rejects
// (BoundsError, ArgumentError) is rejected before any other shard saw it, so the shards never diverge
int32_t dfdb_group_query_add_range(dfdb_gquery* gq, int64_t start, int64_t step, int64_t stop) {
  return gguard([&] { GNEEDQ(gq); gq_invalidate(gq); for (dfdb_query* q : gq->shard) rethrow_rc(dfdb_query_add_range(q, start, step, stop)); });
}
int32_t dfdb_group_query_add_indices(dfdb_gquery* gq, const int64_t* idx, int64_t n) {
  return gguard([&] { GNEEDQ(gq); gq_invalidate(gq); for (dfdb_query* q : gq->shard) rethrow_rc(dfdb_query_add_indices(q, idx, n)); });
}
int32_t dfdb_group_query_add_integer(dfdb_gquery* gq, int64_t i) {
  return gguard([&] { GNEEDQ(gq); gq_invalidate(gq); for (dfdb_query* q : gq->shard) rethrow_rc(dfdb_query_add_integer(q, i)); });
}
int32_t dfdb_group_query_add_predicate(dfdb_gquery* gq, const uint8_t* ir, size_t len) {
  return gguard([&] { GNEEDQ(gq); gq_invalidate(gq); for (dfdb_query* q : gq->shard) rethrow_rc(dfdb_query_add_predicate(q, ir, len)); });
}
int32_t dfdb_group_query_set_projection(dfdb_gquery* gq, int32_t n, const char* const* names, const uint8_t* const* irs, const size_t* lens) {
  return gguard([&] { GNEEDQ(gq); for (dfdb_query* q : gq->shard) rethrow_rc(dfdb_query_set_projection(q, n, names, irs, lens)); });
}
int32_t dfdb_group_query_hint_aggregate(dfdb_gquery* gq, int32_t op, int32_t proj_col) {
  return gguard([&] { GNEEDQ(gq); for (dfdb_query* q : gq->shard) rethrow_rc(dfdb_query_hint_aggregate(q, op, proj_col)); });
}
int32_t dfdb_group_query_hint_materialize(dfdb_gquery* gq, int32_t on) {
  return gguard([&] { GNEEDQ(gq); gq->count_enqueued = false; gq->count = -1; for (dfdb_query* q : gq->shard) rethrow_rc(dfdb_query_hint_materialize(q, on)); });
}
int32_t dfdb_group_query_reset(dfdb_gquery* gq) {
  return gguard([&] { GNEEDQ(gq); gq->count_enqueued = false; gq->count = -1; for (dfdb_query* q : gq->shard) { q->executed_stages = -1; q->count = -1; q->prefix_valid = false; } });
}

/* nrow(v) over the whole table: per-shard scans, the stage-base exchanges a range-after-predicate needs, one all-reduce.
 * n == NULL: only enqueue (no host wait; the reduced count stays on the devices until a later call asks for it) */
int32_t dfdb_group_count(dfdb_gquery* gq, int64_t* n) {
  return gguard([&] { GNEEDQ(gq); fresh(gq->gt->g); if (!n) { group_count_enqueue(gq, false); return; } *n = group_count(gq); });
}
/* selected rows on every rank, in rank order (world values): what a caller needs to place sharded results */
int32_t dfdb_group_shard_counts(dfdb_gquery* gq, int64_t* counts) {
  return gguard([&] {
    GNEEDQ(gq); GNEED(counts);
    dfdb_group* g = gq->gt->g;
    fresh(g);
    group_count(gq);
    for_shards_deferred(g, [&](int l) {
      dfdb_query* q = gq->shard[(size_t)l];
      if (query_out_of_core(q)) { put_slot(g, l, 1, ooc_count(q)); return; }
      const int64_t ntiles = ceil_div(q->t->nrows, kTileRows);
      HIP_CHECK(hipMemcpyAsync(g->xbuf[(size_t)l].as<uint64_t>() + 1, q->prefix.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToDevice, q->t->ctx->stream));
    });
    const std::vector<int64_t> all = exchange_gather(g, 1);
    for (int r = 0; r < g->world; r++) counts[r] = all[(size_t)r];
  });
}

/* sum / minimum / maximum / count of projection column i over the whole table (exact for integers: wrapping 64-bit sums are
 * associative; Float64 sums add the per-GPU partial sums, within the tolerance of DESIGN.md §5) */
int32_t dfdb_group_aggregate(dfdb_gquery* gq, int32_t op, int32_t i, int64_t* out_i, double* out_f) {
  return gguard([&] {
    GNEEDQ(gq);
    dfdb_group* g = gq->gt->g;
    fresh(g);
    if (op == DFDB_AGG_COUNT) { const int64_t n = group_count(gq); if (out_i) *out_i = n; if (out_f) *out_f = (double)n; return; }
    if (op != DFDB_AGG_SUM && op != DFDB_AGG_MIN && op != DFDB_AGG_MAX) fail(DFDB_ERR_ARGUMENT, "unknown aggregate %d", op);
    plan_stage_bases(gq);
    std::vector<int> dts((size_t)g->nlocal(), DFDB_I64);
    for_shards_deferred(g, [&](int l) {
      dfdb_query* q = gq->shard[(size_t)l];
      if (query_out_of_core(q)) { uint64_t vb[2]; dts[(size_t)l] = ooc_aggregate_bits(q, op, i, vb); put_slot(g, l, 8, (int64_t)vb[0]); put_slot(g, l, 9, (int64_t)vb[1]); return; }
      dts[(size_t)l] = query_aggregate_device(q, op, i);          // {value, count} in q->red_result, identity when the shard selects nothing
      HIP_CHECK(hipMemcpyAsync(g->xbuf[(size_t)l].as<uint64_t>() + 8, q->red_result.p, 16, hipMemcpyDeviceToDevice, q->t->ctx->stream));
    });
    // the accumulator type follows from the projection's dtype, which every shard shares (a shard that failed before it could say still knows it)
    const int32_t pdt = i >= 0 && (size_t)i < gq->shard[0]->proj.size() ? dt_base(gq->shard[0]->proj[(size_t)i].expr->dtype) : DFDB_I64;
    const int dt = dt_isfloat(pdt) ? DFDB_F64 : (pdt == DFDB_U64 ? DFDB_U64 : DFDB_I64);      // = what query_aggregate_device returns
    for (int d : dts) if (d != dt && g->fault_key == ~0ull) fail(DFDB_ERR_DEVICE, "shards disagree on the accumulator type");
    int64_t res[2] = {0, 0};
    if (dt == DFDB_F64 && op != DFDB_AGG_SUM && (g->exchange == DFDB_EXCHANGE_RCCL || g->exchange == DFDB_EXCHANGE_CALLBACK) && exchanges(g)) {
      // Julia's minimum / maximum propagate NaN, ncclMin / ncclMax need not: gather the per-rank partials and fold them on the host
      const std::vector<int64_t> vals = exchange_gather(g, 8);
      uint64_t acc = (uint64_t)vals[0];
      for (int r = 1; r < g->world; r++) acc = fold_bits(acc, (uint64_t)vals[(size_t)r], DFDB_F64, op);
      exchange_reduce(g, {XSpec{9, 1, DFDB_I64, DFDB_AGG_SUM}});
      get_slot0(g, 9, 1, &res[1]);
      res[0] = (int64_t)acc;
    } else {
      exchange_reduce(g, {XSpec{8, 1, dt, op}, XSpec{9, 1, DFDB_I64, DFDB_AGG_SUM}});    // ONE exchange for {value, count}
      get_slot0(g, 8, 2, res);
    }
    // the count of an aggregate is the count of the selection (nullable columns are not aggregated): nrow(v) after sum(col) costs nothing more
    bool current = true;
    for (dfdb_query* q : gq->shard) current = current && !shard_needs_exec(q);
    if (current) gq->count = res[1];
    if (res[1] == 0 && op != DFDB_AGG_SUM) fail(DFDB_ERR_ARGUMENT, "ArgumentError: reducing over an empty collection is not allowed");
    if (dt == DFDB_F64) { double d; memcpy(&d, &res[0], 8); if (out_f) *out_f = d; if (out_i) *out_i = (int64_t)d; }
    else { if (out_i) *out_i = res[0]; if (out_f) *out_f = dt == DFDB_U64 ? (double)(uint64_t)res[0] : (double)res[0]; }
  });
}

/* ascending 1-based TABLE row numbers of the local shards' selected rows, each shard into its own DEVICE buffer
 * (outs[l], caps[l]); asynchronous */
int32_t dfdb_group_select_indices_device(dfdb_gquery* gq, int64_t* const* outs, const int64_t* caps) {
  return gguard([&] {
    GNEEDQ(gq); GNEED(outs); GNEED(caps);
    plan_stage_bases(gq);
    for_shards(gq->gt->g, [&](int l) { shard_select_indices(gq->shard[(size_t)l], outs[l], caps[l], DFDB_MEM_DEVICE, nullptr); });
  });
}
/* the same into ONE host buffer: the local shards' row numbers concatenated in rank order (= table order).  *n = rows written
 * by this process; a one-process group therefore gets the whole result */
int32_t dfdb_group_select_indices(dfdb_gquery* gq, int64_t* out, int64_t cap, int64_t* n) {
  return gguard([&] {
    GNEEDQ(gq); if (cap > 0) GNEED(out);
    dfdb_group* g = gq->gt->g;
    plan_stage_bases(gq);
    std::vector<int64_t> cnt((size_t)g->nlocal(), 0), base((size_t)g->nlocal() + 1, 0);
    for_shards(g, [&](int l) { cnt[(size_t)l] = shard_count(gq->shard[(size_t)l]); });
    for (int l = 0; l < g->nlocal(); l++) base[(size_t)l + 1] = base[(size_t)l] + cnt[(size_t)l];
    if (n) *n = base[(size_t)g->nlocal()];
    for_shards(g, [&](int l) {
      const int64_t room = std::max<int64_t>(0, std::min(cnt[(size_t)l], cap - base[(size_t)l]));
      if (room > 0) shard_select_indices(gq->shard[(size_t)l], out + base[(size_t)l], room, DFDB_MEM_HOST, nullptr);
    });
  });
}

/* string bytes projection column i needs over the local shards */
int32_t dfdb_group_result_string_bytes(dfdb_gquery* gq, int32_t i, int64_t* nbytes) {
  return gguard([&] {
    GNEEDQ(gq); GNEED(nbytes);
    dfdb_group* g = gq->gt->g;
    plan_stage_bases(gq);
    std::vector<int64_t> nb((size_t)g->nlocal(), 0);
    for_shards(g, [&](int l) { nb[(size_t)l] = shard_string_bytes(gq->shard[(size_t)l], i); });
    *nbytes = 0; for (int64_t b : nb) *nbytes += b;
  });
}

/* materialize(v) into caller-owned HOST buffers sized for the local shards' rows (dfdb_group_count for a one-process group):
 * every shard writes its rows at its rank-order offset, so the buffers hold the view in table order */
int32_t dfdb_group_materialize(dfdb_gquery* gq, dfdb_outcol* outs, int32_t ncols) {
  return gguard([&] {
    GNEEDQ(gq); if (ncols > 0) GNEED(outs);
    dfdb_group* g = gq->gt->g;
    const int nl = g->nlocal();
    for (int32_t p = 0; p < ncols; p++) if (outs[p].memkind != DFDB_MEM_HOST) fail(DFDB_ERR_ARGUMENT, "dfdb_group_materialize writes host buffers (use the shard queries for device outputs)");
    plan_stage_bases(gq);
    std::vector<int64_t> cnt((size_t)nl, 0), base((size_t)nl + 1, 0);
    std::vector<std::vector<int64_t>> sb((size_t)nl, std::vector<int64_t>((size_t)ncols, 0));
    for_shards(g, [&](int l) {
      dfdb_query* q = gq->shard[(size_t)l];
      if (ncols != (int32_t)q->proj.size()) fail(DFDB_ERR_ARGUMENT, "ArgumentError: view has %zu columns, %d outputs given", q->proj.size(), ncols);
      cnt[(size_t)l] = shard_count(q);
      for (int32_t p = 0; p < ncols; p++) if (dt_base(q->proj[(size_t)p].expr->dtype) == DFDB_STRING) sb[(size_t)l][(size_t)p] = shard_string_bytes(q, p);
    });
    for (int l = 0; l < nl; l++) base[(size_t)l + 1] = base[(size_t)l] + cnt[(size_t)l];
    std::vector<std::vector<int64_t>> boff((size_t)nl + 1, std::vector<int64_t>((size_t)ncols, 0));
    for (int l = 0; l < nl; l++) for (int32_t p = 0; p < ncols; p++) boff[(size_t)l + 1][(size_t)p] = boff[(size_t)l][(size_t)p] + sb[(size_t)l][(size_t)p];
    for (int32_t p = 0; p < ncols; p++)
      if (boff[(size_t)nl][(size_t)p] > outs[p].bytes_cap && boff[(size_t)nl][(size_t)p] > 0)
        fail(DFDB_ERR_ARGUMENT, "output column %d needs %lld string bytes, capacity is %lld", p, (long long)boff[(size_t)nl][(size_t)p], (long long)outs[p].bytes_cap);
    std::vector<std::vector<dfdb_outcol>> so((size_t)nl, std::vector<dfdb_outcol>((size_t)ncols));
    for_shards(g, [&](int l) {
      dfdb_query* q = gq->shard[(size_t)l];
      for (int32_t p = 0; p < ncols; p++) {
        dfdb_outcol o = outs[p];
        const int32_t dt = q->proj[(size_t)p].expr->dtype;
        const int w = dt_base(dt) == DFDB_STRING ? 4 : dt_width(dt);
        if (o.data) o.data = (char*)o.data + base[(size_t)l] * w;
        if (o.bytes) { o.bytes += boff[(size_t)l][(size_t)p]; o.bytes_cap = sb[(size_t)l][(size_t)p]; }
        if (o.missing) o.missing += base[(size_t)l];
        so[(size_t)l][(size_t)p] = o;
      }
      shard_materialize(q, so[(size_t)l].data(), ncols);
    });
    for (int32_t p = 0; p < ncols; p++) {
      outs[p].dtype = so[0][(size_t)p].dtype; outs[p].count = base[(size_t)nl]; outs[p].nbytes = boff[(size_t)nl][(size_t)p];
    }
  });
}


/* string bytes projection column i needs on EACH local shard (nlocal values): what sizes the per-shard device buffers of
 * dfdb_group_materialize_device */
int32_t dfdb_group_shard_string_bytes(dfdb_gquery* gq, int32_t i, int64_t* nbytes) {
  return gguard([&] {
    GNEEDQ(gq); GNEED(nbytes);
    dfdb_group* g = gq->gt->g;
    plan_stage_bases(gq);
    for_shards(g, [&](int l) { nbytes[l] = shard_string_bytes(gq->shard[(size_t)l], i); });
  });
}

/* materialize(v) (materialization.jl:27-40) with the result left SHARDED on the devices: outs[l * ncols + p] describes output column p of local
 * shard l in that shard's own HBM (memkind DFDB_MEM_DEVICE; sized from dfdb_group_shard_counts / dfdb_group_shard_string_bytes).  Asynchronous on
 * the shards' engine streams; no byte crosses PCIe or xGMI.  Rank order = table order: shard r's rows follow shard r-1's. */
int32_t dfdb_group_materialize_device(dfdb_gquery* gq, dfdb_outcol* outs, int32_t ncols) {
  return gguard([&] {
    GNEEDQ(gq); if (ncols > 0) GNEED(outs);
    dfdb_group* g = gq->gt->g;
    for (int l = 0; l < g->nlocal(); l++)
      for (int32_t p = 0; p < ncols; p++)
        if (outs[(size_t)l * ncols + p].memkind != DFDB_MEM_DEVICE) fail(DFDB_ERR_ARGUMENT, "dfdb_group_materialize_device writes device buffers (dfdb_group_materialize for host buffers)");
    plan_stage_bases(gq);
    for_shards(g, [&](int l) { shard_materialize(gq->shard[(size_t)l], outs + (size_t)l * ncols, ncols); });
  });
}

}  // extern "C"

// ------------------------------------------------------------------ unique / groupreduce over the shards
namespace dfdb {
namespace {
void put_i64(std::vector<uint8_t>& b, int64_t v) { const size_t o = b.size(); b.resize(o + 8); memcpy(b.data() + o, &v, 8); }
void put_vec(std::vector<uint8_t>& b, const void* p, size_t n) { put_i64(b, (int64_t)n); const size_t o = b.size(); b.resize(o + n); if (n) memcpy(b.data() + o, p, n); }
std::vector<uint8_t> pack_part(const GroupPart& p) {
  std::vector<uint8_t> b;
  put_i64(b, p.ng);
  put_vec(b, p.key_data.data(), p.key_data.size()); put_vec(b, p.key_missing.data(), p.key_missing.size()); put_vec(b, p.key_bytes.data(), p.key_bytes.size());
  put_vec(b, p.counts.data(), p.counts.size() * 8); put_vec(b, p.vals.data(), p.vals.size() * 8);
  return b;
}
GroupPart unpack_part(const uint8_t* b, size_t n) {
  GroupPart p; size_t o = 0;
  auto i64 = [&]() { if (o + 8 > n) fail(DFDB_ERR_DEVICE, "group exchange: truncated record"); int64_t v; memcpy(&v, b + o, 8); o += 8; return v; };
  auto vec = [&](std::vector<uint8_t>& v) { const int64_t m = i64(); if (m < 0 || o + (size_t)m > n) fail(DFDB_ERR_DEVICE, "group exchange: truncated record"); v.assign(b + o, b + o + m); o += (size_t)m; };
  p.ng = i64();
  vec(p.key_data); vec(p.key_missing); vec(p.key_bytes);
  std::vector<uint8_t> c, v; vec(c); vec(v);
  p.counts.resize(c.size() / 8); if (!c.empty()) memcpy(p.counts.data(), c.data(), c.size());
  p.vals.resize(v.size() / 8); if (!v.empty()) memcpy(p.vals.data(), v.data(), v.size());
  return p;
}
}  // namespace

// every rank's parts, rank order.  One process that holds every shard has them already; one process per GPU all-gathers the packed records
// (first their sizes, then the bytes padded to the largest: they are small — one record per distinct key)
static std::vector<GroupPart> all_parts(dfdb_group* g, std::vector<GroupPart>& local) {
  // (group option "group_force_exchange" = 1 makes a process that holds every shard exchange its records all the same: how a 1-GPU box runs the
  // RCCL all-gather of this path for real, with a one-rank group)
  const bool forced = g->exchange == DFDB_EXCHANGE_RCCL && ctx_option(g->ctx[0], "group_force_exchange", 0) != 0;
  if (g->nlocal() == g->world && !forced) { settle_fault(g, g->fault_key); return std::move(local); }
  if (g->exchange == DFDB_EXCHANGE_CALLBACK) {
    std::vector<uint8_t> blob = pack_part(local[0]);
    HIP_CHECK(hipSetDevice(g->ctx[0]->device)); put_slot(g, 0, 1, (int64_t)blob.size());
    const std::vector<int64_t> sizes = exchange_gather(g, 1);          // (raises the fault the ranks agreed on, if a shard failed)
    int64_t maxb = 8;
    for (int64_t b : sizes) maxb = std::max(maxb, b);
    blob.resize((size_t)maxb, 0);
    std::vector<uint8_t> all((size_t)maxb * (size_t)g->world);
    const int32_t rc = g->fns.allgather(g->fns.user, blob.data(), all.data(), maxb);
    if (rc != 0) fail(DFDB_ERR_DEVICE, "the caller's allgather failed with %d", rc);
    std::vector<GroupPart> parts;
    for (int rk = 0; rk < g->world; rk++) parts.push_back(unpack_part(all.data() + (size_t)rk * (size_t)maxb, (size_t)sizes[(size_t)rk]));
    return parts;
  }
  if (g->exchange != DFDB_EXCHANGE_RCCL) fail(DFDB_ERR_DEVICE, "a host-exchange group holds every shard in one process");
  const int nl = g->nlocal();
  std::vector<std::vector<uint8_t>> blobs((size_t)nl);
  for (int l = 0; l < nl; l++) { blobs[(size_t)l] = pack_part(local[(size_t)l]); HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); put_slot(g, l, 1, (int64_t)blobs[(size_t)l].size()); }
  const std::vector<int64_t> sizes = exchange_gather(g, 1);          // (raises the fault the ranks agreed on, if a shard failed)
  int64_t maxb = 8;
  for (int64_t b : sizes) maxb = std::max(maxb, b);
  maxb = round_up(maxb, 8);
  Rccl& r = rccl();
  std::vector<DevBuf> send((size_t)nl), recv((size_t)nl);
  for (int l = 0; l < nl; l++) {
    HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device));
    send[(size_t)l].ensure((size_t)maxb); recv[(size_t)l].ensure((size_t)maxb * (size_t)g->world);
    blobs[(size_t)l].resize((size_t)maxb, 0);
    HIP_CHECK(hipMemcpyAsync(send[(size_t)l].p, blobs[(size_t)l].data(), (size_t)maxb, hipMemcpyHostToDevice, g->ctx[(size_t)l]->stream));
  }
  rccl_bracket(r, g->comm_state, [&] {
    for (int l = 0; l < nl; l++) RCCL_IN(r, r.AllGather(send[(size_t)l].p, recv[(size_t)l].p, (size_t)maxb, ncclInt8, g->comm[(size_t)l], g->ctx[(size_t)l]->stream));
  });
  std::vector<uint8_t> all((size_t)maxb * (size_t)g->world);
  HIP_CHECK(hipSetDevice(g->ctx[0]->device));
  HIP_CHECK(hipMemcpyAsync(all.data(), recv[0].p, all.size(), hipMemcpyDeviceToHost, g->ctx[0]->stream));
  for (int l = 0; l < nl; l++) { HIP_CHECK(hipSetDevice(g->ctx[(size_t)l]->device)); HIP_CHECK(hipStreamSynchronize(g->ctx[(size_t)l]->stream)); }   // send / recv / blobs die here
  std::vector<GroupPart> parts;
  for (int rk = 0; rk < g->world; rk++) parts.push_back(unpack_part(all.data() + (size_t)rk * (size_t)maxb, (size_t)sizes[(size_t)rk]));
  return parts;
}

static void group_reduce_all(dfdb_gquery* gq, int32_t key_p, int32_t val_p, int32_t op, bool with_stats) {
  dfdb_group* g = gq->gt->g;
  fresh(g);
  gq->merged = GroupMerged{};
  plan_stage_bases(gq);
  const int nl = g->nlocal();
  std::vector<GroupPart> local((size_t)nl);
  std::vector<int> kinds((size_t)nl, 0);
  for_shards_deferred(g, [&](int l) {
    dfdb_query* q = gq->shard[(size_t)l];
    GroupPart& part = local[(size_t)l];
    if (query_out_of_core(q)) { kinds[(size_t)l] = ooc_group_part(q, key_p, val_p, op, part); return; }   // (its chunks merged in chunk order: one part)
    int64_t ng = 0, kb = 0;
    query_groupreduce(q, key_p, val_p, op, &ng, &kb);     // the shard's own device reduction (k_unique.hip / k_dict.hip); the selection is the group's
    kinds[(size_t)l] = q->gr_kind;
    fetch_group_part(q, key_p, ng, kb, false, part);      // (puts the shard's full selection back)
  });
  const std::vector<GroupPart> parts = all_parts(g, local);
  const int32_t kdt = gq->shard[0]->proj[(size_t)key_p].expr->dtype;
  const int kind = kinds[0];
  GroupMerged& m = gq->merged;
  m.key_dtype = kdt; m.kind = kind; m.op = op; m.with_stats = with_stats;
  GroupMerger mg;
  for (const GroupPart& p : parts) mg.add(m, p);           // rank order = table order: a key keeps the place of its first appearance
  m.valid = true;
}

static void group_reduce_fetch(dfdb_gquery* gq, dfdb_outcol* keys, int64_t* counts, int64_t* vals_i, double* vals_f) {
  GroupMerged& m = gq->merged;
  if (!m.valid) fail(DFDB_ERR_ARGUMENT, "ArgumentError: dfdb_group_query_unique / _groupreduce has not been called");
  merged_fetch(m, keys, counts, vals_i, vals_f);
}
}  // namespace dfdb

extern "C" {
/* unique(col) over the whole table (Base.unique over Base.iterate(::DFColumn), column.jl:102-126; docs/src/index.md:171-182,479-487): every shard
 * reduces its own rows on its own device, one record per distinct key crosses to the other ranks (all-gather), and the records are merged by key in
 * rank order — first appearance = lowest rank, then lowest row — so the distinct values come out in Julia's order.  Call 1 returns their number and
 * the string bytes they need; the fetch copies them into a caller-owned HOST column.  Every rank gets the whole answer. */
int32_t dfdb_group_query_unique(dfdb_gquery* gq, int32_t proj_col, int64_t* ndistinct, int64_t* string_bytes) {
  return gguard([&] {
    GNEEDQ(gq);
    group_reduce_all(gq, proj_col, -1, DFDB_AGG_COUNT, false);
    if (ndistinct) *ndistinct = gq->merged.ng;
    if (string_bytes) *string_bytes = (int64_t)gq->merged.key_bytes.size();
  });
}
int32_t dfdb_group_query_unique_fetch(dfdb_gquery* gq, dfdb_outcol* keys) {
  return gguard([&] { GNEEDQ(gq); GNEED(keys); group_reduce_fetch(gq, keys, nullptr, nullptr, nullptr); gq->merged = GroupMerged{}; });
}
/* groupreduce(view, (:key,); out = :val => Stat()) over the whole table (aggregate.jl:1-36, completed as dfdb_query_groupreduce completes it):
 * per-shard groups merged by key in rank order; counts and sums add (Int sums wrap as on one device, Float64 sums are sums of the shards' sums:
 * the tolerance of DESIGN.md section 5), minimum / maximum fold with Julia's NaN and signed-zero rules.  Same two calls as the single-GPU form. */
int32_t dfdb_group_query_groupreduce(dfdb_gquery* gq, int32_t key_col, int32_t val_col, int32_t stat, int64_t* ngroups, int64_t* key_string_bytes) {
  return gguard([&] {
    GNEEDQ(gq);
    group_reduce_all(gq, key_col, val_col, stat, true);
    if (ngroups) *ngroups = gq->merged.ng;
    if (key_string_bytes) *key_string_bytes = (int64_t)gq->merged.key_bytes.size();
  });
}
int32_t dfdb_group_query_groupreduce_fetch(dfdb_gquery* gq, dfdb_outcol* keys, int64_t* counts, int64_t* values_i, double* values_f) {
  return gguard([&] { GNEEDQ(gq); group_reduce_fetch(gq, keys, counts, values_i, values_f); gq->merged = GroupMerged{}; });
}
/* host-side self-tests that need neither a GPU nor RCCL (include/dfdb.h) */
namespace {
int st_starts, st_ends, st_calls, st_fail_at;
ncclResult_t st_group_start() { st_starts++; return st_fail_at == -1 ? ncclInternalError : ncclSuccess; }
ncclResult_t st_group_end() { st_ends++; return st_fail_at == -2 ? ncclInternalError : ncclSuccess; }
ncclResult_t st_allreduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) { return ++st_calls == st_fail_at ? ncclInternalError : ncclSuccess; }
const char* st_errstr(ncclResult_t) { return "stub failure"; }
}  // namespace
int32_t dfdb_selftest(const char* name, int64_t arg, int64_t* out, int32_t nout) {
  return gguard([&] {
    GNEED(name); GNEED(out);
    if (!strcmp(name, "rccl_bracket")) {
      // the collective bracket of every RCCL exchange, driven by a stub table: three all-reduces, number `arg` fails (1-based; 0 = none, -1 = ncclGroupStart,
      // -2 = ncclGroupEnd); then a SECOND exchange on the same state.  out: [GroupStart calls, GroupEnd calls, collectives issued, first status, second status,
      // communicator dead]
      if (nout < 6) fail(DFDB_ERR_ARGUMENT, "ArgumentError: rccl_bracket fills 6 values");
      Rccl r; r.GroupStart = st_group_start; r.GroupEnd = st_group_end; r.AllReduce = st_allreduce; r.GetErrorString = st_errstr;
      st_starts = st_ends = st_calls = 0; st_fail_at = (int)arg;
      CommState cs;
      auto one = [&]() -> int64_t {
        try { rccl_bracket(r, cs, [&] { for (int i = 0; i < 3; i++) RCCL_IN(r, r.AllReduce(nullptr, nullptr, 1, ncclInt64, ncclSum, nullptr, nullptr)); }); return 0; }
        catch (const Error& e) { set_last_error(e.what()); return e.code; }
      };
      const int64_t first = one();
      st_fail_at = 0;
      const int64_t second = one();
      out[0] = st_starts; out[1] = st_ends; out[2] = st_calls; out[3] = first; out[4] = second; out[5] = cs.dead ? 1 : 0;
      return;
    }
    fail(DFDB_ERR_ARGUMENT, "ArgumentError: no self-test named %s", name);
  });
}
}  // extern "C"
