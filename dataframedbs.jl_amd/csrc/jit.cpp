// jit.cpp — expression kernels compiled at run time by hipRTC: the second tier of the interpreter.
//
// The reference never interprets a predicate: `BlockBroadcasting` hands Julia's compiler a fused broadcast per (function, argument types) and runs the
// machine code (src/tables/broadcast.jl:60-68; the block iterator even `precompile`s its executors, src/io/blocksiterator.jl:42).  Ahead-of-time HIP
// cannot do that, so an expression outside the specialised scan kernels starts in the device interpreter (k_interp.hip): no waiting, ~0.5 of the HBM
// peak for four or more instructions because every instruction is a wave-uniform decode + branch on the CU's one scalar unit.  Meanwhile this file
// compiles THE SAME SOURCE (k_interp_device.inc + k_interp_step.inc + k_interp_handlers.inc, embedded in the library at build time) for that one
// program shape — handler ids, operand sources, conversions, wrap widths, column slots and dtypes become literals, the dispatch loop becomes
// straight-line code — on a background thread; executions of the same shape use the compiled kernel as soon as it exists.  Constants of the expression
// (`x > 5` vs `x > 6`), pattern bytes and set members stay run-time data read from the program image, so they share one kernel.
//   ctx option "jit": 0 = never, 1 = in the background (default), 2 = wait for the compiler (tests, benchmarks)
//   ctx option "jit_min_rows": tables smaller than this are left to the interpreter (default 2^22 rows: below that a launch is microseconds either way)
// hipRTC is loaded lazily (dlopen); without it — or if a compile fails — the interpreter simply stays in charge: same results, its speed.
#include "engine.hpp"
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace dfdb {

// the interpreter's device source, as text (Makefile: build/jit_embed.inc from the four files themselves)
#include "build/jit_embed.inc"

namespace {
struct Rtc {
  void* h = nullptr;
  hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char**) = nullptr;
  hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
  hiprtcResult (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
  hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
  hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
  hiprtcResult (*DestroyProgram)(hiprtcProgram*) = nullptr;
  bool ok = false;
};
Rtc& rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) return;
    auto sym = [&](const char* n) { return dlsym(r.h, n); };
    r.CreateProgram = (decltype(r.CreateProgram))sym("hiprtcCreateProgram");
    r.CompileProgram = (decltype(r.CompileProgram))sym("hiprtcCompileProgram");
    r.GetProgramLogSize = (decltype(r.GetProgramLogSize))sym("hiprtcGetProgramLogSize");
    r.GetProgramLog = (decltype(r.GetProgramLog))sym("hiprtcGetProgramLog");
    r.GetCodeSize = (decltype(r.GetCodeSize))sym("hiprtcGetCodeSize");
    r.GetCode = (decltype(r.GetCode))sym("hiprtcGetCode");
    r.DestroyProgram = (decltype(r.DestroyProgram))sym("hiprtcDestroyProgram");
    r.ok = r.CreateProgram && r.CompileProgram && r.GetProgramLogSize && r.GetProgramLog && r.GetCodeSize && r.GetCode && r.DestroyProgram;
  });
  return r;
}

void table_fn(std::string& s, const char* type, const char* name, const std::vector<int64_t>& v, bool hex) {
  char buf[96];
  s += "__device__ constexpr "; s += type; s += " "; s += name; s += "(int i) {\n  switch (i) {\n";
  for (size_t i = 0; i < v.size(); i++) {
    if (hex) snprintf(buf, sizeof buf, "    case %zu: return 0x%llxu;\n", i, (unsigned long long)(uint32_t)v[i]);
    else snprintf(buf, sizeof buf, "    case %zu: return %lld;\n", i, (long long)v[i]);
    s += buf;
  }
  s += "  }\n  return 0;\n}\n";
}
}  // namespace

struct JitKernel {
  std::string key, source, log, arch;      // arch: the gcnArchName the code object is compiled for (part of the key: one kernel per shape AND target)
  std::vector<char> code;
  std::atomic<int> state{0};                   // 0: queued / compiling, 1: ready, -1: failed
  double compile_ms = 0;
  std::mutex mu;                               // guards `loaded`
  std::unordered_map<int, hipFunction_t> loaded;   // device -> function (one module per device, kept for the life of the process)
};

void jit_shutdown();
namespace {
void jit_at_exit();
struct JitCache {
  std::mutex mu; std::condition_variable cv;
  std::unordered_map<std::string, std::shared_ptr<JitKernel>> map;
  std::deque<std::shared_ptr<JitKernel>> queue;
  std::thread worker; bool started = false;
  bool busy = false, stopping = false;          // the worker is inside hipRTC / the process is exiting (guarded by mu)
  std::atomic<int64_t> compiled{0}, failed{0}, from_disk{0};
};
constexpr size_t kMaxShapes = 4096;                                   // ~10-20 KB of code object each
JitCache& cache() { static JitCache* c = new JitCache; return *c; }   // (leaked on purpose: the worker may outlive static destruction)

// ---- the code objects on disk (round 4): a shape compiled once by any process of this user is read back by the next one (0.13-0.17 s of hipRTC per shape and
// process otherwise).  File name = two 64-bit FNV-1a hashes over everything the code depends on (the generated source, the embedded interpreter source, the
// target, the options); content = the code object followed by an 8-byte hash of it, written to a temporary name and renamed.  $DFDB_JIT_CACHE_DIR names the
// directory (default $XDG_CACHE_HOME/dfdb-jit or ~/.cache/dfdb-jit), DFDB_JIT_CACHE=0 turns the cache off; every failure (no home, no space, a torn file) just
// means compiling as before.
uint64_t fnv1a(const void* p, size_t n, uint64_t h) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001B3ull; } return h; }
// The cache directory is TRUSTED INPUT: a code object read from it runs inside this process's GPU context, and the FNV trailer only says the file is whole, not
// who wrote it.  So the directory must be a real directory (no symlink) owned by the effective user with no group / other write permission — anything else
// (a shared or world-writable $DFDB_JIT_CACHE_DIR, somebody else's directory) turns the cache off — and files are opened without following symlinks and must
// be regular files of the same owner.  *why (optional) says what was refused.
std::string disk_dir_checked(std::string* why) {
  auto no = [&](const std::string& w) { if (why) *why = w; return std::string(); };
  const char* off = getenv("DFDB_JIT_CACHE");
  if (off && off[0] == '0') return no("DFDB_JIT_CACHE=0");
  std::string d;
  if (const char* e = getenv("DFDB_JIT_CACHE_DIR")) d = e;
  else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/dfdb-jit";
  else if (const char* h = getenv("HOME")) { d = std::string(h) + "/.cache"; (void)mkdir(d.c_str(), 0700); d += "/dfdb-jit"; }
  if (d.empty()) return no("no DFDB_JIT_CACHE_DIR, XDG_CACHE_HOME or HOME");
  (void)mkdir(d.c_str(), 0700);
  struct stat st;
  if (lstat(d.c_str(), &st) != 0) return no(d + ": cannot be created or examined");
  if (!S_ISDIR(st.st_mode)) return no(d + ": not a directory (a symbolic link is not followed)");
  if (st.st_uid != geteuid()) return no(d + ": owned by another user");
  if (st.st_mode & (S_IWGRP | S_IWOTH)) return no(d + ": writable by group or others");
  return d;
}
std::string disk_dir() {
  std::string why;
  const std::string d = disk_dir_checked(&why);
  if (d.empty() && getenv("DFDB_JIT_DEBUG")) {
    static std::atomic<bool> said{false};
    if (!said.exchange(true)) fprintf(stderr, "[jit] no disk cache: %s\n", why.c_str());
  }
  return d;
}
std::string disk_path(const JitKernel& k, const std::string& arch, const char* const* opts, int nopts) {
  const std::string dir = disk_dir();
  if (dir.empty()) return "";
  uint64_t h1 = 0xCBF29CE484222325ull, h2 = 0x84222325CBF29CE4ull;
  auto mix = [&](const void* p, size_t n) { h1 = fnv1a(p, n, h1); h2 = fnv1a(p, n, h2 ^ 0x9E3779B97F4A7C15ull); };
  mix(k.source.data(), k.source.size());
  for (const char* t : {src_device_utils_hpp, src_k_interp_handlers_inc, src_k_interp_step_inc, src_k_interp_device_inc, src_dfdb_ir_h}) mix(t, strlen(t));
  mix(arch.data(), arch.size());
  for (int i = 0; i < nopts; i++) mix(opts[i], strlen(opts[i]));
  char name[64];
  snprintf(name, sizeof name, "/%016llx%016llx.co", (unsigned long long)h1, (unsigned long long)h2);
  return dir + name;
}
bool disk_read(const std::string& path, std::vector<char>& code) {
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) { close(fd); return false; }   // not ours: never loaded, never touched
  std::vector<char> buf;
  char chunk[65536]; ssize_t n;
  while ((n = read(fd, chunk, sizeof chunk)) > 0) buf.insert(buf.end(), chunk, chunk + n);
  close(fd);
  if (n < 0 || buf.size() <= 8) return false;
  uint64_t want; memcpy(&want, buf.data() + buf.size() - 8, 8);
  buf.resize(buf.size() - 8);
  if (fnv1a(buf.data(), buf.size(), 0xCBF29CE484222325ull) != want) { (void)unlink(path.c_str()); return false; }      // a torn file of ours: out of the way
  code.swap(buf);
  return true;
}
void disk_write(const std::string& path, const std::vector<char>& code) {
  char tmp[32]; snprintf(tmp, sizeof tmp, ".%d.tmp", (int)getpid());
  const std::string t = path + tmp;
  const int fd = open(t.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  if (fd < 0) return;
  const uint64_t h = fnv1a(code.data(), code.size(), 0xCBF29CE484222325ull);
  auto put = [&](const void* p, size_t n) { const char* b = (const char*)p; while (n) { const ssize_t w = write(fd, b, n); if (w <= 0) return false; b += w; n -= (size_t)w; } return true; };
  const bool ok = put(code.data(), code.size()) && put(&h, 8);
  if (close(fd) != 0 || !ok || rename(t.c_str(), path.c_str()) != 0) (void)unlink(t.c_str());
}

void compile_one(JitKernel& k) {
  const std::string& arch = k.arch;
  Rtc& r = rtc();
  const auto t0 = std::chrono::steady_clock::now();
  const std::string archopt = "--offload-arch=" + arch;
  // -ffp-contract=off: the engine's floating-point results are Julia's, operation by operation (the library itself is built with it, see the Makefile)
  const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-pass-failed"};
  const std::string path = disk_path(k, arch, opts, 5);
  if (!path.empty() && disk_read(path, k.code)) {
    k.compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("DFDB_JIT_DEBUG")) fprintf(stderr, "[jit] %zu bytes from %s in %.1f ms: %s\n", k.code.size(), path.c_str(), k.compile_ms, k.key.c_str());
    else { std::string().swap(k.source); std::string().swap(k.log); }
    cache().from_disk++; k.state = 1; return;
  }
  hiprtcProgram prog = nullptr;
  const char* headers[] = {src_device_utils_hpp, src_k_interp_handlers_inc, src_k_interp_step_inc, src_k_interp_device_inc, src_dfdb_ir_h, nullptr};
  const char* names[] = {"device_utils.hpp", "k_interp_handlers.inc", "k_interp_step.inc", "k_interp_device.inc", "dfdb_ir.h", "jit_steps.inc"};
  // the per-program include (`#define PC n / #include "k_interp_step.inc"` for every instruction) travels at the front of the source, up to a marker line
  const size_t cut = k.source.find("//@@STEPS-END\n");
  const std::string steps = k.source.substr(0, cut), main_src = k.source.substr(cut + 14);
  headers[5] = steps.c_str();
  if (r.CreateProgram(&prog, main_src.c_str(), "dfdb_jit.hip", 6, headers, names) != HIPRTC_SUCCESS) { k.log = "hiprtcCreateProgram failed"; k.state = -1; return; }
  const hiprtcResult rc = r.CompileProgram(prog, 5, opts);
  size_t n = 0;
  if (r.GetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) { k.log.resize(n); r.GetProgramLog(prog, &k.log[0]); }
  if (rc == HIPRTC_SUCCESS && r.GetCodeSize(prog, &n) == HIPRTC_SUCCESS && n > 0) {
    k.code.resize(n);
    if (r.GetCode(prog, k.code.data()) != HIPRTC_SUCCESS) k.code.clear();
  }
  r.DestroyProgram(&prog);
  k.compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (k.code.empty()) {
    if (getenv("DFDB_JIT_DEBUG")) fprintf(stderr, "[jit] compile FAILED (%.0f ms) for %s:\n%s\n", k.compile_ms, k.key.c_str(), k.log.c_str());
    cache().failed++; k.state = -1; return;
  }
  if (!path.empty()) disk_write(path, k.code);
  if (getenv("DFDB_JIT_DEBUG")) fprintf(stderr, "[jit] compiled %zu bytes in %.0f ms: %s\n", k.code.size(), k.compile_ms, k.key.c_str());
  else { std::string().swap(k.source); std::string().swap(k.log); }      // (60 KB of text per shape: only the code object is kept)
  cache().compiled++; k.state = 1;
}

void worker_main() {
  JitCache& c = cache();
  bool registered_after_compile = false;
  for (;;) {
    std::shared_ptr<JitKernel> k;
    {
      std::unique_lock<std::mutex> lk(c.mu);
      c.cv.wait(lk, [&] { return !c.queue.empty() || c.stopping; });
      if (c.stopping) return;
      k = c.queue.front(); c.queue.pop_front();
      c.busy = true;
    }
    compile_one(*k);
    // ONE more registration, after the first compile: LLVM builds its statics (and registers their destructors) during that compile, so only a handler
    // registered afterwards runs before them at exit().  Once: atexit slots are finite, and a process compiles up to kMaxShapes shapes.
    if (!registered_after_compile) { registered_after_compile = true; std::atexit(jit_at_exit); }
    { std::lock_guard<std::mutex> lk(c.mu); c.busy = false; }      // (a waiter that has just found state == 0 is inside cv.wait by now: the notification cannot slip past it)
    c.cv.notify_all();
  }
}
// exit(): the compiler thread must not be inside hipRTC / comgr while their static objects are destroyed.  Registered after hipRTC was loaded, so it runs before
// their own exit handlers: pending shapes are dropped, a compile in flight is given ten seconds to finish.
void jit_at_exit() { jit_shutdown(); }
}  // namespace

// No more compiles: pending shapes are dropped, a compile in flight is given ten seconds to finish, later requests are answered "interpret".  The host
// binding calls this (dfdb_shutdown) from ITS exit hook — Python's atexit, Julia's atexit — which runs before the C runtime starts destroying static
// objects.  The std::atexit handler below is only the second line: LLVM registers the destructors of its lazily built statics during the first compile,
// i.e. AFTER this library could register anything, so at exit() they run BEFORE a handler of ours and the compiler thread dies in the rubble
// ("LLVM ERROR: Invalid size request on a scalable vector", SIGSEGV: a third of the exits that caught the compiler busy on the boxes of round 4).
void jit_shutdown() {
  JitCache& c = cache();
  std::unique_lock<std::mutex> lk(c.mu);
  c.stopping = true;
  for (auto& k : c.queue) k->state = -1;
  c.queue.clear();
  c.cv.notify_all();
  // a compile in flight is waited for: 0.1-0.3 s as a rule.  The bound is only there so that a wedged compiler cannot keep a process from ending for ever;
  // ten seconds (round 4) could run out on a loaded box and leave the worker inside LLVM while its statics were destroyed
  c.cv.wait_for(lk, std::chrono::seconds(120), [&] { return !c.busy; });
}

// the directory the code-object cache uses, "" when it is off or was refused (*why then says which check failed): dfdb_jit_cache_dir
std::string jit_cache_dir(std::string* why) { return disk_dir_checked(why); }

// the kernel for this program shape: ready, or nullptr (still compiling, hipRTC missing, or the compile failed).  wait = true blocks until the compiler is done.
std::shared_ptr<JitKernel> jit_request(dfdb_ctx* ctx, const JitShape& sh, bool wait) {
  if (!rtc().ok) return nullptr;
  // ---- the key: every literal of the generated source
  std::string key;
  {
    char b[64];
    key += ctx->prop.gcnArchName; key += "|";      // a code object is one target's: on a box with mixed parts every target compiles its own
    snprintf(b, sizeof b, "m%d s%d n%d a%d l%d r%d t%d|", sh.mode, sh.str, sh.nul, sh.and_existing, sh.stack_levels, sh.result_dtype, sh.nstr); key += b;
    for (size_t i = 0; i < sh.w0.size(); i++) { snprintf(b, sizeof b, "%x.%x.%x.%d.%d;", sh.w0[i], sh.w1[i], sh.w2[i], sh.slot[i], sh.aslot[i]); key += b; }
    key += "|";
    for (int32_t d : sh.col_dtype) { snprintf(b, sizeof b, "%x,", d); key += b; }
    key += "|";
    for (int i = 0; i < sh.nstr; i++) { snprintf(b, sizeof b, "%d,", sh.str_slot[i]); key += b; }
  }
  JitCache& c = cache();
  std::shared_ptr<JitKernel> k;
  {
    std::unique_lock<std::mutex> lk(c.mu);
    auto it = c.map.find(key);
    if (it != c.map.end()) k = it->second;
    else if (c.map.size() >= kMaxShapes || c.stopping) return nullptr;            // a process that has seen this many different program shapes keeps interpreting new ones
    else {
      k = std::make_shared<JitKernel>();
      k->key = key;
      k->arch = ctx->prop.gcnArchName;
      // ---- the source
      std::string s;
      char b[160];
      // a column slot that more than one instruction loads (as the fused A operand: aslot - 1, or as a B_COL operand: slot with bsrc == 2) is loaded once and
      // kept: the first loader saves into jcN, the later ones copy from it.  String columns are read by their own handlers, not through these loads.
      std::vector<int> uses(sh.col_dtype.size(), 0), seen(sh.col_dtype.size(), 0);
      for (size_t i = 0; i < sh.w0.size(); i++) {
        if (sh.aslot[i] > 0 && (size_t)(sh.aslot[i] - 1) < uses.size()) uses[(size_t)sh.aslot[i] - 1]++;
        if (((sh.w0[i] >> 8) & 0xff) == 2 && sh.slot[i] >= 0 && (size_t)sh.slot[i] < uses.size()) uses[(size_t)sh.slot[i]]++;
      }
      std::string decls;
      for (size_t c = 0; c < uses.size(); c++) if (uses[c] >= 2) { snprintf(b, sizeof b, "uint64_t jc%zu[kW]; ", c); decls += b; }
      for (size_t i = 0; i < sh.w0.size(); i++) {
        std::string pre, post;
        auto mark = [&](int col, const char* which) {
          if (col < 0 || (size_t)col >= uses.size() || uses[(size_t)col] < 2) return;
          snprintf(b, sizeof b, "#define JIT_%s_%s jc%d\n", which, seen[(size_t)col] ? "FROM" : "SAVE", col); pre += b;
          snprintf(b, sizeof b, "#undef JIT_%s_%s\n", which, seen[(size_t)col] ? "FROM" : "SAVE"); post += b;
          seen[(size_t)col] = 1;
        };
        if (sh.aslot[i] > 0) mark(sh.aslot[i] - 1, "A");            // (the A load comes first inside a step)
        if (((sh.w0[i] >> 8) & 0xff) == 2) mark(sh.slot[i], "B");
        snprintf(b, sizeof b, "#define PC %zu\n{\n#include \"k_interp_step.inc\"\n}\n#undef PC\n", i);
        s += pre; s += b; s += post;
      }
      s += "//@@STEPS-END\n";
      s += "#define JIT_CACHE_DECLS " + decls + "\n";
      snprintf(b, sizeof b, "#define DFDB_JIT 1\n#define DFDB_JIT_AND_EXISTING %d\n#define DFDB_JIT_STACK_LEVELS %d\n#define DFDB_JIT_RESULT_DTYPE %d\n#define DFDB_JIT_NSTR %d\n",
               sh.and_existing, sh.stack_levels, sh.result_dtype, sh.nstr); s += b;
      // hipRTC keeps the fixed-width integer types in a namespace of its own: name them as <cstdint> does on this ABI (int64_t is long)
      s += "typedef signed char int8_t; typedef short int16_t; typedef int int32_t; typedef long int64_t;\n"
           "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef unsigned long uint64_t;\n"
           "typedef unsigned long uintptr_t;\n";
      s += "#ifndef INT64_MIN\n#define INT64_MIN (-9223372036854775807L - 1)\n#endif\n";
      s += "#include \"dfdb_ir.h\"\n#include \"device_utils.hpp\"\n";
      std::vector<int64_t> v;
      auto tab = [&](const char* type, const char* name, auto&& src, bool hex) { v.clear(); for (auto x : src) v.push_back((int64_t)x); table_fn(s, type, name, v, hex); };
      tab("unsigned", "jit_w0", sh.w0, true); tab("unsigned", "jit_w1", sh.w1, true); tab("unsigned", "jit_w2", sh.w2, true);
      tab("int", "jit_slot", sh.slot, false); tab("int", "jit_aslot", sh.aslot, false); tab("int", "jit_col_dtype", sh.col_dtype, false);
      { std::vector<int32_t> ss(sh.str_slot, sh.str_slot + 4); tab("int", "jit_str_slot", ss, false); }
      s += "namespace dfdb {\n#include \"k_interp_device.inc\"\n}\n";
      snprintf(b, sizeof b, "  dfdb::interp_body<%d, %s, %s>(prog, bitmap, tile_counts, prefix, out, out_cap, nrows, ntiles, and_existing, err, stack_levels, out_missing);\n",
               sh.mode, sh.str ? "true" : "false", sh.nul ? "true" : "false");
      s += "extern \"C\" __global__ __launch_bounds__(256) void dfdb_jit_kernel(const dfdb::IProgram* __restrict__ prog, uint64_t* __restrict__ bitmap,\n"
           "    uint32_t* __restrict__ tile_counts, const uint64_t* __restrict__ prefix, void* __restrict__ out, int64_t out_cap, int64_t nrows, int64_t ntiles,\n"
           "    int and_existing, int* __restrict__ err, int stack_levels, uint8_t* __restrict__ out_missing) {\n";
      s += b;
      s += "}\n";
      k->source = std::move(s);
      c.map.emplace(key, k);
      c.queue.push_back(k);
      if (!c.started) { c.started = true; c.worker = std::thread(worker_main); c.worker.detach(); std::atexit(jit_at_exit); }
      c.cv.notify_all();
    }
    if (wait) c.cv.wait(lk, [&] { return k->state.load() != 0; });
  }
  return k->state.load() == 1 ? k : nullptr;
}

// launch on ctx's stream; the module is loaded on this device the first time the kernel is used there.  false: the module could not be loaded (the caller interprets)
bool jit_launch(JitKernel& k, dfdb_ctx* ctx, unsigned grid, size_t lds_bytes, void** args) {
  hipFunction_t fn = nullptr;
  {
    std::lock_guard<std::mutex> lk(k.mu);
    auto it = k.loaded.find(ctx->device);
    if (it != k.loaded.end()) fn = it->second;
    else {
      hipModule_t mod = nullptr;
      if (hipModuleLoadData(&mod, k.code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "dfdb_jit_kernel") != hipSuccess) {
        (void)hipGetLastError();
        if (getenv("DFDB_JIT_DEBUG")) fprintf(stderr, "[jit] device %d (%s) cannot load the code object compiled for %s: this shape stays interpreted there\n", ctx->device, ctx->prop.gcnArchName, k.arch.c_str());
        k.loaded.emplace(ctx->device, nullptr);
        return false;
      }
      k.loaded.emplace(ctx->device, fn);
    }
  }
  if (!fn) return false;
  HIP_CHECK(hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, (unsigned)lds_bytes, ctx->stream, args, nullptr));
  return true;
}

void jit_stats(int64_t* compiled, int64_t* failed, int64_t* pending, int64_t* from_disk) {
  JitCache& c = cache();
  std::lock_guard<std::mutex> lk(c.mu);
  if (from_disk) *from_disk = c.from_disk.load();
  if (compiled) *compiled = c.compiled.load();
  if (failed) *failed = c.failed.load();
  if (pending) *pending = (int64_t)c.queue.size();
}
const char* jit_source_of(JitKernel& k) { return k.source.c_str(); }
const char* jit_log_of(JitKernel& k) { return k.log.c_str(); }

}  // namespace dfdb
