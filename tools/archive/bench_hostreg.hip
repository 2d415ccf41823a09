// Can the DMA engines read a column file's page-cache pages directly?  mmap a /dev/shm file, hipHostRegister the mapping, time the registration and an H2D
// copy out of it, against pread -> pinned bounce buffer -> H2D.  hipcc -O2 tools/bench_hostreg.hip -o tools/bench_hostreg && tools/bench_hostreg [MB]
#include <hip/hip_runtime.h>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atol(argv[1]) : 1024, n = mb << 20;
  const char* path = "/dev/shm/dfdb_hostreg_test.bin";
  { int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600); std::vector<char> buf(1 << 24, 7); for (size_t o = 0; o < n; o += buf.size()) { if (write(fd, buf.data(), buf.size()) < 0) return 2; } close(fd); }
  void* dev; CK(hipMalloc(&dev, n));
  void* pin; CK(hipHostMalloc(&pin, n, hipHostMallocDefault));
  int fd = open(path, O_RDONLY);
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now(); size_t got = 0; while (got < n) { ssize_t r = pread(fd, (char*)pin + got, n - got, got); if (r <= 0) return 3; got += r; } double t1 = now();
    CK(hipMemcpy(dev, pin, n, hipMemcpyHostToDevice)); double t2 = now();
    printf("pread 1 thread %.1f ms (%.1f GB/s), H2D from pinned %.1f ms (%.1f GB/s)\n", t1 - t0, n / (t1 - t0) / 1e6, t2 - t1, n / (t2 - t1) / 1e6);
  }
  for (int flags : {0, 1}) {
    void* m = mmap(nullptr, n, PROT_READ, MAP_SHARED | (flags ? MAP_POPULATE : 0), fd, 0);
    if (m == MAP_FAILED) { printf("mmap failed\n"); return 4; }
    double t0 = now();
    hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
    double t1 = now();
    if (e != hipSuccess) { printf("hipHostRegister(populate=%d) -> %s\n", flags, hipGetErrorString(e)); (void)hipGetLastError(); munmap(m, n); continue; }
    CK(hipMemcpy(dev, m, n, hipMemcpyHostToDevice)); double t2 = now();
    CK(hipMemcpy(dev, m, n, hipMemcpyHostToDevice)); double t3 = now();
    CK(hipHostUnregister(m)); double t4 = now();
    printf("populate=%d: register %.1f ms (%.1f GB/s), H2D %.1f ms (%.1f GB/s), again %.1f ms, unregister %.1f ms\n", flags, t1 - t0, n / (t1 - t0) / 1e6, t2 - t1, n / (t2 - t1) / 1e6, t3 - t2, t4 - t3);
    munmap(m, n);
  }
  // registration from several threads at once: does it scale?  (T threads, each registers its own 64-MB pieces of one mapping, copies them, unregisters)
  for (int T : {1, 2, 4, 8}) {
    void* m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
    const size_t piece = 64u << 20, np = n / piece;
    std::vector<std::thread> th; std::vector<double> reg_ms(T, 0), unreg_ms(T, 0);
    std::vector<hipStream_t> st(T);
    for (auto& s : st) CK(hipStreamCreate(&s));
    double t0 = now();
    for (int k = 0; k < T; k++) th.emplace_back([&, k] {
      (void)hipSetDevice(0);
      for (size_t p = k; p < np; p += T) {
        char* a = (char*)m + p * piece;
        double r0 = now();
        if (hipHostRegister(a, piece, hipHostRegisterDefault) != hipSuccess) { printf("register failed\n"); return; }
        double r1 = now(); reg_ms[k] += r1 - r0;
        (void)hipMemcpyAsync((char*)dev + p * piece, a, piece, hipMemcpyHostToDevice, st[k]);
        (void)hipStreamSynchronize(st[k]);
        double u0 = now(); (void)hipHostUnregister(a); unreg_ms[k] += now() - u0;
      }
    });
    for (auto& t : th) t.join();
    double t1 = now();
    printf("T=%d threads register+copy+unregister 64-MB pieces: %.1f ms total (%.1f GB/s); per thread: register %.1f ms, unregister %.1f ms\n", T, t1 - t0, n / (t1 - t0) / 1e6, reg_ms[0], unreg_ms[0]);
    for (auto& s : st) (void)hipStreamDestroy(s);
    munmap(m, n);
  }
  // what a streamed scan's copies can reach: L loaders, each its own stream and its own pinned buffer, 32-MB pieces queued back to back (no host copy at all),
  // then the same with R threads doing what the loaders' preads do (page cache -> another pinned buffer) beside the DMA
  for (int cfg = 0; cfg < 9; cfg++) {
    const int R = cfg < 3 ? (cfg == 0 ? 0 : cfg == 1 ? 8 : 24) : 0;
    const int L = cfg < 3 ? 3 : (cfg == 3 ? 1 : cfg == 4 ? 2 : cfg == 5 ? 1 : cfg == 6 ? 3 : cfg == 7 ? 7 : 2);
    const size_t piece = (cfg < 3 ? 32u : cfg == 3 ? 32u : cfg == 4 ? 32u : cfg == 5 ? 128u : cfg == 6 ? 128u : cfg == 7 ? 32u : 8u) << 20, per = n / L / piece * piece;
    std::vector<void*> pb(L); std::vector<hipStream_t> st(L);
    for (int k = 0; k < L; k++) { CK(hipHostMalloc(&pb[k], per, hipHostMallocDefault)); memset(pb[k], k + 1, per); CK(hipStreamCreate(&st[k])); }
    std::atomic<bool> stop{false}; std::atomic<long> copied{0};
    std::vector<std::thread> noise;
    void* sink = nullptr; if (R) CK(hipHostMalloc(&sink, (size_t)R << 25, hipHostMallocDefault));
    for (int r = 0; r < R; r++) noise.emplace_back([&, r] { char* d = (char*)sink + ((size_t)r << 25); size_t off = ((size_t)r << 25) % n; while (!stop) { ssize_t g = pread(fd, d, 1 << 25, off); if (g > 0) copied += g; off = (off + ((size_t)R << 25)) % (n - (1 << 25)); } });
    double t0 = now();
    for (int rep = 0; rep < 2; rep++)
      for (size_t o = 0; o < per; o += piece) for (int k = 0; k < L; k++) (void)hipMemcpyAsync((char*)dev + k * per + o, (char*)pb[k] + o, piece, hipMemcpyHostToDevice, st[k]);
    for (int k = 0; k < L; k++) (void)hipStreamSynchronize(st[k]);
    double t1 = now();
    stop = true; for (auto& t : noise) t.join();
    printf("%d loaders x %zu-MB pieces on their own streams, %d pread threads beside them: %.1f GB/s of H2D (preads moved %.1f GB/s meanwhile)\n", L, piece >> 20, R, 2.0 * L * per / (t1 - t0) / 1e6, copied / (t1 - t0) / 1e6);
    for (int k = 0; k < L; k++) { (void)hipHostFree(pb[k]); (void)hipStreamDestroy(st[k]); }
    if (sink) (void)hipHostFree(sink);
  }
  // unregistered mapping straight into hipMemcpy (the runtime stages it)
  { void* m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0); double t0 = now(); CK(hipMemcpy(dev, m, n, hipMemcpyHostToDevice)); double t1 = now(); printf("H2D from a plain mapping %.1f ms (%.1f GB/s)\n", t1 - t0, n / (t1 - t0) / 1e6); munmap(m, n); }
  close(fd); unlink(path);
  return 0;
}
