// Can the DMA engines read a column file's page-cache pages directly?  mmap a /dev/shm file, hipHostRegister the mapping, time the registration and an H2D
// copy out of it, against pread -> pinned bounce buffer -> H2D.  hipcc -O2 tools/bench_hostreg.hip -o tools/bench_hostreg && tools/bench_hostreg [MB]
#include <hip/hip_runtime.h>
#include <vector>
#include <thread>
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
  // unregistered mapping straight into hipMemcpy (the runtime stages it)
  { void* m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0); double t0 = now(); CK(hipMemcpy(dev, m, n, hipMemcpyHostToDevice)); double t1 = now(); printf("H2D from a plain mapping %.1f ms (%.1f GB/s)\n", t1 - t0, n / (t1 - t0) / 1e6); munmap(m, n); }
  close(fd); unlink(path);
  return 0;
}
