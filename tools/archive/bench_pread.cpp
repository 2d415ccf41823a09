// How fast can the host move a page-cached file into pinned memory?  T threads pread distinct 32-MB slices of a /dev/shm file (the loaders' reads), T = 4 .. 96.
// g++ -O2 -pthread tools/bench_pread.cpp -o tools/bench_pread -I/opt/rocm/include -L/opt/rocm/lib -lamdhip64 -D__HIP_PLATFORM_AMD__
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atol(argv[1]) : 4096, n = mb << 20, piece = 32u << 20;
  const char* path = "/dev/shm/dfdb_pread_test.bin";
  { int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600); std::vector<char> buf(1 << 24, 7); for (size_t o = 0; o < n; o += buf.size()) { for (size_t i = 0; i < buf.size(); i += 4096) buf[i] = (char)(o >> 12); if (write(fd, buf.data(), buf.size()) < 0) return 2; } close(fd); }
  void* pin = nullptr;
  if (hipHostMalloc(&pin, n, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
  memset(pin, 1, n);
  int fd = open(path, O_RDONLY);
  for (int T : {4, 8, 16, 24, 32, 48, 64, 96}) {
    for (int rep = 0; rep < 2; rep++) {
      std::atomic<size_t> next{0};
      std::vector<std::thread> th;
      const double t0 = now();
      for (int k = 0; k < T; k++) th.emplace_back([&] {
        for (;;) { const size_t o = next.fetch_add(piece); if (o >= n) break; size_t got = 0; while (got < piece) { ssize_t r = pread(fd, (char*)pin + o + got, piece - got, o + got); if (r <= 0) return; got += r; } }
      });
      for (auto& t : th) t.join();
      const double t1 = now();
      if (rep) printf("T=%2d threads: %.1f ms = %.1f GB/s\n", T, t1 - t0, n / (t1 - t0) / 1e6);
    }
  }
  close(fd); unlink(path);
  return 0;
}
