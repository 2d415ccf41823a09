// bench_gather.hip — what does a sparse 8-byte gather cost at the memory side on gfx950?  1e9-row uint64 column, one selected row per 10 rows
// (random offset inside its group of 10), several load flavours; run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum (and FETCH_SIZE)
// to see whether any flavour makes the L2 fetch less than the whole 128-byte line.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench_gather.hip -o tools/bench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void k_fill(uint64_t* p, int64_t n) { for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (uint64_t)i * 3u; }

template <int V> __device__ __forceinline__ uint64_t ld(const uint64_t* p) {
  if (V == 0) return *p;
  if (V == 1) return __builtin_nontemporal_load(p);
  uint64_t v;
  if (V == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (V == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (V == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (V == 5) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int V> __global__ __launch_bounds__(256) void k_gather(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, int64_t nout, int group) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nout; k += stride * 4) {
    uint64_t v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int64_t kk = k + u * stride; v[u] = kk < nout ? ld<V>(src + kk * group + (int64_t)(mix((uint64_t)kk) % (uint64_t)group)) : 0; }
#pragma unroll
    for (int u = 0; u < 4; u++) { const int64_t kk = k + u * stride; if (kk < nout) dst[kk] = v[u]; }
  }
}
template <int V> void run(const char* name, const uint64_t* src, uint64_t* dst, int64_t nout, int group) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<V>), dim3(8192), dim3(256), 0, 0, src, dst, nout, group);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-28s %.3f ms  (%.2f GB/s of selected+written bytes)\n", name, best, nout * 16.0 / best / 1e6);
}
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000000ll; const int group = argc > 2 ? atoi(argv[2]) : 10;
  const int64_t nout = n / group;
  uint64_t *src, *dst; CK(hipMalloc(&src, n * 8)); CK(hipMalloc(&dst, nout * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, src, n); CK(hipDeviceSynchronize());
  run<0>("plain", src, dst, nout, group); run<1>("nontemporal builtin", src, dst, nout, group); run<2>("sc1 (asm, serialised)", src, dst, nout, group);
  run<3>("sc0 sc1 nt (asm, serialised)", src, dst, nout, group); run<4>("sc0 (asm, serialised)", src, dst, nout, group); run<5>("nt (asm, serialised)", src, dst, nout, group);
  return 0;
}
