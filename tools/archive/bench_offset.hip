// bench_offset.hip — is K1's allocation-to-allocation spread (1.25-1.45 ms) a function of WHERE the bitmap sits relative to the
// column?  One 8-GB column, one bitmap arena with slack; the scan (shape of the shipped k_scan_cmp: four tiles per trip, one 512-byte bitmap store)
// is timed with the bitmap at arena + k * step for several steps.  A pattern in k would mean the engine can pick the phase; noise means it cannot.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench_offset.hip -o tools/bench_offset
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void k_gen(int64_t* out, int64_t n) { for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (int64_t)(splitmix64(0x9E3779B97F4A7C15ull + (uint64_t)i) % 1000000ull); }
template <int WT>
__global__ __launch_bounds__(256) void k_scan4(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts, int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const int64_t ngroups = ntiles / 4;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    uint64_t my = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t* p = col + (g * 4 + k) * 1024 + lane;
      int64_t v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
      for (int j = 0; j < 16; j++) { uint64_t m = __ballot(v[j] > c); if (lane == 16 * k + j) my = m; }
    }
    uint32_t cn = (uint32_t)__popcll(my);
    for (int d = 8; d >= 1; d >>= 1) cn += __shfl_xor(cn, d, 64);
    if (WT) __hip_atomic_store(&bitmap[g * 64 + lane], my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else bitmap[g * 64 + lane] = my;
    if ((lane & 15) == 0) counts[g * 4 + (lane >> 4)] = cn;
  }
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int64_t n = 1000000000LL, ntiles = n / 1024 / 4 * 4;
  int64_t* col; CK(hipMalloc(&col, n * 8));
  const size_t slack = (size_t)160 << 20;
  uint8_t* arena; CK(hipMalloc(&arena, (size_t)ntiles * 128 + slack));
  uint32_t* cnt; CK(hipMalloc(&cnt, ntiles * 4 + 4096));
  hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, col, n); CK(hipDeviceSynchronize());
  printf("col %p  arena %p  counts %p\n", (void*)col, (void*)arena, (void*)cnt);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_at = [&](size_t off, int wt) {
    std::vector<float> ms;
    for (int r = 0; r < 7; r++) {
      CK(hipEventRecord(e0, nullptr));
      if (wt) hipLaunchKernelGGL(k_scan4<1>, dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, (uint64_t*)(arena + off), cnt, ntiles);
      else hipLaunchKernelGGL(k_scan4<0>, dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, (uint64_t*)(arena + off), cnt, ntiles);
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[3];
  };
  if (argc > 2) {
    // matrix: columns and bitmaps allocated alternately, every column timed against every bitmap
    const int m = atoi(argv[2]);
    std::vector<int64_t*> cols; std::vector<uint8_t*> bms;
    for (int k = 0; k < m; k++) {
      int64_t* c2; CK(hipMalloc(&c2, (size_t)n * 8)); cols.push_back(c2);
      hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, c2, n); CK(hipDeviceSynchronize());
      uint8_t* b2; CK(hipMalloc(&b2, (size_t)ntiles * 128 + 4096)); bms.push_back(b2);
    }
    printf("rows: column allocation, columns: bitmap allocation\n");
    uint8_t* keep_arena = arena; int64_t* keep_col = col;
    for (int i = 0; i < m; i++) {
      printf("col %2d %p:", i, (void*)cols[i]);
      for (int j = 0; j < m; j++) { col = cols[i]; arena = bms[j]; printf(" %.3f", time_at(0, 1)); }
      printf("\n");
    }
    col = keep_col; arena = keep_arena;
    return 0;
  }
  for (int wt = 1; wt >= 1; wt--) {
    printf("== %s bitmap stores\n", wt ? "write-through" : "plain");
    for (size_t step : {(size_t)4096, (size_t)8 << 20}) {
      printf("step %8zu:", step);
      const int nk = step >= ((size_t)8 << 20) ? 16 : 24;
      for (int k = 0; k < nk; k++) printf(" %.3f", time_at(step * (size_t)k, wt));
      printf("\n");
    }
  }
  // the same question for the COLUMN: a 9.5-GB arena, the column at arena + k * step (regenerated in place each time)
  {
    int64_t* big; CK(hipMalloc(&big, (size_t)n * 8 + ((size_t)1536 << 20)));
    printf("column arena %p\n", (void*)big);
    for (size_t step : {(size_t)2 << 20}) {
      printf("column step %9zu:", step);
      for (int k = 0; k < 16; k++) {
        int64_t* c2 = (int64_t*)((uint8_t*)big + step * (size_t)k);
        hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, c2, n); CK(hipDeviceSynchronize());
        int64_t* keep = col; col = c2; printf(" %.3f", time_at(0, 1)); col = keep;
      }
      printf("\n");
    }
    CK(hipFree(big));
    // fresh allocations held simultaneously (the round-1 observation), measured forward and then in reverse order: a per-allocation
    // effect repeats, a drift does not
    std::vector<int64_t*> held; std::vector<float> fwd;
    const int na = argc > 1 ? atoi(argv[1]) : 20;
    for (int k = 0; k < na; k++) {
      int64_t* c2; if (hipMalloc(&c2, (size_t)n * 8) != hipSuccess) break; held.push_back(c2);
      hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, c2, n); CK(hipDeviceSynchronize());
      int64_t* keep = col; col = c2; fwd.push_back(time_at(0, 1)); col = keep;
    }
    printf("allocation  address          forward  reverse   read-only\n");
    std::vector<float> rev(held.size());
    for (int k = (int)held.size() - 1; k >= 0; k--) { int64_t* keep = col; col = held[k]; rev[k] = time_at(0, 1); col = keep; }
    for (size_t k = 0; k < held.size(); k++) printf("%10zu  %p  %.3f    %.3f\n", k, (void*)held[k], fwd[k], rev[k]);
    for (auto* p : held) CK(hipFree(p));
  }
  // repeatability of one offset
  printf("repeat offset 0:"); for (int r = 0; r < 8; r++) printf(" %.3f", time_at(0, 1)); printf("\n");
  return 0;
}
