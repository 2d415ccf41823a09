// r3_k1_phases.hip — does it help K1 to keep its bitmap on chip and write it in chip-wide bursts?
// The scan's 125 MB of bitmap are 1.5 % of its traffic and cost 7-19 % of its time; captured values cost the same share of a scan
// however long the scan is (tools/r3_scan2.py).  If what hurts is writes MIXED into the read stream, writing them in short bursts that every CU issues at
// the same moment should give most of it back.  Variants, interleaved rounds in one process:
//   direct     the shipped form: four tiles per trip, one write-through 512-byte bitmap store + four counts per trip
//   nostore    the same loads and ballots, nothing written (the ceiling)
//   lds full   bitmap words and counts collect in LDS (NB trips per wave) and leave when the buffer is full (bursts per wave, not aligned)
//   lds time   the same, but every wave also flushes when the 100-MHz real-time counter crosses a multiple of the period: aligned bursts chip-wide
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/r3_k1_phases.hip -o tools/r3_k1_phases
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__global__ void k_gen(int64_t* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (int64_t)(splitmix64(0x9E3779B97F4A7C15ull + (uint64_t)i) % 1000000ull);
}

// MODE 0 direct, 1 nostore, 2 lds full, 3 lds time
template <int MODE, int NB>
__global__ __launch_bounds__(256) void k_scan(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts,
                                              int64_t ngroups, uint32_t period_ticks, uint64_t* __restrict__ sink) {
  __shared__ uint64_t wbuf[MODE >= 2 ? 4 * NB * 64 : 1];
  __shared__ uint32_t cbuf[MODE >= 2 ? 4 * NB * 4 : 1];
  __shared__ int64_t gbuf[MODE >= 2 ? 4 * NB : 1];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  uint64_t* wb = wbuf + (MODE >= 2 ? wib * NB * 64 : 0);
  uint32_t* cb = cbuf + (MODE >= 2 ? wib * NB * 4 : 0);
  int64_t* gb = gbuf + (MODE >= 2 ? wib * NB : 0);
  const int64_t wave = (int64_t)blockIdx.x * 4 + wib, nwaves = (int64_t)gridDim.x * 4;
  uint32_t nbuf = 0;
  uint64_t last_phase = MODE == 3 ? __builtin_amdgcn_s_memrealtime() / period_ticks : 0;
  uint64_t acc = 0;
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t k = 0; k < nbuf; k++) {
      const int64_t g = gb[k];
      __hip_atomic_store(&bitmap[g * 64 + lane], wb[k * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (lane < 4) counts[g * 4 + lane] = cb[k * 4 + lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    nbuf = 0;
  };
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    uint64_t my = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t* p = col + (g * 4 + k) * 1024 + lane;
      int64_t v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
      for (int j = 0; j < 16; j++) { const uint64_t m = __ballot(v[j] > c); if (lane == 16 * k + j) my = m; }
    }
    uint32_t cnt = (uint32_t)__popcll(my);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (MODE == 0) {
      __hip_atomic_store(&bitmap[g * 64 + lane], my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((lane & 15) == 0) counts[g * 4 + (lane >> 4)] = cnt;
    } else if (MODE == 1) {
      acc += my + cnt;
    } else {
      wb[nbuf * 64 + lane] = my;
      if ((lane & 15) == 0) cb[nbuf * 4 + (lane >> 4)] = cnt;
      if (lane == 0) gb[nbuf] = g;
      nbuf++;
      bool doit = nbuf == NB;
      if (MODE == 3) {
        const uint64_t ph = __builtin_amdgcn_s_memrealtime() / period_ticks;
        if (ph != last_phase) { doit = true; last_phase = ph; }
      }
      if (doit) flush();
    }
  }
  if (MODE >= 2 && nbuf) flush();
  if (MODE == 1 && acc == 0x1234567ull) sink[0] = acc;
}

__global__ void k_checksum(const uint64_t* bm, int64_t nw, const uint32_t* cnt, int64_t nt, unsigned long long* out) {
  unsigned long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nw; i += (int64_t)gridDim.x * 256) s += bm[i] * (unsigned long long)(2 * i + 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nt; i += (int64_t)gridDim.x * 256) s += cnt[i] * (unsigned long long)(2 * i + 3);
  atomicAdd(out, s);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000000LL;
  const int rounds = argc > 2 ? atoi(argv[2]) : 9;
  const int64_t ngroups = n / 4096, ntiles = ngroups * 4;
  int64_t* col; uint64_t* bm; uint32_t* cnt; unsigned long long* cs; uint64_t* sink;
  CK(hipMalloc(&col, n * 8 + 4096)); CK(hipMalloc(&bm, ntiles * 128 + 4096)); CK(hipMalloc(&cnt, ntiles * 4 + 64)); CK(hipMalloc(&cs, 8)); CK(hipMalloc(&sink, 64));
  hipLaunchKernelGGL(k_gen, dim3(8192), dim3(256), 0, 0, col, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { const char* name; int mode, nb; uint32_t period_us; std::vector<float> ms; unsigned long long sum; };
  std::vector<V> vs = {
      {"direct", 0, 0, 0}, {"nostore", 1, 0, 0},
      {"lds full NB=4", 2, 4, 0}, {"lds full NB=8", 2, 8, 0},
      {"lds time NB=8 T=100us", 3, 8, 100}, {"lds time NB=8 T=200us", 3, 8, 200}, {"lds time NB=8 T=50us", 3, 8, 50},
      {"lds time NB=4 T=100us", 3, 4, 100}, {"lds time NB=4 T=50us", 3, 4, 50}, {"lds time NB=4 T=25us", 3, 4, 25},
  };
  const int64_t c = 899999;
  const int grid = 2048;
  for (int r = 0; r < rounds + 1; r++) {
    for (auto& v : vs) {
      CK(hipMemsetAsync(cs, 0, 8, 0));
      if (r == rounds) { CK(hipMemsetAsync(bm, 0, ntiles * 128, 0)); CK(hipMemsetAsync(cnt, 0, ntiles * 4, 0)); }
      const uint32_t ticks = v.period_us * 100u;
      CK(hipEventRecord(e0, 0));
      if (v.mode == 0) hipLaunchKernelGGL((k_scan<0, 1>), dim3(grid), dim3(256), 0, 0, col, c, bm, cnt, ngroups, 1u, sink);
      else if (v.mode == 1) hipLaunchKernelGGL((k_scan<1, 1>), dim3(grid), dim3(256), 0, 0, col, c, bm, cnt, ngroups, 1u, sink);
      else if (v.mode == 2 && v.nb == 4) hipLaunchKernelGGL((k_scan<2, 4>), dim3(grid), dim3(256), 0, 0, col, c, bm, cnt, ngroups, 1u, sink);
      else if (v.mode == 2) hipLaunchKernelGGL((k_scan<2, 8>), dim3(grid), dim3(256), 0, 0, col, c, bm, cnt, ngroups, 1u, sink);
      else if (v.nb == 4) hipLaunchKernelGGL((k_scan<3, 4>), dim3(grid), dim3(256), 0, 0, col, c, bm, cnt, ngroups, ticks, sink);
      else hipLaunchKernelGGL((k_scan<3, 8>), dim3(grid), dim3(256), 0, 0, col, c, bm, cnt, ngroups, ticks, sink);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms);
      if (r == rounds) {
        hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, 0, bm, ntiles * 16, cnt, ntiles, cs);
        CK(hipMemcpy(&v.sum, cs, 8, hipMemcpyDeviceToHost));
      }
    }
  }
  printf("rows=%lld rounds=%d grid=%d\n", (long long)n, rounds, grid);
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    printf("%-26s median %.4f ms  min %.4f ms  checksum %016llx\n", v.name, v.ms[v.ms.size() / 2], v.ms[0], v.sum);
  }
  return 0;
}
