// bench_placement.hip — does the K1 scan's time depend on WHERE hipMalloc put the 8-GB column?  Twelve columns are allocated one after
// another (all kept, so each lands on different physical memory), the same kernel is timed on each, twice.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench_placement.hip -o tools/bench_placement
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
__global__ __launch_bounds__(256) void k_scan(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts, int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t* p = col + tile * 1024 + lane;
    int64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
    uint64_t my = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { uint64_t m = __ballot(v[j] > c); if (lane == j) my = m; }
    uint32_t cn = lane < 16 ? (uint32_t)__popcll(my) : 0u;
    for (int d = 8; d >= 1; d >>= 1) cn += __shfl_xor(cn, d, 64);
    if (lane < 16) bitmap[tile * 16 + lane] = my;
    if (lane == 0) counts[tile] = cn;
  }
}
// four consecutive tiles per trip: ONE 512-byte bitmap store (tile k's 16 words live in lanes 16k..16k+15) instead of four 128-byte ones
template <int WRITE>
__global__ __launch_bounds__(256) void k_scan4(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts, int64_t ntiles) {
  uint64_t keep = 0;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const int64_t ngroups = ntiles / 4;     // (harness: ntiles is a multiple of 4)
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
    for (int d = 8; d >= 1; d >>= 1) cn += __shfl_xor(cn, d, 64);      // every 16-lane group reduces its own tile
    if (WRITE & 1) bitmap[g * 64 + lane] = my;
    if ((WRITE & 2) && (lane & 15) == 0) counts[g * 4 + (lane >> 4)] = cn;
    if (WRITE == 0) keep += my + cn;
  }
  if (WRITE == 0 && keep == 0x123456789abcull) bitmap[0] = keep;
}
// G groups of 4 tiles per trip: the bitmap leaves in bursts of G x 512 bytes per wave; NT: nontemporal stores
template <int G, int ST>
__global__ __launch_bounds__(256) void k_scanG(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts, int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const int64_t nsuper = ntiles / (4 * G);
  for (int64_t sg = wave; sg < nsuper; sg += nwaves) {
    uint64_t my[G]; uint32_t cn[G];
#pragma unroll
    for (int q = 0; q < G; q++) {
      my[q] = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int64_t* p = col + ((sg * G + q) * 4 + k) * 1024 + lane;
        int64_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
        for (int j = 0; j < 16; j++) { uint64_t m = __ballot(v[j] > c); if (lane == 16 * k + j) my[q] = m; }
      }
      cn[q] = (uint32_t)__popcll(my[q]);
      for (int d = 8; d >= 1; d >>= 1) cn[q] += __shfl_xor(cn[q], d, 64);
    }
#pragma unroll
    for (int q = 0; q < G; q++) {
      uint64_t* d = bitmap + (sg * G + q) * 64 + lane;
      if (ST == 1) __builtin_nontemporal_store(my[q], d);
      else if (ST == 2) __hip_atomic_store(d, my[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else *d = my[q];
      if ((lane & 15) == 0) counts[(sg * G + q) * 4 + (lane >> 4)] = cn[q];
    }
  }
}
// PHASED: the bitmap of G tiles per wave stays in LDS while the whole grid reads; then a grid-wide barrier, every wave writes its
// G x 128 bytes, another barrier: the bitmap is never written INSIDE the read stream.  All workgroups must be resident.
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned nblocks, unsigned& epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch++;
    __threadfence();
    atomicAdd(ctr, 1u);
    for (int it = 0; it < (1 << 17) && atomicAdd(ctr, 0u) < epoch * nblocks; it++) __builtin_amdgcn_s_sleep(2);   // (bounded: a harness must not hang the box)
    __threadfence();
  }
  __syncthreads();
}
template <int G>
__global__ __launch_bounds__(256) void k_scan_phased(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts,
                                                     int64_t ntiles, unsigned* ctr) {
  __shared__ uint64_t bm_sh[4][G * 16];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wib, nwaves = (int64_t)gridDim.x * 4;
  unsigned epoch = 0;
  const int64_t per_phase = nwaves * G;
  for (int64_t t0 = 0; t0 < ntiles; t0 += per_phase) {
#pragma unroll 1
    for (int g = 0; g < G; g++) {
      const int64_t tile = t0 + (int64_t)g * nwaves + wave;
      if (tile < ntiles) {
        const int64_t* p = col + tile * 1024 + lane;
        int64_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
        uint64_t my = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) { uint64_t m = __ballot(v[j] > c); if (lane == j) my = m; }
        uint32_t cn = lane < 16 ? (uint32_t)__popcll(my) : 0u;
        for (int d = 8; d >= 1; d >>= 1) cn += __shfl_xor(cn, d, 64);
        if (lane < 16) bm_sh[wib][g * 16 + lane] = my;
        if (lane == 0) counts[tile] = cn;
      }
    }
    grid_barrier(ctr, gridDim.x, epoch);
#pragma unroll 1
    for (int g = 0; g < G; g++) {
      const int64_t tile = t0 + (int64_t)g * nwaves + wave;
      if (tile < ntiles && lane < 16) bitmap[tile * 16 + lane] = bm_sh[wib][g * 16 + lane];
    }
    grid_barrier(ctr, gridDim.x, epoch);
  }
}
__global__ __launch_bounds__(256) void k_read(const int64_t* __restrict__ col, int64_t ntiles, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  unsigned long long acc = 0;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t* p = col + tile * 1024 + lane;
    int64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = __builtin_nontemporal_load(p + j * 64);
#pragma unroll
    for (int j = 0; j < 16; j++) acc += (unsigned long long)v[j];
  }
  if (acc == 0x123456789abcull) out[0] = acc;
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int64_t n = 1000000000LL, ntiles = n / 1024;
  const int ncols = argc > 1 ? atoi(argv[1]) : 12;
  uint64_t* bm; uint32_t* cnt;
  const bool uncached = argc > 3 && atoi(argv[3]) != 0;
  if (uncached) CK(hipExtMallocWithFlags((void**)&bm, ntiles * 128 + 4096, hipDeviceMallocUncached)); else CK(hipMalloc(&bm, ntiles * 128 + 4096));
  CK(hipMalloc(&cnt, ntiles * 4 + 64));
  printf("bitmap buffer: %s\n", uncached ? "hipDeviceMallocUncached" : "hipMalloc");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<int64_t*> cols;
  auto time_one = [&](int64_t* col) {
    std::vector<float> ms;
    for (int r = 0; r < 7; r++) {
      CK(hipEventRecord(e0, nullptr));
      hipLaunchKernelGGL(k_scan, dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, ntiles);
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[3];
  };
  auto time_scan4 = [&](int64_t* col, int wr) {
    std::vector<float> ms;
    for (int r = 0; r < 7; r++) {
      CK(hipEventRecord(e0, nullptr));
      if (wr == 3) hipLaunchKernelGGL(k_scan4<3>, dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, ntiles / 4 * 4);
      else if (wr == 1) hipLaunchKernelGGL(k_scan4<1>, dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, ntiles / 4 * 4);
      else if (wr == 2) hipLaunchKernelGGL(k_scan4<2>, dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, ntiles / 4 * 4);
      else hipLaunchKernelGGL(k_scan4<0>, dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, ntiles / 4 * 4);
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[3];
  };
  auto time_k = [&](int64_t* col, int which) {
    std::vector<float> ms;
    const int64_t nt = ntiles / 64 * 64;
    for (int r = 0; r < 7; r++) {
      CK(hipEventRecord(e0, nullptr));
      switch (which) {
        case 0: hipLaunchKernelGGL((k_scanG<1, 1>), dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, nt); break;
        case 1: hipLaunchKernelGGL((k_scanG<4, 0>), dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, nt); break;
        case 2: hipLaunchKernelGGL((k_scanG<1, 2>), dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, nt); break;
        default: hipLaunchKernelGGL((k_scanG<4, 1>), dim3(2048), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, nt); break;
      }
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[3];
  };
  unsigned* gctr; CK(hipMalloc(&gctr, 64));
  int occ16 = 0, occ32 = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ16, k_scan_phased<16>, 256, 0));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ32, k_scan_phased<32>, 256, 0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("phased kernels: %d / %d resident workgroups per CU, %d CUs\n", occ16, occ32, prop.multiProcessorCount);
  const int phased_grid = argc > 4 ? atoi(argv[4]) : 1024;
  auto time_phased = [&](int64_t* col, int G) {
    const int grid = phased_grid;
    std::vector<float> ms;
    for (int r = 0; r < 7; r++) {
      CK(hipMemsetAsync(gctr, 0, 64, nullptr));
      CK(hipEventRecord(e0, nullptr));
      if (G == 16) hipLaunchKernelGGL((k_scan_phased<16>), dim3(grid), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, ntiles, gctr);
      else hipLaunchKernelGGL((k_scan_phased<32>), dim3(grid), dim3(256), 0, 0, col, (int64_t)899999, bm, cnt, ntiles, gctr);
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
      if (r == 0) { unsigned h = 0; CK(hipMemcpy(&h, gctr, 4, hipMemcpyDeviceToHost)); printf("[phased G=%d grid %d: counter ends at %u = %.2f x grid] ", G, grid, h, (double)h / grid); }
    }
    std::sort(ms.begin(), ms.end());
    return ms[3];
  };
  unsigned long long* sink; CK(hipMalloc(&sink, 64));
  auto time_read = [&](int64_t* col) {
    std::vector<float> ms;
    for (int r = 0; r < 7; r++) {
      CK(hipEventRecord(e0, nullptr));
      hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, col, ntiles, sink);
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[3];
  };
  const int mode = argc > 2 ? atoi(argv[2]) : 0;      // 0 hipMalloc; k > 0: virtual-memory API, physical chunks of k x 256 MB mapped back to back
  for (int i = 0; i < ncols; i++) {
    int64_t* col;
    if (mode == 0) CK(hipMalloc(&col, n * 8 + 4096 + (size_t)i * (3u << 20)));
    else {
      hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
      const size_t chunk = (size_t)mode * (256u << 20);
      const size_t total = ((size_t)n * 8 + 4096 + chunk - 1) / chunk * chunk;
      void* va = nullptr; CK(hipMemAddressReserve(&va, total, gran, nullptr, 0));
      for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
      }
      hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, total, &ad, 1));
      col = (int64_t*)va;
      if (i == 0) printf("VMM: granularity %zu, chunk %zu MB\n", gran, chunk >> 20);
    }
    hipLaunchKernelGGL(k_gen, dim3(8192), dim3(256), 0, 0, col, n); CK(hipDeviceSynchronize());
    cols.push_back(col);
    printf("column %2d: scan %.4f  scan4 %.4f  scan4+wt %.4f  PHASED G=16 %.4f  G=32 %.4f  no writes %.4f  read %.4f ms\n", i, time_one(col), time_scan4(col, 3), time_k(col, 2), time_phased(col, 16), time_phased(col, 32), time_scan4(col, 0), time_read(col));
  }
  for (int i = 0; i < ncols; i++) printf("again  %2d: %.4f ms\n", i, time_one(cols[i]));
  // the other way round: the same columns against eight differently placed bitmap buffers
  uint64_t* bm0 = bm;
  for (int b = 0; b < 8; b++) {
    uint64_t* nb; CK(hipMalloc(&nb, ntiles * 128 + 4096 + (size_t)b * (5u << 20)));
    bm = nb;
    printf("bitmap buffer %d at %p:", b, (void*)nb);
    for (int i = 0; i < ncols && i < 4; i++) printf("  col %d %.4f", i, time_one(cols[i]));
    printf("\n");
  }
  bm = bm0;
  return 0;
}
