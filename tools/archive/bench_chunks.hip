// bench_chunks.hip — read bandwidth of individual physical chunks (hipMemCreate) of device memory, in allocation order.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench_chunks.hip -o tools/bench_chunks
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
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
  const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 256) << 20;
  const int nchunks = argc > 2 ? atoi(argv[2]) : 64;
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  void* va = nullptr; CK(hipMemAddressReserve(&va, chunk * nchunks, gran, nullptr, 0));
  for (int i = 0; i < nchunks; i++) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); CK(hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, h, 0)); }
  hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, chunk * nchunks, &ad, 1));
  CK(hipMemset(va, 1, chunk * nchunks));
  unsigned long long* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  for (int i = 0; i < nchunks; i++) {
    const int64_t* p = (const int64_t*)((char*)va + (size_t)i * chunk);
    hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, p, (int64_t)(chunk / 8192), out);
    CK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, p, (int64_t)(chunk / 8192), out);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    printf("%5.2f ", (double)chunk * reps / (t * 1e-3) / 1e12);
    if (i % 16 == 15) printf("\n");
  }
  printf("\n(TB/s per %zu-MB chunk, allocation order)\n", chunk >> 20);
  // and the whole range in one launch
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, (const int64_t*)va, (int64_t)(chunk * nchunks / 8192), out);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    printf("whole range: %.3f ms = %.2f TB/s\n", t, (double)chunk * nchunks / (t * 1e-3) / 1e12);
  }
  return 0;
}
