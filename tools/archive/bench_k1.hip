// bench_k1.hip — standalone A/B harness for variants of the K1 predicate scan (x > c -> bitmap + counts).
// Interleaved rounds in one process (cdna guide rule 24), median per variant, checksum of the bitmap so a
// wrong variant is visible.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench_k1.hip -o tools/bench_k1
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

__device__ __forceinline__ uint32_t tile_popcount(uint64_t w, int lane) {
  uint32_t c = lane < 16 ? (uint32_t)__popcll(w) : 0u;
  for (int d = 8; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  return c;
}

// ---- V0: shipped kernel: 16 x 8-byte loads per lane, ballot = bitmap word
template <bool NT>
__global__ __launch_bounds__(256) void k_v0(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts,
                                            int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t* p = col + tile * 1024 + lane;
    int64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
    uint64_t my = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { uint64_t m = __ballot(v[j] > c); if (lane == j) my = m; }
    const uint32_t cnt = tile_popcount(my, lane);
    if (lane < 16) bitmap[tile * 16 + lane] = my;
    if (lane == 0) counts[tile] = cnt;
  }
}

// ---- V2: 8 x 16-byte loads per lane (rows 2l, 2l+1 of a 128-row group) + ds_bpermute transposition
typedef long long ll2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void k_v2(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts,
                                            int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const int srcA = (lane >> 1) << 2, srcB = (32 + (lane >> 1)) << 2;   // byte addresses for ds_bpermute
  const int sh = lane & 1;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const ll2* p = (const ll2*)(col + tile * 1024) + lane;
    ll2 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = NT ? __builtin_nontemporal_load(p + j * 64) : p[j * 64];
    uint64_t my = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int two = (v[j].x > c ? 1 : 0) | (v[j].y > c ? 2 : 0);
      const int a = __builtin_amdgcn_ds_bpermute(srcA, two), b = __builtin_amdgcn_ds_bpermute(srcB, two);
      const uint64_t mA = __ballot((a >> sh) & 1), mB = __ballot((b >> sh) & 1);
      if (lane == 2 * j) my = mA;
      if (lane == 2 * j + 1) my = mB;
    }
    const uint32_t cnt = tile_popcount(my, lane);
    if (lane < 16) bitmap[tile * 16 + lane] = my;
    if (lane == 0) counts[tile] = cnt;
  }
}

// ---- V3: two tiles (32 loads) in flight per wave
__global__ __launch_bounds__(256) void k_v3(const int64_t* __restrict__ col, int64_t c, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts,
                                            int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t2 = wave * 2; t2 < ntiles; t2 += nwaves * 2) {
    const int64_t* p = col + t2 * 1024 + lane;
    int64_t v[32];
    const bool two = t2 + 1 < ntiles;
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = p[j * 64];
    if (two) {
#pragma unroll
      for (int j = 16; j < 32; j++) v[j] = p[j * 64];
    }
    uint64_t my = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) { uint64_t m = __ballot(j < 16 || two ? v[j] > c : false); if (lane == j) my = m; }
    uint32_t cn = lane < 32 ? (uint32_t)__popcll(my) : 0u;
    for (int d = 8; d >= 1; d >>= 1) cn += __shfl_xor(cn, d, 64);
    if (lane < (two ? 32 : 16)) bitmap[t2 * 16 + lane] = my;
    if (lane == 0) counts[t2] = cn;
    if (lane == 16 && two) counts[t2 + 1] = cn;
  }
}

// ---- copy / read-only references: what the memory system gives a plain streaming kernel
__global__ __launch_bounds__(256) void k_read16(const float4* __restrict__ in, float* __restrict__ out, int64_t n16) {
  float acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read8(const double* __restrict__ in, double* __restrict__ out, int64_t n8) {
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) acc += in[i];
  if (acc == 123.456) out[0] = acc;
}

__global__ void k_checksum(const uint64_t* bm, int64_t nw, unsigned long long* out) {
  unsigned long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nw; i += (int64_t)gridDim.x * 256) s += bm[i] * (unsigned long long)(2 * i + 1);
  atomicAdd(out, s);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000000LL;
  const int rounds = argc > 2 ? atoi(argv[2]) : 7;
  const int64_t ntiles = n / 1024;
  int64_t* col; uint64_t* bm; uint32_t* cnt; unsigned long long* cs; float* sink;
  CK(hipMalloc(&col, n * 8 + 4096)); CK(hipMalloc(&bm, ntiles * 128 + 4096)); CK(hipMalloc(&cnt, ntiles * 4 + 64)); CK(hipMalloc(&cs, 8)); CK(hipMalloc(&sink, 64));
  hipLaunchKernelGGL(k_gen, dim3(8192), dim3(256), 0, 0, col, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { const char* name; int grid; double bytes; std::vector<float> ms; unsigned long long sum; };
  const double scan_bytes = (double)n * (8 + 0.125 + 4.0 / 1024);
  std::vector<V> vs = {
      {"v0 x2 g2048", 2048, scan_bytes}, {"v0 x2 g1024", 1024, scan_bytes}, {"v0 x2 g4096", 4096, scan_bytes}, {"v0 x2 g8192", 8192, scan_bytes},
      {"v0 x2 nt g2048", 2048, scan_bytes}, {"v2 x4 bperm g2048", 2048, scan_bytes}, {"v2 x4 bperm nt g2048", 2048, scan_bytes},
      {"v3 x2 2tiles g1024", 1024, scan_bytes}, {"v3 x2 2tiles g2048", 2048, scan_bytes},
      {"read16 g2048", 2048, (double)n * 8}, {"read16 g8192", 8192, (double)n * 8}, {"read8 g2048", 2048, (double)n * 8}, {"read8 g8192", 8192, (double)n * 8},
  };
  const int64_t c = 899999;
  for (int r = 0; r < rounds + 1; r++) {
    for (size_t i = 0; i < vs.size(); i++) {
      V& v = vs[i];
      CK(hipMemsetAsync(cs, 0, 8, 0));
      CK(hipEventRecord(e0, 0));
      switch (i) {
        case 0: case 1: case 2: case 3: hipLaunchKernelGGL((k_v0<false>), dim3(v.grid), dim3(256), 0, 0, col, c, bm, cnt, ntiles); break;
        case 4: hipLaunchKernelGGL((k_v0<true>), dim3(v.grid), dim3(256), 0, 0, col, c, bm, cnt, ntiles); break;
        case 5: hipLaunchKernelGGL((k_v2<false>), dim3(v.grid), dim3(256), 0, 0, col, c, bm, cnt, ntiles); break;
        case 6: hipLaunchKernelGGL((k_v2<true>), dim3(v.grid), dim3(256), 0, 0, col, c, bm, cnt, ntiles); break;
        case 7: case 8: hipLaunchKernelGGL(k_v3, dim3(v.grid), dim3(256), 0, 0, col, c, bm, cnt, ntiles); break;
        case 9: case 10: hipLaunchKernelGGL(k_read16, dim3(v.grid), dim3(256), 0, 0, (const float4*)col, sink, n / 2); break;
        default: hipLaunchKernelGGL(k_read8, dim3(v.grid), dim3(256), 0, 0, (const double*)col, (double*)sink, n); break;
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms);
      if (i < 9 && r == rounds) {
        hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, 0, bm, ntiles * 16, cs);
        CK(hipMemcpy(&v.sum, cs, 8, hipMemcpyDeviceToHost));
      }
    }
  }
  printf("rows=%lld rounds=%d\n", (long long)n, rounds);
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    printf("%-26s median %.4f ms  min %.4f ms  %.1f GB/s (median)  %.1f GB/s (best)  checksum %016llx\n", v.name, med, mn, v.bytes / med / 1e6,
           v.bytes / mn / 1e6, v.sum);
  }
  return 0;
}
