// What does a store pattern cost on MI355X?  The partition pass of unique-by-radix (csrc/k_radix.hip) writes, per 8192-row tile and per partition, one short RUN
// of records at the partition's running position: ~16 records (P = 512), wherever the previous tile's run ended.  This program writes the same 1e9 records
// with no sorting at all — addresses are arithmetic — in several layouts, beside an 8-byte-per-record read stream like the pass's own:
//   layout 0  two arrays: 8-byte keys, 4-byte rows (what the pass does)      layout 1  one array of 12-byte records      layout 2  one array of 16-byte records
//   align A   every run starts at a multiple of A records (the run padded up to it); 1 = runs follow each other without holes
//   P         partitions (runs per tile); P = 0: the tile's 8192 records in row order (one fully coalesced run)
//   chunk-major  the (partition, chunk) regions laid out [chunk][partition] — a workgroup's 512 running positions inside ~12 MB — instead of [partition][chunk]
//             (512 positions ~20 MB apart: as many 2-MB pages as partitions, per array)
//   shared    2 = one running position per (XCD, partition), XCD = workgroup number mod 8;  1 = ONE running position per partition for all workgroups (a global atomicAdd per tile and partition reserves the run) instead of one per
//             (partition, workgroup): 512 write streams instead of 131 072
//   hipcc -O3 --offload-arch=gfx950 scatter_runs.hip -o scatter_runs && ./scatter_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kTile = 8192, kBlock = 1024;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(kBlock) void k_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ keys, uint32_t* __restrict__ rows, int P, int layout, int align,
                                                    int tiles_per_chunk, uint32_t cap /* records per (partition, chunk) */, int do_read, int chunk_major, int shared_mode, uint32_t* gfront /* shared mode: one running position per partition, all workgroups */) {
  __shared__ uint32_t lens[1024], lstart[1024], frontier[1024], wsum[16];
  __shared__ uint16_t owner[kTile];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = blockIdx.x, C = gridDim.x;
  if (!gfront && tid < (P ? P : 1)) frontier[tid] = P ? (chunk_major ? ((uint32_t)c * P + tid) * cap : ((uint32_t)tid * C + c) * cap) : (uint32_t)c * (uint32_t)tiles_per_chunk * kTile;
  __syncthreads();
  uint64_t acc = 0;
  for (int t = 0; t < tiles_per_chunk; t++) {
    uint64_t v[8];
    const size_t rbase = ((size_t)c * tiles_per_chunk + t) * kTile;
    if (do_read) for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(in + rbase + k * kBlock + tid); else for (int k = 0; k < 8; k++) v[k] = rbase + k;
    if (P) {
      // run lengths: 8192 / P on average, +- 25 %, the tile's total exactly 8192 (pairs of partitions trade records)
      const uint32_t L = kTile / P;
      if (tid < P) { const uint32_t j = mix((uint32_t)(c * 131071 + t) * 2654435761u + (tid >> 1)) % (L / 2 + 1); lens[tid] = (tid & 1) ? L + j - L / 4 : L - j + L / 4; }
      __syncthreads();
      uint32_t h = tid < P ? lens[tid] : 0, incl = h;
      for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if (lane >= d) incl += x; }
      if (lane == 63) wsum[wv] = incl;
      __syncthreads();
      uint32_t before = 0; for (int w = 0; w < wv; w++) before += wsum[w];
      if (tid < P) { const uint32_t ex = before + incl - h; lstart[tid] = ex; for (uint32_t i = 0; i < h; i++) owner[ex + i] = (uint16_t)tid;
                     if (gfront) frontier[tid] = atomicAdd(&gfront[(shared_mode == 2 ? (c & 7) * P : 0) + tid], (h + align - 1) / align * align); }
      __syncthreads();
    }
    for (int k = 0; k < 8; k++) {
      const uint32_t s = k * kBlock + tid;
      uint32_t dst;
      if (P) { const uint32_t p = owner[s]; dst = frontier[p] + (s - lstart[p]); } else dst = frontier[0] + s;
      if (layout == 0) { keys[dst] = v[k]; rows[dst] = s; }
      else if (layout == 1) { uint32_t* r = (uint32_t*)keys + (size_t)dst * 3; r[0] = (uint32_t)v[k]; r[1] = (uint32_t)(v[k] >> 32); r[2] = s; }
      else { uint4 q; q.x = (uint32_t)v[k]; q.y = (uint32_t)(v[k] >> 32); q.z = s; q.w = 0; ((uint4*)keys)[dst] = q; }
      acc += v[k];
    }
    __syncthreads();
    if (P) { if (!gfront && tid < P) frontier[tid] += (lens[tid] + align - 1) / align * align; } else if (tid == 0) frontier[0] += kTile;
    __syncthreads();
  }
  if (acc == 0x1234567ull) rows[0] = 1;
}

int main(int argc, char** argv) {
  const int C = 1024, tiles_per_chunk = 120;                      // 1024 x 120 x 8192 = 1.0066e9 records
  const size_t n = (size_t)C * tiles_per_chunk * kTile;
  uint64_t* in; uint64_t* keys; uint32_t* rows;
  const size_t slack = 3;                                         // room for padded runs
  CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&keys, n * 16 * slack)); CK(hipMalloc(&rows, n * 4 * slack));
  CK(hipMemset(in, 1, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { int P, layout, align, read, shared, cm; };
  uint32_t* gfront; CK(hipMalloc(&gfront, 8 * 4096)); static uint32_t hfront[8 * 1024];
  const Cfg cfgs[] = {{0, 0, 1, 1, 0, 0}, {0, 1, 1, 1, 0, 0},
                      {512, 0, 1, 1, 0, 0}, {512, 1, 1, 1, 0, 0}, {512, 0, 1, 1, 1, 0}, {512, 1, 1, 1, 1, 0}, {512, 0, 1, 1, 2, 0}, {512, 1, 1, 1, 2, 0},
                      {256, 1, 1, 1, 0, 0}, {256, 1, 1, 1, 2, 0}, {1024, 1, 1, 1, 0, 0}, {1024, 1, 1, 1, 2, 0}, {1024, 0, 1, 1, 2, 0}, {512, 1, 1, 1, 0, 0}, {512, 1, 1, 1, 2, 0}};
  printf("%zu records; ms are per pass (best of 3); 'alg GB' = records x (8 read + 12 written)\n", n);
  printf("%6s %7s %6s %5s %7s %9s %12s\n", "P", "layout", "align", "read", "shared", "ms", "written GB/s  (last column: 1 = a chunk's runs side by side, [chunk][partition])");
  for (const Cfg& g : cfgs) {
    const uint32_t L = g.P ? kTile / g.P : 0;
    const uint32_t cap = g.P ? (uint32_t)tiles_per_chunk * ((L + L / 4 + 1 + g.align - 1) / g.align * g.align + g.align) : 0;
    if (g.P && (size_t)g.P * C * cap > n * slack) { printf("%6d %7d %6d: no room\n", g.P, g.layout, g.align); continue; }
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      if (g.shared == 1) for (int p = 0; p < g.P; p++) hfront[p] = (uint32_t)p * C * cap;
      if (g.shared == 2) for (int x = 0; x < 8; x++) for (int p = 0; p < g.P; p++) hfront[x * g.P + p] = ((uint32_t)p * 8 + x) * (C / 8) * cap;
      if (g.shared) CK(hipMemcpy(gfront, hfront, 8 * 4096, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_scatter, dim3(C), dim3(kBlock), 0, 0, in, keys, rows, g.P, g.layout, g.align, tiles_per_chunk, cap, g.read, g.cm, g.shared, g.shared ? gfront : nullptr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%6d %7d %6d %5d %7d %9.3f %12.0f   %d\n", g.P, g.layout, g.align, g.read, g.shared, best, n * (g.layout == 2 ? 16.0 : 12.0) / best / 1e6, g.cm);
  }
  return 0;
}
