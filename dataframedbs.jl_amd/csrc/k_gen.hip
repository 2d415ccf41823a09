// k_gen.hip — device-side synthetic column fill (SURVEY.md §8d "Data generation"), so benchmark columns
// are born in HBM with no PCIe copy.  Row i (0-based, global) is a pure function of
// h = splitmix64(seed + i); the tests compare these columns bit for bit with the same formula evaluated
// on the CPU.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace dfdb {

constexpr int kBlock = 256;
static inline int gen_grid(int64_t n) { int64_t b = (n + kBlock - 1) / kBlock; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }

__global__ __launch_bounds__(kBlock) void k_gen_i64_mod1m(int64_t* __restrict__ out, uint64_t seed, int64_t row_first, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    out[i] = (int64_t)(splitmix64(seed + (uint64_t)(row_first + i)) % 1000000ull);
}
__global__ __launch_bounds__(kBlock) void k_gen_i64_iota(int64_t* __restrict__ out, int64_t row_first, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = row_first + i + 1;
}
__global__ __launch_bounds__(kBlock) void k_gen_f64_u2000(double* __restrict__ out, uint64_t seed, int64_t row_first, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double u = (double)(splitmix64(seed + (uint64_t)(row_first + i)) >> 11) * (1.0 / 9007199254740992.0);
    out[i] = u * 2000.0;
  }
}

// brands10 (docs/src/index.md:58 extended to 10 entries), packed 9 bytes each
__constant__ char k_brands[10][10] = {"apple", "samsung", "huawei", "microsoft", "dell", "xbox", "sony", "intel", "lenovo", "asus"};
__constant__ int k_brand_len[10] = {5, 7, 6, 9, 4, 4, 4, 5, 6, 4};

// with_missing: Union{String,Missing} — the row is missing (size -1, no bytes) when (h >> 32) mod 8 == 7: one row in eight, like the docs' real data set whose
// columns are all Union{Missing,String} (docs/src/index.md:264-272)
__global__ __launch_bounds__(kBlock) void k_gen_brand_sizes(int32_t* __restrict__ sizes, uint64_t seed, int64_t row_first, int64_t n, int with_missing) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t h = splitmix64(seed + (uint64_t)(row_first + i));
    sizes[i] = (with_missing && ((h >> 32) & 7ull) == 7ull) ? -1 : k_brand_len[h % 10ull];
  }
}

// one wave per 1024-row string tile: running byte offset = tile_off[tile] + in-tile prefix of sizes
__global__ __launch_bounds__(kBlock) void k_gen_brand_bytes(const int32_t* __restrict__ sizes, const int64_t* __restrict__ tile_off,
                                                            uint8_t* __restrict__ bytes, uint64_t seed, int64_t row_first, int64_t n,
                                                            int64_t ntiles) {
  const int lane = lane_id();
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    int64_t run = tile_off[tile];
    for (int j = 0; j < 16; j++) {
      const int64_t i = tile * 1024 + j * 64 + lane;
      const int32_t raw = i < n ? sizes[i] : 0;
      const uint32_t sz = raw > 0 ? (uint32_t)raw : 0u;                        // (-1 = missing: no bytes)
      const uint32_t incl = wave_incl_scan(sz);
      if (i < n) {
        const int b = (int)(splitmix64(seed + (uint64_t)(row_first + i)) % 10ull);
        uint8_t* d = bytes + run + (incl - sz);
        for (uint32_t k = 0; k < sz; k++) d[k] = (uint8_t)k_brands[b][k];
      }
      run += __shfl(incl, 63, 64);
    }
  }
}

void launch_gen_i64_mod1m(hipStream_t s, int64_t* out, uint64_t seed, int64_t row_first, int64_t n) {
  if (n > 0) hipLaunchKernelGGL(k_gen_i64_mod1m, dim3(gen_grid(n)), dim3(kBlock), 0, s, out, seed, row_first, n);
}
void launch_gen_i64_iota(hipStream_t s, int64_t* out, int64_t row_first, int64_t n) {
  if (n > 0) hipLaunchKernelGGL(k_gen_i64_iota, dim3(gen_grid(n)), dim3(kBlock), 0, s, out, row_first, n);
}
void launch_gen_f64_u2000(hipStream_t s, double* out, uint64_t seed, int64_t row_first, int64_t n) {
  if (n > 0) hipLaunchKernelGGL(k_gen_f64_u2000, dim3(gen_grid(n)), dim3(kBlock), 0, s, out, seed, row_first, n);
}
void launch_gen_brand_sizes(hipStream_t s, int32_t* sizes, uint64_t seed, int64_t row_first, int64_t n, bool with_missing) {
  if (n > 0) hipLaunchKernelGGL(k_gen_brand_sizes, dim3(gen_grid(n)), dim3(kBlock), 0, s, sizes, seed, row_first, n, with_missing ? 1 : 0);
}
void launch_gen_brand_bytes(hipStream_t s, const int32_t* sizes, const int64_t* tile_off, uint8_t* bytes, uint64_t seed, int64_t row_first,
                            int64_t n) {
  const int64_t ntiles = (n + 1023) / 1024;
  if (ntiles == 0) return;
  int64_t blocks = (ntiles + 3) / 4; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_gen_brand_bytes, dim3((unsigned)blocks), dim3(kBlock), 0, s, sizes, tile_off, bytes, seed, row_first, n, ntiles);
}

}  // namespace dfdb
