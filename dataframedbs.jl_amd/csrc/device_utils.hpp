// device_utils.hpp — wave64 helpers shared by the gfx950 kernels (no CUDA/other-arch paths).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dfdb {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// orders one wave's LDS writes before its following LDS reads (LDS ops of a wave execute in order; this
// only stops the compiler from moving them across each other)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// reinterpret the low sizeof(T) bytes of a 64-bit pattern as T (constants travel as bit patterns)
template <typename T>
__host__ __device__ __forceinline__ T from_bits(uint64_t bits) { T v; __builtin_memcpy(&v, &bits, sizeof(T)); return v; }

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

}  // namespace dfdb
