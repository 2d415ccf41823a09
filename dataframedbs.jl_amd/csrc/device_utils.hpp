// device_utils.hpp — wave64 helpers shared by the gfx950 kernels (no CUDA/other-arch paths).
#pragma once
#ifndef __HIPCC_RTC__            // (hipRTC brings the HIP device runtime and the fixed-width integer types itself: jit.cpp compiles this header at run time)
#include <hip/hip_runtime.h>
#include <cstdint>
#endif

namespace dfdb {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// inclusive prefix sum across the 64 lanes of a wave, in-register DPP form (no LDS crossbar): three row_shr
// on the input, row_shr 4/8 on the partials (bank-masked), then row_bcast15 / row_bcast31 to carry across the
// four 16-lane rows.  7 v_add_u32 with DPP modifiers instead of 6 x (ds_bpermute + select + add).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  uint32_t r = x;
  r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
  r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
  r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);   // row_shr:3
  r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xe, false);   // row_shr:4, banks 1-3
  r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xc, false);   // row_shr:8, banks 2-3
  r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
  r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
  return r;
}
// reference form (cross-checked against the DPP form by tests/test_gpu_parity.py through every compaction)
__device__ __forceinline__ uint32_t wave_incl_scan_shfl(uint32_t v) {
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

// vec with lane `lane` (wave-uniform) replaced by the wave-uniform value sval (v_writelane_b32)
__device__ __forceinline__ uint32_t write_lane(uint32_t vec, uint32_t sval, int lane) {
  // one constant-bus operand per VALU instruction on gfx9: the lane select travels in M0 (saved and restored: the
  // compiler treats M0 as reserved and does not accept it in a clobber list)
  uint32_t keep;
  asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
      : "+v"(vec), "=&s"(keep) : "s"(__builtin_amdgcn_readfirstlane(sval)), "s"(__builtin_amdgcn_readfirstlane(lane)));
  return vec;
}

}  // namespace dfdb
