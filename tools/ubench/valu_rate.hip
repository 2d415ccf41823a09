// VALU issue rates on gfx950, one instruction kind at a time: every SIMD of the chip runs 8 waves of 8 independent chains of the same instruction
// (tools/ubench: measurements behind the radix kernels' instruction choices; prints cycles per wave-instruction per SIMD).
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 2048;
#define BODY8(ASM32) \
  for (int i = 0; i < kIters; i++) { \
    asm volatile(ASM32(0) ASM32(1) ASM32(2) ASM32(3) ASM32(4) ASM32(5) ASM32(6) ASM32(7) \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k), "s"(sk) : "vcc", "s20", "s21", "s22"); }
#define KERNEL32(NAME, ASM32) \
  __global__ __launch_bounds__(512) void NAME(unsigned* out, unsigned k, unsigned sk) { \
    unsigned a[8]; for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 8 + j + k; \
    BODY8(ASM32) \
    unsigned r = 0; for (int j = 0; j < 8; j++) r ^= a[j]; if (r == 0x12345u) out[0] = r; }
#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_MUL(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_MULS(n) "v_mul_lo_u32 %" #n ", %" #n ", %9\n"
#define A_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define A_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define A_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %" #n "\n"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %8\n"
#define A_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %8\n"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %8\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %8\n"
#define A_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %8\n"
#define A_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 9\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", 13\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_CNDS(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
#define A_CMPCND(n) "v_cmp_gt_u32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_CMPCND2(n) "v_cmp_gt_u32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define A_CMPCND4(n) "v_cmp_gt_u32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_cndmask_b32 %" #n ", %8, %" #n ", vcc\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define A_CMPSCND2(n) "v_cmp_gt_u32 s[20:21], %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\nv_cndmask_b32 %" #n ", %8, %" #n ", s[20:21]\n"
#define A_MIN(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define A_LSHRV(n) "v_lshrrev_b32 %" #n ", %8, %" #n "\n"
#define A_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define A_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %8\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %" #n ", %8, %8\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define A_READLANE(n) "v_readlane_b32 s22, %" #n ", 5\n"
#define A_MADU32(n) "v_mad_u64_u32 v[40:41], vcc, %" #n ", %8, v[40:41]\n"
KERNEL32(k_add, A_ADD) KERNEL32(k_mul, A_MUL) KERNEL32(k_muls, A_MULS) KERNEL32(k_mulhi, A_MULHI) KERNEL32(k_mul24, A_MUL24) KERNEL32(k_mad24, A_MAD24)
KERNEL32(k_xor, A_XOR) KERNEL32(k_lshladd, A_LSHLADD) KERNEL32(k_lshlor, A_LSHLOR) KERNEL32(k_andor, A_ANDOR) KERNEL32(k_add3, A_ADD3) KERNEL32(k_xad, A_XAD)
KERNEL32(k_bfe, A_BFE) KERNEL32(k_alignbit, A_ALIGNBIT) KERNEL32(k_cndmask, A_CNDMASK) KERNEL32(k_cnds, A_CNDS) KERNEL32(k_cmpcnd, A_CMPCND) KERNEL32(k_cmpcnd2, A_CMPCND2) KERNEL32(k_cmpcnd4, A_CMPCND4) KERNEL32(k_cmpscnd2, A_CMPSCND2) KERNEL32(k_min, A_MIN) KERNEL32(k_and, A_AND)
KERNEL32(k_lshr, A_LSHR) KERNEL32(k_lshrv, A_LSHRV) KERNEL32(k_sub, A_SUB) KERNEL32(k_addco, A_ADDCO) KERNEL32(k_perm, A_PERM) KERNEL32(k_bfi, A_BFI) KERNEL32(k_mov, A_MOV) KERNEL32(k_readlane, A_READLANE)
// 64-bit kinds: four chains of register pairs
#define KERNEL64(NAME, ASM) \
  __global__ __launch_bounds__(512) void NAME(unsigned* out, unsigned k, unsigned sk) { \
    unsigned long long a[4]; for (int j = 0; j < 4; j++) a[j] = threadIdx.x * 8 + j + k; unsigned long long kk = k; \
    for (int i = 0; i < kIters; i++) { asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(0) ASM(1) ASM(2) ASM(3) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(kk), "v"(k)); } \
    unsigned long long r = 0; for (int j = 0; j < 4; j++) r ^= a[j]; if (r == 0x12345u) out[0] = (unsigned)r; }
#define B_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 3, %4\n"
#define B_LSHR64(n) "v_lshrrev_b64 %" #n ", 3, %" #n "\n"
#define B_LSHL64(n) "v_lshlrev_b64 %" #n ", 3, %" #n "\n"
#define B_CMP64(n) "v_cmp_eq_u64 vcc, %" #n ", %4\n"
#define B_CMPGT64(n) "v_cmp_gt_u64 vcc, %" #n ", %4\n"
#define B_CMP32(n) "v_cmp_eq_u32 vcc, %5, %5\n"
#define B_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %5, %5, %" #n "\n"
#define B_PKADD(n) "v_pk_add_u16 %5, %5, %5\n"
KERNEL64(k_lshladd64, B_LSHLADD64) KERNEL64(k_lshr64, B_LSHR64) KERNEL64(k_lshl64, B_LSHL64) KERNEL64(k_cmp64, B_CMP64) KERNEL64(k_cmpgt64, B_CMPGT64) KERNEL64(k_cmp32, B_CMP32) KERNEL64(k_mad64, B_MAD64)

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1000.0;
  unsigned* out; CK(hipMalloc(&out, 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("CUs %d, clock %.0f MHz; 8 waves per SIMD, 8 independent instructions per loop trip\n", cus, mhz);
  struct K { const char* name; void (*f)(unsigned*, unsigned, unsigned); int per_trip; };
  std::vector<K> ks = {{"v_add_u32", k_add, 8}, {"v_mul_lo_u32", k_mul, 8}, {"v_mul_lo_u32 (sgpr operand)", k_muls, 8}, {"v_mul_hi_u32", k_mulhi, 8}, {"v_mul_u32_u24", k_mul24, 8}, {"v_mad_u32_u24", k_mad24, 8},
                       {"v_xor_b32", k_xor, 8}, {"v_lshl_add_u32", k_lshladd, 8}, {"v_lshl_or_b32", k_lshlor, 8}, {"v_and_or_b32", k_andor, 8}, {"v_add3_u32", k_add3, 8}, {"v_xad_u32", k_xad, 8},
                       {"v_bfe_u32", k_bfe, 8}, {"v_alignbit_b32", k_alignbit, 8}, {"v_cndmask_b32 (vcc)", k_cndmask, 8}, {"v_cndmask_b32 (sgpr pair)", k_cnds, 8}, {"v_cmp_gt_u32 + v_cndmask", k_cmpcnd, 16}, {"v_cmp + 2 v_cndmask (vcc)", k_cmpcnd2, 24}, {"v_cmp + 4 v_cndmask (vcc)", k_cmpcnd4, 40}, {"v_cmp + 2 v_cndmask (sgpr pair)", k_cmpscnd2, 24},
                       {"v_min_u32", k_min, 8}, {"v_and_b32", k_and, 8}, {"v_lshrrev_b32 (imm)", k_lshr, 8}, {"v_lshrrev_b32 (vgpr)", k_lshrv, 8}, {"v_sub_u32", k_sub, 8}, {"v_add_co_u32", k_addco, 8},
                       {"v_perm_b32", k_perm, 8}, {"v_bfi_b32", k_bfi, 8}, {"v_mov_b32", k_mov, 8}, {"v_readlane_b32", k_readlane, 8},
                       {"v_lshl_add_u64", k_lshladd64, 8}, {"v_lshrrev_b64", k_lshr64, 8}, {"v_lshlrev_b64", k_lshl64, 8}, {"v_cmp_eq_u64", k_cmp64, 8}, {"v_cmp_gt_u64", k_cmpgt64, 8}, {"v_cmp_eq_u32", k_cmp32, 8},
                       {"v_mad_u64_u32", k_mad64, 8}};
  for (auto& k : ks) {
    const int grid = cus * 4;                       // 4 workgroups of 512 threads per CU = 32 waves per CU = 8 per SIMD
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k.f, dim3(grid), dim3(512), 0, 0, out, 3u, 5u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    const double insts_per_simd = 8.0 * kIters * k.per_trip;      // wave-instructions one SIMD issues
    printf("%-30s %8.3f ms   %.2f cycles per wave-instruction per SIMD\n", k.name, best, best * 1e-3 * mhz * 1e6 / insts_per_simd);
  }
  return 0;
}
