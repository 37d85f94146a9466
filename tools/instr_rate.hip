// tools/instr_rate.hip -- developer micro-benchmark: issue cost of individual gfx950 VALU instructions,
// in "slots" relative to a full-rate 32-bit VALU op.  16 independent dependency chains per work-item,
// 8 waves per SIMD, so the number is throughput, not latency.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
typedef uint64_t u64; typedef uint32_t u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define KERNEL32(NAME, ASM)                                                     \
  __global__ void __launch_bounds__(256) NAME(u32* out, u32 y, int iters) {     \
    u32 x[16];                                                                  \
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;                        \
    for (int it = 0; it < iters; it++)                                          \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(x[i]) : "v"(y) : "vcc"); \
    u32 acc = 0;                                                                \
    for (int i = 0; i < 16; i++) acc ^= x[i];                                   \
    out[blockIdx.x * 256 + threadIdx.x] = acc;                                  \
  }
#define KERNEL64(NAME, ASM)                                                     \
  __global__ void __launch_bounds__(256) NAME(u32* out, u32 y32, int iters) {   \
    u64 x[16];                                                                  \
    u64 y = ((u64)y32 << 32) | threadIdx.x;                                     \
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;                        \
    for (int it = 0; it < iters; it++)                                          \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(x[i]) : "v"(y), "v"(y32) : "vcc"); \
    u64 acc = 0;                                                                \
    for (int i = 0; i < 16; i++) acc ^= x[i];                                   \
    out[blockIdx.x * 256 + threadIdx.x] = (u32)acc ^ (u32)(acc >> 32);          \
  }

KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, %1, %0")
KERNEL32(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_not, "v_not_b32 %0, %0")
KERNEL32(k_cnd_s, "v_cndmask_b32 %0, %0, %1, s[20:21]")
KERNEL32(k_cmp_only, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_cmp_s, "v_cmp_lt_u32 s[20:21], %0, %1")
KERNEL32(k_add_co_s, "v_add_co_u32 %0, s[20:21], %0, %1")
KERNEL32(k_fma, "v_fma_f32 %0, %0, %1, %0")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
KERNEL32(k_min, "v_min_u32 %0, %0, %1")
KERNEL32(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1\n\tv_add_u32 %0, %0, %1")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL32(k_xad, "v_xad_u32 %0, %0, %1, %1")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL64(k_cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %1")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %2, %2, %0")
KERNEL64(k_mad_u64_u32_s, "v_mad_u64_u32 %0, s[10:11], %2, %2, %0")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %1")
#define KERNEL32X2(NAME, ASM)                                                   \
  __global__ void __launch_bounds__(256) NAME(u32* out, u32 y, int iters) {     \
    u32 x[16], z[16];                                                           \
    for (int i = 0; i < 16; i++) { x[i] = threadIdx.x + i; z[i] = i; }          \
    for (int it = 0; it < iters; it++)                                          \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(x[i]), "+v"(z[i]) : "v"(y) : "vcc"); \
    u32 acc = 0;                                                                \
    for (int i = 0; i < 16; i++) acc ^= x[i] ^ z[i];                            \
    out[blockIdx.x * 256 + threadIdx.x] = acc;                                  \
  }
KERNEL32X2(k_add_pair, "v_add_co_u32 %0, vcc, %0, %2\n\ts_nop 0\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc")
KERNEL32X2(k_add_pair_nonop, "v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc")
KERNEL32X2(k_add_pair_nop1, "v_add_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc")

static float run(void (*k)(u32*, u32, int), u32* d, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, 3u, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, 3u, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f;
}
int main() {
  u32* d;
  CK(hipMalloc(&d, 2048 * 256 * 4));
  const int iters = 2048;
  float base = run(k_add_u32, d, iters);
  printf("baseline v_add_u32: %.1f us for %d x 16 instr x 524288 lanes -> %.1f Ginstr-lanes/s\n", base, iters,
         (double)iters * 16 * 524288 / base * 1e-3);
#define R(K, N) printf("  %-22s %8.1f us  %.2f slots%s\n", #K, run(K, d, iters), run(K, d, iters) / base, N)
  R(k_and, ""); R(k_xor, ""); R(k_lshl, ""); R(k_lshr, ""); R(k_sub, ""); R(k_mov, ""); R(k_not, ""); R(k_cnd_s, ""); R(k_cmp_only, ""); R(k_cmp_s, ""); R(k_add_co_s, ""); R(k_fma, ""); R(k_bfe, ""); R(k_min, "");
  R(k_add_u32, " (baseline again)");
  R(k_add_co, ""); R(k_addc_co, ""); R(k_mul_lo, ""); R(k_mul_hi, ""); R(k_mul_u24, ""); R(k_mad_u24, "");
  R(k_alignbit, ""); R(k_add3, ""); R(k_cndmask, ""); R(k_cmp_u32, " (cmp + add)"); R(k_lshl_or, ""); R(k_xad, "");
  R(k_lshl_add_u64, ""); R(k_cmp_lt_u64, ""); R(k_mad_u64_u32, ""); R(k_mad_u64_u32_s, ""); R(k_lshlrev_b64, "");
  R(k_mov_b64, ""); R(k_add_pair, " (add_co; s_nop 0; addc_co)"); R(k_add_pair_nonop, " (add_co; addc_co, no nop: timing only)"); R(k_add_pair_nop1, " (add_co; s_nop 1; addc_co)");
  R(k_add_u32, " (baseline at end)");
  return 0;
}
