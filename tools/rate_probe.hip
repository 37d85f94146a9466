// tools/rate_probe.hip -- developer probe (round 4): issue cost of single VALU opcodes on gfx950, every one written as inline
// asm (tools/instr_rate.hip measured C expressions, which the compiler may lower differently).  16 independent chains per lane,
// 2048 x 256 threads (8 waves per SIMD), cost relative to v_add_u32.  One line per opcode: "slots".
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rate_probe.hip -o gpurun_bin/rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint64_t u64; typedef uint32_t u32;
#define COMMA ,
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define PROBE(NAME, DECL, ASM, OPS, CLOB)                                                        \
  __global__ void __launch_bounds__(256) NAME(u64* out, u32 y32, int iters) {                    \
    DECL                                                                                         \
    for (int it = 0; it < iters; it++)                                                           \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(x[i]) : OPS : CLOB); \
    u64 acc = 0;                                                                                 \
    for (int i = 0; i < 16; i++) acc ^= (u64)x[i];                                               \
    if (acc == 0x123456789ull) out[threadIdx.x] = acc;                                           \
  }
#define D32 u32 x[16]; u32 y = y32 + threadIdx.x; u32 z = y32 * 3 + 1; (void)z; for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
#define D64 u64 x[16]; u64 y = ((u64)y32 << 32) | threadIdx.x; u32 z = y32 * 3 + 1; (void)z; for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
PROBE(p_add_u32, D32, "v_add_u32 %0, %0, %1", "v"(y), "memory")
PROBE(p_sub_u32, D32, "v_sub_u32 %0, %0, %1", "v"(y), "memory")
PROBE(p_and, D32, "v_and_b32 %0, %0, %1", "v"(y), "memory")
PROBE(p_or, D32, "v_or_b32 %0, %0, %1", "v"(y), "memory")
PROBE(p_xor, D32, "v_xor_b32 %0, %0, %1", "v"(y), "memory")
PROBE(p_not, D32, "v_not_b32 %0, %0", "v"(y), "memory")
PROBE(p_mov, D32, "v_mov_b32 %0, %1", "v"(y), "memory")
PROBE(p_lshl, D32, "v_lshlrev_b32 %0, 3, %0", "v"(y), "memory")
PROBE(p_lshr, D32, "v_lshrrev_b32 %0, 3, %0", "v"(y), "memory")
PROBE(p_ashr, D32, "v_ashrrev_i32 %0, 3, %0", "v"(y), "memory")
PROBE(p_lshl_v, D32, "v_lshlrev_b32 %0, %1, %0", "v"(y), "memory")
PROBE(p_lshr_v, D32, "v_lshrrev_b32 %0, %1, %0", "v"(y), "memory")
PROBE(p_bfe, D32, "v_bfe_u32 %0, %0, 3, 12", "v"(y), "memory")
PROBE(p_bfi, D32, "v_bfi_b32 %0, %1, %0, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_and_or, D32, "v_and_or_b32 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_or3, D32, "v_or3_b32 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_xad, D32, "v_xad_u32 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_add3, D32, "v_add3_u32 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_lshl_add, D32, "v_lshl_add_u32 %0, %0, 3, %1", "v"(y), "memory")
PROBE(p_add_lshl, D32, "v_add_lshl_u32 %0, %0, %1, 3", "v"(y), "memory")
PROBE(p_lshl_or, D32, "v_lshl_or_b32 %0, %0, 3, %1", "v"(y), "memory")
PROBE(p_alignbit, D32, "v_alignbit_b32 %0, %0, %1, 12", "v"(y), "memory")
PROBE(p_alignbyte, D32, "v_alignbyte_b32 %0, %0, %1, 1", "v"(y), "memory")
PROBE(p_perm, D32, "v_perm_b32 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_min, D32, "v_min_u32 %0, %0, %1", "v"(y), "memory")
PROBE(p_max, D32, "v_max_u32 %0, %0, %1", "v"(y), "memory")
PROBE(p_cndmask_vcc, D32, "v_cndmask_b32 %0, %0, %1, vcc", "v"(y), "memory")
PROBE(p_add_co, D32, "v_add_co_u32 %0, vcc, %0, %1", "v"(y), "vcc")
PROBE(p_addc_co, D32, "v_addc_co_u32 %0, vcc, %0, %1, vcc", "v"(y), "vcc")
PROBE(p_sub_co, D32, "v_sub_co_u32 %0, vcc, %0, %1", "v"(y), "vcc")
PROBE(p_cmp_u32, D32, "v_cmp_lt_u32 vcc, %0, %1", "v"(y), "vcc")
PROBE(p_mul_lo, D32, "v_mul_lo_u32 %0, %0, %1", "v"(y), "memory")
PROBE(p_mul_hi, D32, "v_mul_hi_u32 %0, %0, %1", "v"(y), "memory")
PROBE(p_mul_u24, D32, "v_mul_u32_u24 %0, %0, %1", "v"(y), "memory")
PROBE(p_mad_u24, D32, "v_mad_u32_u24 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_mul_hi_u24, D32, "v_mul_hi_u32_u24 %0, %0, %1", "v"(y), "memory")
PROBE(p_fma_f32, D32, "v_fma_f32 %0, %0, %1, %0", "v"(y), "memory")
PROBE(p_add_f32, D32, "v_add_f32 %0, %0, %1", "v"(y), "memory")
PROBE(p_mul_f32, D32, "v_mul_f32 %0, %0, %1", "v"(y), "memory")
PROBE(p_cvt_f32_u32, D32, "v_cvt_f32_u32 %0, %0", "v"(y), "memory")
PROBE(p_cvt_u32_f32, D32, "v_cvt_u32_f32 %0, %0", "v"(y), "memory")
PROBE(p_pk_add_u16, D32, "v_pk_add_u16 %0, %0, %1", "v"(y), "memory")
PROBE(p_pk_mul_lo_u16, D32, "v_pk_mul_lo_u16 %0, %0, %1", "v"(y), "memory")
PROBE(p_pk_mad_u16, D32, "v_pk_mad_u16 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_dot4_u8, D32, "v_dot4_u32_u8 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_sad_u8, D32, "v_sad_u8 %0, %0, %1, %2", "v"(y) COMMA "v"(z), "memory")
PROBE(p_mov_dpp_row_shr, D32, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v"(y), "memory")
PROBE(p_mov_dpp_quad, D32, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v"(y), "memory")
PROBE(p_add_dpp_quad, D32, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v"(y), "memory")
PROBE(p_add_dpp_row_ror, D32, "v_add_u32_dpp %0, %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf", "v"(y), "memory")
PROBE(p_xor_dpp_bcast, D32, "v_xor_b32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf", "v"(y), "memory")
PROBE(p_add_sdwa, D32, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1", "v"(y), "memory")
PROBE(p_readlane_bcast, D32, "v_readfirstlane_b32 s20, %0\n\tv_mov_b32 %0, s20", "v"(y), "s20" COMMA "memory")
PROBE(p_lshl_add_u64, D64, "v_lshl_add_u64 %0, %0, 0, %1", "v"(y), "memory")
PROBE(p_mad_u64_u32, D64, "v_mad_u64_u32 %0, vcc, %2, %2, %0", "v"(y) COMMA "v"(y32), "vcc")
PROBE(p_mad_i64_i32, D64, "v_mad_i64_i32 %0, vcc, %2, %2, %0", "v"(y) COMMA "v"(y32), "vcc")
PROBE(p_cmp_lt_u64, D64, "v_cmp_lt_u64 vcc, %0, %1", "v"(y), "vcc")
PROBE(p_lshlrev_b64, D64, "v_lshlrev_b64 %0, 3, %0", "v"(y), "memory")
PROBE(p_lshrrev_b64, D64, "v_lshrrev_b64 %0, 3, %0", "v"(y), "memory")
PROBE(p_ashrrev_i64, D64, "v_ashrrev_i64 %0, 3, %0", "v"(y), "memory")
PROBE(p_mov_b64, D64, "v_mov_b64 %0, %1", "v"(y), "memory")
PROBE(p_pk_mov, D64, "v_pk_mov_b32 %0, %1, %1", "v"(y), "memory")
PROBE(p_pk_add_f32, D64, "v_pk_add_f32 %0, %0, %1", "v"(y), "memory")
PROBE(p_pk_fma_f32, D64, "v_pk_fma_f32 %0, %0, %1, %0", "v"(y), "memory")
PROBE(p_pk_mul_f32, D64, "v_pk_mul_f32 %0, %0, %1", "v"(y), "memory")
PROBE(p_add_f64, D64, "v_add_f64 %0, %0, %1", "v"(y), "memory")
PROBE(p_fma_f64, D64, "v_fma_f64 %0, %0, %1, %0", "v"(y), "memory")
PROBE(p_mul_f64, D64, "v_mul_f64 %0, %0, %1", "v"(y), "memory")

typedef void (*kern_t)(u64*, u32, int);
static double run(kern_t k, int blocks, int iters) {
  u64* d; CK(hipMalloc(&d, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 7u, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 7u, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipFree(d));
  return ms * 1e3;
}
int main() {
#define K(n) {#n, n}
  struct { const char* n; kern_t k; } ks[] = {
    K(p_add_u32), K(p_sub_u32), K(p_and), K(p_or), K(p_xor), K(p_not), K(p_mov), K(p_lshl), K(p_lshr), K(p_ashr), K(p_lshl_v), K(p_lshr_v),
    K(p_bfe), K(p_bfi), K(p_and_or), K(p_or3), K(p_xad), K(p_add3), K(p_lshl_add), K(p_add_lshl), K(p_lshl_or), K(p_alignbit), K(p_alignbyte),
    K(p_perm), K(p_min), K(p_max), K(p_cndmask_vcc), K(p_add_co), K(p_addc_co), K(p_sub_co), K(p_cmp_u32), K(p_mul_lo), K(p_mul_hi),
    K(p_mul_u24), K(p_mad_u24), K(p_mul_hi_u24), K(p_fma_f32), K(p_add_f32), K(p_mul_f32), K(p_cvt_f32_u32), K(p_cvt_u32_f32),
    K(p_pk_add_u16), K(p_pk_mul_lo_u16), K(p_pk_mad_u16), K(p_dot4_u8), K(p_sad_u8), K(p_mov_dpp_row_shr), K(p_mov_dpp_quad),
    K(p_add_dpp_quad), K(p_add_dpp_row_ror), K(p_xor_dpp_bcast), K(p_add_sdwa), K(p_readlane_bcast), K(p_lshl_add_u64), K(p_mad_u64_u32),
    K(p_mad_i64_i32), K(p_cmp_lt_u64), K(p_lshlrev_b64), K(p_lshrrev_b64), K(p_ashrrev_i64), K(p_mov_b64), K(p_pk_mov), K(p_pk_add_f32),
    K(p_pk_fma_f32), K(p_pk_mul_f32), K(p_add_f64), K(p_fma_f64), K(p_mul_f64), K(p_add_u32)};
  const int iters = 2048;
  const double base = run(p_add_u32, 2048, iters);
  printf("baseline v_add_u32: %.1f us for 2048 blocks x 256 threads x %d x 16 instructions (8 waves per SIMD)\n", base, iters);
  for (auto& k : ks) {
    const double us = run(k.k, 2048, iters);
    printf("  %-22s %8.1f us  %5.2f slots\n", k.n + 2, us, us / base);
  }
  return 0;
}
