// tools/clock_probe.hip -- developer probe: what clock does the chip run at under an all-CU VALU load, and how many
// shader cycles does one wave64 VALU instruction take?  s_memtime (clock64) counts shader-clock cycles, s_memrealtime
// (wall_clock64) a constant 100 MHz reference; hipEvents give wall time.  One result line per (instruction, grid).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint64_t u64; typedef uint32_t u32;
#define COMMA ,
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define PROBE(NAME, DECL, ASM, OPS)                                                              \
  __global__ void __launch_bounds__(256) NAME(u64* out, u32 y32, int iters) {                    \
    DECL                                                                                         \
    u64 t0 = clock64(), r0 = wall_clock64();                                                     \
    for (int it = 0; it < iters; it++)                                                           \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(x[i]) : OPS : "vcc"); \
    u64 t1 = clock64(), r1 = wall_clock64();                                                     \
    u64 acc = 0;                                                                                 \
    for (int i = 0; i < 16; i++) acc ^= (u64)x[i];                                               \
    if ((threadIdx.x & 63) == 0) {                                                               \
      u32 w = (blockIdx.x * 256 + threadIdx.x) >> 6;                                             \
      out[2 * w] = t1 - t0; out[2 * w + 1] = (r1 - r0) ^ (acc == 0x123456789ull);                \
    }                                                                                            \
  }
#define D32 u32 x[16]; u32 y = y32; for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
#define D64 u64 x[16]; u64 y = ((u64)y32 << 32) | threadIdx.x; for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
PROBE(p_add_u32, D32, "v_add_u32 %0, %0, %1", "v"(y))
PROBE(p_xor, D32, "v_xor_b32 %0, %0, %1", "v"(y))
PROBE(p_lshl, D32, "v_lshlrev_b32 %0, 3, %0", "v"(y))
PROBE(p_min, D32, "v_min_u32 %0, %0, %1", "v"(y))
PROBE(p_fma, D32, "v_fma_f32 %0, %0, %1, %0", "v"(y))
PROBE(p_add_co, D32, "v_add_co_u32 %0, vcc, %0, %1", "v"(y))
PROBE(p_mul_lo, D32, "v_mul_lo_u32 %0, %0, %1", "v"(y))
PROBE(p_lshl_add_u64, D64, "v_lshl_add_u64 %0, %0, 0, %1", "v"(y))
PROBE(p_mad_u64_u32, D64, "v_mad_u64_u32 %0, vcc, %2, %2, %0", "v"(y) COMMA "v"(y32))
PROBE(p_cmp_u64, D64, "v_cmp_lt_u64 vcc, %0, %1", "v"(y))
PROBE(p_pk_add_f32, D64, "v_pk_add_f32 %0, %0, %1", "v"(y))
PROBE(p_pk_fma_f32, D64, "v_pk_fma_f32 %0, %0, %1, %0", "v"(y))
PROBE(p_add_f64, D64, "v_add_f64 %0, %0, %1", "v"(y))
PROBE(p_fma_f64, D64, "v_fma_f64 %0, %0, %1, %0", "v"(y))
PROBE(p_mov_b64, D64, "v_mov_b64 %0, %1", "v"(y))
PROBE(p_pk_mov, D64, "v_pk_mov_b32 %0, %1, %1", "v"(y))
PROBE(p_lshlrev_b64, D64, "v_lshlrev_b64 %0, 3, %0", "v"(y))
PROBE(p_lshrrev_b64, D64, "v_lshrrev_b64 %0, 3, %0", "v"(y))

typedef void (*kern_t)(u64*, u32, int);
static void run(const char* name, kern_t k, int blocks, int iters) {
  u64* d; const int waves = blocks * 4;
  CK(hipMalloc(&d, (size_t)waves * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 7u, iters);   // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 7u, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  u64* h = (u64*)malloc((size_t)waves * 16);
  CK(hipMemcpy(h, d, (size_t)waves * 16, hipMemcpyDeviceToHost));
  double cyc = 0, ref = 0;
  for (int w = 0; w < waves; w++) { cyc += (double)h[2 * w]; ref += (double)(h[2 * w + 1] & ~1ull); }
  cyc /= waves; ref /= waves;
  const double ninst = (double)iters * 16;
  // per wave: cyc shader cycles for ninst instructions while (resident waves per SIMD) waves share the SIMD
  const double wps = blocks >= 2048 ? 8.0 : (double)blocks * 4 / 1024.0;   // waves per SIMD (2048 blocks x 4 waves = 8 per SIMD)
  printf("%-16s blocks %5d  wall %8.1f us  memtime/wave %10.0f  realtime/wave %8.0f (x10ns)  -> sclk %.3f GHz; memtime per wave-instr %.2f; per SIMD-issued instr %.2f; ns per SIMD instr %.3f\n",
         name, blocks, ms * 1e3, cyc, ref, cyc / (ref * 10.0), cyc / ninst, cyc / ninst / (wps < 1 ? 1 : wps), ms * 1e6 / ninst / (wps < 1 ? 1 : wps));
  free(h); CK(hipFree(d));
}

int main() {
  struct { const char* n; kern_t k; } ks[] = {
    {"v_add_u32", p_add_u32}, {"v_xor_b32", p_xor}, {"v_lshlrev_b32", p_lshl}, {"v_min_u32", p_min}, {"v_fma_f32", p_fma},
    {"v_add_co_u32", p_add_co}, {"v_mul_lo_u32", p_mul_lo}, {"v_lshl_add_u64", p_lshl_add_u64}, {"v_mad_u64_u32", p_mad_u64_u32},
    {"v_cmp_lt_u64", p_cmp_u64}, {"v_pk_add_f32", p_pk_add_f32}, {"v_pk_fma_f32", p_pk_fma_f32}, {"v_add_f64", p_add_f64},
    {"v_fma_f64", p_fma_f64}, {"v_mov_b64", p_mov_b64}, {"v_pk_mov_b32", p_pk_mov}, {"v_lshlrev_b64", p_lshlrev_b64}, {"v_lshrrev_b64", p_lshrrev_b64}};
  for (auto& k : ks) {
    run(k.n, k.k, 2048, 2048);   // 8 waves per SIMD on every CU
    run(k.n, k.k, 256, 2048);    // 1 wave per SIMD on every CU
    run(k.n, k.k, 8, 2048);      // a handful of CUs: no power limit
  }
  return 0;
}
