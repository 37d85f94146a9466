// tools/variants.hip -- developer probe (round 4): the arithmetic alternatives that rounds 2-3 dismissed on paper, MEASURED.
// Math only (registers, no memory), 2048 workgroups x 256 work-items (8 waves per SIMD) unless said otherwise; results are
// issue slots per coefficient (1 slot = one full-rate v_add_u32), against the shipped forms measured in the same run.
//
//   A  the shipped 16-point round: Dif<16> (shift twiddles) + 15 table multiplies                    [baseline]
//   B  128-bit lazy limbs inside a 16-point round: add / sub as four-limb carry chains without canonicalisation, shift
//      twiddles as limb shifts with the 2^96 = -1 wrap, ONE reduction to a canonical residue per coefficient and round
//      (what "3 x 32-bit lazy accumulation" needs once the range is honest: a shifted 64-bit value fills 96 bits and four
//      butterfly stages add four more)
//   C  a cross-lane radix-2 stage (the building block of a (64, 32) pass whose sixth stage pairs lanes): each lane fetches its
//      partner's coefficient through DPP and computes ONE output of the butterfly -- sum in even lanes, twiddled difference in
//      odd lanes
//   D  32 coefficients per lane: Dif<32> + 31 table multiplies at 4 waves per SIMD (the registers allow no more), against two
//      shipped 16-point rounds at 8
//   E  (round 5) a twiddle layer whose row digit is WAVE-UNIFORM (the review's proposal: a pass as [16 . 4] . [4 . 8], the digit that
//      meets the register index taken from the top bits of the lane's row group, so omega_64^(a k) = 2^(39 a k) is selected by a
//      scalar branch among compile-time mul_2exp<K> bodies): E1 the shift layer against E0 the table layer it would replace, and
//      E2 what the extra round costs -- one more LDS exchange (16 ds_write_b64 + barrier + 16 ds_read_b64 per lane).  A pass of
//      2^11 rows is (16, 16, 8) = 2 table layers + 2 exchanges today; [16 . 4] . [4 . 8] is 1 table layer + 2 shift layers +
//      3 exchanges.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/variants.hip -o gpurun_bin/variants
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../ronkathon_amd/csrc/ntt_tile.h"

using namespace ronk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- A / D: shipped rounds
template <int N>
__global__ void __launch_bounds__(256) round_kernel(u64* out, const u64* tw, int iters) {
  u64 x[N];
  for (int i = 0; i < N; i++) x[i] = (u64)(threadIdx.x * 7919u + blockIdx.x + i + 1) * 0x9E3779B97F4A7C15ull % gl64::P;
  const u32 tb = (threadIdx.x & 63) << 3;
  for (int it = 0; it < iters; it++) {
    Dif<N, false, true>::run(x);
#pragma unroll
    for (int i = 1; i < N; i++) x[i] = gl64::mul(x[i], ld_tabb(tw, (tb * i + it) & 0x3FF8));
  }
  u64 acc = 0;
  for (int i = 0; i < N; i++) acc ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---------------------------------------------------------------- B: 128-bit lazy limbs
struct L128 { u32 w[4]; };   // two's complement, value = sum w[i] 2^(32 i) (mod p: 2^96 = -1, 2^64 = 2^32 - 1)
__device__ __forceinline__ L128 l_from(u64 x) { return L128{{(u32)x, (u32)(x >> 32), 0u, 0u}}; }
__device__ __forceinline__ L128 l_add(L128 a, L128 b) {
  L128 r; u32 c;
  r.w[0] = __builtin_addc(a.w[0], b.w[0], 0u, &c);
  r.w[1] = __builtin_addc(a.w[1], b.w[1], c, &c);
  r.w[2] = __builtin_addc(a.w[2], b.w[2], c, &c);
  r.w[3] = a.w[3] + b.w[3] + c;
  return r;
}
__device__ __forceinline__ L128 l_sub(L128 a, L128 b) {
  L128 r; u32 c;
  r.w[0] = __builtin_subc(a.w[0], b.w[0], 0u, &c);
  r.w[1] = __builtin_subc(a.w[1], b.w[1], c, &c);
  r.w[2] = __builtin_subc(a.w[2], b.w[2], c, &c);
  r.w[3] = a.w[3] - b.w[3] - c;
  return r;
}
// x * 2^K mod (2^96 + 1), 0 < K < 96, for a value that fits 100 bits signed: shift inside 128 bits, fold the part above bit 96
// back negated.  K = 32 q + s.
template <int K>
__device__ __forceinline__ L128 l_shl(L128 x) {
  constexpr int q = K / 32, s = K % 32;
  u32 y[5];   // x << s, five limbs (sign bits above)
  const u32 sg = (u32)((int)x.w[3] >> 31);
  if constexpr (s == 0) { y[0] = x.w[0]; y[1] = x.w[1]; y[2] = x.w[2]; y[3] = x.w[3]; y[4] = sg; }
  else {
    y[0] = x.w[0] << s;
    y[1] = __builtin_amdgcn_alignbit(x.w[1], x.w[0], 32 - s);
    y[2] = __builtin_amdgcn_alignbit(x.w[2], x.w[1], 32 - s);
    y[3] = __builtin_amdgcn_alignbit(x.w[3], x.w[2], 32 - s);
    y[4] = __builtin_amdgcn_alignbit(sg, x.w[3], 32 - s);
  }
  // limbs move up by q; limb index >= 3 wraps to index - 3 with a minus sign (2^96 = -1)
  L128 lo{{0, 0, 0, 0}}, hi{{0, 0, 0, 0}};
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int j = i + q;
    if (j < 3) lo.w[j] = y[i];
    else if (j - 3 < 3) hi.w[j - 3] = y[i];
  }
  hi.w[3] = (u32)((int)hi.w[2] >> 31);   // (sign of the folded part)
  return l_sub(lo, hi);
}
// 128-bit two's complement -> canonical residue: w3 2^96 + w2 2^64 + lo64 = lo64 + w2 (2^32 - 1) - w3 (signed limbs above)
__device__ __forceinline__ u64 l_reduce(L128 x) {
  const u64 lo = ((u64)x.w[1] << 32) | x.w[0];
  // top limb is a small signed number (|w3| < 2^6 here): make everything non-negative by adding 2^6 (2^96 + ...) = -2^6
  const u32 t3 = x.w[3] + 64u;                    // in [0, 128)
  u64 r = gl64::mad_eps_canon(x.w[2], lo);        // + w2 * EPS
  r = gl64::sub32(r, t3);                         // - t3 (2^96 = -1) ...
  return gl64::add(r, 64u);                       // ... + the 64 added above
}
template <int N, int J>
__device__ __forceinline__ void l_stage(L128* x) {
  if constexpr (J < N / 2) {
    L128 a = x[J], b = x[J + N / 2];
    x[J] = l_add(a, b);
    constexpr int E = root_exp(N, J, false);
    if constexpr (E == 0) x[J + N / 2] = l_sub(a, b);
    else if constexpr (E >= 96) x[J + N / 2] = l_shl<E - 96 ? E - 96 : 1>(l_sub(b, a));
    else x[J + N / 2] = l_shl<E>(l_sub(a, b));
    l_stage<N, J + 1>(x);
  }
}
template <int N>
__device__ __forceinline__ void l_dif(L128* x) {
  if constexpr (N >= 2) {
    l_stage<N, 0>(x);
    l_dif<N / 2>(x);
    l_dif<N / 2>(x + N / 2);
  }
}
// NOTE: range.  A shifted value is < 2^96 in magnitude and each later stage can double it: the fourth stage would need 100
// bits.  This probe does not re-normalise between stages (an honest implementation would, at extra cost), so its RESULTS are not
// checked -- it measures the instruction stream a lazy round would at least have to execute.
__global__ void __launch_bounds__(256) lazy_round_kernel(u64* out, const u64* tw, int iters) {
  u64 x[16];
  for (int i = 0; i < 16; i++) x[i] = (u64)(threadIdx.x * 7919u + blockIdx.x + i + 1) * 0x9E3779B97F4A7C15ull % gl64::P;
  const u32 tb = (threadIdx.x & 63) << 3;
  for (int it = 0; it < iters; it++) {
    L128 l[16];
#pragma unroll
    for (int i = 0; i < 16; i++) l[i] = l_from(x[i]);
    l_dif<16>(l);
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = l_reduce(l[i]);
#pragma unroll
    for (int i = 1; i < 16; i++) x[i] = gl64::mul(x[i], ld_tabb(tw, (tb * i + it) & 0x3FF8));
  }
  u64 acc = 0;
  for (int i = 0; i < 16; i++) acc ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---------------------------------------------------------------- C: cross-lane stage through DPP
__device__ __forceinline__ u64 dpp_partner(u64 v) {   // the value of lane ^ 1
  const u32 lo = __builtin_amdgcn_mov_dpp((u32)v, 0xB1, 0xF, 0xF, true);          // quad_perm [1, 0, 3, 2]
  const u32 hi = __builtin_amdgcn_mov_dpp((u32)(v >> 32), 0xB1, 0xF, 0xF, true);
  return ((u64)hi << 32) | lo;
}
// register stage for comparison: 8 butterflies on 16 registers, twiddle 2^24 on the differences (a mid-cost shift)
__global__ void __launch_bounds__(256) reg_stage_kernel(u64* out, int iters) {
  u64 x[16];
  for (int i = 0; i < 16; i++) x[i] = (u64)(threadIdx.x * 7919u + blockIdx.x + i + 1) * 0x9E3779B97F4A7C15ull % gl64::P;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u64 a = x[i], b = x[i + 8];
      x[i] = gl64::add(a, b);
      x[i + 8] = gl64::mul_2exp<24>(gl64::sub(a, b));
    }
  u64 acc = 0;
  for (int i = 0; i < 16; i++) acc ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) xlane_stage_kernel(u64* out, int iters) {
  u64 x[16];
  for (int i = 0; i < 16; i++) x[i] = (u64)(threadIdx.x * 7919u + blockIdx.x + i + 1) * 0x9E3779B97F4A7C15ull % gl64::P;
  const bool odd = threadIdx.x & 1;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u64 own = x[i], oth = dpp_partner(own);
      // even lane: own + other; odd lane: (other - own) * 2^24.  Both lanes execute one instruction stream: the sum and the
      // twiddled difference are both computed, each lane keeps one (the cheapest honest form: a select between two results;
      // predicating on EXEC instead would serialise the two halves, same cost)
      const u64 s = gl64::add(own, oth);
      const u64 d = gl64::mul_2exp<24>(gl64::sub(oth, own));
      x[i] = odd ? d : s;
    }
  u64 acc = 0;
  for (int i = 0; i < 16; i++) acc ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---------------------------------------------------------------- E: wave-uniform shift layer / table layer / LDS exchange
// x[i] *= omega_64^(a * k_i), k_i = brev4(i): a in [0, 4) is wave-uniform -> one scalar branch per layer, compile-time shifts inside
template <int A, int I>
__device__ __forceinline__ void shift_layer_regs(u64* x) {
  if constexpr (I < 16) {
    constexpr int K = brev(I, 4);
    constexpr int E = (39 * A * K) % 192;
    if constexpr (E >= 96) x[I] = gl64::neg(gl64::mul_2exp<E - 96>(x[I]));
    else if constexpr (E > 0) x[I] = gl64::mul_2exp<E>(x[I]);
    shift_layer_regs<A, I + 1>(x);
  }
}
__global__ void __launch_bounds__(256) shift_layer_kernel(u64* out, int iters) {
  u64 x[16];
  for (int i = 0; i < 16; i++) x[i] = (u64)(threadIdx.x * 7919u + blockIdx.x + i + 1) * 0x9E3779B97F4A7C15ull % gl64::P;
  const u32 a = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // the wave index: 0..3, in an SGPR
  for (int it = 0; it < iters; it++) {
    switch ((a + it) & 3) {
      case 0: shift_layer_regs<0, 0>(x); break;
      case 1: shift_layer_regs<1, 0>(x); break;
      case 2: shift_layer_regs<2, 0>(x); break;
      default: shift_layer_regs<3, 0>(x); break;
    }
  }
  u64 acc = 0;
  for (int i = 0; i < 16; i++) acc ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) table_layer_kernel(u64* out, const u64* tw, int iters) {
  u64 x[16];
  for (int i = 0; i < 16; i++) x[i] = (u64)(threadIdx.x * 7919u + blockIdx.x + i + 1) * 0x9E3779B97F4A7C15ull % gl64::P;
  const u32 tb = (threadIdx.x & 63) << 3;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 1; i < 16; i++) x[i] = gl64::mul(x[i], ld_tabb(tw, (tb * i + it) & 0x3FF8));
  u64 acc = 0;
  for (int i = 0; i < 16; i++) acc ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// one LDS exchange of the tile kernel's kind: 16 ds_write_b64 at one per-lane base + immediates, barrier, 16 ds_read_b64 at
// another (256 lanes x 16 x 8 B = 32 KiB per workgroup + the dummy rows; two workgroups per CU: 4 waves per SIMD as in the kernel)
__global__ void __launch_bounds__(256) exchange_kernel(u64* out, int iters) {
  extern __shared__ u64 lds[];
  u64 x[16];
  for (int i = 0; i < 16; i++) x[i] = (u64)(threadIdx.x * 7919u + blockIdx.x + i + 1) * 0x9E3779B97F4A7C15ull % gl64::P;
  const u32 t = threadIdx.x, wbase = t + (t >> 4), rbase = 17 * (t >> 4) * 16 + (t & 15);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) lds[wbase + i * 272] = x[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = lds[(rbase + i * 17) % 4352] + (u64)it;
    __syncthreads();
  }
  u64 acc = 0;
  for (int i = 0; i < 16; i++) acc ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---------------------------------------------------------------- plain v_add_u32 reference (1 slot)
__global__ void __launch_bounds__(256) slot_kernel(u64* out, u32 y, int iters) {
  u32 x[16];
  for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
  u32 acc = 0;
  for (int i = 0; i < 16; i++) acc ^= x[i];
  if (acc == 0x12345u) out[threadIdx.x] = acc;
}

template <class F>
static double time_us(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3;
}

int main() {
  u64 *d_out, *d_tw;
  CK(hipMalloc(&d_out, 2048 * 256 * 8)); CK(hipMalloc(&d_tw, 16384 + 64));
  { u64 h[2056]; u64 s = 99; for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl64::P; }
    CK(hipMemcpy(d_tw, h, sizeof h, hipMemcpyHostToDevice)); }
  const int G = 2048, T = 256;
  const double lanes = (double)G * T;
  const int si = 4096;
  const double slot_us = time_us([&] { hipLaunchKernelGGL(slot_kernel, dim3(G), dim3(T), 0, 0, d_out, 7u, si); }) / ((double)si * 16);   // us per slot per lane-set
  printf("one issue slot (v_add_u32, %d x %d lanes, 8 waves per SIMD): %.4f ns per instruction per grid\n", G, T, slot_us * 1e3);
  auto slots_per_coeff = [&](double us, double coeffs_per_lane_per_iter, int iters) { return us / (slot_us * iters * coeffs_per_lane_per_iter); };
  const int it = 64;
  const double a16 = time_us([&] { hipLaunchKernelGGL((round_kernel<16>), dim3(G), dim3(T), 0, 0, d_out, d_tw, it); });
  printf("A  shipped 16-point round + 15 table multiplies          %8.1f us  %6.1f slots per coefficient\n", a16, slots_per_coeff(a16, 16, it));
  const double b16 = time_us([&] { hipLaunchKernelGGL(lazy_round_kernel, dim3(G), dim3(T), 0, 0, d_out, d_tw, it); });
  printf("B  128-bit lazy round + 16 reductions + 15 multiplies    %8.1f us  %6.1f slots per coefficient\n", b16, slots_per_coeff(b16, 16, it));
  const int it2 = 256;
  const double cr = time_us([&] { hipLaunchKernelGGL(reg_stage_kernel, dim3(G), dim3(T), 0, 0, d_out, it2); });
  const double cx = time_us([&] { hipLaunchKernelGGL(xlane_stage_kernel, dim3(G), dim3(T), 0, 0, d_out, it2); });
  printf("C  one radix-2 stage in registers (shift 2^24)           %8.1f us  %6.1f slots per coefficient\n", cr, slots_per_coeff(cr, 16, it2));
  printf("C  the same stage across lanes (DPP partner, select)     %8.1f us  %6.1f slots per coefficient\n", cx, slots_per_coeff(cx, 16, it2));
  // D: 32 per lane at 4 waves per SIMD (1024 workgroups x 256: half the waves, the same coefficients)
  const double d32 = time_us([&] { hipLaunchKernelGGL((round_kernel<32>), dim3(G / 2), dim3(T), 0, 0, d_out, d_tw, it); });
  const double a16x2 = time_us([&] { hipLaunchKernelGGL((round_kernel<16>), dim3(G), dim3(T), 0, 0, d_out, d_tw, 2 * it); });
  printf("D  32-point round + 31 multiplies, 4 waves per SIMD      %8.1f us  for 5 stages + 1 multiply layer per coefficient\n", d32);
  printf("D  two shipped 16-point rounds, 8 waves per SIMD         %8.1f us  for 8 stages + 2 multiply layers per coefficient\n", a16x2);
  printf("   per stage-equivalent: 32-point %.2f us, 16-point pair %.2f us (same %.0f coefficients)\n", d32 / 5.0, a16x2 / 8.0, lanes * 16);
  // E: the review's wave-uniform shift layer
  const double e0 = time_us([&] { hipLaunchKernelGGL(table_layer_kernel, dim3(G), dim3(T), 0, 0, d_out, d_tw, it2); });
  const double e1 = time_us([&] { hipLaunchKernelGGL(shift_layer_kernel, dim3(G), dim3(T), 0, 0, d_out, it2); });
  const double e2 = time_us([&] { hipLaunchKernelGGL(exchange_kernel, dim3(G), dim3(T), 4352 * 8, 0, d_out, it2); });
  printf("E0 table-twiddle layer (15 multiplies + table loads)      %8.1f us  %6.1f slots per coefficient\n", e0, slots_per_coeff(e0, 16, it2));
  printf("E1 wave-uniform shift layer omega_64^(a k), a in SGPR    %8.1f us  %6.1f slots per coefficient\n", e1, slots_per_coeff(e1, 16, it2));
  printf("E2 one LDS exchange (16 writes, barrier, 16 reads)       %8.1f us  %6.1f slot-equivalents per coefficient (LDS pipe + barrier, not VALU)\n", e2, slots_per_coeff(e2, 16, it2));
  printf("   a 2^11-row pass today: 2 x E0 + 2 x E2 = %.1f; as [16.4].[4.8]: E0 + 2 x E1 + 3 x E2 = %.1f slot-equivalents per coefficient (butterflies equal)\n",
         2 * slots_per_coeff(e0, 16, it2) + 2 * slots_per_coeff(e2, 16, it2),
         slots_per_coeff(e0, 16, it2) + 2 * slots_per_coeff(e1, 16, it2) + 3 * slots_per_coeff(e2, 16, it2));
  (void)lanes;
  return 0;
}
