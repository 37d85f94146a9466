// fr_scan_kernels.h -- division of a polynomial over BN254's scalar field by a linear divisor (x - z): the polynomial half of
// kzg::open on a production curve (reference src/kzg/setup.rs:63-78: `poly.div([-eval_point, ONE])`, i.e.
// quotient_and_remainder, src/polynomial/mod.rs:170-225, with a monic degree-1 divisor; the quotient has as many entries as
// the dividend, the top one ZERO, and the remainder's constant term is poly(z)).
//
// With S(x) = sum_{i >= x} c_i z^(i-x) (the value of the coefficient suffix starting at x): quot[j] = S(j+1), rem = S(0).
// Same chunked suffix scan as scan_kernels.h / lindiv_kernels.h, on 256-bit elements (bn254_fr.h):
//   1  fr_chunk_sum_kernel   chunk b = 1024 coefficients (256 lanes x 4): Horner over the lane's own four, suffix scan over the
//                            lanes with the multipliers z^(4 2^s) -> H_b = value of the chunk at z
//   2  fr_carry_kernel       ONE workgroup: the same division of the polynomial sum_b H_b X^b by (X - Y), Y = z^1024: its quotient
//                            entry b is the carry into chunk b, G_b = sum_{j > b} H_j Y^(j-b-1) (blocks of 1024 chunks, top first)
//   3  fr_apply_kernel       chunk b again: S(first coefficient of lane t+1) = W_(t+1) + z^(4 (255 - t)) G_b, then the recurrence
//                            q[j-1] = c_j + z q[j] down the lane's four coefficients
// Coefficients stay in STANDARD form; only the multipliers (powers of z, built on the host: FrScanTab) are in Montgomery form.
// 32 bytes per coefficient read twice and written once; a few microseconds per 2^16 coefficients next to a millisecond of MSM.
#pragma once
#include <hip/hip_runtime.h>

#include "bn254_fr.h"

namespace ronk {

using bn254::Fr;

constexpr int FR_PL = 4;                 // coefficients per lane
constexpr int FR_CHUNK = 256 * FR_PL;    // per workgroup

// multipliers for a scan with ratio mu (z for the coefficients, z^1024 for the chunk sums), all in Montgomery form
struct FrScanTab {
  Fr mu;          // mu
  Fr mup[8];      // mu^(PL 2^s): lane-scan steps
  Fr muq[256];    // mu^(PL k): carry into lane 255 - k
  Fr muchunk;     // mu^1024
};

// host: table for ratio mu (standard form, canonical).  (A product of two Montgomery-form values is again in Montgomery form.)
inline void fr_build_tab(const Fr& mu_std, FrScanTab* t) {
  using namespace bn254;
  const Fr mu = fr_to_mont(mu_std);
  t->mu = mu;
  Fr m4 = fr_const_one_mont();
  for (int i = 0; i < FR_PL; i++) m4 = fr_mul(m4, mu);                // mu^PL
  Fr x = fr_const_one_mont();
  for (int k = 0; k < 256; k++) { t->muq[k] = x; x = fr_mul(x, m4); }  // mu^(PL k)
  t->muchunk = x;                                                      // mu^(PL 256) = mu^1024
  for (int s = 0; s < 8; s++) t->mup[s] = t->muq[1 << s];
}
// Montgomery form -> standard form (the chunk ratio Y = z^1024 is handed to fr_build_tab in standard form)
inline Fr fr_from_mont_host(const Fr& a) {
  Fr one = bn254::fr_zero();
  one.l[0] = 1;
  return bn254::fr_mul(a, one);
}

// value at the start of this lane's run within the chunk: W_t = U_t + mu^PL W_(t+1); e[] = the lane's PL entries (standard form)
__device__ __forceinline__ Fr fr_lane_scan(const Fr (&e)[FR_PL], const FrScanTab& tab, Fr* sc, int tid) {
  using namespace bn254;
  Fr U = e[FR_PL - 1];
#pragma unroll
  for (int m = FR_PL - 2; m >= 0; m--) U = fr_add(fr_mul(U, tab.mu), e[m]);
  sc[tid] = U;
  __syncthreads();
  for (int s = 0; s < 8; s++) {
    const int off = 1 << s;
    Fr w = U;
    if (tid + off < 256) w = fr_add(U, fr_mul(sc[tid + off], tab.mup[s]));
    __syncthreads();
    sc[tid] = U = w;
    __syncthreads();
  }
  return U;
}

__device__ __forceinline__ void fr_load_run(const uint64_t* __restrict__ c, size_t n, size_t base, int tid, Fr (&e)[FR_PL]) {
#pragma unroll
  for (int m = 0; m < FR_PL; m++) {
    const size_t i = base + (size_t)FR_PL * tid + m;
    e[m] = i < n ? bn254::fr_canon(bn254::fr_load(c + 4 * i)) : bn254::fr_zero();
  }
}

// launch 1: H[b] (4 words each)
__global__ void __launch_bounds__(256) fr_chunk_sum_kernel(const uint64_t* __restrict__ c, size_t n, const FrScanTab* __restrict__ tab,
                                                            uint64_t* __restrict__ H) {
  __shared__ Fr sc[256];
  const int tid = threadIdx.x;
  Fr e[FR_PL];
  fr_load_run(c, n, (size_t)blockIdx.x * FR_CHUNK, tid, e);
  const Fr W = fr_lane_scan(e, *tab, sc, tid);
  if (tid == 0) bn254::fr_store(H + 4 * (size_t)blockIdx.x, W);
}

// quotient of the lane's run given the value S just above it; returns the value at the run's first entry
__device__ __forceinline__ Fr fr_run_down(const Fr (&e)[FR_PL], Fr r, const Fr& mu, Fr (&o)[FR_PL]) {
  using namespace bn254;
#pragma unroll
  for (int m = FR_PL - 1; m >= 0; m--) { o[m] = r; r = fr_add(fr_mul(r, mu), e[m]); }
  return r;
}

// launch 2 (one workgroup): G[b] = sum_{j > b} H_j Y^(j-b-1) for every chunk b, *rem = sum_b H_b Y^b = poly(z).  tabY: ratio Y.
__global__ void __launch_bounds__(256) fr_carry_kernel(const uint64_t* __restrict__ H, size_t nchunks, const FrScanTab* __restrict__ tabY,
                                                        uint64_t* __restrict__ G, uint64_t* __restrict__ rem) {
  using namespace bn254;
  __shared__ Fr sc[256];
  __shared__ Fr carry_s;
  const int tid = threadIdx.x;
  Fr carry = fr_zero();                                   // value of everything above the current block, at the block's end
  const size_t nblocks = (nchunks + FR_CHUNK - 1) / FR_CHUNK;
  for (size_t blk = nblocks; blk-- > 0;) {
    const size_t base = blk * FR_CHUNK;
    Fr e[FR_PL], o[FR_PL];
    fr_load_run(H, nchunks, base, tid, e);
    const Fr W = fr_lane_scan(e, *tabY, sc, tid);         // sc[t] = W_t afterwards
    const Fr wn = tid < 255 ? sc[tid + 1] : fr_zero();
    const Fr above = fr_add(wn, fr_mul(carry, tabY->muq[255 - tid]));   // S(first entry of lane t+1)
    const Fr first = fr_run_down(e, above, tabY->mu, o);
#pragma unroll
    for (int m = 0; m < FR_PL; m++) {
      const size_t i = base + (size_t)FR_PL * tid + m;
      if (i < nchunks) fr_store(G + 4 * i, o[m]);
    }
    __syncthreads();
    if (tid == 0) carry_s = first;                        // S(base): what the next (lower) block sees above itself
    __syncthreads();
    carry = carry_s;
    (void)W;
  }
  if (tid == 0 && rem) fr_store(rem, carry);
}

// launch 3: the quotient
__global__ void __launch_bounds__(256) fr_apply_kernel(const uint64_t* __restrict__ c, size_t n, const FrScanTab* __restrict__ tab,
                                                        const uint64_t* __restrict__ G, uint64_t* __restrict__ quot) {
  using namespace bn254;
  __shared__ Fr sc[256];
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * FR_CHUNK;
  Fr e[FR_PL], o[FR_PL];
  fr_load_run(c, n, base, tid, e);
  (void)fr_lane_scan(e, *tab, sc, tid);
  const Fr cin = fr_load(G + 4 * (size_t)blockIdx.x);
  const Fr wn = tid < 255 ? sc[tid + 1] : fr_zero();
  const Fr above = fr_add(wn, fr_mul(cin, tab->muq[255 - tid]));
  (void)fr_run_down(e, above, tab->mu, o);
#pragma unroll
  for (int m = 0; m < FR_PL; m++) {
    const size_t i = base + (size_t)FR_PL * tid + m;
    if (i < n) fr_store(quot + 4 * i, o[m]);
  }
}

}  // namespace ronk
