// interp_kernels.h -- Reed-Solomon Message::decode (src/codes/reed_solomon.rs:54-106) on the GPU.
//
// The reference interpolates the first K coordinates (x_j, y_j) with Lagrange's formula written through sums
// over `combinations`; the signs fold to the textbook form
//   data(X) = sum_j  y_j / prod_{m != j} (x_j - x_m)  *  M(X) / (X - x_j),      M(X) = prod_m (X - x_m)
// (coefficient i of prod_{m != j}(X - x_m) is (-1)^(K-1-i) e_(K-1-i), the reference multiplies e by (-1)^i and
// divides by prod (x_m - x_j) = (-1)^(K-1) prod (x_j - x_m): the same number).  O(K^2) like the oracle, organised
// for the GPU: one lane per node for the weights and for the synthetic division M / (X - x_j); sums over nodes
// are wave reductions.  Correct, not tuned (K <= 2^14).  Coincident nodes are the reference's panic
// (numerator / ZERO -> inverse().unwrap()): *flag is raised.
#pragma once
#include <hip/hip_runtime.h>

#include "field_kernels.h"

namespace ronk {

constexpr size_t RS_DECODE_MAX_K = (size_t)1 << 14;

// w_j = y_j / prod_{m != j} (x_j - x_m)
template <class Ops>
__global__ void __launch_bounds__(256) rs_weights_kernel(Ops ops, const u64* __restrict__ xs, const u64* __restrict__ ys,
                                                          size_t k, u64* __restrict__ w, int* flag) {
  __shared__ u64 chunk[256];
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const u64 xj = j < k ? xs[j] : 0;
  u64 den = ops.one();
  for (size_t m0 = 0; m0 < k; m0 += 256) {
    __syncthreads();
    if (m0 + threadIdx.x < k) chunk[threadIdx.x] = xs[m0 + threadIdx.x];
    __syncthreads();
    const size_t lim = k - m0 < 256 ? k - m0 : 256;
    if (j < k)
      for (size_t m = 0; m < lim; m++)
        if (m0 + m != j) {
          const u64 df = ops.sub(xj, chunk[m]);
          if (df == 0) *flag = 1;
          den = ops.mul(den, df);
        }
  }
  if (j < k) w[j] = ops.mul(ys[j], ops.pow(den, ops.order() - 2));
}

// M(X) = prod_{s < k} (X - x_s), k+1 coefficients in m[] (global, L2-resident).  One workgroup; step s turns
// the degree-s product into the degree-(s+1) one: new[i] = old[i-1] - x_s * old[i].
template <class Ops>
__global__ void __launch_bounds__(1024) master_poly_kernel(Ops ops, const u64* __restrict__ xs, size_t k, u64* __restrict__ m) {
  const int tid = threadIdx.x;
  constexpr int PER = (int)(RS_DECODE_MAX_K / 1024) + 1;  // indices tid + 1024 r, r < PER, cover 0..k
  for (size_t i = tid; i <= k; i += 1024) m[i] = i == 0 ? ops.one() : 0;
  __syncthreads();
  for (size_t s = 0; s < k; s++) {
    const u64 x = xs[s];
    u64 nv[PER];
#pragma unroll
    for (int r = 0; r < PER; r++) {
      const size_t i = (size_t)tid + 1024 * (size_t)r;
      if (i <= s + 1) {
        const u64 lo = i ? m[i - 1] : 0, cur = i <= s ? m[i] : 0;
        nv[r] = ops.sub(lo, ops.mul(x, cur));
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; r++) {
      const size_t i = (size_t)tid + 1024 * (size_t)r;
      if (i <= s + 1) m[i] = nv[r];
    }
    __syncthreads();
  }
}

// partial[blk][i] = sum over the block's nodes j of w_j * N_j[i],  N_j = M / (X - x_j) by synthetic division
// (N_j[k-1] = 1, N_j[i-1] = m[i] + x_j N_j[i]), i from k-1 down; 32 coefficients per reduction round.
template <class Ops>
__global__ void __launch_bounds__(256) rs_accumulate_kernel(Ops ops, const u64* __restrict__ xs, const u64* __restrict__ w,
                                                             const u64* __restrict__ m, size_t k, u64* __restrict__ partial) {
  constexpr int TI = 32;
  __shared__ u64 red[4][TI];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t j = blockIdx.x * (size_t)256 + tid;
  const bool valid = j < k;
  const u64 xj = valid ? xs[j] : 0, wj = valid ? w[j] : 0;
  u64 q = ops.one();
  for (size_t hi = k; hi > 0;) {
    const int cnt = hi < (size_t)TI ? (int)hi : TI;
    for (int u = 0; u < cnt; u++) {
      const size_t i = hi - 1 - u;
      u64 val = ops.mul(wj, q);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) val = ops.add(val, (u64)__shfl_xor((unsigned long long)val, off));
      if (lane == 0) red[wave][u] = val;
      q = ops.add(m[i], ops.mul(xj, q));
    }
    __syncthreads();
    if (tid < cnt)
      partial[blockIdx.x * k + (hi - 1 - tid)] = ops.add(ops.add(red[0][tid], red[1][tid]), ops.add(red[2][tid], red[3][tid]));
    __syncthreads();
    hi -= cnt;
  }
}

template <class Ops>
__global__ void __launch_bounds__(256) rs_finish_kernel(Ops ops, const u64* __restrict__ partial, size_t nblk, size_t k,
                                                         u64* __restrict__ out) {
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i >= k) return;
  u64 acc = 0;
  for (size_t b = 0; b < nblk; b++) acc = ops.add(acc, partial[b * k + i]);
  out[i] = acc;
}

}  // namespace ronk
