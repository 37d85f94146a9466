// ntt_aux_kernels.h -- small non-template kernels around the transforms (included by ronk_plan.hip only):
// batched zero padding and the chirps of Bluestein's algorithm.
#pragma once
#include <hip/hip_runtime.h>

#include "field_kernels.h"

namespace ronk {

// From<[F;N]> zero padding (polynomial/mod.rs:503-515) for a batch: out[b][i] = i < k ? in[b][i] : ZERO
__global__ void __launch_bounds__(256) pad_rows_kernel(const u64* __restrict__ in, size_t k, u64* __restrict__ out, size_t n,
                                                        size_t total) {
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t b = t / n, i = t - b * n;
    out[t] = i < k ? in[b * k + i] : 0;
  }
}

// Bluestein chirps (Goldilocks), T[i] = w^i, C(m) = m(m-1)/2:
//   a_rev[n-1-j] = x[j] * w^-C(j)  (j < n),   b[m] = w^C(m)  (m < 2n-1);   out[k] = conv[n-1+k] * w^-C(k)
__device__ __forceinline__ size_t tri_mod(size_t m, size_t n) { return (size_t)(((u64)m * (u64)(m ? m - 1 : 0) / 2) % (u64)n); }
__global__ void __launch_bounds__(256) bluestein_pre_kernel(const u64* __restrict__ x, const u64* __restrict__ T, size_t n,
                                                             u64* __restrict__ a_rev, u64* __restrict__ b) {
  const size_t lb = 2 * n - 1;
  for (size_t m = blockIdx.x * (size_t)blockDim.x + threadIdx.x; m < lb; m += (size_t)gridDim.x * blockDim.x) {
    const size_t e = tri_mod(m, n);
    b[m] = T[e];
    if (m < n) a_rev[n - 1 - m] = gl64::mul(x[m], T[e ? n - e : 0]);
  }
}
__global__ void __launch_bounds__(256) bluestein_post_kernel(const u64* __restrict__ conv, const u64* __restrict__ T, size_t n,
                                                              u64* __restrict__ out) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
    const size_t e = tri_mod(k, n);
    out[k] = gl64::mul(conv[n - 1 + k], T[e ? n - e : 0]);
  }
}

}  // namespace ronk
