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
                                                          size_t k, u64* __restrict__ w, int* flag, const int* skip_if_zero) {
  if (skip_if_zero && *skip_if_zero == 0) return;   // the O(K log K) form applies (rsf_check_kernel)
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
__global__ void __launch_bounds__(1024) master_poly_kernel(Ops ops, const u64* __restrict__ xs, size_t k, u64* __restrict__ m, const int* skip_if_zero) {
  if (skip_if_zero && *skip_if_zero == 0) return;   // the O(K log K) form applies (rsf_check_kernel)
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
                                                             const u64* __restrict__ m, size_t k, u64* __restrict__ partial, const int* skip_if_zero) {
  if (skip_if_zero && *skip_if_zero == 0) return;   // the O(K log K) form applies (rsf_check_kernel)
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
                                                         u64* __restrict__ out, const int* skip_if_zero) {
  if (skip_if_zero && *skip_if_zero == 0) return;   // the O(K log K) form applies (rsf_check_kernel)
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i >= k) return;
  u64 acc = 0;
  for (size_t b = 0; b < nblk; b++) acc = ops.add(acc, partial[b * k + i]);
  out[i] = acc;
}

// ---- the same decode in O(K log K) for the node sequences Message::encode produces: x_j = q^j (q = omega_N, any order > K).
// With B_n = prod_{d=1..n} (1 - q^d)  (one prefix product):
//   M'(x_j) = prod_{m != j} (q^j - q^m) = q^{j(K-1)} (-1)^j B_j B_{K-1-j} / q^{j(j+1)/2}            -> c_j = y_j / M'(x_j)
//   M(X)    = prod_{m < K} (X - q^m)   = sum_i (-1)^{K-i} q^{(K-i)(K-i-1)/2} B_K / (B_i B_{K-i}) X^i   (q-binomial theorem)
//   data(X) = M(X) * sum_j c_j / (X - q^j) = polynomial part of  M(X) * sum_t s_t X^{-t-1},   s_t = sum_j c_j q^{jt}
// s_t (t < K) is a chirp-z transform with base q -- jt = C(j+t) - C(j) - C(t), C(m) = m(m-1)/2, so only integer powers of q
// occur and q's order never has to be known -- i.e. one linear convolution, and the polynomial part is a second one:
//   data_e = sum_t s_t M_{e+1+t} = (reverse(s) * M)[e + K].
// Every step is an exact identity in F_p: the K coefficients are those of the unique interpolating polynomial, i.e. the
// reference's Message::decode value for value.  rsf_check_kernel verifies what the identities need (x_0 = 1, x_j = x_{j-1} x_1,
// x_j != 1 for 0 < j < K, q^K != 1); otherwise bit 2 of *flag is set.  Goldilocks only (the convolutions run on the NTT path).
__global__ void __launch_bounds__(256) rsf_check_kernel(const u64* __restrict__ xs, size_t k, int* flag) {
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (j >= k) return;
  bool ok;
  // (q = x_1 = 0 would pass every other test -- x_j = x_{j-1} * 0 = 0 != 1 -- with coincident nodes: those belong to the
  // general kernels, which report the reference's division by zero)
  if (j == 0) ok = xs[0] == 1 && (k < 2 || (xs[1] != 0 && gl64::mul(xs[k - 1], xs[1]) != 1));
  else ok = xs[j] == gl64::mul(xs[j - 1], xs[1]) && xs[j] != 1;
  if (!ok) atomicOr(flag, 4);
}
// factors F[d] = 1 - q^d for d = 1 .. k (F[0] = 1): q^d = xs[d] below k, q^k = xs[k-1] * q
__global__ void __launch_bounds__(256) rsf_factors_kernel(const u64* __restrict__ xs, size_t k, u64* __restrict__ F) {
  for (size_t d = blockIdx.x * (size_t)blockDim.x + threadIdx.x; d <= k; d += (size_t)gridDim.x * blockDim.x)
    F[d] = d == 0 ? 1 : gl64::sub(1, d < k ? xs[d] : gl64::mul(xs[k - 1], k > 1 ? xs[1] : 1));
}
// inclusive prefix PRODUCT in place, three launches like the MSM's prefix sum: per-block products (256 lanes x PER
// consecutive entries), one workgroup scanning the <= 1024 block products, every block redone from its base
constexpr u32 RSF_PER = 16;   // 4096 entries per workgroup: 1024 blocks cover 2^22 entries
__global__ void __launch_bounds__(256) rsf_scan_totals_kernel(const u64* __restrict__ F, size_t m, u64* __restrict__ tot) {
  __shared__ u64 red[256];
  const u32 tid = threadIdx.x;
  const size_t base = ((size_t)blockIdx.x * 256 + tid) * RSF_PER;
  u64 v = 1;
  for (u32 j = 0; j < RSF_PER; j++) if (base + j < m) v = gl64::mul(v, F[base + j]);
  red[tid] = v;
  __syncthreads();
  for (u32 s = 1; s < 256; s <<= 1) {          // ordered tree: red[i] *= red[i + s] keeps the left-to-right order (commutative anyway)
    if ((tid & (2 * s - 1)) == 0) red[tid] = gl64::mul(red[tid], red[tid + s]);
    __syncthreads();
  }
  if (tid == 0) tot[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(1024) rsf_scan_mid_kernel(u64* __restrict__ tot, u32 nb) {
  __shared__ u64 part[1024];
  const u32 tid = threadIdx.x;
  const u64 v0 = tid < nb ? tot[tid] : 1;
  part[tid] = v0;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    const u64 v = tid >= off ? part[tid - off] : 1;
    __syncthreads();
    part[tid] = gl64::mul(part[tid], v);
    __syncthreads();
  }
  if (tid < nb) tot[tid] = tid ? part[tid - 1] : 1;      // exclusive: the product of the blocks before this one
}
__global__ void __launch_bounds__(256) rsf_scan_apply_kernel(u64* __restrict__ F, size_t m, const u64* __restrict__ tot) {
  __shared__ u64 part[256];
  const u32 tid = threadIdx.x;
  const size_t base = ((size_t)blockIdx.x * 256 + tid) * RSF_PER;
  u64 v = 1;
  for (u32 j = 0; j < RSF_PER; j++) if (base + j < m) v = gl64::mul(v, F[base + j]);
  part[tid] = v;
  __syncthreads();
  for (u32 off = 1; off < 256; off <<= 1) {
    const u64 w = tid >= off ? part[tid - off] : 1;
    __syncthreads();
    part[tid] = gl64::mul(part[tid], w);
    __syncthreads();
  }
  u64 run = gl64::mul(tot[blockIdx.x], tid ? part[tid - 1] : 1);
  for (u32 j = 0; j < RSF_PER; j++)
    if (base + j < m) { run = gl64::mul(run, F[base + j]); F[base + j] = run; }
}
// c_j (chirped and reversed for the convolution: a_rev[k-1-j] = c_j q^-C(j)), the chirp b[m] = q^C(m), m < 2k-1, and M_i
__global__ void __launch_bounds__(256) rsf_prepare_kernel(const u64* __restrict__ xs, const u64* __restrict__ ys,
                                                           const u64* __restrict__ B, size_t k, u64* __restrict__ a_rev,
                                                           u64* __restrict__ b, u64* __restrict__ M) {
  const size_t lb = 2 * k - 1;
  const u64 q = xs[1], qinv = gl64::inv(q);      // (k >= 2; a q of 0 only comes with a failed structure check)
  for (size_t m = blockIdx.x * (size_t)blockDim.x + threadIdx.x; m < lb; m += (size_t)gridDim.x * blockDim.x) {
    const u64 tri = (u64)m * (u64)(m ? m - 1 : 0) / 2;                     // C(m) < 2^43
    b[m] = gl64::pow(q, tri);
    if (m < k) {
      const u64 j = m;
      // 1 / M'(x_j) = q^{j(j+1)/2} (-1)^j / (q^{j(k-1)} B_j B_{k-1-j})
      const u64 den = gl64::mul(gl64::pow(q, j * (u64)(k - 1)), gl64::mul(B[j], B[k - 1 - j]));
      u64 c = gl64::mul(gl64::mul(ys[j], gl64::pow(q, j * (j + 1) / 2)), gl64::inv(den));
      if (j & 1) c = gl64::neg(c);
      a_rev[k - 1 - j] = gl64::mul(c, gl64::pow(qinv, tri));
    }
    if (m <= k) {
      const u64 i = m, r = k - i;
      u64 v = gl64::mul(gl64::mul(gl64::pow(q, r * (r ? r - 1 : 0) / 2), B[k]), gl64::inv(gl64::mul(B[i], B[r])));
      if (r & 1) v = gl64::neg(v);
      M[i] = v;
    }
  }
}
// s_rev[k-1-t] = conv[k-1+t] * q^-C(t)    (s_t, reversed for the second convolution)
__global__ void __launch_bounds__(256) rsf_unchirp_kernel(const u64* __restrict__ xs, const u64* __restrict__ conv, size_t k,
                                                           u64* __restrict__ s_rev) {
  const u64 qinv = gl64::inv(xs[1]);
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < k; t += (size_t)gridDim.x * blockDim.x)
    s_rev[k - 1 - t] = gl64::mul(conv[k - 1 + t], gl64::pow(qinv, (u64)t * (u64)(t ? t - 1 : 0) / 2));
}
// out[e] = conv2[e + k] unless the structure check failed
__global__ void __launch_bounds__(256) rsf_extract_kernel(const u64* __restrict__ conv2, size_t k, const int* __restrict__ flag,
                                                           u64* __restrict__ out) {
  if (*flag & 4) return;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < k; e += (size_t)gridDim.x * blockDim.x) out[e] = conv2[e + k];
}

}  // namespace ronk
