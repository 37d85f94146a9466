// field_kernels.h -- element-wise field kernels and the generic (any prime) polynomial kernels.
//
// Templated on an Ops type that supplies canonical-in / canonical-out add, sub, mul:
//   GlOps    Goldilocks fast path (gl64.h)
//   MontOps  any odd prime p < 2^64 (mont64.h) -- the reference's PrimeField<P> for F_101, F_17, ...
//   Mod2Ops  p = 2 (the reference's AESField = PrimeField<2>)
// These restate ronkathon's per-element semantics (src/algebra/field/prime/arithmetic.rs:3-71,
// src/polynomial/mod.rs:133-258, src/polynomial/arithmetic.rs:16-119) as grid-stride HIP kernels.
// HBM-bound streaming work: 8-byte loads, lanes along the array, no LDS needed except reductions.
#pragma once
#include <hip/hip_runtime.h>

#include "gl64.h"
#include "mont64.h"
#include "longdiv_kernel.h"

namespace ronk {

typedef uint64_t u64;
typedef uint32_t u32;

struct GlOps {
  __device__ __forceinline__ u64 add(u64 a, u64 b) const { return gl64::add(a, b); }
  __device__ __forceinline__ u64 sub(u64 a, u64 b) const { return gl64::sub(a, b); }
  __device__ __forceinline__ u64 mul(u64 a, u64 b) const { return gl64::mul(a, b); }
  __device__ __forceinline__ u64 neg(u64 a) const { return gl64::neg(a); }
  __device__ __forceinline__ u64 pow(u64 a, u64 e) const { return gl64::pow(a, e); }
  __device__ __forceinline__ u64 one() const { return 1; }
  __device__ __forceinline__ u64 order() const { return gl64::P; }
};

struct MontOps {
  mont64::Field f;
  __device__ __forceinline__ u64 add(u64 a, u64 b) const { return mont64::add(f, a, b); }
  __device__ __forceinline__ u64 sub(u64 a, u64 b) const { return mont64::sub(f, a, b); }
  __device__ __forceinline__ u64 mul(u64 a, u64 b) const { return mont64::mul(f, a, b); }
  __device__ __forceinline__ u64 neg(u64 a) const { return mont64::neg(f, a); }
  __device__ __forceinline__ u64 pow(u64 a, u64 e) const { return mont64::pow(f, a, e); }
  __device__ __forceinline__ u64 one() const { return 1; }
  __device__ __forceinline__ u64 order() const { return f.p; }
};

struct Mod2Ops {
  __device__ __forceinline__ u64 add(u64 a, u64 b) const { return a ^ b; }
  __device__ __forceinline__ u64 sub(u64 a, u64 b) const { return a ^ b; }
  __device__ __forceinline__ u64 mul(u64 a, u64 b) const { return a & b; }
  __device__ __forceinline__ u64 neg(u64 a) const { return a; }
  __device__ __forceinline__ u64 pow(u64 a, u64 e) const { return e ? a : 1; }
  __device__ __forceinline__ u64 one() const { return 1; }
  __device__ __forceinline__ u64 order() const { return 2; }
};

enum VecOp { VEC_ADD, VEC_SUB, VEC_MUL };

// out[i] = a[i] (op) b[i]; b is read as ZERO beyond nb (Polynomial Add/Sub zero-extension,
// polynomial/arithmetic.rs:23-34)
template <class Ops, int OP>
__global__ void __launch_bounds__(256) vec_binary_kernel(Ops ops, const u64* __restrict__ a, const u64* __restrict__ b,
                                                          u64* __restrict__ out, size_t n, size_t nb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u64 x = a[i], y = i < nb ? b[i] : 0;
    out[i] = OP == VEC_ADD ? ops.add(x, y) : OP == VEC_SUB ? ops.sub(x, y) : ops.mul(x, y);
  }
}

// the same on 16-byte accesses (two elements per lane per step): n even, b as long as a, 16-byte aligned arrays
template <class Ops, int OP>
__global__ void __launch_bounds__(256) vec_binary2_kernel(Ops ops, const ulonglong2* a, const ulonglong2* b, ulonglong2* out,
                                                           size_t npairs) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npairs; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong2 x = a[i], y = b[i];
    ulonglong2 r;
    r.x = OP == VEC_ADD ? ops.add(x.x, y.x) : OP == VEC_SUB ? ops.sub(x.x, y.x) : ops.mul(x.x, y.x);
    r.y = OP == VEC_ADD ? ops.add(x.y, y.y) : OP == VEC_SUB ? ops.sub(x.y, y.y) : ops.mul(x.y, y.y);
    out[i] = r;
  }
}

template <class Ops>
__global__ void __launch_bounds__(256) vec_neg_kernel(Ops ops, const u64* __restrict__ a, u64* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = ops.neg(a[i]);
}

// Field::pow (prime/mod.rs:74-84); with e = p-2 and flag != null it is Field::inverse
// (prime/mod.rs:62-72): a zero input raises *flag (the reference returns None / panics on unwrap)
template <class Ops>
__global__ void __launch_bounds__(256) vec_pow_kernel(Ops ops, const u64* __restrict__ a, u64 e, u64* __restrict__ out,
                                                       size_t n, int* flag) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u64 x = a[i];
    if (flag && x == 0) *flag = 1;
    out[i] = ops.pow(x, e);
  }
}

// FieldExt::euler_criterion (field/mod.rs:79-84, prime/mod.rs:172): out[i] = (a[i]^((p-1)/2) == 1)
template <class Ops>
__global__ void __launch_bounds__(256) vec_euler_kernel(Ops ops, const u64* __restrict__ a, u64* __restrict__ out, size_t n) {
  const u64 e = (ops.order() - 1) / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = ops.pow(a[i], e) == 1 ? 1 : 0;
}

// FieldExt::sqrt (prime/mod.rs:174-226), Tonelli-Shanks per element with the constants of the prime from the host
// (p - 1 = q 2^s, c0 = z^q for the first non-residue z >= 2): ZERO -> (0, 0); a non-residue raises *flag (the reference's
// assert); the pair is stored smaller root first, as the reference returns it.
template <class Ops>
__global__ void __launch_bounds__(256) vec_sqrt_kernel(Ops ops, const u64* __restrict__ a, u64* __restrict__ r0, u64* __restrict__ r1,
                                                        size_t n, u64 q, u32 s, u64 c0, int* flag) {
  const u64 half = (ops.order() - 1) / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const u64 x = a[i];
    if (x == 0) { r0[i] = 0; r1[i] = 0; continue; }
    if (ops.pow(x, half) != 1) { if (flag) *flag = 1; r0[i] = 0; r1[i] = 0; continue; }
    u32 m = s;
    u64 c = c0, t = ops.pow(x, q), r = ops.pow(x, (q + 1) / 2);
    while (t != 1) {
      u32 j = 1;
      u64 tp = ops.mul(t, t);
      while (tp != 1) { tp = ops.mul(tp, tp); j++; }
      u64 b = c;
      for (u32 k = 0; k + j + 1 < m; k++) b = ops.mul(b, b);   // c^(2^(m - j - 1))
      m = j;
      c = ops.mul(b, b);
      t = ops.mul(t, c);
      r = ops.mul(r, b);
    }
    const u64 nr = ops.neg(r);
    r0[i] = nr < r ? nr : r;
    r1[i] = nr < r ? r : nr;
  }
}

// t[i] = w^i (Lagrange::new's nodes, polynomial/mod.rs:363)
template <class Ops>
__global__ void __launch_bounds__(256) power_table_kernel(Ops ops, u64 w, u64* __restrict__ t, size_t n) {
  size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
  if (i0 >= n) return;
  u64 x = ops.pow(w, i0), ws = ops.pow(w, step);
  for (size_t i = i0; i < n; i += step) { t[i] = x; x = ops.mul(x, ws); }
}

// Polynomial::dft (polynomial/mod.rs:240-258): out[i] = sum_j c[j] * w^(i*j).  One work-item per
// output, the input staged through LDS in chunks.  O(n^2): only for n without a fast path.
template <class Ops>
__global__ void __launch_bounds__(256) dft_naive_kernel(Ops ops, const u64* __restrict__ in, u64* __restrict__ out,
                                                         size_t n, u64 w) {
  __shared__ u64 chunk[256];
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const u64 wi = ops.pow(w, i < n ? i : 0);
  u64 acc = 0, wij = 1;
  for (size_t j0 = 0; j0 < n; j0 += 256) {
    __syncthreads();
    if (j0 + threadIdx.x < n) chunk[threadIdx.x] = in[j0 + threadIdx.x];
    __syncthreads();
    const size_t lim = n - j0 < 256 ? n - j0 : 256;
    for (size_t j = 0; j < lim; j++) {
      acc = ops.add(acc, ops.mul(chunk[j], wij));
      wij = ops.mul(wij, wi);
    }
  }
  if (i < n) out[i] = acc;
}

// impl Mul (polynomial/arithmetic.rs:97-119), schoolbook: out[k] = sum_{i+j=k} a[i] b[j]
template <class Ops>
__global__ void __launch_bounds__(256) poly_mul_schoolbook_kernel(Ops ops, const u64* __restrict__ a, size_t d,
                                                                   const u64* __restrict__ b, size_t d2,
                                                                   u64* __restrict__ out) {
  const size_t m = d + d2 - 1;
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < m; k += (size_t)gridDim.x * blockDim.x) {
    const size_t lo = k >= d2 ? k - d2 + 1 : 0, hi = k < d ? k : d - 1;
    u64 acc = 0;
    for (size_t i = lo; i <= hi; i++) acc = ops.add(acc, ops.mul(a[i], b[k - i]));
    out[k] = acc;
  }
}

// quotient_and_remainder (polynomial/mod.rs:170-225), the long-division kernel: longdiv_kernel.h (its body also runs on host
// fibers in the CPU suite)

// ---- the same evaluate for the node tables Lagrange::new builds (polynomial/mod.rs:358-365: nodes[i] = omega^i, omega of
// order n), in O(n): prod_{m != j} (x_j - x_m) = n * x_j^(n-1) = n / x_j and prod_i (x - x_i) = x^n - 1, so
//   L(x) = (x^n - 1) / n * sum_j c_j x_j / (x - x_j)
// -- the same field element as the general formula above (at a node both give l(x) * (...) = 0, like the reference's
// fold).  lagrange_check_kernel verifies the structure the identity needs: nodes[0] == 1, nodes[j] == nodes[j-1] * nodes[1],
// nodes[1]^n == 1 and nodes[1]^(n/q) != 1 for every prime q | n (order exactly n, hence n distinct nodes); otherwise bit 2 of
// *status is set and the result is meaningless (the entry point reports RONK_ERR_UNSUPPORTED).
struct LagPrimes { u64 q[16]; int count; };
template <class Ops>
__global__ void __launch_bounds__(256) lagrange_check_kernel(Ops ops, const u64* __restrict__ nodes, size_t n, LagPrimes pr,
                                                              int* status) {
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (j >= n) return;
  bool ok = true;
  if (j == 0) {
    ok = nodes[0] == ops.one() % ops.order();
    if (n > 1) {
      const u64 w = nodes[1];
      ok = ok && ops.pow(w, (u64)n) == 1;
      for (int i = 0; i < pr.count; i++) ok = ok && ops.pow(w, (u64)n / pr.q[i]) != 1;
    }
  } else {
    ok = nodes[j] == ops.mul(nodes[j - 1], nodes[1]);
  }
  if (!ok) atomicOr(status, 4);
}
template <class Ops>
__global__ void __launch_bounds__(256) lagrange_fast_terms_kernel(Ops ops, const u64* __restrict__ c, const u64* __restrict__ nodes,
                                                                   size_t n, u64 x, u64* __restrict__ part_sum) {
  __shared__ u64 rs[256];
  u64 acc = 0;
  for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
    const u64 xj = nodes[j], fac = ops.sub(x, xj);
    // fac == 0 (x is a node): pow gives 0, the term vanishes, and the factor x^n - 1 makes the whole value 0 anyway
    acc = ops.add(acc, ops.mul(ops.mul(c[j], xj), ops.pow(fac, ops.order() - 2)));
  }
  rs[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) rs[threadIdx.x] = ops.add(rs[threadIdx.x], rs[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) part_sum[blockIdx.x] = rs[0];
}
template <class Ops>
__global__ void __launch_bounds__(256) lagrange_fast_finish_kernel(Ops ops, const u64* __restrict__ part_sum, size_t nparts,
                                                                    size_t n, u64 x, u64* out, const int* only_if_zero) {
  __shared__ u64 rs[256];
  if (only_if_zero && *only_if_zero != 0) return;   // not an omega^i table: the general kernels write the value
  u64 s = 0;
  for (size_t i = threadIdx.x; i < nparts; i += 256) s = ops.add(s, part_sum[i]);
  rs[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) rs[threadIdx.x] = ops.add(rs[threadIdx.x], rs[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const u64 l = ops.sub(ops.pow(x, (u64)n), 1 % ops.order());                 // x^n - 1
    const u64 ninv = ops.pow((u64)n % ops.order(), ops.order() - 2);
    *out = ops.mul(ops.mul(l, ninv), rs[0]);
  }
}

// ---- Polynomial::<Lagrange>::evaluate (polynomial/mod.rs:382-415), barycentric form --------------------
//   w_j = prod_{m != j} 1/(x_j - x_m),  l(x) = prod_i (x - x_i),  L(x) = l(x) * sum_j c_j w_j / (x - x_j)
// One work-item per node j: den_j = (x - x_j) * prod_{m != j} (x_j - x_m) with the nodes staged through LDS,
// term_j = c_j / den_j (ONE inversion per node instead of the reference's n), then block reductions of the
// sum of terms and of the product of (x - x_j).  If x is a node, l(x) == 0 and the reference's fold returns
// l(x) * (...) == ZERO (its `return c` only replaces the accumulator): den_j == 0 there is inverted to 0, no
// panic, result 0 -- same value.  Coincident nodes make the reference panic (ONE.div(ZERO)): flag -> -2.
template <class Ops>
__global__ void __launch_bounds__(256) lagrange_terms_kernel(Ops ops, const u64* __restrict__ c, const u64* __restrict__ nodes,
                                                              size_t n, u64 x, u64* __restrict__ part_sum,
                                                              u64* __restrict__ part_prod, int* flag, const int* skip_if_zero) {
  __shared__ u64 chunk[256];
  if (skip_if_zero && *skip_if_zero == 0) return;   // the O(n) form applies (lagrange_check_kernel) and computes the value
  __shared__ u64 rs[256];
  __shared__ u64 rp[256];
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const u64 xj = j < n ? nodes[j] : 0;
  u64 den = 1;
  for (size_t m0 = 0; m0 < n; m0 += 256) {
    __syncthreads();
    if (m0 + threadIdx.x < n) chunk[threadIdx.x] = nodes[m0 + threadIdx.x];
    __syncthreads();
    const size_t lim = n - m0 < 256 ? n - m0 : 256;
    if (j < n)
      for (size_t m = 0; m < lim; m++)
        if (m0 + m != j) {
          const u64 d = ops.sub(xj, chunk[m]);
          if (d == 0) *flag = 1;
          den = ops.mul(den, d);
        }
  }
  u64 term = 0, fac = 1;
  if (j < n) {
    fac = ops.sub(x, xj);
    den = ops.mul(den, fac);
    term = ops.mul(c[j], ops.pow(den, ops.order() - 2));
  }
  rs[threadIdx.x] = term;
  rp[threadIdx.x] = fac;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      rs[threadIdx.x] = ops.add(rs[threadIdx.x], rs[threadIdx.x + s]);
      rp[threadIdx.x] = ops.mul(rp[threadIdx.x], rp[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_sum[blockIdx.x] = rs[0]; part_prod[blockIdx.x] = rp[0]; }
}
template <class Ops>
__global__ void __launch_bounds__(256) lagrange_finish_kernel(Ops ops, const u64* __restrict__ part_sum,
                                                               const u64* __restrict__ part_prod, size_t nparts, u64* out,
                                                               const int* skip_if_zero) {
  __shared__ u64 rs[256];
  __shared__ u64 rp[256];
  if (skip_if_zero && *skip_if_zero == 0) return;
  u64 s = 0, p = 1;
  for (size_t i = threadIdx.x; i < nparts; i += 256) { s = ops.add(s, part_sum[i]); p = ops.mul(p, part_prod[i]); }
  rs[threadIdx.x] = s;
  rp[threadIdx.x] = p;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      rs[threadIdx.x] = ops.add(rs[threadIdx.x], rs[threadIdx.x + st]);
      rp[threadIdx.x] = ops.mul(rp[threadIdx.x], rp[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = ops.mul(rp[0], rs[0]);
}

// ---- generic power-of-two NTT for fields without the Goldilocks fast path -------------------
// bit-reversal copy, then log2(n) radix-2 decimation-in-time stages over HBM (the reference's
// recursion, polynomial/mod.rs:295-323, unrolled bottom-up); twiddles from a w^i table.
__device__ __forceinline__ u32 bitrev32(u32 x, int bits) { return __brev(x) >> (32 - bits); }

template <class Ops>
__global__ void __launch_bounds__(256) bitrev_copy_kernel(Ops, const u64* __restrict__ in, u64* __restrict__ out,
                                                           int log2n, size_t total) {
  const size_t n = (size_t)1 << log2n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t poly = i >> log2n, j = i & (n - 1);
    out[(poly << log2n) + (log2n ? bitrev32((u32)j, log2n) : 0)] = in[i];
  }
}
// stage s (half = 2^s): pairs (i, i+half) inside blocks of 2*half, twiddle w^(j * n/(2 half))
template <class Ops>
__global__ void __launch_bounds__(256) radix2_stage_kernel(Ops ops, u64* __restrict__ x, const u64* __restrict__ wtab,
                                                            int log2n, int s, size_t total_pairs, u64 scale) {
  const size_t half = (size_t)1 << s;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total_pairs; t += (size_t)gridDim.x * blockDim.x) {
    const size_t poly = t >> (log2n - 1), r = t & (((size_t)1 << (log2n - 1)) - 1);
    const size_t j = r & (half - 1), blk = r >> s;
    const size_t i0 = (poly << log2n) + (blk << (s + 1)) + j;
    const u64 w = wtab[j << (log2n - 1 - s)];
    const u64 u = x[i0], v = ops.mul(x[i0 + half], w);
    u64 a = ops.add(u, v), b = ops.sub(u, v);
    if (scale != 1) { a = ops.mul(a, scale); b = ops.mul(b, scale); }
    x[i0] = a;
    x[i0 + half] = b;
  }
}

}  // namespace ronk
