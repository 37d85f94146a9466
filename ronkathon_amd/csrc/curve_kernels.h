// curve_kernels.h -- the multi-scalar multiplication behind kzg::commit (src/kzg/setup.rs:45-60), SURVEY.md 8(f) N4.
//
//   commit = sum_i srs[i] * coeff[i]      (AffinePoint Mul<ScalarField>, curve/mod.rs:152-166; Sum = reduce, :213-217)
//
// over the reference's curve family: y^2 = x^3 + a x + b on the quadratic extension F_p[u]/(u^2 - nr) of a SMALL
// prime field (p < 2^32: PlutoExtendedCurve is p = 101, X^2 + 2, a = 0, b = 3; src/curve/pluto_curve.rs:39-51,
// src/algebra/field/extension/gf_101_2.rs:12-18).  Affine points with an Infinity variant, 5 words each
// (x0 x1 y0 y1 inf).  One lane per term: on-curve check (AffinePoint::new's assert, curve/mod.rs:77-81),
// [k]P by double-and-add through the reference's Add (same group element as its k-1 repeated additions), then a
// workgroup tree of point additions and a second launch over the partial sums.  This row is about closing the KZG
// caller on the device with the reference's own vectors; it is not a production MSM (that is a bucket method over
// a 254/381-bit field -- a different kernel family, out of scope for this path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ronk {

struct CurveCtx { uint64_t p, nr, a, b; };
struct Fp2 { uint64_t c0, c1; };
struct CPoint { Fp2 x, y; uint64_t inf; };

enum { CURVE_ERR_INVERSE = 1, CURVE_ERR_NOT_ON_CURVE = 2 };

__device__ __forceinline__ uint64_t cm_mul(const CurveCtx& c, uint64_t a, uint64_t b) { return (a * b) % c.p; }  // a, b < p < 2^32
__device__ __forceinline__ uint64_t cm_add(const CurveCtx& c, uint64_t a, uint64_t b) { uint64_t s = a + b; return s >= c.p ? s - c.p : s; }
__device__ __forceinline__ uint64_t cm_sub(const CurveCtx& c, uint64_t a, uint64_t b) { return a >= b ? a - b : a + c.p - b; }
__device__ __forceinline__ uint64_t cm_pow(const CurveCtx& c, uint64_t a, uint64_t e) {
  uint64_t r = 1 % c.p;
  while (e) { if (e & 1) r = cm_mul(c, r, a); a = cm_mul(c, a, a); e >>= 1; }
  return r;
}
__device__ __forceinline__ Fp2 f2_add(const CurveCtx& c, Fp2 a, Fp2 b) { return Fp2{cm_add(c, a.c0, b.c0), cm_add(c, a.c1, b.c1)}; }
__device__ __forceinline__ Fp2 f2_sub(const CurveCtx& c, Fp2 a, Fp2 b) { return Fp2{cm_sub(c, a.c0, b.c0), cm_sub(c, a.c1, b.c1)}; }
__device__ __forceinline__ Fp2 f2_neg(const CurveCtx& c, Fp2 a) { return Fp2{cm_sub(c, 0, a.c0), cm_sub(c, 0, a.c1)}; }
__device__ __forceinline__ bool f2_eq(Fp2 a, Fp2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
// gf_101_2.rs:83-97: product reduced by u^2 = nr
__device__ __forceinline__ Fp2 f2_mul(const CurveCtx& c, Fp2 a, Fp2 b) {
  return Fp2{cm_add(c, cm_mul(c, a.c0, b.c0), cm_mul(c, c.nr, cm_mul(c, a.c1, b.c1))),
             cm_add(c, cm_mul(c, a.c0, b.c1), cm_mul(c, a.c1, b.c0))};
}
// gf_101_2.rs:34-47: conjugate over the norm a0^2 - nr a1^2; ZERO has no inverse (Div's expect("invalid inverse"))
__device__ __forceinline__ Fp2 f2_inv(const CurveCtx& c, Fp2 a, int* err) {
  if (a.c0 == 0 && a.c1 == 0) { *err |= CURVE_ERR_INVERSE; return a; }
  const uint64_t norm = cm_sub(c, cm_mul(c, a.c0, a.c0), cm_mul(c, c.nr, cm_mul(c, a.c1, a.c1)));
  const uint64_t s = cm_pow(c, norm, c.p - 2);
  return Fp2{cm_mul(c, a.c0, s), cm_mul(c, cm_sub(c, 0, a.c1), s)};
}
__device__ __forceinline__ Fp2 f2_small(const CurveCtx& c, uint64_t k) { return Fp2{k % c.p, 0}; }

// curve/mod.rs:129-138
__device__ __forceinline__ bool on_curve(const CurveCtx& c, const CPoint& P) {
  if (P.inf) return true;
  const Fp2 rhs = f2_add(c, f2_add(c, f2_mul(c, f2_mul(c, P.x, P.x), P.x), f2_mul(c, f2_small(c, c.a), P.x)), f2_small(c, c.b));
  return f2_eq(f2_mul(c, P.y, P.y), rhs);
}
__device__ __forceinline__ CPoint c_inf() { return CPoint{Fp2{0, 0}, Fp2{0, 0}, 1}; }

// impl Add (curve/mod.rs:176-211), the cases in the reference's order
__device__ inline CPoint c_add(const CurveCtx& c, const CPoint& P, const CPoint& Q, int* err) {
  if (P.inf) return Q;
  if (Q.inf) return P;
  if (f2_eq(P.x, Q.x) && f2_eq(P.y, f2_neg(c, Q.y))) return c_inf();
  Fp2 lambda;
  if (f2_eq(P.x, Q.x) && f2_eq(P.y, Q.y)) {
    const Fp2 num = f2_add(c, f2_mul(c, f2_mul(c, f2_small(c, 3), P.x), P.x), f2_small(c, c.a));
    lambda = f2_mul(c, num, f2_inv(c, f2_mul(c, f2_small(c, 2), P.y), err));
  } else {
    lambda = f2_mul(c, f2_sub(c, Q.y, P.y), f2_inv(c, f2_sub(c, Q.x, P.x), err));
  }
  CPoint R;
  R.x = f2_sub(c, f2_sub(c, f2_mul(c, lambda, lambda), P.x), Q.x);
  R.y = f2_sub(c, f2_mul(c, lambda, f2_sub(c, P.x, R.x)), P.y);
  R.inf = 0;
  return R;
}

__device__ __forceinline__ CPoint c_load(const uint64_t* w) { return CPoint{Fp2{w[0], w[1]}, Fp2{w[2], w[3]}, w[4] ? 1ull : 0ull}; }
__device__ __forceinline__ void c_store(uint64_t* w, const CPoint& P) {
  w[0] = P.inf ? 0 : P.x.c0; w[1] = P.inf ? 0 : P.x.c1; w[2] = P.inf ? 0 : P.y.c0; w[3] = P.inf ? 0 : P.y.c1; w[4] = P.inf;
}

__device__ inline CPoint block_point_sum(const CurveCtx& c, CPoint v, CPoint* sh, int* err) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) sh[tid] = c_add(c, sh[tid], sh[tid + s], err);
    __syncthreads();
  }
  return sh[0];
}

// partial[blk] = sum over the block's terms of points[i] * scalars[i]
__global__ void __launch_bounds__(256) msm_terms_kernel(CurveCtx c, const uint64_t* __restrict__ points,
                                                        const uint64_t* __restrict__ scalars, size_t n,
                                                        uint64_t* __restrict__ partial, int* flag) {
  __shared__ CPoint sh[256];
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  int err = 0;
  CPoint acc = c_inf();
  if (i < n) {
    const CPoint P = c_load(points + 5 * i);
    if (!on_curve(c, P)) err |= CURVE_ERR_NOT_ON_CURVE;
    const uint64_t k = scalars[i];
    if (!err && k) {
      const int top = 63 - __clzll((long long)k);
      for (int bit = top; bit >= 0; bit--) {
        acc = c_add(c, acc, acc, &err);
        if ((k >> bit) & 1) acc = c_add(c, acc, P, &err);
      }
    }
  }
  const CPoint tot = block_point_sum(c, acc, sh, &err);
  if (err) atomicOr(flag, err);
  if (threadIdx.x == 0) c_store(partial + 5 * (size_t)blockIdx.x, tot);
}

// out = sum of `count` points (one workgroup)
__global__ void __launch_bounds__(256) msm_reduce_kernel(CurveCtx c, const uint64_t* __restrict__ pts, size_t count,
                                                         uint64_t* __restrict__ out, int* flag) {
  __shared__ CPoint sh[256];
  int err = 0;
  CPoint acc = c_inf();
  for (size_t i = threadIdx.x; i < count; i += 256) acc = c_add(c, acc, c_load(pts + 5 * i), &err);
  const CPoint tot = block_point_sum(c, acc, sh, &err);
  if (err) atomicOr(flag, err);
  if (threadIdx.x == 0) c_store(out, tot);
}

}  // namespace ronk
