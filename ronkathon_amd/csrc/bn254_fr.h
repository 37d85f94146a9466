// bn254_fr.h -- the SCALAR field of BN254 (alt_bn128), r = 36x^4 + 36x^3 + 18x^2 + 6x + 1 (254 bits, the order of G1): what
// kzg::open's polynomial lives over when the commitment group is a production curve.  The reference's open
// (src/kzg/setup.rs:63-78) is `poly.div([-eval_point, ONE])` over its scalar field followed by `commit(quotient, g1_srs)`;
// the MSM half over BN254 G1 is msm_kernels.h / bn254.h, this header is the field of the division half (fr_scan_kernels.h).
//
// Elements: 8 x 32-bit limbs, little endian, CANONICAL, STANDARD form in memory (the same bytes as the 4 x 64-bit scalars
// ronk_msm_bn254 takes).  Products are Montgomery products (R = 2^256, CIOS on 32-bit limbs) whose SECOND operand is in
// Montgomery form: fr_mul(x, wR) = x * w, standard -> standard, so the data never change form (the scan multiplies by powers
// of the evaluation point only, and those come from the host in Montgomery form) -- the same device as the NTT's table-form
// twiddles (field_policy.h).  Plain C++: host (table construction, CPU tests) and device.  Not tuned (the division is < 5 %
// of an opening; the MSM's field, bn254.h, is the tuned one).
#pragma once
#include <stdint.h>

#include "bn254_consts.h"

#ifndef RONK_HD
#if defined(__HIPCC__)
#define RONK_HD __host__ __device__ __forceinline__
#else
#define RONK_HD inline
#endif
#endif

namespace bn254 {

typedef uint32_t u32;
typedef uint64_t u64;

struct Fr { u32 l[8]; };

RONK_HD u32 fr_mod(int i) { constexpr u32 c[8] = BN254_FR_LIMBS; return c[i]; }
RONK_HD Fr fr_zero() { Fr r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }
RONK_HD Fr fr_const_one_mont() { constexpr u32 c[8] = BN254_FR_ONE_LIMBS; Fr r; for (int i = 0; i < 8; i++) r.l[i] = c[i]; return r; }
RONK_HD Fr fr_const_r2() { constexpr u32 c[8] = BN254_FR_R2_LIMBS; Fr r; for (int i = 0; i < 8; i++) r.l[i] = c[i]; return r; }

// a - r if a >= r (a < 2r given as 8 limbs + carry bit `top`)
RONK_HD Fr fr_cond_sub(const Fr& a, u32 top) {
  Fr d;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)a.l[i] - fr_mod(i) - borrow;
    d.l[i] = (u32)t;
    borrow = (t >> 32) & 1;
  }
  const bool take = top != 0 || borrow == 0;   // a >= r
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = take ? d.l[i] : a.l[i];
  return r;
}
// any 256-bit integer -> canonical (r > 2^253: at most four subtractions)
RONK_HD Fr fr_canon(Fr a) {
  for (int k = 0; k < 5; k++) a = fr_cond_sub(a, 0);
  return a;
}
RONK_HD Fr fr_add(const Fr& a, const Fr& b) {   // a, b < r
  Fr s;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)a.l[i] + b.l[i] + c;
    s.l[i] = (u32)t;
    c = t >> 32;
  }
  return fr_cond_sub(s, (u32)c);
}
RONK_HD Fr fr_sub(const Fr& a, const Fr& b) {   // a, b < r
  Fr d;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)a.l[i] - b.l[i] - borrow;
    d.l[i] = (u32)t;
    borrow = (t >> 32) & 1;
  }
  const u32 m = (u32)0 - (u32)borrow;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)d.l[i] + (fr_mod(i) & m) + c;
    d.l[i] = (u32)t;
    c = t >> 32;
  }
  return d;
}
RONK_HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
// Montgomery product a * b / 2^256 mod r.  a: ANY 256-bit integer, b < r (then a*b < 2^256 r and the result, < 2r before the
// final subtraction, comes out canonical).
RONK_HD Fr fr_mul(const Fr& a, const Fr& b) {
  u32 t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u64 s = (u64)t[j] + (u64)a.l[j] * b.l[i] + c;
      t[j] = (u32)s;
      c = s >> 32;
    }
    u64 s = (u64)t[8] + c;
    t[8] = (u32)s;
    t[9] = (u32)(s >> 32);
    const u32 m = t[0] * BN254_FR_N0INV;
    c = ((u64)t[0] + (u64)m * fr_mod(0)) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      s = (u64)t[j] + (u64)m * fr_mod(j) + c;
      t[j - 1] = (u32)s;
      c = s >> 32;
    }
    s = (u64)t[8] + c;
    t[7] = (u32)s;
    t[8] = t[9] + (u32)(s >> 32);
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = t[i];
  return fr_cond_sub(r, t[8]);
}
RONK_HD Fr fr_to_mont(const Fr& a) { return fr_mul(a, fr_const_r2()); }   // a * 2^256 mod r
// 4 x u64 little endian (the C ABI's layout) <-> limbs
RONK_HD Fr fr_load(const u64* w) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 4; i++) { r.l[2 * i] = (u32)w[i]; r.l[2 * i + 1] = (u32)(w[i] >> 32); }
  return r;
}
RONK_HD void fr_store(u64* w, const Fr& a) {
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = ((u64)a.l[2 * i + 1] << 32) | a.l[2 * i];
}

}  // namespace bn254
