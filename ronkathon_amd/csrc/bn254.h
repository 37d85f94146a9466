// bn254.h -- BN254 (alt_bn128) base-field and G1 arithmetic for the bucket-method MSM behind kzg::commit on a real curve
// (the reference's commit is the fold `g1_srs.zip(coeffs).map(|(g, c)| g * c).sum()`,
// src/kzg/setup.rs:48-60, over its toy curve -- same sum, production-size group).
//
//   p = 36x^4 + 36x^3 + 24x^2 + 6x + 1,  x = 4965661367192848881  (254 bits),   E: y^2 = x^3 + 3,   G = (1, 2)
//
// Field elements: 8 x 32-bit limbs, little endian, MONTGOMERY form (R = 2^256), WEAKLY REDUCED -- any representative in
// [0, 2p) -- inside kernels (4p < R, so a Montgomery product of two such values is again below 2p without the final
// conditional subtraction, and sums of two fit 256 bits; zero tests accept 0 and p; the conversion back canonicalises); the C ABI speaks the
// standard form as 4 x 64-bit little-endian limbs (the same bytes).  gfx950 has no 64x64 multiplier: a limb product is one
// v_mad_u64_u32 (32 x 32 + 64 -> 64), which is exactly the CIOS inner step t + a*b + carry.
// Points: affine (x, y) on input ((0, 0) = the point at infinity, which is not on the curve), buckets and partial sums
// in XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ = 0 <=> infinity): mixed addition costs 8M + 2S, no
// inversion anywhere but the final conversion.
// Plain C++: the same code runs on the device and on the host (tail of the host-pointer entry point).
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

struct Fp { u32 l[8]; };

// constants live in functions (constexpr arrays in device code without relocatable globals)
RONK_HD u32 P_limb(int i) { constexpr u32 c[8] = BN254_P_LIMBS; return c[i]; }
RONK_HD u32 P2_limb(int i) { constexpr u32 c[8] = BN254_2P_LIMBS; return c[i]; }
RONK_HD Fp fp_const_one() { constexpr u32 c[8] = BN254_R_LIMBS; Fp r; for (int i = 0; i < 8; i++) r.l[i] = c[i]; return r; }
RONK_HD Fp fp_const_r2() { constexpr u32 c[8] = BN254_R2_LIMBS; Fp r; for (int i = 0; i < 8; i++) r.l[i] = c[i]; return r; }
RONK_HD Fp fp_const_b() { constexpr u32 c[8] = BN254_B_MONT_LIMBS; Fp r; for (int i = 0; i < 8; i++) r.l[i] = c[i]; return r; }
RONK_HD Fp fp_zero() { Fp r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }

// the integer zero (the encoding of the point at infinity's coordinates)
RONK_HD bool fp_is_zero_exact(const Fp& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.l[i];
  return o == 0;
}
// a == 0 (mod p) for a weakly reduced a: the integers 0 and p
RONK_HD bool fp_is_zero(const Fp& a) {
  u32 o = 0, q = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { o |= a.l[i]; q |= a.l[i] ^ P_limb(i); }
  return o == 0 || q == 0;
}
// a >= p (as integers)
RONK_HD bool fp_geq_p(const Fp& a) {
  bool ge = true;   // equal so far -> a == p counts as >=
#pragma unroll
  for (int i = 0; i < 8; i++) {   // from the least significant limb: the most significant difference decides last
    const u32 pi = P_limb(i);
    if (a.l[i] != pi) ge = a.l[i] > pi;
  }
  return ge;
}
// r = a - M if a >= M, M = p (TWO == false) or 2p
template <bool TWO>
RONK_HD Fp fp_cond_sub(const Fp& a) {
  Fp d;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)a.l[i] - (TWO ? P2_limb(i) : P_limb(i)) - borrow;
    d.l[i] = (u32)t;
    borrow = (t >> 32) & 1;
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
  return r;
}
// weakly reduced -> canonical
RONK_HD Fp fp_canon(const Fp& a) { return fp_cond_sub<false>(fp_cond_sub<false>(a)); }
RONK_HD Fp fp_add(const Fp& a, const Fp& b) {
  Fp s;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)a.l[i] + b.l[i] + c;
    s.l[i] = (u32)t;
    c = t >> 32;
  }
  return fp_cond_sub<true>(s);   // a + b < 4p < 2^256: no carry out; back below 2p
}
RONK_HD Fp fp_sub(const Fp& a, const Fp& b) {
  Fp d;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)a.l[i] - b.l[i] - borrow;
    d.l[i] = (u32)t;
    borrow = (t >> 32) & 1;
  }
  // + 2p on borrow: the difference of two values below 2p lies in (-2p, 2p)
  const u32 m = (u32)0 - (u32)borrow;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)d.l[i] + (P2_limb(i) & m) + c;
    d.l[i] = (u32)t;
    c = t >> 32;
  }
  return d;
}
RONK_HD bool fp_eq(const Fp& a, const Fp& b) { return fp_is_zero(fp_sub(a, b)); }
RONK_HD Fp fp_neg(const Fp& a) { return fp_is_zero_exact(a) ? a : fp_sub(fp_zero(), a); }   // 0 - a + 2p in (0, 2p); the integer 0 stays 0
RONK_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

// ---- Montgomery product a*b/R mod p.
// Product scanning (FIPS): column i of the 16-limb sum a*b + m*p is accumulated in ONE 96-bit register triple
// (lo:64, hi:32), then its low limb is dropped.  On the device every product is
//     v_mad_u64_u32  lo, carry, x, y, lo        (32 x 32 + 64 -> 64, carry-out to an SGPR pair)
//     v_addc_co_u32  hi, _, 0, hi, carry
// i.e. two VALU instructions per limb product and no register moves: the row-wise (CIOS) form, whose inner step is
// t + a*b + c with two 32-bit addends, compiles to mad + 64-bit add + THREE v_mov for the zero-extended operand pairs
// (5 534 v_mov of 11 375 VALU instructions in the first accumulate kernel).  Products are issued in pairs so that the
// two wait states gfx950 wants between a VALU writing an SGPR and a VALU reading it are real work (mad, mad, s_nop 0, addc, addc).
// The compiler never uses the mad's carry-out, hence the inline asm; the host build (tests, the MSM's host tail) runs the
// same algorithm on unsigned __int128.
struct Acc96 { u64 lo; u32 hi; };
RONK_HD void acc_mad(Acc96& t, u32 x, u32 y) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 c;
  asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, %1, %2"
      : "+v"(t.lo), "+v"(t.hi), "=&s"(c)
      : "v"(x), "v"(y)
      : "vcc");
#else
  const unsigned __int128 s = (((unsigned __int128)t.hi << 64) | t.lo) + (unsigned __int128)((u64)x * y);
  t.lo = (u64)s; t.hi = (u32)(s >> 64);
#endif
}
// t += x1*y1 + x2*y2  (y2 may sit in an SGPR: the modulus limbs are constants)
RONK_HD void acc_mad2(Acc96& t, u32 x1, u32 y1, u32 x2, u32 y2) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 c1, c2;
  asm("v_mad_u64_u32 %0, %2, %4, %5, %0\n\t"
      "v_mad_u64_u32 %0, %3, %6, %7, %0\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, %1, %3"
      : "+v"(t.lo), "+v"(t.hi), "=&s"(c1), "=&s"(c2)
      : "v"(x1), "v"(y1), "v"(x2), "s"(y2)
      : "vcc");
#else
  acc_mad(t, x1, y1);
  acc_mad(t, x2, y2);
#endif
}
RONK_HD void acc_shift(Acc96& t) { t.lo = (t.lo >> 32) | ((u64)t.hi << 32); t.hi = 0; }

RONK_HD Fp fp_mul(const Fp& a, const Fp& b) {
  u32 m[8];
  Fp r;
  Acc96 t;
  t.lo = 0; t.hi = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int j = 0; j < i; j++) acc_mad2(t, a.l[j], b.l[i - j], m[j], P_limb(i - j));
    acc_mad(t, a.l[i], b.l[0]);
    m[i] = (u32)t.lo * BN254_N0INV;
    acc_mad(t, m[i], P_limb(0));          // the low limb becomes zero
    acc_shift(t);
  }
#pragma unroll
  for (int i = 8; i < 16; i++) {
#pragma unroll
    for (int j = i - 7; j < 8; j++) acc_mad2(t, a.l[j], b.l[i - j], m[j], P_limb(i - j));
    r.l[i - 8] = (u32)t.lo;
    acc_shift(t);
  }
  return r;   // < 2p for inputs < 2p (4p < R): no final subtraction; the 17th limb (u32)t.lo is 0
}
RONK_HD Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }

RONK_HD Fp fp_to_mont(const Fp& a) { return fp_mul(a, fp_const_r2()); }
RONK_HD Fp fp_from_mont(const Fp& a) {   // -> standard form, canonical
  Fp one = fp_zero();
  one.l[0] = 1;
  return fp_canon(fp_mul(a, one));
}
// a^(p-2) (a != 0), Montgomery in and out
RONK_HD Fp fp_inv(const Fp& a) {
  constexpr u32 e[8] = BN254_PM2_LIMBS;
  Fp r = fp_const_one();
  for (int i = 7; i >= 0; i--) {
    for (int b = 31; b >= 0; b--) {
      r = fp_sqr(r);
      if ((e[i] >> b) & 1) r = fp_mul(r, a);
    }
  }
  return r;
}
// 4 x u64 little-endian (the C ABI's layout) <-> limbs
RONK_HD Fp fp_load(const u64* w) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 4; i++) { r.l[2 * i] = (u32)w[i]; r.l[2 * i + 1] = (u32)(w[i] >> 32); }
  return r;
}
RONK_HD void fp_store(u64* w, const Fp& a) {
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = (u64)a.l[2 * i] | ((u64)a.l[2 * i + 1] << 32);
}

// ---- G1 ---------------------------------------------------------------------------------------------------------------
struct Affine { Fp x, y; };            // Montgomery form; (0, 0) = infinity
struct Xyzz { Fp X, Y, ZZ, ZZZ; };     // ZZ == 0 <=> infinity

RONK_HD bool affine_is_inf(const Affine& p) { return fp_is_zero_exact(p.x) && fp_is_zero_exact(p.y); }
RONK_HD Xyzz xyzz_inf() { Xyzz r; r.X = fp_zero(); r.Y = fp_zero(); r.ZZ = fp_zero(); r.ZZZ = fp_zero(); return r; }
RONK_HD bool xyzz_is_inf(const Xyzz& p) { return fp_is_zero(p.ZZ); }
// y^2 == x^3 + 3
RONK_HD bool affine_on_curve(const Affine& p) {
  const Fp x2 = fp_sqr(p.x);
  return fp_eq(fp_sqr(p.y), fp_add(fp_mul(x2, p.x), fp_const_b()));
}
// 2 * (affine P), P not infinity (y != 0 on this curve: no points of order 2)   -- mdbl-2008-s-1 with a = 0
RONK_HD Xyzz xyzz_dbl_affine(const Affine& p) {
  const Fp U = fp_dbl(p.y), V = fp_sqr(U), W = fp_mul(U, V), S = fp_mul(p.x, V);
  const Fp x2 = fp_sqr(p.x), M = fp_add(fp_dbl(x2), x2);
  Xyzz r;
  r.X = fp_sub(fp_sqr(M), fp_dbl(S));
  r.Y = fp_sub(fp_mul(M, fp_sub(S, r.X)), fp_mul(W, p.y));
  r.ZZ = V;
  r.ZZZ = W;
  return r;
}
// 2 * P   -- dbl-2008-s-1 with a = 0
RONK_HD Xyzz xyzz_dbl(const Xyzz& p) {
  if (xyzz_is_inf(p)) return p;
  const Fp U = fp_dbl(p.Y), V = fp_sqr(U), W = fp_mul(U, V), S = fp_mul(p.X, V);
  const Fp x2 = fp_sqr(p.X), M = fp_add(fp_dbl(x2), x2);
  Xyzz r;
  r.X = fp_sub(fp_sqr(M), fp_dbl(S));
  r.Y = fp_sub(fp_mul(M, fp_sub(S, r.X)), fp_mul(W, p.Y));
  r.ZZ = fp_mul(V, p.ZZ);
  r.ZZZ = fp_mul(W, p.ZZZ);
  return r;
}
// acc += (neg ? -q : q), q affine   -- madd-2008-s
RONK_HD void xyzz_madd(Xyzz& acc, const Affine& q, bool neg) {
  if (affine_is_inf(q)) return;
  const Fp qy = neg ? fp_neg(q.y) : q.y;
  if (xyzz_is_inf(acc)) {
    acc.X = q.x; acc.Y = qy; acc.ZZ = fp_const_one(); acc.ZZZ = fp_const_one();
    return;
  }
  const Fp U2 = fp_mul(q.x, acc.ZZ), S2 = fp_mul(qy, acc.ZZZ);
  const Fp P = fp_sub(U2, acc.X), R = fp_sub(S2, acc.Y);
  if (fp_is_zero(P)) {
    if (fp_is_zero(R)) { Affine t; t.x = q.x; t.y = qy; acc = xyzz_dbl_affine(t); }
    else acc = xyzz_inf();
    return;
  }
  const Fp PP = fp_sqr(P), PPP = fp_mul(P, PP), Q = fp_mul(acc.X, PP);
  const Fp X3 = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Q));
  acc.Y = fp_sub(fp_mul(R, fp_sub(Q, X3)), fp_mul(acc.Y, PPP));
  acc.X = X3;
  acc.ZZ = fp_mul(acc.ZZ, PP);
  acc.ZZZ = fp_mul(acc.ZZZ, PPP);
}
// a + b   -- add-2008-s
RONK_HD Xyzz xyzz_add(const Xyzz& a, const Xyzz& b) {
  if (xyzz_is_inf(a)) return b;
  if (xyzz_is_inf(b)) return a;
  const Fp U1 = fp_mul(a.X, b.ZZ), U2 = fp_mul(b.X, a.ZZ), S1 = fp_mul(a.Y, b.ZZZ), S2 = fp_mul(b.Y, a.ZZZ);
  const Fp P = fp_sub(U2, U1), R = fp_sub(S2, S1);
  if (fp_is_zero(P)) return fp_is_zero(R) ? xyzz_dbl(a) : xyzz_inf();
  const Fp PP = fp_sqr(P), PPP = fp_mul(P, PP), Q = fp_mul(U1, PP);
  Xyzz r;
  r.X = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Q));
  r.Y = fp_sub(fp_mul(R, fp_sub(Q, r.X)), fp_mul(S1, PPP));
  r.ZZ = fp_mul(fp_mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = fp_mul(fp_mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}
// -> affine standard form as 8 x u64 (x limbs, y limbs); infinity = all zero
RONK_HD void xyzz_store_affine(const Xyzz& p, u64* out) {
  if (xyzz_is_inf(p)) { for (int i = 0; i < 8; i++) out[i] = 0; return; }
  // 1/ZZZ, then 1/ZZ = ZZ^2 / ZZZ^2 ... simply: one inversion of ZZ*ZZZ
  const Fp zi = fp_inv(fp_mul(p.ZZ, p.ZZZ));
  const Fp zz_inv = fp_mul(zi, p.ZZZ), zzz_inv = fp_mul(zi, p.ZZ);
  fp_store(out, fp_from_mont(fp_mul(p.X, zz_inv)));
  fp_store(out + 4, fp_from_mont(fp_mul(p.Y, zzz_inv)));
}

}  // namespace bn254
