// mont64.h -- arithmetic modulo an arbitrary odd prime p < 2^64 (Montgomery, R = 2^64).
//
// This is the device-side counterpart of ronkathon's generic `PrimeField<const P: usize>`
// (src/algebra/field/prime/mod.rs:39-52, arithmetic.rs:3-71): the small fields the
// reference's own tests use (F_101, F_17, F_127) run through these kernels so that the
// reference's golden vectors are checked on the GPU path itself.  Not the hot path --
// the 64-bit Goldilocks field (gl64.h) is.  Interface values are canonical residues;
// Montgomery form is internal to a kernel.
#pragma once
#include <stdint.h>

#ifndef RONK_HD
#if defined(__HIPCC__)
#define RONK_HD __host__ __device__ __forceinline__
#else
#define RONK_HD inline
#endif
#endif

namespace mont64 {

typedef uint64_t u64;
typedef uint32_t u32;

struct Field {
  u64 p;     // odd modulus
  u64 pinv;  // -p^{-1} mod 2^64
  u64 r2;    // 2^128 mod p
  u64 one;   // 2^64 mod p (Montgomery form of 1)
};

RONK_HD void mul64(u64 a, u64 b, u64& lo, u64& hi) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 p00 = (u64)a0 * b0;
  u64 p01 = (u64)a0 * b1 + (p00 >> 32);
  u64 p10 = (u64)a1 * b0 + (u32)p01;
  hi = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
  lo = (p10 << 32) | (u32)p00;
}

// prime/arithmetic.rs:3-7 without the `%`: a, b < p.  p may exceed 2^63: the sum carries into bit 64.
// Device form (6 VALU + 2 SALU on gfx950): s = a + b with carry C, d = s - p with borrow B on 32-bit limbs; the sum is >= p
// exactly when C or not B, and the two candidates are selected by that one mask (no 64-bit compare: those cost 1.6 issue
// slots each, tools/rate_probe.hip).
RONK_HD u64 add(const Field& f, u64 a, u64 b) {
#if defined(__clang__)
  u32 c1, c2, b1, b2;
  const u32 sl = __builtin_addc((u32)a, (u32)b, 0u, &c1);
  const u32 sh = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c1, &c2);
  const u32 dl = __builtin_subc(sl, (u32)f.p, 0u, &b1);
  const u32 dh = __builtin_subc(sh, (u32)(f.p >> 32), b1, &b2);
  const bool take = c2 | !b2;
  return take ? (((u64)dh << 32) | dl) : (((u64)sh << 32) | sl);
#else
  u64 s = a + b;
  return (s < a || s >= f.p) ? s - f.p : s;
#endif
}
// prime/arithmetic.rs:19-28.  Valid for ANY a < 2^64 and b < p with a - b > -p (what redc needs).  Device form: borrow chain,
// then + (B ? p : 0) selected by the borrow itself (6 VALU).
RONK_HD u64 sub(const Field& f, u64 a, u64 b) {
#if defined(__clang__)
  u32 b1, b2, c1, c2;
  const u32 lo = __builtin_subc((u32)a, (u32)b, 0u, &b1);
  const u32 hi = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), b1, &b2);
  const u32 pl = b2 ? (u32)f.p : 0u, ph = b2 ? (u32)(f.p >> 32) : 0u;
  const u32 rl = __builtin_addc(lo, pl, 0u, &c1);
  const u32 rh = __builtin_addc(hi, ph, c1, &c2);
  return ((u64)rh << 32) | rl;
#else
  u64 d = a - b;
  return (a < b) ? d + f.p : d;
#endif
}
RONK_HD u64 neg(const Field& f, u64 a) { return a ? f.p - a : 0; }

// REDC(hi:lo) = (hi:lo) * 2^-64 mod p, for hi < p.  With the POSITIVE inverse q = p^-1 mod 2^64 and m = lo * q mod 2^64 the
// product m * p has exactly the low half `lo`, so hi:lo - m*p = (hi - hi64(m p)) * 2^64 with NO carry to track: the result is
// hi - hi64(m p), plus p when that is negative (both terms are below p).  26 VALU per Montgomery product on gfx950 (4 + 4
// v_mad_u64_u32 / v_mul_hi for the two wide products, 1 mad + 2 v_mul_lo for m, the borrow-chain subtraction) against 36
// for the textbook form with -p^-1, the carry of the low halves and a compare against p.
RONK_HD u64 redc(const Field& f, u64 lo, u64 hi) {
  const u64 q = (u64)0 - f.pinv;                 // + p^-1 (wave-uniform: scalar ALU)
  const u32 t0 = (u32)lo, t1 = (u32)(lo >> 32), q0 = (u32)q, q1 = (u32)(q >> 32);
  const u64 ml = (u64)t0 * q0;
  const u32 m0 = (u32)ml, m1 = (u32)(ml >> 32) + t0 * q1 + t1 * q0;   // m = lo * q mod 2^64
  const u32 p0 = (u32)f.p, p1 = (u32)(f.p >> 32);
  const u64 c00 = (u64)m0 * p0;                  // only its high word is needed (v_mul_hi_u32)
  const u64 c01 = (u64)m0 * p1 + (c00 >> 32);
  const u64 c10 = (u64)m1 * p0 + (u32)c01;
  const u64 mhi = (u64)m1 * p1 + (c01 >> 32) + (c10 >> 32);
  return sub(f, hi, mhi);
}
// Montgomery product: a*b*2^-64 mod p.  a: ANY 64-bit value, b < p (then hi64(a b) < p); result canonical.
RONK_HD u64 mmul(const Field& f, u64 a, u64 b) {
  u64 lo, hi;
  mul64(a, b, lo, hi);
  return redc(f, lo, hi);
}
RONK_HD u64 to_mont(const Field& f, u64 a) { return mmul(f, a, f.r2); }
RONK_HD u64 from_mont(const Field& f, u64 a) { return redc(f, a, 0); }

// prime/arithmetic.rs:34-38 on canonical values: (a*b) mod p
RONK_HD u64 mul(const Field& f, u64 a, u64 b) { return mmul(f, to_mont(f, a), b); }

// prime/mod.rs:74-84 on canonical values
RONK_HD u64 pow(const Field& f, u64 a, u64 e) {
  u64 r = f.one, am = to_mont(f, a);
  while (e) {
    if (e & 1) r = mmul(f, r, am);
    am = mmul(f, am, am);
    e >>= 1;
  }
  return from_mont(f, r);
}

// Host-side constructor (plan creation).  p odd.
inline Field make_field(u64 p) {
  Field f;
  f.p = p;
  u64 inv = 1;  // Newton: inv = p^{-1} mod 2^64
  for (int i = 0; i < 7; i++) inv *= 2 - p * inv;
  f.pinv = (u64)0 - inv;
  unsigned __int128 r = ((unsigned __int128)1 << 64) % p;
  f.one = (u64)r;
  f.r2 = (u64)((r * r) % p);
  return f;
}

}  // namespace mont64
