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

// prime/arithmetic.rs:3-7 without the `%`: a, b < p
RONK_HD u64 add(const Field& f, u64 a, u64 b) {
  u64 s = a + b;
  return (s < a || s >= f.p) ? s - f.p : s;
}
// prime/arithmetic.rs:19-28
RONK_HD u64 sub(const Field& f, u64 a, u64 b) {
  u64 d = a - b;
  return (a < b) ? d + f.p : d;
}
RONK_HD u64 neg(const Field& f, u64 a) { return a ? f.p - a : 0; }

// REDC(hi:lo) = (hi:lo) * 2^-64 mod p, for hi < p
RONK_HD u64 redc(const Field& f, u64 lo, u64 hi) {
  u64 m = lo * f.pinv;
  u64 mlo, mhi;
  mul64(m, f.p, mlo, mhi);
  // lo + mlo == 0 mod 2^64; carry out is 1 unless lo == 0
  u64 carry = lo != 0;
  u64 t = hi + mhi;
  bool over = t < hi;
  u64 t2 = t + carry;
  over |= t2 < t;
  return (over || t2 >= f.p) ? t2 - f.p : t2;
}
// Montgomery product: a*b*2^-64 mod p
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
