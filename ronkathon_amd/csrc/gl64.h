// gl64.h -- Goldilocks field arithmetic (p = 2^64 - 2^32 + 1) for CDNA4 device code.
//
// This is the 64-bit `FiniteField` implementor that sits behind ronkathon's `Field` /
// `FiniteField` trait surface (src/algebra/field/mod.rs:17-76): canonical residues in
// [0, p), the same observable results as PrimeField's `%`-based operators
// (src/algebra/field/prime/arithmetic.rs:3-71) but without a divide: gfx950 has no
// 64x64->128 multiply and no integer divide, so products are built from four
// v_mad_u64_u32 and reduced with 2^64 = 2^32 - 1, 2^96 = -1 (mod p).
//
// All functions take and return CANONICAL values unless the name says otherwise.
// Plain C++ on uint32/uint64 only, so the same header also builds for the host-side
// kernel emulator under tests/emu (test infrastructure; never part of the product .so).
#pragma once
#include <stdint.h>

#ifndef RONK_HD
#if defined(__HIPCC__)
#define RONK_HD __host__ __device__ __forceinline__
#else
#define RONK_HD inline
#endif
#endif

namespace gl64 {

typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u64 P = 0xFFFFFFFF00000001ull;
static constexpr u64 EPS = 0xFFFFFFFFull;  // 2^64 mod p = 2^32 - 1
static constexpr u64 GENERATOR = 7;        // explicit PRIMITIVE_ELEMENT

// x < 2^64 arbitrary -> canonical
RONK_HD u64 canon(u64 x) { return x >= P ? x - P : x; }

// prime/arithmetic.rs:3-7
RONK_HD u64 add(u64 a, u64 b) {
  u64 s = a + b;
  // a, b < p: a wrapped sum (s < a) or s >= p both mean "subtract p", and s - p == s + EPS mod 2^64
  return (s < a || s >= P) ? s + EPS : s;
}

// prime/arithmetic.rs:19-28 (borrow -> + ORDER)
RONK_HD u64 sub(u64 a, u64 b) {
  u64 d = a - b;
  return (a < b) ? d - EPS : d;  // d + p == d - EPS mod 2^64
}

// prime/arithmetic.rs:61-65
RONK_HD u64 neg(u64 a) { return a ? P - a : 0; }

// 128-bit value hi:lo -> canonical residue.  hi = hh*2^32 + hl:
//   x = lo + hl*(2^32-1) - hh   (mod p)
RONK_HD u64 reduce128(u64 lo, u64 hi) {
  u32 hh = (u32)(hi >> 32), hl = (u32)hi;
  u64 t0 = lo - hh;
  if (lo < hh) t0 -= EPS;                  // borrow: + p
  u64 t1 = ((u64)hl << 32) - hl;           // hl * EPS, < p
  u64 r = t0 + t1;
  if (r < t1) r += EPS;                    // carry: 2^64 = EPS; cannot carry twice
  return canon(r);
}

// prime/arithmetic.rs:34-38.  Schoolbook on 32-bit limbs; each line is one v_mad_u64_u32.
RONK_HD u64 mul(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 p00 = (u64)a0 * b0;
  u64 p01 = (u64)a0 * b1 + (p00 >> 32);
  u64 p10 = (u64)a1 * b0 + (u32)p01;
  u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
  u64 lo = (p10 << 32) | (u32)p00;
  return reduce128(lo, p11);
}

RONK_HD u64 sqr(u64 a) { return mul(a, a); }

// prime/mod.rs:74-84 (value of the recursion: a^e, pow(_,0) == 1)
RONK_HD u64 pow(u64 a, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = mul(r, a);
    a = sqr(a);
    e >>= 1;
  }
  return r;
}

// prime/mod.rs:62-72: a^(p-2); caller handles a == 0 (None)
RONK_HD u64 inv(u64 a) { return pow(a, P - 2); }

// x * 2^K for a compile-time 0 <= K < 96.  2 generates the order-192 subgroup and, with
// the reference's convention omega_n = 7^((p-1)/n), omega_64 == 2^39: every twiddle
// inside a <=64-point sub-transform is +-2^K, i.e. shifts instead of multiplies.
// With y = x << (K%32) as limbs (y2,y1,y0) placed K/32 limbs up:
//   q=0: (y1:y0) + y2*EPS      q=1: (y0<<32) + y1*EPS - y2      q=2: y0*EPS - (y2:y1)
template <int K>
RONK_HD u64 mul_2exp(u64 x) {
  static_assert(K >= 0 && K < 96, "shift out of range");
  if (K == 0) return x;
  constexpr int q = K / 32, s = K % 32;
  u32 x0 = (u32)x, x1 = (u32)(x >> 32);
  u32 y0, y1, y2;
  if constexpr (s == 0) {
    y0 = x0; y1 = x1; y2 = 0;
  } else {
    y0 = x0 << s;
    y1 = (x1 << s) | (x0 >> (32 - s));
    y2 = x1 >> (32 - s);
  }
  if (q == 0) {
    u64 n = ((u64)y1 << 32) | y0;          // may be >= p
    u64 t = ((u64)y2 << 32) - y2;          // y2 * EPS < p
    u64 r = n + t;
    if (r < t) r += EPS;
    return canon(r);
  } else if (q == 1) {
    u64 n = (u64)y0 << 32;                 // < p
    u64 t = ((u64)y1 << 32) - y1;          // < p
    return sub(add(n, t), (u64)y2);
  } else {
    u64 t = ((u64)y0 << 32) - y0;          // < p
    u64 m = ((u64)y2 << 32) | y1;          // y2 < 2^31 -> < p
    return sub(t, m);
  }
}

}  // namespace gl64
