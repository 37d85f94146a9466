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
// The forms below are chosen by VALU instruction count on gfx950 (tools/instr_rate.hip,
// tools/isa_loop_count.py): every 64-bit add / compare / select / v_mad_u64_u32 costs the
// same ~1.6 issue slots, so "value + (cond ? EPS : 0)" (5 VALU per add) beats a 64-bit
// select between two candidates (6), and a borrow chain with the +p folded into the limbs (4 VALU) beats
// subtract + compare for sub; a product is 4 mads + one combined final correction (18 VALU).
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

// Experiment switch (tools/ubench only; the product always builds the default): bit 0 = sub() with the +p folded into the
// limbs (4 VALU + s_xor) instead of mask-and-subtract (5 VALU); bit 1 = mad_eps_canon as the 4-instruction asm block
// (v_mad_u64_u32 carry-out); bit 2 = sub32 as the asm block.
#ifndef RONK_GL64_VARIANT
#define RONK_GL64_VARIANT 7
#endif
// bit 3 (TIMING EXPERIMENTS ONLY, results undefined): drop the hazard wait states inside the asm blocks
#if RONK_GL64_VARIANT & 8
#define RONK_ASM_NOP1 ""
#else
#define RONK_ASM_NOP1 "s_nop 1\n\t"
#endif

namespace gl64 {

typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u64 P = 0xFFFFFFFF00000001ull;
static constexpr u64 EPS = 0xFFFFFFFFull;  // 2^64 mod p = 2^32 - 1
static constexpr u64 GENERATOR = 7;        // explicit PRIMITIVE_ELEMENT

// x < 2^64 arbitrary -> canonical  (x - p == x + EPS mod 2^64)
RONK_HD u64 canon(u64 x) { return x + ((x >= P) ? EPS : 0); }

// prime/arithmetic.rs:3-7.  a, b < p: a wrapped sum (s < a) or s >= p both mean "subtract p"
RONK_HD u64 add(u64 a, u64 b) {
  u64 s = a + b;
#ifdef RONK_GL64_ALL_LAZY   // tools/census.hip only: instruction-count upper bound for deferred canonicalisation (WRONG values)
  return s + ((s < a) ? EPS : 0);
#else
  return s + ((s < a || s >= P) ? EPS : 0);
#endif
}

// h*EPS + t for a 32-bit h and ANY 64-bit t, canonical.  The sum wraps at most once (h*EPS <= 2^64 - 2^33 + 1) and
// one correction covers both cases: after a wrap r < 2^64 - 2^33 + 1, so r + EPS < p (no second wrap); without a
// wrap, r >= p gives r + EPS - 2^64 = r - p < 2^32.  On the device this is 4 VALU: v_mad_u64_u32 takes the addend and
// delivers the wrap as its carry-out (the compiler never uses that output, and splits mad + add when the wrap is
// tested in C), the correction is a second mad by a 0/1 lane value.  SGPR timing inside the block: the mad's carry is
// read by SALU (no hazard) and by the v_cndmask two instructions later (the gfx950 VALU-writes-SGPR -> VALU-reads
// rule wants 2 wait states, which the v_cmp and the s_or supply).
RONK_HD u64 mad_eps_canon(u32 h, u64 t) {
#if defined(__HIP_DEVICE_COMPILE__) && (RONK_GL64_VARIANT & 2)
  u64 r, sc;
  u32 m;
  const u64 pm1 = P - 1;
  asm("v_mad_u64_u32 %0, %2, %3, -1, %4\n\t"
      "v_cmp_lt_u64_e32 vcc, %5, %0\n\t"
      "s_or_b64 %2, vcc, %2\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, %2\n\t"
      "v_mad_u64_u32 %0, vcc, %1, -1, %0"
      : "=&v"(r), "=&v"(m), "=&s"(sc)
      : "v"(h), "v"(t), "s"(pm1)
      : "vcc", "scc");
  return r;
#else
  u64 r = (u64)h * 0xFFFFFFFFu + t;
  return r + ((r < t || r >= P) ? EPS : 0);
#endif
}

RONK_HD u64 sub(u64 a, u64 b);

// a + b for canonical a, b when the consumer is a multiplication (mul / mul_2exp accept any 64-bit representative):
// only the wrap is folded back (2^64 = EPS), the ">= p" test of add() is skipped.  Result in [0, 2^64), == a + b (mod p).
RONK_HD u64 add_lazy(u64 a, u64 b) {
  u64 s = a + b;
  return s + ((s < a) ? EPS : 0);   // after a wrap s <= p - 2, so + EPS cannot wrap again
}

// a - h (+ p on borrow) for a 32-bit h: the upper limb only sees the borrow.  (Written out because the compiler turns
// "hi - 0 - borrow" into v_cndmask + v_sub_co; v_subbrev_co with the borrow as carry-in is one instruction.)
RONK_HD u64 sub32(u64 a, u32 h) {
#if defined(__HIP_DEVICE_COMPILE__) && (RONK_GL64_VARIANT & 4)
  u32 lo, hi;
  u64 sc;
  asm("v_sub_co_u32_e32 %0, vcc, %3, %5\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32_e32 %1, vcc, 0, %4, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_xor_b64 vcc, vcc, %2\n\t"
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "=&v"(lo), "=&v"(hi), "=&s"(sc)
      : "v"((u32)a), "v"((u32)(a >> 32)), "v"(h)
      : "vcc", "scc");
  return ((u64)hi << 32) | lo;
#else
  return sub(a, (u64)h);
#endif
}

// a - b, + p on borrow (prime/arithmetic.rs:19-28).  Also valid for ANY a < 2^64 and b <= p; the
// result is then some representative in [0, 2^64).
RONK_HD u64 sub(u64 a, u64 b) {
#if defined(__clang__) && !(RONK_GL64_VARIANT & 1)
  u32 b1, b2, b3, b4;
  u32 lo = __builtin_subc((u32)a, (u32)b, 0u, &b1);
  u32 hi = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), b1, &b2);
  u32 e = 0u - b2;  // 0xFFFFFFFF on borrow: d + p == d - EPS (mod 2^64)
  lo = __builtin_subc(lo, e, 0u, &b3);
  hi = __builtin_subc(hi, 0u, b3, &b4);
  return ((u64)hi << 32) | lo;
#elif defined(__clang__)
  // borrow chain, then d + B*p = d + B - B*2^32 on the limbs: lo + B (carry c2 only if lo == 2^32-1), hi - B + c2
  // = hi - (B xor c2).  4 VALU (v_sub_co, v_subb_co, v_addc_co, v_subbrev_co) + one s_xor on the carry masks.
  u32 b1, b2, c2, b4;
  u32 lo = __builtin_subc((u32)a, (u32)b, 0u, &b1);
  u32 hi = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), b1, &b2);
  lo = __builtin_addc(lo, 0u, b2, &c2);
  hi = __builtin_subc(hi, 0u, b2 ^ c2, &b4);
  return ((u64)hi << 32) | lo;
#else
  u64 d = a - b;
  return (a < b) ? d - EPS : d;
#endif
}

// prime/arithmetic.rs:61-65
RONK_HD u64 neg(u64 a) { return a ? P - a : 0; }

// 128-bit value hi:lo -> canonical residue.  hi = hh*2^32 + hl:
//   x = lo - hh + hl*(2^32-1)   (mod p)
RONK_HD u64 reduce128(u64 lo, u64 hi) {
  u32 hh = (u32)(hi >> 32), hl = (u32)hi;
  u64 t0 = sub32(lo, hh);                     // any representative; borrow -> + p
  return mad_eps_canon(hl, t0);
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
// Input: canonical (any 64-bit value works); output canonical.
template <int K>
RONK_HD u64 mul_2exp(u64 x) {
  static_assert(K >= 0 && K < 96, "shift out of range");
  if (K == 0) return x;
  constexpr int q = K / 32, s = K % 32;
  // x << s as limbs (y2, y1, y0) on 32-bit registers (shift, funnel shift v_alignbit_b32, shift): the halves of x
  // usually sit in unrelated registers (they come out of a borrow chain), where a 64-bit shift would need extra moves
  const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
  u32 y0, y1, y2;
  if constexpr (s == 0) {
    y0 = x0; y1 = x1; y2 = 0;
  } else {
    y0 = x0 << s;
    y1 = (x1 << s) | (x0 >> ((32 - s) & 31));
    y2 = x1 >> ((32 - s) & 31);
  }
  if constexpr (q == 0) {
    const u64 n = ((u64)y1 << 32) | y0;                       // may be >= p
    return mad_eps_canon(y2, n);                              // + y2*EPS (phi^2 = phi - 1)
  } else if constexpr (q == 1) {
    const u64 n = (u64)y0 << 32;                              // y0*phi
    const u64 r = mad_eps_canon(y1, n);                       // + y1*EPS, canonical
    return s ? sub32(r, y2) : r;                              // - y2 (phi^3 = -1)
  } else {
    const u64 t = (u64)y0 * 0xFFFFFFFFu;                      // y0*EPS < p
    const u64 m = ((u64)y2 << 32) | y1;                       // y2 < 2^31 -> < p
    return sub(t, m);                                         // canonical: t, m < p
  }
}

// -x * 2^K for a compile-time 0 <= K < 96 (i.e. x * 2^(K + 96): the upper half of the powers of two, 2^96 = -1).  The negation
// rides on the last subtraction of mul_2exp where there is one (K >= 32: operands swapped, no extra instruction); K < 32 pays
// a negation (4 VALU).  Input: any 64-bit value; output canonical.
template <int K>
RONK_HD u64 mul_2exp_neg(u64 x) {
  static_assert(K >= 0 && K < 96, "shift out of range");
  constexpr int q = K / 32, s = K % 32;
  const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
  u32 y0, y1, y2;
  if constexpr (s == 0) {
    y0 = x0; y1 = x1; y2 = 0;
  } else {
    y0 = x0 << s;
    y1 = (x1 << s) | (x0 >> ((32 - s) & 31));
    y2 = x1 >> ((32 - s) & 31);
  }
  if constexpr (q == 0) {
    const u64 n = ((u64)y1 << 32) | y0;
    return sub(0, mad_eps_canon(y2, n));                      // 0 - r: r canonical
  } else if constexpr (q == 1) {
    const u64 r = mad_eps_canon(y1, (u64)y0 << 32);           // canonical
    return sub((u64)y2, r);                                   // y2 - r (+ p)
  } else {
    const u64 t = (u64)y0 * 0xFFFFFFFFu;                      // < p
    const u64 m = ((u64)y2 << 32) | y1;                       // < p
    return sub(m, t);
  }
}

}  // namespace gl64
