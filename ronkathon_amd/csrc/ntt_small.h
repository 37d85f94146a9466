// ntt_small.h -- the LATENCY form of a tile pass: 4 coefficients per work-item, radix-4 rounds in place in LDS.
//
// ntt_tile.h gives a work-item 16 coefficients (fewest LDS exchanges and table twiddles per coefficient: the right trade
// when the chip is full).  A single 2^16 transform (BASELINE config 2) is only 64 waves of that kind on 1024 SIMDs, and its
// time is one wave's dependent instruction stream: ~2 000 VALU instructions per pass at ~9.5 cycles each when a wave has
// its SIMD to itself (tools/clock_probe.hip) = 8 us per launch, 35 us per forward + inverse.  Here the same pass is
// R*C/4 work-items doing log4(R) rounds of one radix-4 butterfly each (omega_4 = 2^48: a shift), i.e. a quarter of the
// instructions per lane and four times the waves; more barriers and one table twiddle per coefficient and round do not
// matter when nothing else competes for the VALU.  Selected by the planner (plan.h) for two-pass plans whose whole batch
// is small (plan.h: at most 2^18 coefficients, 2^19 for n <= 2^17, 2^20 for n <= 2^14).
//
// Same TileArgs contract as tile_body (strides, blocked rows, inter-pass twiddle by two-level table or full matrix, scale,
// second operand and implicit padding / truncation of the fused polynomial multiply; no staged I/O), same results: X[k] = sum_j x[j] omega^{jk}, natural order in and out
// (reference src/polynomial/mod.rs:273-323, :430-484).
//
// One column of R = 2^LOGR points, in place, decimation in frequency: round s works on blocks of L = R/4^s points,
//   a_i = y[B + j + i*L/4]   ->   y[B + j + r*L/4] = (sum_i a_i omega_4^{ir}) * omega_L^{jr},   r = 0..3
// (a last radix-2 round when LOGR is odd); afterwards position p = d_0*R/4 + d_1*R/16 + ... holds X[d_0 + 4 d_1 + ...].
// Round 0 reads its inputs from HBM, the last round writes to HBM; LDS image: [R + R/4][C] (one dummy row per 4 rows keeps
// the short strides of the last rounds off the same banks).
#pragma once
#include "ntt_tile.h"

namespace ronk {

RONK_HD u32 small_row(u32 p) { return p + (p >> 2); }

// the value is materialised here (device: an empty asm that reads and writes its register)
RONK_HD void small_keep(u64& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}

// omega_4^{+-1} * x: omega_4 = omega_64^16 = 2^(39*16 mod 192) = 2^48; the inverse is 2^144 = -2^48
// (Montgomery primes: one product by the table entry omega_16^{+-4}, field_policy.h)
template <bool INV, class FLD>
RONK_HD u64 mul_w4_of_diff(const FLD& f, u64 a, u64 b) {   // (a - b) * omega_4^{+-1}
  return f.template sub_mul_root<4, 1, INV>(a, b);
}

template <int LOGR, bool INV, class FLD = GlField, class Barrier>
RONK_HD void small_body(const TileArgs& a, u64* lds, u32 tid, u32 bid, Barrier&& barrier) {
  constexpr u32 R = 1u << LOGR;
  constexpr int S4 = LOGR / 2;              // radix-4 rounds
  constexpr bool ODD = (LOGR & 1) != 0;     // + one radix-2 round
  static_assert(LOGR >= 4 && LOGR <= 10, "small pass size");
  const FLD f(a.fc);
  const u32 logc = a.logc, C = 1u << logc;
  const u32 c = tid & (C - 1), u = tid >> logc;   // u in [0, R/4)
  const u32 t = bid % a.tiles, bb = bid / a.tiles, b1 = bb % a.nb1, b2 = bb / a.nb1;
  const u32 col = (t << logc) + c;
  const bool live = col < a.ncols;
  const u64* in = a.in + (i64)b1 * a.in_sb1 + (i64)b2 * a.in_sb2 + (i64)t * a.in_st + (i64)c * a.in_sc;
  u64* out = a.out + (i64)b1 * a.out_sb1 + (i64)b2 * a.out_sb2 + (i64)t * a.out_st + (i64)c * a.out_sc;
  auto in_row = [&](u32 j) -> i64 {
    return a.js_log < 31 ? (i64)(j >> a.js_log) * a.in_sj_hi + (i64)(j & ((1u << a.js_log) - 1)) * a.in_sj : (i64)j * a.in_sj;
  };
  auto cell = [&](u32 p) -> u32 { return (small_row(p) << logc) + c; };

  // natural output index of position p: reverse the base-4 digits (and the last binary digit when LOGR is odd)
  auto natural = [&](u32 p) -> u32 {
    u32 k = 0;
    if (ODD) {
      k = (p & 1) << (2 * S4);
      p >>= 1;
    }
#pragma unroll
    for (int s = S4 - 1; s >= 0; s--) { k |= (p & 3) << (2 * s); p >>= 2; }
    return k;
  };
  // ---- EVERY global load of the pass is issued here, before anything waits for one: the four inputs (and the second
  // operand's), the table twiddles of all rounds, the inter-pass twiddles of the four outputs.  None of the addresses
  // depends on data, and a launch of this kernel is a handful of memory round trips long: taken one after the other -- a
  // load behind every `if`, as the first version compiled -- they were most of its 6.3 us (profiles/r03_small_kernel_loads.txt).
  // Loads that a lane must not make (padding, dead columns) go to a safe address and are discarded; round twiddles with
  // exponent 0 read omega^0 = 1 from the table instead of being skipped.
  u64 x[4], x2[4];
  bool ok[4];
  {
    constexpr u32 q = R / 4;   // round 0: one block, j = u
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const i64 off = in_row(u + i * q);
      // lin = offset inside the polynomial (everything but the b1 term): >= in_valid reads as ZERO (From<[F;N]> padding)
      const u64 lin = (u64)((i64)b2 * a.in_sb2 + (i64)t * a.in_st + (i64)c * a.in_sc + off);
      const u64 valid = (b1 && a.in_valid1 != ~(u64)0) ? a.in_valid1 : a.in_valid;
      ok[i] = live && (valid == ~(u64)0 || lin < valid);
      x[i] = *(ok[i] ? in + off : a.in);
    }
    if (a.in2) {                                             // fused pointwise product
#pragma unroll
      for (int i = 0; i < 4; i++) x2[i] = *(ok[i] ? a.in2 + (in - a.in) + in_row(u + i * q) : a.in2);
    }
  }
  constexpr int TWR = ODD ? S4 : S4 - 1;                     // rounds followed by table twiddles (none after L == 4, j == 0)
  u64 twr[TWR > 0 ? TWR : 1][3];
#pragma unroll
  for (int s = 0; s < TWR; s++) {
    const u32 L = R >> (2 * s), q = L / 4;
    const u32 step = (u % q) * (R / L);
#pragma unroll
    for (int r = 1; r < 4; r++) twr[s][r - 1] = ld_tab(a.wr, (step * r) & (R - 1));
  }
  u32 pos[4];
#pragma unroll
  for (int i = 0; i < 4; i++) pos[i] = ODD ? 4 * u + i : u * 4 + i;   // (odd sizes: positions 2v, 2v+1 of v = 2u, 2u+1)
  const u32 nmask = a.tw_log >= 32 ? 0xFFFFFFFFu : ((1u << a.tw_log) - 1);
  const u32 lmask = (1u << a.tw_lo_bits) - 1;
  const u32 twX = (u32)a.xc * col + (u32)a.xb1 * b1 + (u32)a.xb2 * b2 + (u32)a.x0;
  u64 two[4], two_hi[4];
  if (a.tw_full) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      two[i] = a.tw_full[live ? (size_t)natural(pos[i]) * a.tf_sk + (size_t)col * a.tf_sc + (size_t)b2 * a.tf_sb2 : (size_t)0];
  } else if (a.tw_log) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32 e = (twX * ((u32)a.yk * natural(pos[i]) + (u32)a.yb1 * b1 + (u32)a.yb2 * b2 + (u32)a.y0)) & nmask;
      two[i] = ld_tab(a.tw_lo, e & lmask);
      two_hi[i] = ld_tab(a.tw_hi, e >> a.tw_lo_bits);
    }
    // (the compiler would sink these eight loads to their use behind the last barrier -- one more exposed round trip)
#pragma unroll
    for (int i = 0; i < 4; i++) { small_keep(two[i]); small_keep(two_hi[i]); }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (!ok[i]) x[i] = 0;
    else if (a.in2) x[i] = f.mul_plain(x[i], x2[i]);
  }
  // ---- radix-4 rounds
#pragma unroll
  for (int s = 0; s < S4; s++) {
    const u32 L = R >> (2 * s), q = L / 4;
    const u32 Bk = u / q, j = u % q;
    const u32 p0 = Bk * L + j;
    if (s > 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = lds[cell(p0 + i * q)];
    }
    const u64 t0 = f.add(x[0], x[2]), t1 = f.sub(x[0], x[2]);
    const u64 t2 = f.add(x[1], x[3]), t3 = mul_w4_of_diff<INV>(f, x[1], x[3]);
    x[0] = f.add(t0, t2); x[1] = f.add(t1, t3); x[2] = f.sub(t0, t2); x[3] = f.sub(t1, t3);
    // twiddles omega_L^{j r} = omega_R^{j r R/L} (fetched above)
    if (s < TWR) {
#pragma unroll
      for (int r = 1; r < 4; r++) x[r] = f.mul(x[r], twr[s < TWR ? s : 0][r - 1]);
    }
    if (s == S4 - 1 && !ODD) break;   // results stay in registers: output below
    // in place: the four positions belong to this work-item alone in this round, so no barrier between its reads and writes
#pragma unroll
    for (int i = 0; i < 4; i++) lds[cell(p0 + i * q)] = x[i];
    barrier();
  }
  if (ODD) {
    // last round: radix 2 on positions 2v, 2v+1; each work-item does two butterflies (v = 2u, 2u + 1)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const u32 v = 2 * u + h;
      const u64 e0 = lds[cell(2 * v)], e1 = lds[cell(2 * v + 1)];
      x[2 * h] = f.add(e0, e1);
      x[2 * h + 1] = f.sub(e0, e1);
    }
  }
  // ---- output: inter-pass twiddle / scale, natural order
  if (!live) return;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const u32 k = natural(pos[i]);
    u64 v = x[i];
    if (a.tw_full) v = f.mul(v, two[i]);
    else if (a.tw_log) v = f.mul(v, f.mul(two[i], two_hi[i]));
    if (a.scale != 1) v = f.mul(v, a.scale);
    const u64 lin = (u64)((i64)b2 * a.out_sb2 + (i64)t * a.out_st + (i64)c * a.out_sc + (i64)k * a.out_sk);
    if (a.out_valid == ~(u64)0 || lin < a.out_valid) out[(i64)k * a.out_sk] = v;
  }
}

}  // namespace ronk
