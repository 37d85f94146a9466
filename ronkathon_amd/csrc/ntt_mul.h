// ntt_mul.h -- the middle of a polynomial multiply as ONE tile body (BASELINE config 3; reference semantics
// src/polynomial/arithmetic.rs:97-119: c[i + j] += a[i] * b[j], D + D2 - 1 coefficients).
//
// a * b = iNTT(NTT(a) . NTT(b)) on N = A * B points (A = B = 2^LOGR here).  With two-pass plans that is four launches per
// product and NTT(a), NTT(b) travel through HBM twice:
//     F1  column pass of a and b (one batch of two)           in -> scratch
//     F2  row pass of both: NTT(a), NTT(b) in natural order   scratch -> 2 N elements
//     I1  column pass of the inverse, a.b multiplied on load  2 N elements -> scratch
//     I2  row pass of the inverse, truncated                  scratch -> out
// The tile a workgroup of F2 produces -- C adjacent k_a, every k_b, at k_a + A k_b -- is exactly the tile a workgroup of I1
// consumes (rows k_b, columns k_a), and after F2's last round lane m of column c holds the 16 outputs k_b = m + M t
// (t = 0..15: keep_row_digit), which are the 16 rows i M + m the FIRST round of I1's column transform starts from.  So:
//     F1;  M = [F2 on the a tile, results in 32 VGPRs] [F2 on the b tile] [product in registers] [I1 from registers];  I2
// three launches, NTT(a) and NTT(b) never exist in HBM (-4 N elements of traffic of 24 N), no exchange between the forward
// and the inverse rounds (a compile-time register renaming).  The LDS image is used by the three transforms one after
// the other; a barrier separates them (the last-round reads of one from the first parks of the next).
#pragma once
#include "ntt_tile.h"

namespace ronk {

// hides a lane index from the optimiser (device build; a no-op for the host emulator)
RONK_HD void mul_mid_opaque(u32& t) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(t));
#else
  (void)t;
#endif
}

// fa: the forward plan's ROW pass (KIND 2 shape, batch of two: polynomial b1 = 0 is a, 1 is b; in = its scratch)
// ia: the inverse plan's COLUMN pass (KIND 1 or 3 shape; out = the inverse plan's scratch; `in` is not read)
// bid in [0, tiles): the same tile number on both sides (same tile width: LOGC)
// FLD: the field policy (field_policy.h) -- Goldilocks, or a Montgomery prime (round 5: the product of two canonical values is
// mul_plain = mmul(mmul(x, y), R^2); fa.fc / ia.fc carry the prime)
template <int LOGR, int LOGC, int KINDI, class FLD = GlField, class Barrier>
RONK_HD void mul_mid_body(const TileArgs& fa, const TileArgs& ia, u64* lds, u32 tid, u32 bid, Barrier&& barrier) {
  static_assert(LOGR >= 9 && LOGR <= 12, "three-round passes");
  typedef TileCfg<LOGC, 2, false, false, FEAT_KEEP> CF;
  typedef TileCfg<LOGC, KINDI> CI;
  u64 x[16], ya[16], y[16];
  // The three transforms address the same tile with the same lane: left alone, the compiler keeps every per-lane offset of
  // the first one (16 load offsets, the LDS cells of each round, the table offsets) live for the other two -- 128 VGPRs and
  // 350 bytes of scratch per lane.  An opaque copy of the lane index per phase makes it recompute them (a few dozen full-rate
  // adds) instead.
  u32 t1 = tid, t2 = tid, t3 = tid;
  mul_mid_opaque(t1);
  {
    const TileCtx cx = tile_ctx<LOGR, CF>(fa, t1, bid);
    tile_load<LOGR, false, 0, CF, FLD>(cx, lds, t1, x, barrier);
    tile_compute<LOGR, false, 0, CF, FLD>(cx, lds, t1, x, barrier);
  }
#pragma unroll
  for (int r = 0; r < 16; r++) ya[keep_row_digit(LOGR, r)] = x[r];
  barrier();   // every lane has read its last-round rows of the a tile before the image is written again
  mul_mid_opaque(t2);
  {
    const TileCtx cx = tile_ctx<LOGR, CF>(fa, t2, bid + fa.tiles);   // b1 = 1
    tile_load<LOGR, false, 0, CF, FLD>(cx, lds, t2, x, barrier);
    tile_compute<LOGR, false, 0, CF, FLD>(cx, lds, t2, x, barrier);
  }
  {
    const FLD f(fa.fc);
#pragma unroll
    for (int r = 0; r < 16; r++) y[keep_row_digit(LOGR, r)] = f.mul_plain(ya[keep_row_digit(LOGR, r)], x[r]);
  }
  barrier();
  mul_mid_opaque(t3);
  {
    const TileCtx cx = tile_ctx<LOGR, CI>(ia, t3, bid);
    tile_compute<LOGR, true, 0, CI, FLD>(cx, lds, t3, y, barrier);
  }
}

// host-side check that two passes can be fused: same tile geometry, the shapes the instantiations know
inline bool mul_mid_matches(const TileArgs& fa, const TileArgs& ia, int logr, int logc, int kindi) {
  return fa.nb1 == 2 && fa.nb2 == 1 && ia.nb1 == 1 && ia.nb2 == 1 && fa.tiles == ia.tiles && fa.logc == ia.logc &&
         tile_features(fa) == 0 && tile_features(ia) == 0 && tile_cfg_matches(fa, logr, logc, 2) &&
         tile_cfg_matches(ia, logr, logc, kindi);
}

}  // namespace ronk
