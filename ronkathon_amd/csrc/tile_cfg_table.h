// tile_cfg_table.h -- the (LOGR, LOGC, KIND) combinations that have a compile-time-specialised tile kernel (TileCfg,
// ntt_tile.h).  One list, used by the launcher (tile_kernels_cfg.hip) and by the host emulator (tests/emu) so that the
// very same instantiations are checked against the oracle on the CPU.
//   KIND 1 (column pass, two-level inter-pass twiddle)   2^19 .. 2^22; first pass of the three-pass plans 2^23 .. 2^28
//   KIND 3 (column pass, full twiddle matrix)            2^16 .. 2^18 (default), 2^21 / 2^22 when the plan asks for it; middle
//                                                        pass of the three-pass plans
//   KIND 2 (row pass)                                    2^16 .. 2^22; last pass of the three-pass plans (flat rows)
#pragma once
#define RONK_CFG_TABLE(X)                                                                     \
  X(8, 4, 1) X(9, 4, 1) X(10, 4, 1) X(10, 3, 1) X(10, 2, 1) X(11, 3, 1) X(11, 2, 1)                    \
  X(8, 4, 3) X(9, 4, 3) X(10, 4, 3) X(11, 3, 3) X(11, 2, 3)                                   \
  X(7, 5, 2) X(8, 4, 2) X(9, 4, 2) X(10, 4, 2) X(10, 3, 2) X(10, 2, 2) X(11, 3, 2) X(11, 2, 2)
// Shapes of the multi-GPU four-step phases (plan.h build_dist_phase1 / 2; KIND 4 = general twiddled pass), for the sizes
// BASELINE config 5 and its neighbours use: 2^26 (R = C = 2^13: passes of 2^7 and 2^6 rows) and 2^24 (R = C = 2^12, one pass per
// phase); (12, 2, 1): the first pass of the single 2^23 transform (2^12 x 2^11).  No HALF variants (tile_kernels_half.hip walks
// RONK_CFG_TABLE only).
#define RONK_CFG_TABLE_DIST(X)                                                                \
  X(7, 5, 4) X(6, 6, 4) X(6, 6, 2) X(12, 2, 4) X(12, 2, 2) X(12, 2, 1)                        \
  X(12, 0, 5) X(11, 0, 5) X(10, 0, 5) X(9, 1, 5) X(8, 2, 5) X(7, 3, 5) X(6, 4, 5)
// (the last row: KIND 5 = whole-polynomial passes, the single-pass plans n = 2^6 .. 2^12 at the tile widths plan.h picks)
// The same shapes with FEATURES (TileCfg::FEAT, ntt_tile.h): X(LOGR, LOGC, KIND, FEAT).  1 = zero-padded input (the forward
// transforms of a polynomial multiply, a batched Reed-Solomon encode), 2 = second operand multiplied in on load (the inverse
// transform of a multiply, first pass), 4 = truncated output (its last pass).  NTT sizes 2^20 .. 2^22 of the multiply; the
// 1024 x 2^16 encode.  (12, 2, 1, 1) / (12, 2, 2, 4): the first and the last launch of the fused multiply at N = 2^23 (round 5).
#define RONK_CFG_TABLE_FEAT(X)                                                                \
  X(10, 2, 1, 1) X(11, 2, 1, 1) X(11, 2, 3, 1) X(8, 4, 3, 1) X(12, 2, 1, 1)                   \
  X(10, 2, 1, 2) X(11, 2, 1, 2) X(11, 3, 1, 2) X(11, 2, 3, 2) X(11, 3, 3, 2)                  \
  X(10, 2, 2, 4) X(10, 3, 2, 4) X(11, 3, 2, 4) X(11, 2, 2, 4) X(12, 2, 2, 4)
