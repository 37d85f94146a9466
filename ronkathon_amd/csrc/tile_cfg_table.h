// tile_cfg_table.h -- the (LOGR, LOGC, KIND) combinations that have a compile-time-specialised tile kernel (TileCfg,
// ntt_tile.h).  One list, used by the launcher (tile_kernels_cfg.hip) and by the host emulator (tests/emu) so that the
// very same instantiations are checked against the oracle on the CPU.
//   KIND 1 (column pass, two-level inter-pass twiddle)   2^19 .. 2^22
//   KIND 3 (column pass, full twiddle matrix)            2^16 .. 2^18 (default), 2^21 / 2^22 when the plan asks for it
//   KIND 2 (row pass)                                    2^16 .. 2^22
#pragma once
#define RONK_CFG_TABLE(X)                                                                     \
  X(10, 4, 1) X(10, 3, 1) X(10, 2, 1) X(11, 3, 1) X(11, 2, 1)                                          \
  X(8, 4, 3) X(9, 4, 3) X(10, 4, 3) X(11, 3, 3) X(11, 2, 3)                                   \
  X(8, 4, 2) X(9, 4, 2) X(10, 4, 2) X(10, 3, 2) X(10, 2, 2) X(11, 3, 2) X(11, 2, 2)
