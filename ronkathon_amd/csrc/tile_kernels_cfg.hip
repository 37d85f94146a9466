// tile_kernels_cfg.hip -- tile kernels that know their pass's shape at compile time (TileCfg, ntt_tile.h): the column
// and row passes of the two-pass plans 2^16 .. 2^22 at the tile widths the planner (plan.h) and bench.py use.
// The list of (LOGR, LOGC, KIND) combinations is tile_cfg_table.h.
// Anything else (other sizes, fused multiply operands, ragged or staged tiles, the multi-GPU phases) runs the generic
// kernels of tile_kernels.hip.
#include "tile_cfg_table.h"
#include "tile_kernel_def.h"

namespace ronk {

#define RONK_CFG_CASE(LR, LC, KD)                                                                         \
  if (logr == LR && (int)a.logc == LC && kind == KD) {                                                    \
    *found = true;                                                                                        \
    return inverse ? launch_one<LR, true, LC, KD>(a, grid, block, lds, s)                                 \
                   : launch_one<LR, false, LC, KD>(a, grid, block, lds, s);                               \
  }

hipError_t launch_tile_cfg(int logr, bool inverse, int kind, const TileArgs& a, u32 grid, u32 block, size_t lds,
                           hipStream_t s, bool* found) {
  RONK_CFG_TABLE(RONK_CFG_CASE)
  RONK_CFG_TABLE_DIST(RONK_CFG_CASE)
  *found = false;
  return hipSuccess;
}

}  // namespace ronk
