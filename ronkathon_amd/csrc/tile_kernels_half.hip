// tile_kernels_half.hip -- the specialised tile kernels of tile_cfg_table.h with TileCfg::HALF: the exchanges between
// rounds go through LDS as two 32-bit phases, so a tile's image is half as large and twice as many workgroups are
// resident per CU (8 instead of 4 waves per SIMD when a pass has that many tiles).  Selected by the launcher
// (tile_kernels.hip) for passes with several tiles per CU.
#include "tile_cfg_table.h"
#include "tile_kernel_def.h"

namespace ronk {

#define RONK_HALF_CASE(LR, LC, KD)                                                                        \
  if (logr == LR && (int)a.logc == LC && kind == KD) {                                                    \
    *found = true;                                                                                        \
    return inverse ? launch_one<LR, true, LC, KD, true>(a, grid, block, lds, s)                           \
                   : launch_one<LR, false, LC, KD, true>(a, grid, block, lds, s);                         \
  }

hipError_t launch_tile_cfg_half(int logr, bool inverse, int kind, const TileArgs& a, u32 grid, u32 block, size_t lds,
                                hipStream_t s, bool* found) {
  RONK_CFG_TABLE(RONK_HALF_CASE)
  *found = false;
  return hipSuccess;
}

}  // namespace ronk
