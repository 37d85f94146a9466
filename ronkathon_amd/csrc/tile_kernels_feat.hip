// tile_kernels_feat.hip -- the specialised tile kernels for passes that carry FEATURES (TileCfg::FEAT, ntt_tile.h): the
// zero-padded forward transforms, the fused pointwise product and the truncated store of a polynomial multiply
// (reference src/polynomial/arithmetic.rs:97-119 through `From<[F;N]>` padding, mod.rs:503-515), and the zero-padded batched
// Reed-Solomon encode (src/codes/reed_solomon.rs:42-52).  List: RONK_CFG_TABLE_FEAT in tile_cfg_table.h.
#include "tile_cfg_table.h"
#include "tile_kernel_def.h"

namespace ronk {

#define RONK_FEAT_CASE(LR, LC, KD, FT)                                                                    \
  if (logr == LR && (int)a.logc == LC && kind == KD && feat == FT) {                                      \
    *found = true;                                                                                        \
    return inverse ? launch_one_feat<LR, true, LC, KD, FT>(a, grid, block, lds, s)                        \
                   : launch_one_feat<LR, false, LC, KD, FT>(a, grid, block, lds, s);                      \
  }

hipError_t launch_tile_cfg_feat(int logr, bool inverse, int kind, int feat, const TileArgs& a, u32 grid, u32 block, size_t lds,
                                hipStream_t s, bool* found) {
  RONK_CFG_TABLE_FEAT(RONK_FEAT_CASE)
  *found = false;
  return hipSuccess;
}

}  // namespace ronk
