// tile_kernels_mont_feat.hip -- Montgomery-prime instantiations of the specialised passes that carry FEATURES (TileCfg::FEAT,
// ntt_tile.h; list RONK_CFG_TABLE_FEAT in tile_cfg_table.h): the pieces of a polynomial multiply over a generic odd 64-bit prime
// (reference src/polynomial/arithmetic.rs:97-119 for any PrimeField<P>) -- zero-padded forward transforms (FEAT 1, forward
// direction only), the second operand multiplied in on load (FEAT 2) and the truncated store (FEAT 4) of the inverse -- and the
// zero-padded batched Reed-Solomon encode (src/codes/reed_solomon.rs:42-52).  Without an instantiation a pass with features runs
// the generic Montgomery kernel (tile_kernels_mont.hip).
#include <hip/hip_runtime.h>

#include "tile_cfg_table.h"
#include "tile_kernel_def.h"

namespace ronk {

template <int LOGR, bool INV, int LOGC, int KIND, int FEAT>
__global__ void __launch_bounds__(1024) ntt_tile_kernel_mont_feat(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  tile_kernel_main<LOGR, INV, LOGC, KIND, false, FEAT, MontField>(a, lds);
}

template <int LOGR, bool INV, int LOGC, int KIND, int FEAT>
static hipError_t launch_one_mont_feat(const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  static bool attr_done[64] = {};   // per (kernel, device), see launch_one
  if (lds > 48 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      e = hipFuncSetAttribute((const void*)ntt_tile_kernel_mont_feat<LOGR, INV, LOGC, KIND, FEAT>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ntt_tile_kernel_mont_feat<LOGR, INV, LOGC, KIND, FEAT>), dim3(grid), dim3(block), lds, s, a);
  return hipGetLastError();
}

// direction a feature occurs in: padding limits belong to forward transforms (multiply operands, encode), the second operand
// and the truncation to the multiply's inverse
#define RONK_MONT_FEAT_CASE(LR, LC, KD, FT)                                                              \
  if (logr == LR && (int)a.logc == LC && feat == FT && inverse == (FT != 1) && tile_cfg_matches(a, LR, LC, KD, FT)) { \
    *found = true;                                                                                       \
    return launch_one_mont_feat<LR, (FT != 1), LC, KD, FT>(a, grid, block, lds, s);                      \
  }

hipError_t launch_tile_mont_feat(int logr, bool inverse, int feat, const TileArgs& a, u32 grid, u32 block, size_t lds,
                                 hipStream_t s, bool* found) {
  RONK_CFG_TABLE_FEAT(RONK_MONT_FEAT_CASE)
  *found = false;
  return hipSuccess;
}

}  // namespace ronk
