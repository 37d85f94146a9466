// tile_kernels_r4.hip -- the column / row passes of 2^9 and 2^10 rows with the R4 round structure (TileCfg::R4, ntt_tile.h):
// rounds of radix 16, 4 and R/64, the twiddle after the first round a wave-uniform SHIFT (omega_64^(a k1) = +-2^K with the
// reference's root convention omega = 7^((p-1)/n), src/algebra/field/mod.rs:70-75), ONE table twiddle per pass.  Same results
// as the (16, 16, 2 | 4) kernels of tile_kernels_cfg.hip.  OPT-IN (RONK_R4MID=1, read by launch_tile): measured in round 5 it
// executes 5.5 % fewer VALU instructions per pass and is not faster (profiles/r05_r4_ab.txt), so the default stays with the
// (16, 16, 2 | 4) kernels; kept instantiated so that the measurement can be repeated and the parity tests keep covering it.
#include <hip/hip_runtime.h>

#include "tile_cfg_table.h"
#include "tile_kernel_def.h"

namespace ronk {

template <int LOGR, bool INV, int LOGC, int KIND>
__global__ void __launch_bounds__(1024) ntt_tile_kernel_r4(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  // XCD-aware renumbering (tile_kernel_main): each XCD works on a contiguous run of tiles
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  tile_body<LOGR, INV, 0, TileCfg<LOGC, KIND, false, false, 0, true>>(a, lds, threadIdx.x, bid, [] { __syncthreads(); });
}

template <int LOGR, bool INV, int LOGC, int KIND>
static hipError_t launch_one_r4(const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  static bool attr_done[64] = {};   // per (kernel, device), see launch_one
  if (lds > 48 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      e = hipFuncSetAttribute((const void*)ntt_tile_kernel_r4<LOGR, INV, LOGC, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ntt_tile_kernel_r4<LOGR, INV, LOGC, KIND>), dim3(grid), dim3(block), lds, s, a);
  return hipGetLastError();
}

template <int LR, int LC, int KD>
static hipError_t launch_r4_case(bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s, bool* found) {
  if constexpr (cfg_r4(LR, LC, KD)) {
    *found = true;
    return inverse ? launch_one_r4<LR, true, LC, KD>(a, grid, block, lds, s) : launch_one_r4<LR, false, LC, KD>(a, grid, block, lds, s);
  } else {
    (void)inverse; (void)a; (void)grid; (void)block; (void)lds; (void)s;
    *found = false;
    return hipSuccess;
  }
}

hipError_t launch_tile_r4(int logr, bool inverse, int kind, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s,
                          bool* found) {
#define RONK_R4_CASE(LR, LC, KD) \
  if (logr == LR && (int)a.logc == LC && kind == KD) return launch_r4_case<LR, LC, KD>(inverse, a, grid, block, lds, s, found);
  RONK_CFG_TABLE(RONK_R4_CASE)
#undef RONK_R4_CASE
  *found = false;
  return hipSuccess;
}

}  // namespace ronk
