// tile_kernels_mul.hip -- gfx950 instantiations of the fused middle of a polynomial multiply (ntt_mul.h): forward row pass of
// both operands, pointwise product in registers, inverse column pass -- one launch instead of two, NTT(a) / NTT(b) never in HBM.
#include <hip/hip_runtime.h>

#include "ntt_mul.h"
#include "tile_launch.h"

namespace ronk {

template <int LOGR, int LOGC, int KINDI>
__global__ void __launch_bounds__(1024) ntt_mul_mid_kernel(const TileArgs fa, const TileArgs ia) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  // XCD-aware renumbering (tile_kernel_def.h): each XCD works on a contiguous run of tiles
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  mul_mid_body<LOGR, LOGC, KINDI>(fa, ia, lds, threadIdx.x, bid, [] { __syncthreads(); });
}

template <int LOGR, int LOGC, int KINDI>
static hipError_t launch_mid(const TileArgs& fa, const TileArgs& ia, u32 grid, u32 block, size_t lds, hipStream_t s) {
  static bool attr_done[64] = {};
  if (lds > 48 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      e = hipFuncSetAttribute((const void*)ntt_mul_mid_kernel<LOGR, LOGC, KINDI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ntt_mul_mid_kernel<LOGR, LOGC, KINDI>), dim3(grid), dim3(block), lds, s, fa, ia);
  return hipGetLastError();
}

// the (rows, tile width, inverse twiddle form) combinations of the multiply's plans, 4- and 8-column tiles: 2^11-row passes = NTT
// size 2^22; 2^10-row passes = NTT size 2^21 (pair plan 2^11 x 2^10, inverse split the other way round) and 2^20 (instantiated,
// but conv_dev keeps four launches there: measured slower fused)
#define RONK_MUL_MID_TABLE(X) X(11, 2, 1) X(11, 2, 3) X(11, 3, 1) X(11, 3, 3) X(10, 2, 1) X(10, 2, 3) X(10, 3, 1) X(10, 3, 3)

bool mul_mid_available(int logr, int logc, int kindi) {
#define RONK_MID_HAS(LR, LC, KD) if (logr == LR && logc == LC && kindi == KD) return true;
  RONK_MUL_MID_TABLE(RONK_MID_HAS)
#undef RONK_MID_HAS
  return false;
}

hipError_t launch_mul_mid(int logr, int kindi, const TileArgs& fa, const TileArgs& ia, u32 grid, u32 block, size_t lds,
                          hipStream_t s, bool* found) {
#define RONK_MID_CASE(LR, LC, KD)                                                                   \
  if (logr == LR && (int)fa.logc == LC && kindi == KD && mul_mid_matches(fa, ia, LR, LC, KD)) {     \
    *found = true;                                                                                  \
    return launch_mid<LR, LC, KD>(fa, ia, grid, block, lds, s);                                     \
  }
  RONK_MUL_MID_TABLE(RONK_MID_CASE)
#undef RONK_MID_CASE
  *found = false;
  return hipSuccess;
}

}  // namespace ronk
