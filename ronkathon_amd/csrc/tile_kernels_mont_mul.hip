// tile_kernels_mont_mul.hip -- the fused middle of a polynomial multiply (ntt_mul.h) over a Montgomery prime: forward row pass of
// both operands, pointwise product in registers (mul_plain: two Montgomery products, operands and result canonical), inverse
// column pass -- one launch instead of two, NTT(a) / NTT(b) never in HBM (reference src/polynomial/arithmetic.rs:97-119 for any
// PrimeField<P>).  Same shapes as the Goldilocks instantiations of tile_kernels_mul.hip that conv_dev uses: 2^11-row passes
// (NTT sizes 2^22, 2^23) and 2^10-row passes (2^21), 4-column tiles, both inverse twiddle forms.
#include <hip/hip_runtime.h>

#include "ntt_mul.h"
#include "tile_launch.h"

namespace ronk {

template <int LOGR, int LOGC, int KINDI>
__global__ void __launch_bounds__(1024) ntt_mul_mid_kernel_mont(const TileArgs fa, const TileArgs ia) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  mul_mid_body<LOGR, LOGC, KINDI, MontField>(fa, ia, lds, threadIdx.x, bid, [] { __syncthreads(); });
}

template <int LOGR, int LOGC, int KINDI>
static hipError_t launch_mid_mont(const TileArgs& fa, const TileArgs& ia, u32 grid, u32 block, size_t lds, hipStream_t s) {
  static bool attr_done[64] = {};
  if (lds > 48 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      e = hipFuncSetAttribute((const void*)ntt_mul_mid_kernel_mont<LOGR, LOGC, KINDI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ntt_mul_mid_kernel_mont<LOGR, LOGC, KINDI>), dim3(grid), dim3(block), lds, s, fa, ia);
  return hipGetLastError();
}

#define RONK_MUL_MID_TABLE_MONT(X) X(11, 2, 1) X(11, 2, 3) X(10, 2, 1) X(10, 2, 3)

bool mul_mid_available_mont(int logr, int logc, int kindi) {
#define RONK_MID_HAS(LR, LC, KD) if (logr == LR && logc == LC && kindi == KD) return true;
  RONK_MUL_MID_TABLE_MONT(RONK_MID_HAS)
#undef RONK_MID_HAS
  return false;
}

hipError_t launch_mul_mid_mont(int logr, int kindi, const TileArgs& fa, const TileArgs& ia, u32 grid, u32 block, size_t lds,
                               hipStream_t s, bool* found) {
#define RONK_MID_CASE(LR, LC, KD)                                                                   \
  if (logr == LR && (int)fa.logc == LC && kindi == KD && mul_mid_matches(fa, ia, LR, LC, KD)) {     \
    *found = true;                                                                                  \
    return launch_mid_mont<LR, LC, KD>(fa, ia, grid, block, lds, s);                                \
  }
  RONK_MUL_MID_TABLE_MONT(RONK_MID_CASE)
#undef RONK_MID_CASE
  *found = false;
  return hipSuccess;
}

}  // namespace ronk
