// tile_kernels_mont.hip -- the tile kernels over ANY odd prime p < 2^64 with enough 2-adicity (Montgomery form, R = 2^64).
//
// ronkathon's PrimeField<P> and Polynomial::fft / ifft are generic over the modulus (reference
// src/algebra/field/prime/mod.rs:39-52, src/polynomial/mod.rs:273-323, :430-484).  The bodies are ntt_tile.h / ntt_small.h
// instantiated with field_policy.h's MontField: the prime, -p^-1 mod 2^64, 2^128 mod p and the eight roots of a 16-point
// register round arrive in TileArgs::fc (kernel arguments -> SGPRs); every table the plan uploads holds w * 2^64 mod p, so a
// twiddle costs one Montgomery product and coefficients stay canonical everywhere (plan.h HostField::tab).
//   generic instantiation (KIND 0)  every pass size 2^4 .. 2^12, every feature (zero padding, second operand, truncation):
//                                   the polynomial multiply, the staged single-pass plans, three-pass plans
//   specialised shapes              RONK_CFG_TABLE (tile_cfg_table.h): the column / row passes of the two-pass plans
//   latency form                    ntt_small.h, 2^4 .. 2^10 rows
#include <hip/hip_runtime.h>

#include "ntt_small.h"
#include "tile_cfg_table.h"
#include "tile_kernel_def.h"
#include "tile_launch.h"

namespace ronk {

template <int LOGR, bool INV, int LOGC, int KIND>
__global__ void __launch_bounds__(1024) ntt_tile_kernel_mont(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  tile_kernel_main<LOGR, INV, LOGC, KIND, false, 0, MontField>(a, lds);
}

template <int LOGR, bool INV, int LOGC, int KIND>
static hipError_t launch_one_mont(const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  static bool attr_done[64] = {};   // per (kernel, device), see launch_one
  if (lds > 48 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      e = hipFuncSetAttribute((const void*)ntt_tile_kernel_mont<LOGR, INV, LOGC, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ntt_tile_kernel_mont<LOGR, INV, LOGC, KIND>), dim3(grid), dim3(block), lds, s, a);
  return hipGetLastError();
}

template <bool INV>
static hipError_t launch_dir_mont(int logr, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  switch (logr) {
    case 4: return launch_one_mont<4, INV, -1, 0>(a, grid, block, lds, s);
    case 5: return launch_one_mont<5, INV, -1, 0>(a, grid, block, lds, s);
    case 6: return launch_one_mont<6, INV, -1, 0>(a, grid, block, lds, s);
    case 7: return launch_one_mont<7, INV, -1, 0>(a, grid, block, lds, s);
    case 8: return launch_one_mont<8, INV, -1, 0>(a, grid, block, lds, s);
    case 9: return launch_one_mont<9, INV, -1, 0>(a, grid, block, lds, s);
    case 10: return launch_one_mont<10, INV, -1, 0>(a, grid, block, lds, s);
    case 11: return launch_one_mont<11, INV, -1, 0>(a, grid, block, lds, s);
    case 12: return launch_one_mont<12, INV, -1, 0>(a, grid, block, lds, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_tile_mont(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  static const bool no_cfg = getenv("RONK_NO_CFG_KERNELS") != nullptr;
  if (!no_cfg && tile_features(a) != 0) {   // the multiply's / the encode's passes (tile_kernels_mont_feat.hip)
    bool found = false;
    hipError_t e = launch_tile_mont_feat(logr, inverse, tile_features(a), a, grid, block, lds, s, &found);
    if (found) return e;
  }
  if (!no_cfg && tile_features(a) == 0 && logr >= 10 && logr <= 12 && a.logc == 2) {   // ntt_tile_wl.h over the Montgomery field policy
    for (int kind : {1, 2, 3}) {
      bool half = false, found = false;
      if (!tile_wl_wanted(kind, &half)) continue;
      hipError_t e = launch_tile_wl(logr, inverse, kind, false, a, grid, s, &found);
      if (found) return e;
    }
  }
  if (!no_cfg && tile_features(a) == 0) {
#define RONK_MONT_CASE(LR, LC, KD)                                                               \
  if (logr == LR && (int)a.logc == LC && tile_cfg_matches(a, LR, LC, KD))                        \
    return inverse ? launch_one_mont<LR, true, LC, KD>(a, grid, block, lds, s)                   \
                   : launch_one_mont<LR, false, LC, KD>(a, grid, block, lds, s);
    RONK_CFG_TABLE(RONK_MONT_CASE)
#undef RONK_MONT_CASE
  }
  return inverse ? launch_dir_mont<true>(logr, a, grid, block, lds, s) : launch_dir_mont<false>(logr, a, grid, block, lds, s);
}

template <int LOGR, bool INV>
__global__ void __launch_bounds__(256) ntt_small_kernel_mont(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  small_body<LOGR, INV, MontField>(a, lds, threadIdx.x, blockIdx.x, [] { __syncthreads(); });
}

template <bool INV>
static hipError_t launch_small_dir_mont(int logr, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  switch (logr) {
#define RONK_SMALL_CASE(LR) \
    case LR: hipLaunchKernelGGL((ntt_small_kernel_mont<LR, INV>), dim3(grid), dim3(block), lds, s, a); return hipGetLastError();
    RONK_SMALL_CASE(4) RONK_SMALL_CASE(5) RONK_SMALL_CASE(6) RONK_SMALL_CASE(7) RONK_SMALL_CASE(8) RONK_SMALL_CASE(9) RONK_SMALL_CASE(10)
#undef RONK_SMALL_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_small_mont(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  return inverse ? launch_small_dir_mont<true>(logr, a, grid, block, lds, s) : launch_small_dir_mont<false>(logr, a, grid, block, lds, s);
}

}  // namespace ronk
