// small_kernels.hip -- gfx950 instantiations of the latency form of a pass (ntt_small.h): 4 coefficients per work-item,
// radix-4 rounds in place in LDS; chosen by the planner for two-pass plans of at most 2^17 coefficients in all.
#include <hip/hip_runtime.h>

#include "ntt_small.h"
#include "tile_launch.h"
#include "tile_kernel_def.h"

namespace ronk {

template <int LOGR, bool INV>
__global__ void __launch_bounds__(256) ntt_small_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  small_body<LOGR, INV>(a, lds, threadIdx.x, blockIdx.x, [] { __syncthreads(); });
}

template <bool INV>
static hipError_t launch_small_dir(int logr, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  switch (logr) {
#define RONK_SMALL_CASE(LR) \
    case LR: hipLaunchKernelGGL((ntt_small_kernel<LR, INV>), dim3(grid), dim3(block), lds, s, a); return hipGetLastError();
    RONK_SMALL_CASE(4) RONK_SMALL_CASE(5) RONK_SMALL_CASE(6) RONK_SMALL_CASE(7) RONK_SMALL_CASE(8) RONK_SMALL_CASE(9) RONK_SMALL_CASE(10)
#undef RONK_SMALL_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_small(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  if (block > 256 || lds > 48 * 1024) return hipErrorInvalidValue;   // make_small (plan.h): 256 work-items, <= 10 KiB
  if (a.fc.p) return launch_small_mont(logr, inverse, a, grid, block, lds, s);
  return inverse ? launch_small_dir<true>(logr, a, grid, block, lds, s) : launch_small_dir<false>(logr, a, grid, block, lds, s);
}

}  // namespace ronk
