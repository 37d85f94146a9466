// tile_kernels_w.hip -- EXPERIMENT (not built into the library; see ntt_tile_w.h): gfx950 instantiations of the tile body with a wave-local first exchange (ntt_tile_w.h): 2^11-row,
// 4-column tiles of the two-pass plans (the two-lane plan of ronk_ntt_forward_many_dev, 2^21 / 2^22 / 2^23 shapes).
#include <hip/hip_runtime.h>

#include "ntt_tile_w.h"
#include "../../ronkathon_amd/csrc/tile_launch.h"

namespace ronk {

template <bool INV, int KIND>
__global__ void __launch_bounds__(512, 2) ntt_tile_w_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  tile_body_w<INV, KIND>(a, lds, threadIdx.x, bid, [] { __syncthreads(); }, [] { __builtin_amdgcn_wave_barrier(); });
}

template <bool INV, int KIND>
static hipError_t launch_w(const TileArgs& a, u32 grid, hipStream_t s) {
  static bool attr_done[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    e = hipFuncSetAttribute((const void*)ntt_tile_w_kernel<INV, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  hipLaunchKernelGGL((ntt_tile_w_kernel<INV, KIND>), dim3(grid), dim3(512), TW_LDS_BYTES, s, a);
  return hipGetLastError();
}

hipError_t launch_tile_w(int logr, bool inverse, int kind, const TileArgs& a, u32 grid, hipStream_t s, bool* found) {
  *found = tile_w_matches(a, logr, kind);
  if (!*found) return hipSuccess;
  switch (kind) {
    case 1: return inverse ? launch_w<true, 1>(a, grid, s) : launch_w<false, 1>(a, grid, s);
    case 2: return inverse ? launch_w<true, 2>(a, grid, s) : launch_w<false, 2>(a, grid, s);
    default: return inverse ? launch_w<true, 3>(a, grid, s) : launch_w<false, 3>(a, grid, s);
  }
}

}  // namespace ronk
