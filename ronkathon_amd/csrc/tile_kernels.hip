// tile_kernels.hip -- gfx950 instantiations of the NTT tile kernel (ntt_tile.h) and their launcher.
//
// One __global__ per (LOGR, direction).  Workgroup = 2^LOGR * C / 16 work-items (<= 1024),
// dynamic LDS = 2^LOGR * C * 8 bytes (<= 128 KiB of the CU's 160 KiB).
#include <hip/hip_runtime.h>

#include "ntt_tile.h"
#include "tile_launch.h"

namespace ronk {

template <int LOGR, bool INV>
__global__ void __launch_bounds__(1024) ntt_tile_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  // The dispatcher hands workgroup b to XCD b % 8 (observed, for speed only): renumber so that
  // each XCD works on a contiguous run of tiles -- neighbouring tiles share 128-byte lines and
  // twiddle rows, which then hit in that XCD's private L2.  Bijective for any grid size.
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  tile_body<LOGR, INV, 0>(a, lds, threadIdx.x, bid, [] { __syncthreads(); });
}

template <int LOGR, bool INV>
static hipError_t launch_one(const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  // HIP keeps this attribute per (kernel, DEVICE): one flag per device ordinal (benign race: the call is idempotent)
  static bool attr_done[64] = {};
  if (lds > 48 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      e = hipFuncSetAttribute((const void*)ntt_tile_kernel<LOGR, INV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ntt_tile_kernel<LOGR, INV>), dim3(grid), dim3(block), lds, s, a);
  return hipGetLastError();
}

template <bool INV>
static hipError_t launch_dir(int logr, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  switch (logr) {
    case 4: return launch_one<4, INV>(a, grid, block, lds, s);
    case 5: return launch_one<5, INV>(a, grid, block, lds, s);
    case 6: return launch_one<6, INV>(a, grid, block, lds, s);
    case 7: return launch_one<7, INV>(a, grid, block, lds, s);
    case 8: return launch_one<8, INV>(a, grid, block, lds, s);
    case 9: return launch_one<9, INV>(a, grid, block, lds, s);
    case 10: return launch_one<10, INV>(a, grid, block, lds, s);
    case 11: return launch_one<11, INV>(a, grid, block, lds, s);
    case 12: return launch_one<12, INV>(a, grid, block, lds, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_tile(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds,
                       hipStream_t s) {
  return inverse ? launch_dir<true>(logr, a, grid, block, lds, s) : launch_dir<false>(logr, a, grid, block, lds, s);
}

}  // namespace ronk
