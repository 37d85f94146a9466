// tile_kernel_def.h -- the __global__ wrapper of ntt_tile.h's tile body and its launch helper, shared by the two
// translation units that instantiate it (tile_kernels.hip: the generic kernels; tile_kernels_cfg.hip: the kernels
// that know the shape of a two-pass plan's passes at compile time, TileCfg in ntt_tile.h).
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_tile.h"

namespace ronk {

// Workgroup = 2^LOGR * C / 16 work-items (<= 1024), dynamic LDS = (2^LOGR + 2^LOGR/16) * C * 8 bytes (<= 136 KiB of the
// CU's 160 KiB).
template <int LOGR, bool INV, int LOGC, int KIND, bool HALF, int FEAT = 0, class FLD = GlField>
__device__ __forceinline__ void tile_kernel_main(const TileArgs& a, u64* lds) {
  // The dispatcher hands workgroup b to XCD b % 8 (observed, for speed only): renumber so that
  // each XCD works on a contiguous run of tiles -- neighbouring tiles share 128-byte lines and
  // twiddle rows, which then hit in that XCD's private L2.  Bijective for any grid size.
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  tile_body<LOGR, INV, 0, TileCfg<LOGC, KIND, !HALF && !FEAT && !FLD::MONT && cfg_ldstw(LOGR, LOGC, KIND), HALF, FEAT>, FLD>(a, lds, threadIdx.x, bid, [] { __syncthreads(); });
}

// the shapes with features (tile_cfg_table.h RONK_CFG_TABLE_FEAT; tile_kernels_feat.hip)
template <int LOGR, bool INV, int LOGC, int KIND, int FEAT>
__global__ void __launch_bounds__(1024) ntt_tile_kernel_feat(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  tile_kernel_main<LOGR, INV, LOGC, KIND, false, FEAT>(a, lds);
}
template <int LOGR, bool INV, int LOGC, int KIND, int FEAT>
static hipError_t launch_one_feat(const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  static bool attr_done[64] = {};
  if (lds > 48 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      e = hipFuncSetAttribute((const void*)ntt_tile_kernel_feat<LOGR, INV, LOGC, KIND, FEAT>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ntt_tile_kernel_feat<LOGR, INV, LOGC, KIND, FEAT>), dim3(grid), dim3(block), lds, s, a);
  return hipGetLastError();
}

template <int LOGR, bool INV, int LOGC, int KIND>
__global__ void __launch_bounds__(1024) ntt_tile_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  tile_kernel_main<LOGR, INV, LOGC, KIND, false>(a, lds);
}

// TileCfg::HALF (two-phase 32-bit LDS exchanges, half the image): built for 8 resident waves per SIMD (<= 64 VGPRs), which
// is the point of halving the image
// waves per SIMD a HALF kernel is built for: 8 (64 VGPRs), except the two-round column pass with the full twiddle matrix
// (16 table entries + 16 coefficients live at the end), which spills 70-80 bytes per lane at 64 and gets 6 (80 VGPRs)
constexpr int half_wpe(int logr, int kind) { return (logr == 8 && kind == 3) ? 6 : 8; }

template <int LOGR, bool INV, int LOGC, int KIND>
__global__ void __launch_bounds__(1024, half_wpe(LOGR, KIND)) ntt_tile_kernel_half(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  tile_kernel_main<LOGR, INV, LOGC, KIND, true>(a, lds);
}

template <int LOGR, bool INV, int LOGC, int KIND, bool HALF = false>
static hipError_t launch_one(const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  // HIP keeps this attribute per (kernel, DEVICE): one flag per device ordinal (benign race: the call is idempotent)
  static bool attr_done[64] = {};
  if (HALF) lds /= 2;   // the image holds 4-byte cells (TileCfg::HALF)
  if (lds > 48 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      const void* fn;
      if constexpr (HALF) fn = (const void*)ntt_tile_kernel_half<LOGR, INV, LOGC, KIND>;
      else fn = (const void*)ntt_tile_kernel<LOGR, INV, LOGC, KIND>;
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  if (!HALF && cfg_ldstw(LOGR, LOGC, KIND)) lds += (size_t)8 << LOGR;   // the staged round-twiddle table behind the image
  if constexpr (HALF) hipLaunchKernelGGL((ntt_tile_kernel_half<LOGR, INV, LOGC, KIND>), dim3(grid), dim3(block), lds, s, a);
  else hipLaunchKernelGGL((ntt_tile_kernel<LOGR, INV, LOGC, KIND>), dim3(grid), dim3(block), lds, s, a);
  return hipGetLastError();
}

// tile_kernels_r4.hip: the 2^9 / 2^10-row shapes of RONK_CFG_TABLE with the [16 . 4] . [8 | 16] round structure (TileCfg::R4:
// one table-twiddle layer and one wave-uniform shift layer per pass instead of two table layers); *found = there is one
hipError_t launch_tile_r4(int logr, bool inverse, int kind, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s,
                          bool* found);
// tile_kernels_mont.hip: the same bodies over a Montgomery prime (field_policy.h MontField; TileArgs::fc.p != 0) -- the generic
// kernel for every pass size and the specialised shapes of RONK_CFG_TABLE; launch_small_mont: the latency form
hipError_t launch_tile_mont(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s);
hipError_t launch_small_mont(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s);
// tile_kernels_mont_feat.hip: the shapes with features over a Montgomery prime; *found says whether there is one
hipError_t launch_tile_mont_feat(int logr, bool inverse, int feat, const TileArgs& a, u32 grid, u32 block, size_t lds,
                                 hipStream_t s, bool* found);
// tile_kernels_cfg.hip: launches the specialised instantiation for (logr, a.logc, kind) if there is one; *found says so
hipError_t launch_tile_cfg(int logr, bool inverse, int kind, const TileArgs& a, u32 grid, u32 block, size_t lds,
                           hipStream_t s, bool* found);
// tile_kernels_feat.hip: the shapes with features (zero-padded input, fused second operand, truncated output)
hipError_t launch_tile_cfg_feat(int logr, bool inverse, int kind, int feat, const TileArgs& a, u32 grid, u32 block, size_t lds,
                                hipStream_t s, bool* found);
// tile_kernels_half.hip: the same shapes with two-phase 32-bit LDS exchanges (TileCfg::HALF; `lds` is the full-size
// image, the launcher halves it)
hipError_t launch_tile_cfg_half(int logr, bool inverse, int kind, const TileArgs& a, u32 grid, u32 block, size_t lds,
                                hipStream_t s, bool* found);

}  // namespace ronk
