// tile_kernels_wl.hip -- gfx950 instantiations of ntt_tile_wl.h: the 2^11-row x 4-column column / row passes of the two-pass
// plans with a wave-local exchange, half the LDS image and two workgroup barriers per pass (3-4 resident workgroups per CU).
// Built for 8 waves per SIMD (<= 64 VGPRs, four workgroups per CU) and for 6 (<= 80 VGPRs, three per CU); the launcher
// (tile_kernels.hip) picks by RONK_WL / RONK_WL_WPE.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "ntt_tile_wl.h"
#include "tile_kernel_def.h"
#include "tile_launch.h"

namespace ronk {

#define RONK_WL_PROLOGUE                                                                       \
  __shared__ __attribute__((aligned(16))) u32 l32[8 * WL_REGION];                              \
  const u32 nb = gridDim.x, b = blockIdx.x;                                                    \
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;                                \
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;               \
  auto bar = [] { __syncthreads(); };                                                          \
  auto wsync = [] {                                                                            \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                     \
    __builtin_amdgcn_wave_barrier();                                                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                     \
  };

template <bool INV, int KIND, int WPE, int MF>
__global__ void __launch_bounds__(WL_THREADS, WPE) ntt_tile_wl_col_kernel(const TileArgs a) {
  RONK_WL_PROLOGUE
  tile_body_wl_col<INV, KIND, MF>(a, l32, threadIdx.x, bid, bar, wsync);
  tile_prefetch_tail(a, bid);
}
template <bool INV, int WPE, int MF>
__global__ void __launch_bounds__(WL_THREADS, WPE) ntt_tile_wl_row_kernel(const TileArgs a) {
  RONK_WL_PROLOGUE
  tile_body_wl_row<INV, MF>(a, l32, threadIdx.x, bid, bar, wsync);
  tile_prefetch_tail(a, bid);
}

// experiments: RONK_WL_PAD = bytes of (unused) dynamic LDS per workgroup, to bound the workgroups per CU from above;
// RONK_WL_MF_COL / RONK_WL_MF_ROW = memory-policy flags (ntt_tile_wl.h WL_NT_*; forward matrix column pass / forward row pass
// of the 6-waves-per-SIMD build only)
static u32 wl_pad() { static const u32 v = [] { const char* e = getenv("RONK_WL_PAD"); return e ? (u32)atoi(e) : 0u; }(); return v; }
static int wl_mf(bool row) {
  static const int c = [] { const char* e = getenv("RONK_WL_MF_COL"); return e ? atoi(e) : 0; }();
  static const int r = [] { const char* e = getenv("RONK_WL_MF_ROW"); return e ? atoi(e) : 0; }();
  return row ? r : c;
}

template <bool INV, int WPE>
static hipError_t launch_wl(int kind, const TileArgs& a, u32 grid, hipStream_t s) {
  const u32 pad = wl_pad();
  if constexpr (!INV && WPE == 6) {
    const int mf = wl_mf(kind == 2);
#define RONK_WL_MF_COL_CASE(M) if (kind == 3 && mf == M) { hipLaunchKernelGGL((ntt_tile_wl_col_kernel<false, 3, 6, M>), dim3(grid), dim3(WL_THREADS), pad, s, a); return hipGetLastError(); }
#define RONK_WL_MF_ROW_CASE(M) if (kind == 2 && mf == M) { hipLaunchKernelGGL((ntt_tile_wl_row_kernel<false, 6, M>), dim3(grid), dim3(WL_THREADS), pad, s, a); return hipGetLastError(); }
    RONK_WL_MF_COL_CASE(1) RONK_WL_MF_COL_CASE(2) RONK_WL_MF_COL_CASE(3)
    RONK_WL_MF_ROW_CASE(2) RONK_WL_MF_ROW_CASE(4) RONK_WL_MF_ROW_CASE(6)
  }
  switch (kind) {
    case 1: hipLaunchKernelGGL((ntt_tile_wl_col_kernel<INV, 1, WPE, 0>), dim3(grid), dim3(WL_THREADS), pad, s, a); break;
    case 3: hipLaunchKernelGGL((ntt_tile_wl_col_kernel<INV, 3, WPE, 0>), dim3(grid), dim3(WL_THREADS), pad, s, a); break;
    default: hipLaunchKernelGGL((ntt_tile_wl_row_kernel<INV, WPE, 0>), dim3(grid), dim3(WL_THREADS), pad, s, a); break;
  }
  return hipGetLastError();
}

hipError_t launch_tile_wl(int logr, bool inverse, int kind, int wpe, const TileArgs& a, u32 grid, hipStream_t s, bool* found) {
  *found = tile_wl_matches(a, logr, kind);
  if (!*found) return hipSuccess;
  if (wpe >= 8) return inverse ? launch_wl<true, 8>(kind, a, grid, s) : launch_wl<false, 8>(kind, a, grid, s);
  return inverse ? launch_wl<true, 6>(kind, a, grid, s) : launch_wl<false, 6>(kind, a, grid, s);
}

}  // namespace ronk
