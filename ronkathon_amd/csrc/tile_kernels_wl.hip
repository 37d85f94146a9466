// tile_kernels_wl.hip -- gfx950 instantiations of ntt_tile_wl.h: the 2^10 / 2^11 / 2^12-row x 4-column column / row passes of the two-pass
// plans with one wave-local and one cross-wave exchange per pass.  FULL image (8-byte cells, one barrier per pass, two
// workgroups per CU): the product's kernels for these shapes, Goldilocks and Montgomery primes.  Half image (4-byte cells,
// two 32-bit phases, built for 6 waves per SIMD): opt-in with RONK_WL_HALF=1, measured slower (ntt_tile_wl.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "ntt_tile_wl.h"
#include "tile_kernel_def.h"
#include "tile_launch.h"

namespace ronk {

#ifdef RONK_WL_NO_PREFETCH
#define RONK_WL_BARRIER __syncthreads()
#else
#define RONK_WL_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#define RONK_WL_PROLOGUE(LOGR, FULL)                                                           \
  __shared__ __attribute__((aligned(16))) u32 l32[wl_waves(LOGR) * WL_REGION * ((FULL) ? 2 : 1)]; \
  const u32 nb = gridDim.x, b = blockIdx.x;                                                    \
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;                                \
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;               \
  /* the exchanges need the workgroup's LDS traffic ordered, nothing else: __syncthreads() would also drain the global loads   \
     in flight (the matrix entries requested ahead of the barrier, ntt_tile_wl.h) */                                            \
  auto bar = [] { RONK_WL_BARRIER; };                                                          \
  auto wsync = [] {                                                                            \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                     \
    __builtin_amdgcn_wave_barrier();                                                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                     \
  };

// built for 4 waves per SIMD (FULL: 16 wavefronts per CU = 4 / 2 / 1 workgroups of 2^10 / 2^11 / 2^12 rows, what their LDS image
// allows anyway; up to 128 VGPRs) or 6 (half image)
template <int LOGR, bool INV, int KIND, bool FULL, class FLD>
__global__ void __launch_bounds__(wl_threads(LOGR), FULL ? 4 : 6) ntt_tile_wl_col_kernel(const TileArgs a) {
  RONK_WL_PROLOGUE(LOGR, FULL)
  tile_body_wl_col<LOGR, INV, KIND, FULL, FLD>(a, l32, threadIdx.x, bid, bar, wsync);
}
template <int LOGR, bool INV, bool FULL, class FLD>
__global__ void __launch_bounds__(wl_threads(LOGR), FULL ? 4 : 6) ntt_tile_wl_row_kernel(const TileArgs a) {
  RONK_WL_PROLOGUE(LOGR, FULL)
  tile_body_wl_row<LOGR, INV, FULL, FLD>(a, l32, threadIdx.x, bid, bar, wsync);
}

// experiments with the half image: RONK_WL_PAD = bytes of (unused) dynamic LDS per workgroup, to bound the workgroups per CU
static u32 wl_pad() { static const u32 v = [] { const char* e = getenv("RONK_WL_PAD"); return e ? (u32)atoi(e) : 0u; }(); return v; }

template <int LOGR, bool INV, bool FULL, class FLD>
static hipError_t launch_wl(int kind, const TileArgs& a, u32 grid, hipStream_t s) {
  const u32 pad = FULL ? 0u : wl_pad();
  switch (kind) {
    case 1: hipLaunchKernelGGL((ntt_tile_wl_col_kernel<LOGR, INV, 1, FULL, FLD>), dim3(grid), dim3(wl_threads(LOGR)), pad, s, a); break;
    case 3: hipLaunchKernelGGL((ntt_tile_wl_col_kernel<LOGR, INV, 3, FULL, FLD>), dim3(grid), dim3(wl_threads(LOGR)), pad, s, a); break;
    default: hipLaunchKernelGGL((ntt_tile_wl_row_kernel<LOGR, INV, FULL, FLD>), dim3(grid), dim3(wl_threads(LOGR)), pad, s, a); break;
  }
  return hipGetLastError();
}
template <int LOGR>
static hipError_t launch_wl_logr(bool inverse, int kind, bool half, const TileArgs& a, u32 grid, hipStream_t s) {
  if (a.fc.p) return inverse ? launch_wl<LOGR, true, true, MontField>(kind, a, grid, s) : launch_wl<LOGR, false, true, MontField>(kind, a, grid, s);
  if constexpr (LOGR == 11) {
    if (half) return inverse ? launch_wl<11, true, false, GlField>(kind, a, grid, s) : launch_wl<11, false, false, GlField>(kind, a, grid, s);
  }
  return inverse ? launch_wl<LOGR, true, true, GlField>(kind, a, grid, s) : launch_wl<LOGR, false, true, GlField>(kind, a, grid, s);
}

hipError_t launch_tile_wl(int logr, bool inverse, int kind, bool half, const TileArgs& a, u32 grid, hipStream_t s, bool* found) {
  // RONK_WL_ROWS: bit mask of the pass sizes served here (1 = 2^10 rows, 2 = 2^11, 4 = 2^12; default all) -- for A/B runs
  static const int rows = [] { const char* e = getenv("RONK_WL_ROWS"); return e ? atoi(e) : 7; }();
  *found = tile_wl_matches(a, logr, kind) && ((rows >> (logr - 10)) & 1);
  if (!*found) return hipSuccess;
  switch (logr) {
    case 10: return launch_wl_logr<10>(inverse, kind, half, a, grid, s);
    case 11: return launch_wl_logr<11>(inverse, kind, half, a, grid, s);
    default: return launch_wl_logr<12>(inverse, kind, half, a, grid, s);
  }
}

// RONK_WL: 0 = the ntt_tile.h kernels for these shapes (A/B), 1 (default) = both passes, 2 = column pass only, 3 = row pass only;
// RONK_WL_HALF=1: the half-image form (Goldilocks)
bool tile_wl_wanted(int kind, bool* half) {
  static const int mode = [] { const char* e = getenv("RONK_WL"); return e ? atoi(e) : 1; }();
  static const bool h = [] { const char* e = getenv("RONK_WL_HALF"); return e && atoi(e) != 0; }();
  *half = h;
  return mode == 1 || (mode == 2 && kind != 2) || (mode == 3 && kind == 2);
}

}  // namespace ronk
