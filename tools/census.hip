// tools/census.hip -- developer tool: compile ntt_tile.h's tile body with the plan flags of the two passes of the
// single 2^22 transform (and of config 4) fixed at compile time, so that the static instruction count of the .s file
// IS the executed path (no dead variants).  Never linked into the product; see tools/census.py.
// MODE 0: column pass (KIND 1), 1: row pass (KIND 2), 3: column pass with the full twiddle matrix (KIND 3).
// ABL: ntt_tile.h's ablation mask (1 no inter-pass twiddle, 2 no round twiddles, 4 no butterflies, 64 twiddle values
// without table loads) -- the component breakdown of DESIGN.md 5.1 comes from differences between these instantiations.
// -DRONK_CENSUS_ALL_LAZY: every canonicalising add replaced by the 4-instruction lazy form (WRONG results; an upper bound
// on what deferred canonicalisation could ever save).
#include <hip/hip_runtime.h>
#ifdef RONK_CENSUS_ALL_LAZY
#define RONK_GL64_ALL_LAZY 1
#endif
#include "../ronkathon_amd/csrc/ntt_tile.h"
using namespace ronk;

template <int LOGR, bool INV, int MODE, int ABL = 0>
__global__ void __launch_bounds__(1024) census_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  constexpr int LOGC = LOGR >= 11 ? 3 : 4;
  constexpr int KIND = MODE == 0 ? 1 : MODE == 1 ? 2 : 3;
  tile_body<LOGR, INV, ABL, TileCfg<LOGC, KIND, cfg_ldstw(LOGR, LOGC, KIND)>>(a, lds, threadIdx.x, blockIdx.x, [] { __syncthreads(); });
}
template __global__ void census_kernel<11, false, 0>(TileArgs);
template __global__ void census_kernel<11, false, 3>(TileArgs);        // full inter-pass twiddle matrix: what 2^21 / 2^22 plans run
template __global__ void census_kernel<11, false, 1>(TileArgs);
template __global__ void census_kernel<8, false, 0>(TileArgs);
template __global__ void census_kernel<8, false, 3>(TileArgs);         // ... and the 1024 x 2^16 plan
template __global__ void census_kernel<8, false, 1>(TileArgs);
#ifdef RONK_CENSUS_BREAKDOWN
template __global__ void census_kernel<11, false, 0, 1>(TileArgs);     // pass 1 without the inter-pass twiddle
template __global__ void census_kernel<11, false, 0, 2>(TileArgs);     // ... without the two round-twiddle layers
template __global__ void census_kernel<11, false, 1, 2>(TileArgs);
template __global__ void census_kernel<11, false, 0, 4>(TileArgs);     // ... without the butterflies (adds, subs, shift twiddles)
template __global__ void census_kernel<11, false, 1, 4>(TileArgs);
template __global__ void census_kernel<11, false, 0, 7>(TileArgs);     // none of the three: addressing, loads, LDS exchange, stores
template __global__ void census_kernel<11, false, 1, 7>(TileArgs);
#endif
