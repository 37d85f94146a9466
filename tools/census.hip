// tools/census.hip -- developer tool: compile ntt_tile.h's tile body with the plan flags of the two passes of the
// single 2^22 transform (and of config 4) fixed at compile time, so that the static instruction count of the .s file
// IS the executed path (no dead variants).  Never linked into the product; see tools/census.py.
#include <hip/hip_runtime.h>
#include "../ronkathon_amd/csrc/ntt_tile.h"
using namespace ronk;

template <int LOGR, bool INV, int MODE>
__global__ void __launch_bounds__(1024) census_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  // MODE 0: column pass (KIND 1), MODE 1: row pass (KIND 2); tile width fixed like the library's hot instantiations
  constexpr int LOGC = LOGR >= 11 ? 3 : 4;
  tile_body<LOGR, INV, 0, TileCfg<LOGC, MODE == 0 ? 1 : 2, cfg_ldstw(LOGR, LOGC, MODE == 0 ? 1 : 2)>>(a, lds, threadIdx.x, blockIdx.x, [] { __syncthreads(); });
}
template __global__ void census_kernel<11, false, 0>(TileArgs);
template __global__ void census_kernel<11, false, 1>(TileArgs);
template __global__ void census_kernel<8, false, 0>(TileArgs);
template __global__ void census_kernel<8, false, 1>(TileArgs);
