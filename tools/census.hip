// tools/census.hip -- developer tool: compile ntt_tile.h's tile body with the plan flags of the two passes of the
// single 2^22 transform (and of config 4) fixed at compile time, so that the static instruction count of the .s file
// IS the executed path (no dead variants).  Never linked into the product; see tools/census.py.
#include <hip/hip_runtime.h>
#include "../ronkathon_amd/csrc/ntt_tile.h"
using namespace ronk;

template <int LOGR, bool INV, int MODE>
__global__ void __launch_bounds__(1024) census_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  a.stage_io = 0; a.in2 = nullptr; a.in_valid = ~(u64)0; a.out_valid = ~(u64)0; a.scale = 1; a.tw_full = nullptr;
  a.nb1 = 1; a.nb2 = 1; a.ncols = 1u << 30;
  if (MODE == 0) {          // pass 1 of a two-pass plan: flat rows, two-level inter-pass twiddle
    a.js_log = 31; a.xb1 = a.xb2 = a.x0 = 0; a.yb1 = a.yb2 = a.y0 = 0; a.xc = 1; a.yk = 1; a.in_sc = 1; a.out_sc = 1;
    a.logc = 3; a.tw_log = 22; a.tw_lo_bits = 11;
  } else {                  // pass 2: blocked rows (tiled scratch), no twiddle
    a.tw_log = 0; a.js_log = 3; a.in_sj = 1; a.out_sc = 1; a.logc = 3;
  }
  tile_body<LOGR, INV, 0>(a, lds, threadIdx.x, blockIdx.x, [] { __syncthreads(); });
}
template __global__ void census_kernel<11, false, 0>(TileArgs);
template __global__ void census_kernel<11, false, 1>(TileArgs);
template __global__ void census_kernel<8, false, 0>(TileArgs);
template __global__ void census_kernel<8, false, 1>(TileArgs);
