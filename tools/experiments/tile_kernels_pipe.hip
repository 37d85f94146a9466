// tools/experiments/tile_kernels_pipe.hip -- EXPERIMENT, not part of the product build (round 3).
//
// The specialised tile kernels (TileCfg, ntt_tile.h) as PERSISTENT, software-pipelined kernels: a workgroup walks over
// tiles and issues the 16 global loads per lane of its next tile before the arithmetic of the current one (register
// prefetch, LDS-only barriers), using the tile_ctx / tile_load / tile_compute split of ntt_tile.h.
// Parity-green on MI355X (25 GPU tests with RONK_PIPE=1), but SLOWER than one workgroup per tile everywhere it was tried
// (same box, HBM-cold protocol, gpurun_out/r03b -> profiles/r03_exp_pipe.txt):
//   2^22 x 16, 8192-coefficient tiles   19.7 k -> 17.1 k NTT/s        2^22 x 16, 16384-coefficient tiles  21.2 k -> 18.7 k
//   1024 x 2^16                         0.489 -> 0.550 ms             polynomial multiply 2^22            156.5 -> 159.0 us
// Why: gfx950 has ONE in-order counter (vmcnt) for vector loads AND stores.  The table-twiddle loads inside a tile's
// arithmetic can only be waited for together with everything issued before them -- the prefetch of the next tile and the
// stores of the previous one -- so the first twiddle multiplication of every tile stalls on exactly the traffic the
// pipeline was built to hide, and the kernels need 100-128 VGPRs instead of 63-67.  With one workgroup per tile the
// hardware overlaps a finishing workgroup's stores and a starting workgroup's loads with the arithmetic of its
// co-resident workgroups for free.  (Round 2 reached the same verdict with the generic body.)
// To rebuild the experiment: add this file to the Makefile's OBJS and call launch_tile_cfg_pipe from launch_tile.
#include "../../ronkathon_amd/csrc/tile_cfg_table.h"
#include "../../ronkathon_amd/csrc/tile_kernel_def.h"

namespace ronk {

// ---- persistent, software-pipelined form of the specialised kernels (TileCfg KIND 1 / 2 / 3) for passes with several
// tiles per workgroup slot (batches): a workgroup walks over tiles v = blockIdx.x, + gridDim.x, ... and issues the 16
// global loads per lane of its NEXT tile before it starts the arithmetic of the current one, so a CU's VALU never waits
// for HBM between tiles; the stores of a tile drain under the next tile's arithmetic.  Costs 32 VGPRs (the prefetched
// coefficients; the kernels stay within the 128 that four waves per SIMD allow).  The barriers between the register
// rounds wait for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would drain the prefetch too.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int LOGR, bool INV, int LOGC, int KIND>
__global__ void __launch_bounds__(1024) ntt_tile_kernel_pipe(const TileArgs a, const u32 total) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  using CFG = TileCfg<LOGC, KIND, false, false>;
  // virtual block v -> tile number, as in tile_kernel_main with a grid of `total` blocks (gridDim.x is a multiple of 8, so
  // every tile a workgroup visits has v % 8 == blockIdx.x % 8: one XCD works on one contiguous run of tiles)
  const u32 q = total >> 3, r = total & 7;
  auto tile_of = [&](u32 v) {
    const u32 xcd = v & 7, idx = v >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  };
  const u32 G = gridDim.x;
  u32 v = blockIdx.x;
  u64 xa[16];
  TileCtx cx = tile_ctx<LOGR, CFG>(a, threadIdx.x, tile_of(v));
  tile_load<LOGR, INV, 0, CFG>(cx, lds, threadIdx.x, xa, [] { lds_only_barrier(); });
  // the first tile's coefficients are waited for HERE: inside the loop `xa` is then plain register data on every path, and
  // the compiler's wait-count bookkeeping never has to wait for the loop's prefetch loads on behalf of these
#pragma unroll
  for (int i = 0; i < 16; i++) asm volatile("" : "+v"(xa[i]));
  for (;;) {
    const u32 vn = v + G;
    const bool more = vn < total;                      // wave-uniform
    u32 tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                      // opaque per iteration: the per-lane address arithmetic of a tile is
                                                       // not hoisted out of the loop (it would be kept live in ~300 VGPRs)
    u64 xb[16];
    TileCtx cn;
    if (more) {
      cn = tile_ctx<LOGR, CFG>(a, tid, tile_of(vn));
      tile_load<LOGR, INV, 0, CFG>(cn, lds, tid, xb, [] { lds_only_barrier(); });
    }
    tile_compute<LOGR, INV, 0, CFG>(cx, lds, tid, xa, [] { lds_only_barrier(); });
    if (!more) break;
    lds_only_barrier();                                // every lane has read its last-round rows: the image may be reused
#pragma unroll
    for (int i = 0; i < 16; i++) xa[i] = xb[i];
    cx = cn;
    v = vn;
  }
}

template <int LOGR, bool INV, int LOGC, int KIND>
static hipError_t launch_one_pipe(const TileArgs& a, u32 total, u32 block, size_t lds, hipStream_t s) {
  static int wg_per_cu[64] = {};   // per device: resident workgroups per CU of this kernel (0 = not asked yet)
  static int cus[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const void* fn = (const void*)ntt_tile_kernel_pipe<LOGR, INV, LOGC, KIND>;
  int occ = (dev >= 0 && dev < 64) ? wg_per_cu[dev] : 0, ncu = (dev >= 0 && dev < 64) ? cus[dev] : 0;
  if (!occ) {
    if (lds > 48 * 1024) {
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, (int)block, lds);
    if (e != hipSuccess) return e;
    if (occ < 1) occ = 1;
    e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) { wg_per_cu[dev] = occ; cus[dev] = ncu; }
  }
  u32 grid = (u32)occ * (u32)ncu;
  grid &= ~7u;                                         // a multiple of 8 (XCD affinity of the walk)
  if (grid < 8) grid = 8;
  if (grid > total) grid = total;                      // (then every workgroup has one tile and nothing to prefetch)
  hipLaunchKernelGGL((ntt_tile_kernel_pipe<LOGR, INV, LOGC, KIND>), dim3(grid), dim3(block), lds, s, a, total);
  return hipGetLastError();
}


#define RONK_CFG_CASE_PIPE(LR, LC, KD)                                                                    \
  if (logr == LR && (int)a.logc == LC && kind == KD) {                                                    \
    *found = true;                                                                                        \
    return inverse ? launch_one_pipe<LR, true, LC, KD>(a, total, block, lds, s)                           \
                   : launch_one_pipe<LR, false, LC, KD>(a, total, block, lds, s);                         \
  }

hipError_t launch_tile_cfg_pipe(int logr, bool inverse, int kind, const TileArgs& a, u32 total, u32 block, size_t lds,
                                hipStream_t s, bool* found) {
  RONK_CFG_TABLE(RONK_CFG_CASE_PIPE)
  *found = false;
  return hipSuccess;
}

}  // namespace ronk
