// tile_launch.h -- launcher for the instantiated tile kernels (tile_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_tile.h"

namespace ronk {
hipError_t launch_tile(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds_bytes,
                       hipStream_t stream);
// ntt_tile_wl.h (tile_kernels_wl.hip): 2^11-row x 4-column passes with one wave-local and one cross-wave exchange (one barrier per
// pass); Goldilocks and Montgomery primes (a.fc).  half = the half-image form (experiments).  *found = the pass has that shape.
hipError_t launch_tile_wl(int logr, bool inverse, int kind, bool half, const TileArgs& a, u32 grid, hipStream_t stream, bool* found);
bool tile_wl_wanted(int kind, bool* half);   // RONK_WL / RONK_WL_HALF
// the latency form of a pass (ntt_small.h: 4 coefficients per work-item), 2^4 .. 2^10 rows
hipError_t launch_small(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds_bytes,
                        hipStream_t stream);
// the fused middle of a polynomial multiply (ntt_mul.h, tile_kernels_mul.hip): fa = the forward plan's row pass over the batch
// of two operands, ia = the inverse plan's column pass; grid = tiles of ONE operand; *found = an instantiation exists
// is there an instantiation for (rows, log2 tile columns, inverse twiddle form)?  Asked BEFORE the forward column pass is enqueued
bool mul_mid_available(int logr, int logc, int kindi);
hipError_t launch_mul_mid(int logr, int kindi, const TileArgs& fa, const TileArgs& ia, u32 grid, u32 block, size_t lds_bytes,
                          hipStream_t stream, bool* found);
// the same over a Montgomery prime (tile_kernels_mont_mul.hip; fa.fc / ia.fc carry it)
bool mul_mid_available_mont(int logr, int logc, int kindi);
hipError_t launch_mul_mid_mont(int logr, int kindi, const TileArgs& fa, const TileArgs& ia, u32 grid, u32 block, size_t lds_bytes,
                               hipStream_t stream, bool* found);
}
