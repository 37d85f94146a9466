// tile_launch.h -- launcher for the instantiated tile kernels (tile_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_tile.h"

namespace ronk {
hipError_t launch_tile(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds_bytes,
                       hipStream_t stream);
}
