// tile_kernels.hip -- gfx950 instantiations of the generic NTT tile kernel (ntt_tile.h) and the launcher.
//
// One __global__ per (LOGR, direction) that reads every stride and flag from TileArgs; passes whose shape
// tile_cfg_matches() recognises go to the specialised kernels of tile_kernels_cfg.hip instead.
#include "tile_kernel_def.h"
#include "tile_launch.h"

namespace ronk {

template <bool INV>
static hipError_t launch_dir(int logr, const TileArgs& a, u32 grid, u32 block, size_t lds, hipStream_t s) {
  switch (logr) {
    case 4: return launch_one<4, INV, -1, 0>(a, grid, block, lds, s);
    case 5: return launch_one<5, INV, -1, 0>(a, grid, block, lds, s);
    case 6: return launch_one<6, INV, -1, 0>(a, grid, block, lds, s);
    case 7: return launch_one<7, INV, -1, 0>(a, grid, block, lds, s);
    case 8: return launch_one<8, INV, -1, 0>(a, grid, block, lds, s);
    case 9: return launch_one<9, INV, -1, 0>(a, grid, block, lds, s);
    case 10: return launch_one<10, INV, -1, 0>(a, grid, block, lds, s);
    case 11: return launch_one<11, INV, -1, 0>(a, grid, block, lds, s);
    case 12: return launch_one<12, INV, -1, 0>(a, grid, block, lds, s);
    default: return hipErrorInvalidValue;
  }
}

// Two-phase 32-bit LDS exchanges (TileCfg::HALF, half the LDS image, kernels built for 6-8 waves per SIMD) pay when a
// pass has far more tiles than the chip holds at once: the workgroups of a CU drift into different phases and the
// additional resident ones fill the load / store phases of the others.  Measured (same box): 1024 x 2^16
// 0.493 -> 0.469 ms, 512 x 2^17 0.561 -> 0.514 ms; row passes with >= 2^10 rows get slower (2^22 x 16 pass 2: 402 -> 484 us)
// and a single transform (one tile per CU) only pays the extra barriers (51.4 -> 53.7 us), so:
//   default  column passes (KIND 1, 3) of any size and row passes (KIND 2) up to 2^9 rows, when the grid has at least
//            twice the threads the chip holds at four waves per SIMD (2 * 256 CUs * 1024)
//   RONK_HALF_LDS = 0 never, 1 always, 2 row passes only (experiments)
//   round 3 (planner: big batches keep 16384-coefficient tiles for 2^11-row passes; HBM-cold sweep, profiles/r03_half_rule_sweep.txt):
//            a 2^11-row x 8-column row pass owns a whole CU (136 KiB) -- there the half image pays as well (2^22 x 16: 21.5 k ->
//            22.5 k NTT/s); 2^10-row row passes stay on the full image (2^21 x 32: 49.9 k with the rule, 48.4 k all-half)
static bool use_half(const TileArgs& a, int logr, u32 grid, u32 block, int kind) {
  static const int mode = [] { const char* e = getenv("RONK_HALF_LDS"); return e ? atoi(e) : -1; }();
  if (mode >= 0) return mode == 1 || (mode == 2 && kind == 2);
  if ((unsigned long long)grid * block < 2ull * 256 * 1024) return false;
  return kind != 2 || logr <= 9 || (logr == 11 && a.logc == 3);
}

hipError_t launch_tile(int logr, bool inverse, const TileArgs& a, u32 grid, u32 block, size_t lds,
                       hipStream_t s) {
  if (a.fc.p) return launch_tile_mont(logr, inverse, a, grid, block, lds, s);   // a Montgomery prime (field_policy.h)
  static const bool no_cfg = getenv("RONK_NO_CFG_KERNELS") != nullptr;   // experiments: force the generic kernels
  if (!no_cfg) {
    const int feat = tile_features(a);
    for (int kind : {1, 2, 3, 4, 5}) {
      if (!tile_cfg_matches(a, logr, (int)a.logc, kind, feat)) continue;
      bool found = false;
      if (feat) {
        hipError_t e = launch_tile_cfg_feat(logr, inverse, kind, feat, a, grid, block, lds, s, &found);
        if (found) return e;
        continue;
      }
      // 2^10 / 2^11 / 2^12-row x 4-column passes (the two-lane plans of 2^20 .. 2^22, one transform of 2^20 / 2^21 / 2^23): one
      // wave-local and one cross-wave exchange, one barrier per pass (ntt_tile_wl.h).  Round 6, same box: two lanes at 2^22
      // 22.4 k -> 23.6 k NTT/s, one stream 58.5 -> 55.5 us.
      static const bool r4_on = [] { const char* e_ = getenv("RONK_R4MID"); return e_ && atoi(e_) != 0; }();   // opt-in, below
      bool wl_half = false;
      if (kind < 4 && logr >= 10 && logr <= 12 && a.logc == 2 && !(r4_on && logr == 10) && tile_wl_wanted(kind, &wl_half)) {
        hipError_t e = launch_tile_wl(logr, inverse, kind, wl_half, a, grid, s, &found);
        if (found) return e;
      }
      if (kind < 4 && use_half(a, logr, grid, block, kind)) {
        hipError_t e = launch_tile_cfg_half(logr, inverse, kind, a, grid, block, lds, s, &found);
        if (found) return e;
      }
      // 2^9 / 2^10-row passes: the [16 . 4] . [8 | 16] round structure (tile_kernels_r4.hip), OPT-IN with RONK_R4MID=1.
      // Measured (round 5, same box, A/B/A/B: profiles/r05_r4_ab.txt): it executes 5.5 % fewer VALU instructions per pass
      // (one table-twiddle layer traded for a wave-uniform shift layer) and is not faster anywhere -- one 2^20 transform
      // 27.7 -> 28.1 us, two lanes 17.05 -> 17.3 us per transform, 64 x 2^20 / 256 x 2^18 / 2^24 .. 2^26 within +-1 % -- so
      // the (16, 16, 2 | 4) kernels stay the default.
      if (r4_on && kind < 4 && (logr == 9 || logr == 10)) {
        hipError_t e = launch_tile_r4(logr, inverse, kind, a, grid, block, lds, s, &found);
        if (found) return e;
      }
      hipError_t e = launch_tile_cfg(logr, inverse, kind, a, grid, block, lds, s, &found);
      if (found) return e;
    }
  }
  return inverse ? launch_dir<true>(logr, a, grid, block, lds, s) : launch_dir<false>(logr, a, grid, block, lds, s);
}

}  // namespace ronk
