// ntt_tile_w.h -- EXPERIMENT (round 4; measured and NOT adopted: profiles/r04_exp_wave_local.txt): the tile body with a
// WAVE-LOCAL first exchange, for 4-column tiles of 2^11 rows.  Parked here; tools/timeline.hip (wlat) still builds it.
//
// Why (profiles/r04_timeline_ntt22.txt): a workgroup's 16 loads per lane land between 0.6 and 7.5 us after the launch, and in
// ntt_tile.h every wavefront then waits at the barrier behind round 1 until the LAST-served wavefront has done its own round
// 1 -- 2-3 us per wavefront and pass, the largest parked block of the kernel -- although the exchange behind round 1 never
// mixes more than 16 lanes of one column.  With rows j = 128 j1 + 8 j2 + j3 (digits of the rounds 16, 16, 8):
//     round 1 (over j1)  lane (j2, j3)   -> outputs k1     exchange 1: (k1; j2, j3) -> lane (k1, j3): partners share j3
//     round 2 (over j2)  lane (k1, j3)   -> outputs k2     exchange 2: (k1, k2; j3) -> lane (k1, k2 pair): partners share k1
//     round 3 (over j3)  lane (k1, k2/2) -> outputs k3,  row k = k1 + 16 k2 + 256 k3
// ntt_tile.h numbers a column's 128 lanes m = 8 j2 + j3, so the 16 partners of exchange 1 sit in 8 different wavefronts.
// Here the lanes of a 4-column tile are numbered wavefront = j3, lane = (j2, column) for rounds 1-2 -- exchange 1 stays
// inside a wavefront: NO workgroup barrier, a wavefront goes from its loads through round 1 AND round 2 (two thirds of its
// arithmetic) before it meets anybody -- and wavefront = k2 pair, lane = (k1, column) for round 3.  Exchange 2 is done IN
// PLACE (round 2 writes the cells it read: no barrier between read and park either); output rows of the 16 lanes of a store
// instruction are then 16 consecutive k (k1 = the lane), so the stores are as coalesced as with ntt_tile.h's digit-swapped
// parking.  ONE workgroup barrier per tile instead of three.
//
// LDS image: 8 regions (digit 3) x 16 blocks (digit 1) x [16 (digit 2) x 4 columns + 4 pad cells]: cell = d3 * 1088 + d1 * 68
// + d2 * 4 + c.  Every LDS instruction's 64 lanes touch either 64 consecutive cells (the parks of round 1) or 16 runs of 4
// cells 68 cells apart (544 bytes = 32 mod 256: each 32-lane group covers all 64 banks once).  69 632 bytes, the size of
// ntt_tile.h's image.
//
// Same TileArgs contract, same results (every output bit-identical: the arithmetic per coefficient is unchanged).  Shapes:
// KIND 1 / 3 (column pass, two-level tables / full matrix) and KIND 2 (row pass) of the two-pass plans, full tiles, no
// features.  The body is plain C++ over (tid, bid, lds, barrier, wave_sync) so that tests/emu runs it on host fibers.
#pragma once
#include "../../ronkathon_amd/csrc/ntt_tile.h"

namespace ronk {

constexpr int TW_LOGR = 11, TW_LOGC = 2;
constexpr u32 TW_BLOCK = 68, TW_REGION = 16 * TW_BLOCK;   // cells
constexpr size_t TW_LDS_BYTES = (size_t)8 * TW_REGION * 8;

inline bool tile_w_matches(const TileArgs& a, int logr, int kind) {
  return logr == TW_LOGR && (kind == 1 || kind == 2 || kind == 3) && tile_cfg_matches(a, TW_LOGR, TW_LOGC, kind, 0);
}

// wave_sync: orders a wavefront's LDS writes before its own later reads.  On the device LDS operations of one wavefront are
// executed in order, so this is only a compiler fence; the host emulator (one fiber per lane) passes its barrier.
struct TileNoProbe { RONK_HD void operator()(int) const {} };
// probe(k): developer hook (tools/timeline.hip stamps the clock there): 0 after the loads are issued, 1 after round 1's park,
// 2 before the workgroup barrier, 3 after it, 4 after the last store is issued; a no-op in the product
template <bool INV, int KIND, class Barrier, class WaveSync, class Probe = TileNoProbe>
RONK_HD void tile_body_w(const TileArgs& a_in, u64* lds, u32 tid, u32 bid, Barrier&& barrier, WaveSync&& wave_sync,
                         Probe&& probe = TileNoProbe()) {
  constexpr int LOGR = TW_LOGR;
  typedef TileCfg<TW_LOGC, KIND> CFG;
  constexpr int R = 1 << LOGR, M = R / 16;
  const u32 c = tid & 3, l = (tid >> 2) & 15, w = tid >> 6;   // column, lane digit, wavefront
  // rounds 1-2: j3 = w, j2 (then k1) = l.  ntt_tile.h's lane index of the same coefficients: m = 8 j2 + j3
  const u32 m_old = l * 8 + w;
  const TileCtx cx = tile_ctx<LOGR, CFG>(a_in, m_old * 4 + c, bid);
  const TileArgs& a = cx.a;
  u64 x[16];
  tile_load<LOGR, INV, 0, CFG>(cx, lds, m_old * 4 + c, x, barrier);   // x[j1] = row 128 j1 + 8 j2 + j3
  probe(0);

  // ---- round 1 (over j1), table twiddle omega_R^{m_old k1}, park at (d3 = j3, d1 = k1, d2 = j2)
  Dif<16, INV, true>::run(x);
  u32 tb[16];
  tb[0] = 0; tb[1] = m_old << 3;
#pragma unroll
  for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
  const u32 reg = w * TW_REGION;
  {
    const u32 base = reg + l * 4 + c;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k1 = brev(i, 4);
      if (k1) x[i] = gl64::mul(x[i], ld_tabb(a.wr, tb[k1]));
      lds[base + k1 * TW_BLOCK] = x[i];
    }
  }
  wave_sync();
  probe(1);
  // ---- round 2 (over j2): lane (k1 = l, j3 = w), in place
  {
    const u32 base = reg + l * TW_BLOCK + c;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = lds[base + i * 4];
    Dif<16, INV, true>::run(x);
    tb[1] = w << 7;   // omega_{R/16}^{j3 k2} = omega_R^{16 j3 k2}
#pragma unroll
    for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k2 = brev(i, 4);
      if (k2) x[i] = gl64::mul(x[i], ld_tabb(a.wr, tb[k2]));
      lds[base + k2 * 4] = x[i];
    }
  }
  probe(2);
  barrier();   // the ONE workgroup barrier: exchange 2 crosses the wavefronts
  probe(3);
  // ---- round 3 (over j3): wavefront p = w owns k2 in {2p, 2p+1}, lane k1 = l; register g*8 + j3
  {
    const u32 base = l * TW_BLOCK + (2 * w) * 4 + c;
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int j = 0; j < 8; j++) x[g * 8 + j] = lds[base + j * TW_REGION + g * 4];
  }
  // natural output row of register (g, i): k = l + 16 (2 w + g) + 256 brev3(i); the 16 lanes of an instruction hold 16
  // consecutive rows
  constexpr int SH = 3;   // NARROW byte offsets (KIND != 0)
  const u32 kbase = l + 32 * w;
  const u32 out_sk = cx.out_sk, out_lane = cx.out_lane;
  u64* __restrict__ const outp = cx.out;
  if constexpr (KIND == 3) {
    const u32 tf_lane = (cx.col * a.tf_sc) << SH, tf_sk = a.tf_sk << SH;
    // all 16 matrix entries first (one group ahead is not enough here: both groups' loads are issued before any store, so no
    // twiddle load queues behind a store on the in-order vmcnt counter)
    u64 wq[16];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int i = 0; i < 8; i++)
        wq[g * 8 + i] = ld_g<true>(a.tw_full, tf_lane + (kbase + 16 * g + 256 * brev(i, 3)) * tf_sk);
#pragma unroll
    for (int g = 0; g < 2; g++) {
      u64* xg = x + g * 8;
      Dif<8, INV, true>::run(xg, false);
#pragma unroll
      for (int i = 0; i < 8; i++) xg[i] = gl64::mul(xg[i], wq[g * 8 + i]);
#pragma unroll
      for (int i = 0; i < 8; i++) st_g<true>(outp, out_lane + (kbase + 16 * g + 256 * brev(i, 3)) * out_sk, xg[i]);
    }
  } else if constexpr (KIND == 1) {
    // two-level inter-pass twiddle omega_N^{col * k}: exponents pre-scaled by 8 (byte offsets), add chain over i
    const u32 nmask = a.tw_log >= 32 ? 0xFFFFFFFFu : ((1u << a.tw_log) - 1);
    const u32 lmask8 = ((1u << a.tw_lo_bits) - 1) << 3, hmask8 = (nmask >> a.tw_lo_bits) << 3;
    const u32 twX = cx.col;
#pragma unroll
    for (int g = 0; g < 2; g++) {
      u64* xg = x + g * 8;
      Dif<8, INV, true>::run(xg, false);
      u32 ej[8];
      ej[0] = (twX * (kbase + 16 * g)) << 3;
      const u32 estep = (twX * 256u) << 3;
#pragma unroll
      for (int j = 1; j < 8; j++) ej[j] = ej[j - 1] + estep;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const u32 ee = ej[brev(i, 3)];
        const u64 tw = gl64::mul(ld_tabb(a.tw_lo, ee & lmask8), ld_tabb(a.tw_hi, (ee >> a.tw_lo_bits) & hmask8));
        xg[i] = gl64::mul(xg[i], tw);
      }
#pragma unroll
      for (int i = 0; i < 8; i++) st_g<true>(outp, out_lane + (kbase + 16 * g + 256 * brev(i, 3)) * out_sk, xg[i]);
    }
  } else {
#pragma unroll
    for (int g = 0; g < 2; g++) {
      u64* xg = x + g * 8;
      Dif<8, INV, false>::run(xg, false);
#pragma unroll
      for (int i = 0; i < 8; i++) st_g<true>(outp, out_lane + (kbase + 16 * g + 256 * brev(i, 3)) * out_sk, xg[i]);
    }
  }
  probe(4);
  (void)M;
}

}  // namespace ronk
