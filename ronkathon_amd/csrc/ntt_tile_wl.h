// ntt_tile_wl.h -- the 2^10 / 2^11 / 2^12-row x 4-column tile passes of the two-pass plans (2^20 .. 2^24: the headline 2^22 among
// them) with ONE wave-local and ONE cross-wave LDS exchange per pass: one workgroup barrier per pass (FULL image) instead of three.
//
// Written out for 2^11 rows (RL = 8; 2^10: RL = 4, 2^12: RL = 16 -- the radix of the last round = the wavefronts of a workgroup):
// with rows j = 128 j1 + 8 j2 + j3 (digits of the rounds 16, 16, 8) the 16 partners of an exchange always share a column and
// one more digit; ntt_tile.h numbers a column's lanes m = 8 j2 + j3 for every round, so both exchanges cross the wavefronts
// (write / barrier / read / barrier, twice).  Here
//   column pass (tile_body_wl_col): wavefront = j3 for rounds 1-2, so the exchange behind round 1 stays inside a wavefront (no
//     barrier: LDS operations of one wavefront execute in order) and is done in the wavefront's own region of the image; the
//     exchange behind round 2 crosses the wavefronts IN PLACE (own region written, the other regions read); the last round's 16
//     lanes per column hold 16 CONSECUTIVE output rows -- whole 512-byte runs of the tiled scratch per store instruction;
//   row pass (tile_body_wl_row): lanes along the tile as in ntt_tile.h (m = 8 j2 + j3: whole 128-byte runs of the tiled scratch
//     per load), the exchange behind round 1 crosses the wavefronts into wavefront = k1 pair, the exchange behind round 2 stays
//     inside the wavefront (own region); stores are 32-byte row segments whatever the numbering.
// A wavefront runs two thirds of its arithmetic between its loads and the one barrier (column pass) or between the one barrier and
// its stores (row pass) without meeting anybody; the round-1 twiddles of the column pass come from a permuted copy of the round
// table (plan.h wr_table) so that its 16 lanes per column read ONE 128-byte line per instruction, and the round-2 twiddles are
// wave-uniform there (scalar loads).
//
// FULL = true (the product): image of 8-byte cells, 69 888 bytes, two workgroups per CU.  Measured on MI355X (round 6,
// profiles/r06_wl_ab.txt): two-lane 2^22 regime 22.4 k -> 23.6 k NTT/s on one box (+5 %), one stream 58.5 -> 55.5 us.
// FULL = false: image of 4-byte cells (34 944 bytes), every exchange in two 32-bit phases -- low words, then high words; the
// cross-wave exchange with TWO barriers instead of three: the high words are written into exactly the cells their writer has
// just read its low words from, so nothing separates "low words read" from "high words parked".  Three to four workgroups per
// CU -- and SLOWER: 2 / 3 / 4 resident workgroups per CU run the two-lane regime at 23.3 k / 22.0 k / 20.9 k NTT/s (same box,
// same bodies, occupancy bounded by an LDS pad).  More wavefronts in flight cost more on the memory side (in-flight tiles
// outgrow the 4 MiB L2 of an XCD) than they return on the VALU, which is ~88 % busy over a wavefront's life already
// (profiles/r06_pmc_base_vs_wl.txt).  Kept as an opt-in for experiments (RONK_WL_HALF=1).
//
// LDS image: 8 regions x 16 blocks x [16 slots x 4 columns + 4 pad cells]: cell = region * 1092 + block * 68 + slot * 4 + c.
// 68 = 4 and 1092 = 4 (mod 32): in every LDS instruction the 32 lanes of a half-wavefront vary 8 values of ONE of (region,
// block, slot) and the 4 columns -> 32 distinct banks (4-byte cells) / bank pairs (8-byte cells): 0 bank conflicts measured.
//
// Same TileArgs contract and the same results as ntt_tile.h, bit for bit (the arithmetic per coefficient is unchanged): KIND 1 /
// 3 (column pass, two-level tables / full matrix) and KIND 2 (row pass), full tiles, no features; Goldilocks and Montgomery
// primes (field_policy.h).  Plain C++ over (tid, bid, lds, barrier, wave_sync), so tests/emu runs the same code on host fibers.
// Reference semantics: Polynomial::fft / ifft, src/polynomial/mod.rs:273-323, :430-484 (omega = g^((p-1)/n), natural order).
#pragma once
#include "ntt_tile.h"

namespace ronk {

constexpr int WL_LOGC = 2;
constexpr u32 WL_BLOCK = 68, WL_REGION = 16 * WL_BLOCK + 4;   // cells; 1092
constexpr bool wl_logr_ok(int logr) { return logr >= 10 && logr <= 12; }
constexpr u32 wl_waves(int logr) { return 1u << (logr - 8); }                 // = RL, the radix of the last round
constexpr u32 wl_threads(int logr) { return 64u * wl_waves(logr); }
constexpr size_t wl_lds_bytes(int logr, bool full) { return (size_t)wl_waves(logr) * WL_REGION * (full ? 8 : 4); }

// the full twiddle matrix may be laid out transposed, [col][k] (plan.h twf_transposed): tf_sc = R, tf_sk = 1
inline bool tile_wl_twf_transposed(const TileArgs& a, int logr) { return a.tw_full && a.tf_sk == 1 && a.tf_sc == (1u << logr); }
inline bool tile_wl_matches(const TileArgs& a, int logr, int kind) {
  if (!wl_logr_ok(logr) || !(kind == 1 || kind == 2 || kind == 3)) return false;
  if (kind == 3 && tile_wl_twf_transposed(a, logr)) {
    TileArgs b = a;
    b.tf_sc = 1; b.tf_sk = (u32)a.ncols;   // what tile_cfg_matches knows as the matrix of a column pass
    return a.logc == (u32)WL_LOGC && a.tf_sb2 == 0 && tile_cfg_matches(b, logr, WL_LOGC, 3, 0);
  }
  return tile_cfg_matches(a, logr, WL_LOGC, kind, 0);
}

// One LDS cell of the image: 8-byte cells (FULL) or the low / high word of a coefficient in a 4-byte cell
struct WlImage {
  u32* l32;
  RONK_HD void put(u32 cell, u64 v) const { reinterpret_cast<u64*>(l32)[cell] = v; }
  RONK_HD u64 get(u32 cell) const { return reinterpret_cast<u64*>(l32)[cell]; }
  RONK_HD void put_lo(u32 cell, u64 v) const { l32[cell] = (u32)v; }
  RONK_HD void put_hi(u32 cell, u64 v) const { l32[cell] = (u32)(v >> 32); }
  RONK_HD u32 get32(u32 cell) const { return l32[cell]; }
};

// wave_sync(): orders a wavefront's LDS accesses against its own later ones.  On the device LDS operations of one wavefront
// execute in order, so it is a compiler fence only; the host emulator (one fiber per lane) passes its barrier.

// ---- column pass: wavefront = j3 ----------------------------------------------------------------------------------
template <int LOGR, bool INV, int KIND, bool FULL, class FLD = GlField, class Barrier, class WaveSync>
RONK_HD void tile_body_wl_col(const TileArgs& a_in, u32* l32, u32 tid, u32 bid, Barrier&& barrier, WaveSync&& wave_sync) {
  static_assert(KIND == 1 || KIND == 3, "column pass");
  static_assert(wl_logr_ok(LOGR) && (FULL || LOGR == 11), "2^10 .. 2^12 rows; the half image is written for 2^11");
  constexpr int LOGL = LOGR - 8, RL = 1 << LOGL, G = 16 / RL;   // last-round radix = wavefronts; groups per lane in the last round
  typedef TileCfg<WL_LOGC, KIND> CFG;
  const WlImage img{l32};
  const u32 c = tid & 3, l = (tid >> 2) & 15, w = wave_uniform(tid >> 6);   // column, lane digit, wavefront (a scalar)
  // rounds 1-2: j3 = w, j2 (then k1) = l.  ntt_tile.h's lane index of the same coefficients: m = RL j2 + j3
  const u32 m_old = l * RL + w;
  const TileCtx cx = tile_ctx<LOGR, CFG>(a_in, m_old * 4 + c, bid);
  const TileArgs& a = cx.a;
  const FLD f(a.fc);
  u64 x[16];
  {   // x[j1] = row 16 RL j1 + RL j2 + j3 (flat rows, as tile_load for KIND 1 / 3)
    const u32 j0 = cx.in_lane + m_old * cx.in_sj, step = (16 * RL) * cx.in_sj;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = ld_g<true>(cx.in, j0 + i * step);
  }

  // ---- round 1 (over j1), table twiddle omega_R^{m_old k1}; exchange 1 inside the wavefront's own region: park (block k1, slot
  // j2), read back as lane k1 = l (block l, slot j2).  The twiddles come from the PERMUTED copy of the round table behind the
  // table itself (plan.h wr_table: entry R + (w * 16 + k1) * 16 + l = omega_R^{(8 l + w) k1}): the 16 lanes of a column read 16
  // consecutive entries -- one 128-byte line per instruction instead of 16 entries 64 k1 bytes apart -- at one per-lane base
  // plus immediates.
  Dif<16, INV, true, FLD>::run(x, f);
  u32 tb[16];
  const u32 reg = w * WL_REGION;
  {
    const u32 wbase = reg + l * 4 + c, rbase = reg + l * WL_BLOCK + c;
    const u32 tperm = (u32)(8 << LOGR) + (w << 11) + (l << 3);   // bytes
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k1 = brev(i, 4);
      if (k1) x[i] = f.mul(x[i], ld_tabb(a.wr, tperm + (k1 << 7)));
      if constexpr (FULL) img.put(wbase + k1 * WL_BLOCK, x[i]); else img.put_lo(wbase + k1 * WL_BLOCK, x[i]);
    }
    wave_sync();
    if constexpr (FULL) {
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] = img.get(rbase + j * 4);
    } else {
      u32 lo[16];
#pragma unroll
      for (int j = 0; j < 16; j++) lo[j] = img.get32(rbase + j * 4);
      wave_sync();
#pragma unroll
      for (int i = 0; i < 16; i++) img.put_hi(wbase + (u32)brev(i, 4) * WL_BLOCK, x[i]);
      wave_sync();
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] = ((u64)img.get32(rbase + j * 4) << 32) | lo[j];
    }
    wave_sync();
  }
  // ---- round 2 (over j2): lane (k1 = l, j3 = w); twiddle omega_{R/16}^{j3 k2} = omega_R^{16 j3 k2} (the same for the whole
  // wavefront).  Exchange 2 crosses the wavefronts: wavefront p of round 3 owns k2 in {2p, 2p + 1}.
  //   FULL        parked in place (own region: block l, slot k2)              | barrier |  gathered from region j3, slot 2p + g
  //   low words   the same
  //   high words  parked in the cells just read: region k2 >> 1, block l, slot 2w + (k2 & 1)   | barrier |  read from the own
  //               region, slot 2 j3 + g  (the writer of a cell is the one lane that read it: no barrier in between)
  // (the matrix entries of the first two output chunks are requested BEFORE the barrier -- the kernels' barrier waits for LDS
  // only, tile_kernels_wl.hip -- so that their trip to memory passes under the barrier wait and the exchange's LDS reads)
  constexpr int SH = 3;   // NARROW byte offsets (KIND != 0)
  // natural output row of register r = g RL + i of the last round: k = k1 + 16 k2 + 256 k3 = l + 16 (G w + g) + 256 brev(i)
  const u32 kbase = l + 16 * G * w;
  auto krow = [](int r) -> u32 { return (u32)(16 * (r / RL) + 256 * brev(r % RL, LOGL)); };
  const u32 tf_lane = KIND == 3 ? (cx.col * a_in.tf_sc) << SH : 0, tf_sk = KIND == 3 ? a_in.tf_sk << SH : 0;
  u64 wq[2][4];
  auto fetch = [&](int q, u64* wv) {   // chunk q: registers 4 q .. 4 q + 3
#pragma unroll
    for (int i = 0; i < 4; i++) wv[i] = ld_g<true>(a.tw_full, tf_lane + (kbase + krow(4 * q + i)) * tf_sk);
  };
  {
    Dif<16, INV, true, FLD>::run(x, f);
    tb[0] = 0; tb[1] = w << 7;
#pragma unroll
    for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
    const u32 own = reg + l * WL_BLOCK + c;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k2 = brev(i, 4);
      if (k2) x[i] = f.mul(x[i], ld_tabb(a.wr, tb[k2]));
      if constexpr (FULL) img.put(own + k2 * 4, x[i]); else img.put_lo(own + k2 * 4, x[i]);
    }
#ifndef RONK_WL_NO_PREFETCH   // (A/B builds: tools/build_variant.sh nopf -DRONK_WL_NO_PREFETCH)
    if constexpr (KIND == 3 && FULL) { fetch(0, wq[0]); fetch(1, wq[1]); }
#endif
    barrier();
    const u32 gat = l * WL_BLOCK + (G * w) * 4 + c;
    if constexpr (FULL) {
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int j = 0; j < RL; j++) x[g * RL + j] = img.get(gat + j * WL_REGION + g * 4);
    } else {
      u32 lo[16];
#pragma unroll
      for (int g = 0; g < 2; g++)
#pragma unroll
        for (int j = 0; j < 8; j++) lo[g * 8 + j] = img.get32(gat + j * WL_REGION + g * 4);
      wave_sync();
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const u32 k2 = brev(i, 4);
        img.put_hi(gat + (k2 >> 1) * WL_REGION + (k2 & 1) * 4, x[i]);
      }
      barrier();
#pragma unroll
      for (int g = 0; g < 2; g++)
#pragma unroll
        for (int j = 0; j < 8; j++) x[g * 8 + j] = ((u64)img.get32(own + (2 * j + g) * 4) << 32) | lo[g * 8 + j];
    }
  }
  // ---- round 3 (over j3): wavefront p = w owns k2 in {2p, 2p+1}, lane k1 = l; register g*8 + j3.  Natural output row of
  // register (g, i): k = l + 16 (2 w + g) + 256 brev3(i): the 16 lanes of a column hold 16 consecutive rows.
  const u32 out_sk = cx.out_sk, out_lane = cx.out_lane;
  u64* __restrict__ const outp = cx.out;
  if constexpr (KIND == 3) {
    // (tf_lane / tf_sk above are the strides of the LAUNCH, not those tile_ctx folds in: the matrix may be transposed -- [col][k],
    // plan.h twf_transposed.)  The matrix entries travel in chunks of four, two chunks in flight: chunk q + 2 is fetched before
    // the stores of chunk q are issued (one in-order vmcnt for loads and stores: a load behind a store waits for the store's
    // whole trip).
#ifndef RONK_WL_NO_PREFETCH
    if constexpr (!FULL) { fetch(0, wq[0]); fetch(1, wq[1]); }
#else
    fetch(0, wq[0]); fetch(1, wq[1]);
#endif
#pragma unroll
    for (int q = 0; q < 4; q++) {
      u64* xq = x + 4 * q;
      if ((4 * q) % RL == 0) {   // a group's sub-transform runs when its first chunk comes up (RL = 4: every chunk is a group)
        Dif<RL, INV, true, FLD>::run(x + (4 * q / RL) * RL, f, false);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) xq[i] = f.mul(xq[i], wq[q & 1][i]);
      if (q + 2 < 4) fetch(q + 2, wq[q & 1]);
#pragma unroll
      for (int i = 0; i < 4; i++) st_g<true>(outp, out_lane + (kbase + krow(4 * q + i)) * out_sk, xq[i]);
    }
  } else {
    // two-level inter-pass twiddle omega_N^{col * k}: exponents pre-scaled by 8 (byte offsets), add chain over i
    const u32 nmask = a.tw_log >= 32 ? 0xFFFFFFFFu : ((1u << a.tw_log) - 1);
    const u32 lmask8 = ((1u << a.tw_lo_bits) - 1) << 3, hmask8 = (nmask >> a.tw_lo_bits) << 3;
    const u32 twX = cx.col;
#pragma unroll
    for (int g = 0; g < G; g++) {
      u64* xg = x + g * RL;
      Dif<RL, INV, true, FLD>::run(xg, f, false);
      u32 ej[RL];
      ej[0] = (twX * (kbase + 16 * g)) << 3;
      const u32 estep = (twX * 256u) << 3;
#pragma unroll
      for (int j = 1; j < RL; j++) ej[j] = ej[j - 1] + estep;
#pragma unroll
      for (int i = 0; i < RL; i++) {
        const u32 ee = ej[brev(i, LOGL)];
        const u64 tw = f.mul(ld_tabb(a.tw_lo, ee & lmask8), ld_tabb(a.tw_hi, (ee >> a.tw_lo_bits) & hmask8));
        xg[i] = f.mul(xg[i], tw);
      }
#pragma unroll
      for (int i = 0; i < RL; i++) st_g<true>(outp, out_lane + (kbase + 16 * g + 256 * brev(i, LOGL)) * out_sk, xg[i]);
    }
  }
}

// ---- row pass: lanes along the tile as in ntt_tile.h, wavefront = k1 group from round 2 on -----------------------------
template <int LOGR, bool INV, bool FULL, class FLD = GlField, class Barrier, class WaveSync>
RONK_HD void tile_body_wl_row(const TileArgs& a_in, u32* l32, u32 tid, u32 bid, Barrier&& barrier, WaveSync&& wave_sync) {
  static_assert(wl_logr_ok(LOGR) && (FULL || LOGR == 11), "2^10 .. 2^12 rows; the half image is written for 2^11");
  constexpr int LOGL = LOGR - 8, RL = 1 << LOGL, G = 16 / RL;   // last-round radix = wavefronts; j2 values / k1 values per wavefront
  typedef TileCfg<WL_LOGC, 2> CFG;
  const WlImage img{l32};
  const u32 c = tid & 3, l = (tid >> 2) & 15, w = wave_uniform(tid >> 6);
  const u32 m = tid >> 2;                       // 16 w + l = RL j2 + j3
  const u32 e = l >> LOGL, j3 = l & (RL - 1);   // j2 = G w + e
  const TileCtx cx = tile_ctx<LOGR, CFG>(a_in, tid, bid);
  const TileArgs& a = cx.a;
  const FLD f(a.fc);
  u64 x[16];
  {   // x[j1] = row 16 RL j1 + m of the tiled scratch (blocked rows, as tile_load for KIND 2)
    const u32 hi = (u32)a.in_sj_hi << 3, jmask = (1u << a.js_log) - 1;
    const u32 j0 = cx.in_lane + (m >> a.js_log) * hi + (m & jmask) * cx.in_sj, step = ((u32)(16 * RL) >> a.js_log) * hi;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = ld_g<true>(cx.in, j0 + i * step);
  }

  // ---- round 1 (over j1), twiddle omega_R^{m k1}.  Exchange 1 crosses the wavefronts: k1 = G q + s goes to wavefront q,
  // lane (s, j3) = block RL s + j3, register j2; the slot of j2 = G w + e is w + RL e (e-major: the half-wavefront that varies e
  // and j3 together -- RL = 4 -- still covers 32 distinct banks).
  //   FULL        scattered to region q, block RL s + j3, slot w + RL e      | barrier |  read from the own region, block l
  //   low words   the same (2^11 rows: RL = 8, G = 2)
  //   high words  parked in the cells just read (own region, block l, slot k1)   | barrier |  gathered from region j2's
  //               wavefront, block of (j2's e, j3), slot of (w, e)
  Dif<16, INV, true, FLD>::run(x, f);
  u32 tb[16];
  tb[0] = 0; tb[1] = m << 3;
#pragma unroll
  for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
  const u32 own = w * WL_REGION + l * WL_BLOCK + c;
  const u32 far = j3 * WL_BLOCK + (w + RL * e) * 4 + c;
  {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k1 = brev(i, 4);
      if (k1) x[i] = f.mul(x[i], ld_tabb(a.wr, tb[k1]));
      const u32 cell = far + (k1 / G) * WL_REGION + (k1 % G) * (RL * WL_BLOCK);
      if constexpr (FULL) img.put(cell, x[i]); else img.put_lo(cell, x[i]);
    }
    barrier();
    // register j2 = G w' + e' sits in slot w' + RL e'
    if constexpr (FULL) {
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] = img.get(own + ((j / G) + RL * (j % G)) * 4);
    } else {
      u32 lo[16];
#pragma unroll
      for (int j = 0; j < 16; j++) lo[j] = img.get32(own + ((j / G) + RL * (j % G)) * 4);
      wave_sync();
      // (the lane writes the 16 cells it has just read: high word of register k1 into the slot numbered k1)
#pragma unroll
      for (int i = 0; i < 16; i++) img.put_hi(own + (u32)brev(i, 4) * 4, x[i]);
      barrier();
      // element (k1 = G w + e of this lane; j2 = G w' + e') was parked by lane (wavefront w', e', j3) = region w', block RL e' + j3,
      // in the slot numbered k1
#pragma unroll
      for (int j = 0; j < 16; j++)
        x[j] = ((u64)img.get32((j / G) * WL_REGION + ((j % G) * RL + j3) * WL_BLOCK + (G * w + e) * 4 + c) << 32) | lo[j];
    }
    wave_sync();
  }
  // ---- round 2 (over j2): lane (k1 = G w + e, j3); twiddle omega_R^{16 j3 k2}.  Exchange 2 stays inside the wavefront.
  //   FULL   after exchange 1 the wavefront's own region is read by nobody else: element (s = e, j3, k2 = RL g + u) goes to block
  //          RL s + u, slot RL g + j3, and lane (s, u) of round 3 reads its block: slot RL g + j3'
  //   halves (2^11 rows) through the cells the wavefront alone has just read (slots 2 w, 2 w + 1 of every block of every region):
  //          element (s = e, j3, k2) at region j3, block k2, slot 2 w + s; lane (s, u) reads k2 = 8 g + u: region j3', block 8 g + u
  {
    Dif<16, INV, true, FLD>::run(x, f);
    tb[1] = j3 << 7;
#pragma unroll
    for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
    if constexpr (FULL) {
      const u32 wb = w * WL_REGION + e * (RL * WL_BLOCK) + j3 * 4 + c;   // + u * WL_BLOCK + g * RL * 4, k2 = RL g + u
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const u32 k2 = brev(i, 4);
        if (k2) x[i] = f.mul(x[i], ld_tabb(a.wr, tb[k2]));
        img.put(wb + (k2 % RL) * WL_BLOCK + (k2 / RL) * (RL * 4), x[i]);
      }
      wave_sync();
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int j = 0; j < RL; j++) x[g * RL + j] = img.get(own + (RL * g + j) * 4);
    } else {
      const u32 wbase = j3 * WL_REGION + (2 * w + e) * 4 + c;          // + k2 * WL_BLOCK
      const u32 rbase = j3 * WL_BLOCK + (2 * w + e) * 4 + c;           // u = j3 of the lane: + j3' * WL_REGION + g * 8 * WL_BLOCK
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const u32 k2 = brev(i, 4);
        if (k2) x[i] = f.mul(x[i], ld_tabb(a.wr, tb[k2]));
        img.put_lo(wbase + k2 * WL_BLOCK, x[i]);
      }
      wave_sync();
      u32 lo[16];
#pragma unroll
      for (int g = 0; g < 2; g++)
#pragma unroll
        for (int j = 0; j < 8; j++) lo[g * 8 + j] = img.get32(rbase + j * WL_REGION + g * (8 * WL_BLOCK));
      wave_sync();
#pragma unroll
      for (int i = 0; i < 16; i++) img.put_hi(wbase + (u32)brev(i, 4) * WL_BLOCK, x[i]);
      wave_sync();
#pragma unroll
      for (int g = 0; g < 2; g++)
#pragma unroll
        for (int j = 0; j < 8; j++) x[g * 8 + j] = ((u64)img.get32(rbase + j * WL_REGION + g * (8 * WL_BLOCK)) << 32) | lo[g * 8 + j];
    }
  }
  // ---- round 3 (over j3): lane (s = e, u = l mod RL); register g RL + j3 = element (k1 = G w + e, k2 = RL g + u, j3);
  // natural output row k = k1 + 16 k2 + 256 brev(i)
  const u32 kbase = (G * w + e) + 16 * j3;   // u = l mod RL
  const u32 out_sk = cx.out_sk, out_lane = cx.out_lane;
  u64* __restrict__ const outp = cx.out;
#pragma unroll
  for (int g = 0; g < G; g++) {
    u64* xg = x + g * RL;
    Dif<RL, INV, false, FLD>::run(xg, f, false);
#pragma unroll
    for (int i = 0; i < RL; i++) st_g<true>(outp, out_lane + (kbase + 16 * RL * g + 256 * brev(i, LOGL)) * out_sk, xg[i]);
  }
}

}  // namespace ronk
