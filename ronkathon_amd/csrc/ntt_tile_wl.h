// ntt_tile_wl.h -- the 2^11-row x 4-column tile passes of the two-pass plans (2^21 .. 2^23: the headline 2^22 among them) with
// HALF the LDS image and TWO workgroup barriers per pass, so that 3-4 eight-wavefront workgroups are resident per CU (24-32
// wavefronts) instead of two (16).
//
// Why (rounds 4-5): between 245 and 280 VALU per coefficient the 2^22 transform's time is set by how many
// wavefronts are in an arithmetic phase at once.  ntt_tile.h parks a whole tile in LDS between rounds -- 8.5 KiB per wavefront,
// 16 wavefronts per CU -- and its half-image form (TileCfg::HALF) pays for 8 per SIMD with two more workgroup barriers per
// exchange.  Here, with rows j = 128 j1 + 8 j2 + j3 (digits of the rounds 16, 16, 8):
//   * ONE of the two exchanges of a pass stays inside a wavefront: it needs no workgroup barrier, and run in two 32-bit phases
//     (low words, then high words) it passes through the wavefront's own 4.3 KiB of the image;
//   * the other exchange crosses the wavefronts in two 32-bit phases with TWO barriers instead of three: the high words are
//     written into exactly the cells their writer has just read its low words from (nobody else reads those), so no barrier
//     separates "low words read" from "high words parked".
// Column pass (tile_body_wl_col): lanes numbered wavefront = j3, so exchange 1 (behind round 1) is the wave-local one and the
//   last round's 16 lanes per column hold 16 CONSECUTIVE output rows -- whole 512-byte runs of the tiled scratch per store.
// Row pass (tile_body_wl_row): lanes numbered as in ntt_tile.h (m = 8 j2 + j3 along the tile: whole 128-byte runs of the tiled
//   scratch per load), exchange 1 crosses the wavefronts into wavefront = k1 pair, exchange 2 is wave-local through the cells
//   the wavefront alone has read last (the d2 slab of its k1 pair); its stores are 32-byte row segments whatever the numbering.
//
// LDS image: 4-byte cells, 8 regions x 16 blocks x [16 x 4 columns + 4 pad]: cell = region * 1092 + block * 68 + slot * 4 + c.
// 68 = 4 (mod 32) and 1092 = 4 (mod 32): in every ds_read_b32 / ds_write_b32 below the 32 lanes of a half-wavefront vary 8
// values of ONE of (region, block) and the 4 columns -> 32 distinct banks; a varying slot never meets a varying block.
// 34 944 bytes per workgroup.
//
// Same TileArgs contract and the same results as ntt_tile.h, bit for bit (the arithmetic per coefficient is unchanged):
// KIND 1 / 3 (column pass, two-level tables / full matrix) and KIND 2 (row pass), full tiles, no features.  Plain C++ over
// (tid, bid, lds, barrier, wave_sync), so tests/emu runs the very same code on host fibers.
// Reference semantics: Polynomial::fft / ifft, src/polynomial/mod.rs:273-323, :430-484 (omega = g^((p-1)/n), natural order).
#pragma once
#include "ntt_tile.h"

namespace ronk {

constexpr int WL_LOGR = 11, WL_LOGC = 2;
constexpr u32 WL_BLOCK = 68, WL_REGION = 16 * WL_BLOCK + 4;   // cells; 1092
constexpr size_t WL_LDS_BYTES = (size_t)8 * WL_REGION * 4;
constexpr u32 WL_THREADS = 512;

inline bool tile_wl_matches(const TileArgs& a, int logr, int kind) {
  return !a.fc.p && logr == WL_LOGR && (kind == 1 || kind == 2 || kind == 3) && tile_cfg_matches(a, WL_LOGR, WL_LOGC, kind, 0);
}

// MF: memory-policy flags of an instantiation (experiments; 0 in the product until measured): 1 = the full twiddle matrix is read
// with non-temporal loads (every entry is used once per transform, by one lane), 2 = non-temporal stores, 4 = non-temporal
// loads of the tile itself (row pass: every scratch line is read by exactly one tile).
constexpr int WL_NT_TWF = 1, WL_NT_ST = 2, WL_NT_LD = 4;
template <bool NT>
RONK_HD u64 ld_gb(const u64* base, u32 byte_off) {
  const u64* q = reinterpret_cast<const u64*>(reinterpret_cast<const char*>(base) + byte_off);
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (NT) return __builtin_nontemporal_load(q);
#endif
  return *q;
}
template <bool NT>
RONK_HD void st_gb(u64* base, u32 byte_off, u64 v) {
  u64* q = reinterpret_cast<u64*>(reinterpret_cast<char*>(base) + byte_off);
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (NT) { __builtin_nontemporal_store(v, q); return; }
#endif
  st_out(q, v);
}

// wave_sync(): orders a wavefront's LDS accesses against its own later ones.  On the device LDS operations of one wavefront
// execute in order, so it is a compiler fence only; the host emulator (one fiber per lane) passes its barrier.

// ---- column pass: wavefront = j3 ----------------------------------------------------------------------------------
template <bool INV, int KIND, int MF = 0, class Barrier, class WaveSync>
RONK_HD void tile_body_wl_col(const TileArgs& a_in, u32* l32, u32 tid, u32 bid, Barrier&& barrier, WaveSync&& wave_sync) {
  static_assert(KIND == 1 || KIND == 3, "column pass");
  constexpr int LOGR = WL_LOGR;
  typedef TileCfg<WL_LOGC, KIND> CFG;
  const u32 c = tid & 3, l = (tid >> 2) & 15, w = wave_uniform(tid >> 6);   // column, lane digit, wavefront (a scalar)
  // rounds 1-2: j3 = w, j2 (then k1) = l.  ntt_tile.h's lane index of the same coefficients: m = 8 j2 + j3
  const u32 m_old = l * 8 + w;
  const TileCtx cx = tile_ctx<LOGR, CFG>(a_in, m_old * 4 + c, bid);
  const TileArgs& a = cx.a;
  u64 x[16];
  {   // x[j1] = row 128 j1 + 8 j2 + j3 (flat rows: what tile_load does for KIND 1 / 3, with the load policy of MF)
    const u32 j0 = cx.in_lane + m_old * cx.in_sj, step = 128 * cx.in_sj;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = ld_gb<(MF & WL_NT_LD) != 0>(cx.in, j0 + i * step);
  }

  // ---- round 1 (over j1), table twiddle omega_R^{m_old k1}; exchange 1 inside the wavefront's own region:
  // park (block k1, slot j2), read back as lane k1 = l (block l, slot j2)
  // The twiddles come from the PERMUTED copy of the round table behind the table itself (plan.h wr_table: entry
  // R + (w * 16 + k1) * 16 + l = omega_R^{(8 l + w) k1}): the 16 lanes of a column read 16 consecutive entries -- one 128-byte
  // line per instruction instead of 16 entries 64 k1 bytes apart -- at one per-lane base plus immediates.
  Dif<16, INV, true>::run(x);
  u32 tb[16];
  const u32 reg = w * WL_REGION;
  {
    const u32 wbase = reg + l * 4 + c, rbase = reg + l * WL_BLOCK + c;
    const u32 tperm = (u32)(8 << LOGR) + (w << 11) + (l << 3);   // bytes
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k1 = brev(i, 4);
      if (k1) x[i] = gl64::mul(x[i], ld_tabb(a.wr, tperm + (k1 << 7)));
      l32[wbase + k1 * WL_BLOCK] = (u32)x[i];
    }
    wave_sync();
    u32 lo[16];
#pragma unroll
    for (int j = 0; j < 16; j++) lo[j] = l32[rbase + j * 4];
    wave_sync();
#pragma unroll
    for (int i = 0; i < 16; i++) l32[wbase + (u32)brev(i, 4) * WL_BLOCK] = (u32)(x[i] >> 32);
    wave_sync();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = ((u64)l32[rbase + j * 4] << 32) | lo[j];
    wave_sync();
  }
  // ---- round 2 (over j2): lane (k1 = l, j3 = w); twiddle omega_{R/16}^{j3 k2} = omega_R^{16 j3 k2} (the same for the whole
  // wavefront).  Exchange 2 crosses the wavefronts: wavefront p of round 3 owns k2 in {2p, 2p + 1}.
  //   low words   parked in place (own region: block l, slot k2)          | barrier |  gathered from region j3, slot 2p + g
  //   high words  parked in the cells just read: region k2 >> 1, block l, slot 2w + (k2 & 1)   | barrier |  read from the own
  //               region, slot 2 j3 + g  (the writer of a cell is the one lane that read it: no barrier in between)
  {
    Dif<16, INV, true>::run(x);
    tb[0] = 0; tb[1] = w << 7;
#pragma unroll
    for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
    const u32 own = reg + l * WL_BLOCK + c;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k2 = brev(i, 4);
      if (k2) x[i] = gl64::mul(x[i], ld_tabb(a.wr, tb[k2]));
      l32[own + k2 * 4] = (u32)x[i];
    }
    barrier();
    u32 lo[16];
    const u32 gat = l * WL_BLOCK + (2 * w) * 4 + c;
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int j = 0; j < 8; j++) lo[g * 8 + j] = l32[gat + j * WL_REGION + g * 4];
    wave_sync();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k2 = brev(i, 4);
      l32[gat + (k2 >> 1) * WL_REGION + (k2 & 1) * 4] = (u32)(x[i] >> 32);
    }
    barrier();
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int j = 0; j < 8; j++) x[g * 8 + j] = ((u64)l32[own + (2 * j + g) * 4] << 32) | lo[g * 8 + j];
  }
  // ---- round 3 (over j3): wavefront p = w owns k2 in {2p, 2p+1}, lane k1 = l; register g*8 + j3.  Natural output row of
  // register (g, i): k = l + 16 (2 w + g) + 256 brev3(i): the 16 lanes of a column hold 16 consecutive rows.
  constexpr int SH = 3;   // NARROW byte offsets (KIND != 0)
  const u32 kbase = l + 32 * w;
  const u32 out_sk = cx.out_sk, out_lane = cx.out_lane;
  u64* __restrict__ const outp = cx.out;
  if constexpr (KIND == 3) {
    const u32 tf_lane = (cx.col * a.tf_sc) << SH, tf_sk = a.tf_sk << SH;
    // The matrix entries travel in chunks of four, two chunks in flight: chunk q + 2 is fetched before the stores of chunk q are
    // issued (one in-order vmcnt for loads and stores: a load behind a store waits for the store's whole trip), and the
    // tail needs 16 instead of 32 registers of twiddles beside the 32 of coefficients (the kernel is built for <= 64).
    // chunk q = (g, h): registers g*8 + 4h .. + 3
    u64 wq[2][4];
    auto fetch = [&](int q, u64* wv) {
#pragma unroll
      for (int i = 0; i < 4; i++) wv[i] = ld_gb<(MF & WL_NT_TWF) != 0>(a.tw_full, tf_lane + (kbase + 16 * (q >> 1) + 256 * brev(4 * (q & 1) + i, 3)) * tf_sk);
    };
    fetch(0, wq[0]);
    fetch(1, wq[1]);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      u64* xq = x + 4 * q;
      if ((q & 1) == 0) Dif<8, INV, true>::run(xq, false);
#pragma unroll
      for (int i = 0; i < 4; i++) xq[i] = gl64::mul(xq[i], wq[q & 1][i]);
      if (q + 2 < 4) fetch(q + 2, wq[q & 1]);
#pragma unroll
      for (int i = 0; i < 4; i++) st_gb<(MF & WL_NT_ST) != 0>(outp, out_lane + (kbase + 16 * (q >> 1) + 256 * brev(4 * (q & 1) + i, 3)) * out_sk, xq[i]);
    }
  } else {
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
      for (int i = 0; i < 8; i++) st_gb<(MF & WL_NT_ST) != 0>(outp, out_lane + (kbase + 16 * g + 256 * brev(i, 3)) * out_sk, xg[i]);
    }
  }
}

// ---- row pass: lanes along the tile as in ntt_tile.h, wavefront = k1 pair from round 2 on -----------------------------
template <bool INV, int MF = 0, class Barrier, class WaveSync>
RONK_HD void tile_body_wl_row(const TileArgs& a_in, u32* l32, u32 tid, u32 bid, Barrier&& barrier, WaveSync&& wave_sync) {
  constexpr int LOGR = WL_LOGR;
  typedef TileCfg<WL_LOGC, 2> CFG;
  const u32 c = tid & 3, l = (tid >> 2) & 15, w = wave_uniform(tid >> 6);
  const u32 m = tid >> 2;                 // 16 w + l = 8 j2 + j3
  const u32 e = l >> 3, j3 = l & 7;       // j2 = 2 w + e
  const TileCtx cx = tile_ctx<LOGR, CFG>(a_in, tid, bid);
  const TileArgs& a = cx.a;
  u64 x[16];
  {   // x[j1] = row 128 j1 + m of the tiled scratch (blocked rows: what tile_load does for KIND 2, with the load policy of MF)
    const u32 hi = (u32)a.in_sj_hi << 3, jmask = (1u << a.js_log) - 1;
    const u32 j0 = cx.in_lane + (m >> a.js_log) * hi + (m & jmask) * cx.in_sj, step = (128u >> a.js_log) * hi;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = ld_gb<(MF & WL_NT_LD) != 0>(cx.in, j0 + i * step);
  }

  // ---- round 1 (over j1), twiddle omega_R^{m k1}.  Exchange 1 crosses the wavefronts: k1 = 2 q + s goes to wavefront q,
  // lane (s, j3), register j2.
  //   low words   scattered to region q, block 8 s + j3, slot j2 = 2 w + e   | barrier |  read from the own region, block l
  //   high words  parked in the cells just read (own region, block l, slot k1)   | barrier |  gathered from region j2 >> 1,
  //               block 8 (j2 & 1) + j3, slot 2 w + e
  Dif<16, INV, true>::run(x);
  u32 tb[16];
  tb[0] = 0; tb[1] = m << 3;
#pragma unroll
  for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
  const u32 own = w * WL_REGION + l * WL_BLOCK + c;
  const u32 far = j3 * WL_BLOCK + (2 * w + e) * 4 + c;
  {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k1 = brev(i, 4);
      if (k1) x[i] = gl64::mul(x[i], ld_tabb(a.wr, tb[k1]));
      l32[far + (k1 >> 1) * WL_REGION + (k1 & 1) * (8 * WL_BLOCK)] = (u32)x[i];
    }
    barrier();
    u32 lo[16];
#pragma unroll
    for (int j = 0; j < 16; j++) lo[j] = l32[own + j * 4];
    wave_sync();
#pragma unroll
    for (int i = 0; i < 16; i++) l32[own + (u32)brev(i, 4) * 4] = (u32)(x[i] >> 32);
    barrier();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = ((u64)l32[far + (j >> 1) * WL_REGION + (j & 1) * (8 * WL_BLOCK)] << 32) | lo[j];
    wave_sync();
  }
  // ---- round 2 (over j2): lane (k1 = 2 w + e, j3); twiddle omega_R^{16 j3 k2}.  Exchange 2 stays inside the wavefront, through
  // the cells it alone has just read (slots 2 w, 2 w + 1 of every block of every region): element (s = e, j3, k2) at region
  // j3, block k2, slot 2 w + s; lane (s, u) of round 3 reads k2 = 8 g + u, i.e. region j3', block 8 g + u.
  {
    Dif<16, INV, true>::run(x);
    tb[1] = j3 << 7;
#pragma unroll
    for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
    const u32 wbase = j3 * WL_REGION + (2 * w + e) * 4 + c;          // + k2 * WL_BLOCK
    const u32 rbase = j3 * WL_BLOCK + (2 * w + e) * 4 + c;           // u = j3 of the lane: + j3' * WL_REGION + g * 8 * WL_BLOCK
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k2 = brev(i, 4);
      if (k2) x[i] = gl64::mul(x[i], ld_tabb(a.wr, tb[k2]));
      l32[wbase + k2 * WL_BLOCK] = (u32)x[i];
    }
    wave_sync();
    u32 lo[16];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int j = 0; j < 8; j++) lo[g * 8 + j] = l32[rbase + j * WL_REGION + g * (8 * WL_BLOCK)];
    wave_sync();
#pragma unroll
    for (int i = 0; i < 16; i++) l32[wbase + (u32)brev(i, 4) * WL_BLOCK] = (u32)(x[i] >> 32);
    wave_sync();
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int j = 0; j < 8; j++) x[g * 8 + j] = ((u64)l32[rbase + j * WL_REGION + g * (8 * WL_BLOCK)] << 32) | lo[g * 8 + j];
  }
  // ---- round 3 (over j3): lane (s = e, u = l & 7); register g*8 + j3 = element (k1 = 2 w + e, k2 = 8 g + u, j3);
  // natural output row k = k1 + 16 k2 + 256 brev3(i)
  const u32 kbase = (2 * w + e) + 16 * j3;   // u = l & 7
  const u32 out_sk = cx.out_sk, out_lane = cx.out_lane;
  u64* __restrict__ const outp = cx.out;
#pragma unroll
  for (int g = 0; g < 2; g++) {
    u64* xg = x + g * 8;
    Dif<8, INV, false>::run(xg, false);
#pragma unroll
    for (int i = 0; i < 8; i++) st_gb<(MF & WL_NT_ST) != 0>(outp, out_lane + (kbase + 128 * g + 256 * brev(i, 3)) * out_sk, xg[i]);
  }
}

}  // namespace ronk
