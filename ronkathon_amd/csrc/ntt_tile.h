// ntt_tile.h -- the one NTT kernel every plan is composed from (Goldilocks, gfx950).
//
// A workgroup transforms a TILE: C adjacent "columns" x R "rows" (R = 2^LOGR, 16..4096),
// i.e. C independent R-point NTTs whose elements sit at arbitrary (row, column) strides in
// HBM.  Lanes always run fastest along the column index, so every 8-16 lane group touches
// one 64-128 B segment on both the load and the store side, whatever the strides are.  A
// thread owns 16 coefficients in VGPRs; the R-point transform is done as up to three
// register-resident radix-16 (last round radix-2..16) decimation-in-frequency rounds with
// the tile staged through LDS between rounds (three barriers at most; the second exchange is digit-swapped):
//
//   round i : 16-point sub-DFT over digit j_i (pure shifts: omega_16^k = +-2^K, gl64.h)
//             then one table twiddle omega_M^{rest * k_i} per coefficient
//   output  : natural index k = k_1 + 16 k_2 + 256 k_3, optional inter-pass twiddle
//             omega_N^{X*Y} (two-level table) and optional scale (n^-1 of the inverse)
//
// which restates, pass by pass, what Polynomial::fft / ifft compute recursively
// (reference src/polynomial/mod.rs:273-323, :430-484): X[k] = sum_j x[j] omega^{jk},
// natural order in, natural order out, omega = g^((p-1)/n).  No MFMA: this is 64-bit
// modular integer work, bounded by HBM traffic and the VALU integer-multiply rate.
//
// LDS image: row-major [R + R/16][C] u64 -- one dummy row after every 16 rows (row' = row + (row>>4)).
// The early rounds touch 8..16 consecutive rows per wave (contiguous, conflict free); in the last
// round lane groups read rows 16u+t for consecutive u, and the dummy row shifts each u by one row
// (64..128 B), so a half-wave covers all 64 banks once: 0 bank conflicts for ds_read_b64/ds_write_b64.
// Unlike an XOR swizzle the map is additive, so every LDS address is one per-lane base plus a
// wave-uniform constant (1 VALU per access).
//
// The body is plain C++ over (tid, bid, lds, barrier) so that tests/emu can run the very
// same code on host threads to check the index algebra without a GPU.
#pragma once
#include "field_policy.h"

namespace ronk {

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;

struct TileArgs {
  const u64* in;
  const u64* in2;  // optional second operand: x = in * in2 on load (fused pointwise product)
  u64* out;
  // element (j, c) of tile t in batch (b1, b2):
  //   in [b1*in_sb1 + b2*in_sb2 + t*in_st + c*in_sc + j*in_sj]
  //   out[b1*out_sb1 + b2*out_sb2 + t*out_st + c*out_sc + k*out_sk]
  // row j may be split in two levels (the multi-GPU receive buffer is a list of blocks):
  //   row offset = (j >> js_log)*in_sj_hi + (j & (2^js_log - 1))*in_sj      (js_log = 31: flat)
  i64 in_sj, in_sc, in_sb1, in_sb2;
  i64 in_st, out_st;  // offset of tile t: t*in_st / t*out_st (C*in_sc for a plain matrix; the scratch buffer of a
                      // two-pass plan is stored tile by tile instead, see plan.h)
  i64 in_sj_hi;
  u32 js_log;
  i64 out_sk, out_sc, out_sb1, out_sb2;
  u32 logc;   // C = 2^logc columns per tile
  u32 tiles;  // tiles per (b1, b2)
  u32 nb1, nb2;
  u64 ncols;  // valid columns per (b1, b2); columns >= ncols of the last tile are skipped
  const u64* wr;  // round twiddles omega_R^e, e in [0, R)
  // output twiddle omega_N^{(X*Y) mod N}, N = 2^tw_log (0 = none), w^e = tw_lo[e & m] * tw_hi[e >> bits]
  u32 tw_log, tw_lo_bits;
  const u64* tw_lo;
  const u64* tw_hi;
  // ... or, when the whole twiddle matrix is affordable, ONE coalesced load instead of two gathers and a
  // multiply: w = tw_full[k*tf_sk + (t*C + c)*tf_sc + b2*tf_sb2] (laid out like the pass's own output tile)
  const u64* tw_full;
  u32 tf_sk, tf_sc, tf_sb2;
  u64 xc, xb1, xb2, x0;  // X = xc*(t*C + c) + xb1*b1 + xb2*b2 + x0
  u64 yk, yb1, yb2, y0;  // Y = yk*k + yb1*b1 + yb2*b2 + y0
  u64 scale;             // 1 = none
  // implicit zero padding / truncation (polynomial multiply): with lin = element offset inside the polynomial
  // (everything but the b1 term), loads with lin >= in_valid read as ZERO and stores with lin >= out_valid are
  // dropped.  ~0 = no limit.
  u64 in_valid, out_valid;
  u64 in_valid1;   // the input limit of batch entries b1 >= 1 when it differs from in_valid (~0: the same): the two operands
                   // of a polynomial multiply transformed as ONE batch of two
  // Staged I/O for single-pass plans of tiny transforms (n = 16, 32; plan.h): the tile -- C whole polynomials -- is one contiguous
  // run of R*C elements on both sides, but a lane's own accesses are only M*8 bytes long (n = 16: one polynomial per
  // lane, 128-byte lane stride, outputs in bit-reversed order).  With stage_io the run is copied HBM <-> LDS with fully
  // coalesced accesses and the per-lane indexing happens against LDS (image e + e/16, e = c*R + row).
  u32 stage_io;
  // the prime of a Montgomery pass (field_policy.h; p == 0: Goldilocks, nothing of it is read)
  FieldConst fc;
};

// ---- compile-time helpers -------------------------------------------------------------

constexpr int brev(int x, int bits) {
  int r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

// one DIF stage on x[0..N): (a, b) -> (a + b, (a - b) * omega_N^j), j = J..N/2-1.
// LAZY (last stage, N == 2, when every output but x[0] is multiplied next): the sums skip the canonicalising compare
// (gl64::add_lazy); `keep0` says whether this pair contains element 0 of the whole sub-transform, which is stored /
// parked without a multiplication and must stay canonical.
template <int N, bool INV, int J, bool LAZY = false, class FLD = GlField>
RONK_HD void dif_stage(u64* x, const FLD& f, bool keep0 = true) {
  if constexpr (J < N / 2) {
    u64 a = x[J], b = x[J + N / 2];
    if constexpr (LAZY) x[J] = keep0 ? f.add(a, b) : f.add_lazy(a, b);
    else x[J] = f.add(a, b);
    x[J + N / 2] = f.template sub_mul_root<N, J, INV>(a, b);
    dif_stage<N, INV, J + 1, LAZY, FLD>(x, f, keep0);
  }
}

// N-point DIF DFT in registers; x[t] ends up holding X[brev(t)].
// LAZY: outputs other than X[0] may be non-canonical representatives (see dif_stage); `first` = this sub-block starts at
// element 0 of the whole transform.
template <int N, bool INV, bool LAZY = false, class FLD = GlField>
struct Dif {
  static RONK_HD void run(u64* x, const FLD& f, bool first = true) {
    if constexpr (N == 2) {
      dif_stage<2, INV, 0, LAZY, FLD>(x, f, first);
    } else {
      dif_stage<N, INV, 0, false, FLD>(x, f);
      Dif<N / 2, INV, LAZY, FLD>::run(x, f, first);
      Dif<N / 2, INV, LAZY, FLD>::run(x + N / 2, f, false);
    }
  }
  // stateless fields (Goldilocks): no policy object to pass
  static RONK_HD void run(u64* x, bool first = true) { run(x, FLD(), first); }
};
template <bool INV, bool LAZY, class FLD>
struct Dif<1, INV, LAZY, FLD> {
  static RONK_HD void run(u64*, const FLD&, bool = true) {}
};

// FEAT_KEEP: after a three-round pass of 2^logr rows, register r of lane m holds output row m + M * keep_row_digit(logr, r)
// (M = 2^logr / 16): group g = r / RLAST is the sub-transform kl = m + g*M, register i = r % RLAST of it the output
// kl + (R / RLAST) * brev(i) = m + M * (g + (16 / RLAST) * brev(i)).  A bijection of 0..15 -- exactly the 16 rows i*M + m that
// the FIRST round of a column pass over the same index wants in x[i] (tile_load).
constexpr int keep_row_digit(int logr, int r) {
  const int loglast = logr - 8, rlast = 1 << loglast;
  return r / rlast + (16 / rlast) * brev(r % rlast, loglast);
}

// LDS row of tile row `row`: one dummy row after every 16 (see the header comment)
RONK_HD u32 swz_row(u32 row) { return row + (row >> 4); }

// table[idx] for a small twiddle table: wave-uniform base + 32-bit BYTE offset, so the load is the
// `global_load ... v_off, s[base:base+1]` form (one VALU shift) instead of a 64-bit per-lane address
RONK_HD u64 ld_tab(const u64* table, u32 idx) {
  return *reinterpret_cast<const u64*>(reinterpret_cast<const char*>(table) + (u32)(idx << 3));
}
// the same with the BYTE offset given
RONK_HD u64 ld_tabb(const u64* table, u32 byte_off) {
  return *reinterpret_cast<const u64*>(reinterpret_cast<const char*>(table) + byte_off);
}

// Global store of one coefficient.  RONK_STORE_MODE: 0 plain (line stays dirty in the XCD's L2 and is written
// back at the kernel boundary), 1 non-temporal, 2 write-through (sc1: a relaxed agent-scope atomic store).
#ifndef RONK_STORE_MODE
#define RONK_STORE_MODE 0
#endif
RONK_HD void st_out(u64* p, u64 v) {
#if defined(__HIP_DEVICE_COMPILE__) && RONK_STORE_MODE == 1
  __builtin_nontemporal_store(v, p);
#elif defined(__HIP_DEVICE_COMPILE__) && RONK_STORE_MODE == 2
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}

// ---- compile-time knowledge about a pass -------------------------------------------------
//
// The generic body reads every stride and flag from TileArgs.  The passes of the two-pass plans (2^13 .. 2^24, the
// headline 2^22 and the batched 2^16 among them) always have the same shape, so the launcher (tile_kernels.hip) picks
// an instantiation that KNOWS it -- the compiler then folds the strides into immediates / SGPR constants and drops the
// unused variants.  tile_cfg_matches() (below) is the host-side check; anything else runs the generic instantiation.
//   KIND 0  generic
//   KIND 1  "column pass" (first pass of a two-pass plan): flat rows, unit column stride on both sides, two-level
//           inter-pass twiddle omega_N^{col * k}, no scale / second operand / limits, full tiles only
//   KIND 2  "row pass" (second pass): rows blocked by the tiled scratch layout, unit row / output-column strides, no
//           twiddle, no scale, full tiles only
//   KIND 3  column pass whose inter-pass twiddle is the full matrix [k][col] (plan.h maybe_full_table; it stays
//           L2-resident up to 2^18 entries: the batched 2^16 shape)
//   KIND 5  whole-polynomial pass (single-pass plans, n = R <= 2^12: the batch is the column axis): rows contiguous, column
//           stride R on both sides (compile-time), no twiddle; the scale of the inverse and a ragged last tile stay run-time
//   KIND 4  general twiddled pass (the phases of the multi-GPU four-step, plan.h build_dist_phase1/2): unit column strides
//           and a two-level output twiddle omega_N^{X*Y} whose X / Y coefficients (column, b2, offsets) stay run-time scalars;
//           rows flat or blocked (the receive buffer of the exchange)
// KIND != 0 also means NARROW addressing: every lane offset fits 32 bits IN BYTES (n*8 < 2^32), so global accesses
// are the `global_load v, v_off, s[base:base+1]` form with offsets built from 32-bit adds.
// LOGC >= 0 fixes the tile width (LDS addresses become immediates); -1 = run-time.
// LDSTW: the round-twiddle table omega_R^e (R entries, 8*R bytes) is staged in LDS behind the tile image at workgroup
// start (the launcher requests 8*R more bytes) and read with ds_read_b64 instead of global loads.  Used where it fits
// beside the image without costing a resident workgroup: the 2^11-row, 8-column tiles (136 + 16 KiB of the CU's 160).
// HALF: the exchanges between rounds go through LDS in two 32-bit phases (low words, then high words) through an image
// of 4-byte cells, so a tile needs half the LDS (68 instead of 136 KiB for 16384 coefficients) and twice as many
// workgroups fit a CU: 8 resident waves per SIMD instead of 4 when the pass has enough tiles (batched plans, several
// streams).  Costs two more barriers per exchange and 32-bit instead of 64-bit LDS instructions; same bank pattern
// (ds_*_b32: 32-lane groups over 32 banks, ds_*_b64: 32-lane groups over 64 banks -- both need the 32 lanes' cell
// indices distinct mod 32).
// FEAT (KIND != 0): what a specialised pass carries beyond the plain shape -- the pieces of a polynomial multiply and of a
// batched Reed-Solomon encode.  A feature that is NOT in the mask is compiled out; one that is must be present at launch.
//   FEAT_IN_VALID   implicit zero padding of the input (in_valid / in_valid1)
//   FEAT_IN2        second operand, pointwise product fused into the load
//   FEAT_OUT_VALID  truncated output
//   FEAT_KEEP       the results stay in the lane's registers -- no output twiddle, no scale, no store (three-round passes
//                   only): register r = g*RLAST + i ends with output row m + M*keep_row_digit(r) of column c.  Never a launch
//                   feature (tile_features() does not report it): the fused multiply (ntt_mul.h) instantiates it directly.
// R4 (round 5): passes of 2^9 / 2^10 rows as [16 . 4] . [8 | 16] instead of (16, 16, 2 | 4) -- three rounds either way, but the
// twiddle after the first round is a SHIFT: with the row group m = a*B + m' (B = R/64 rows per 64-point block, a in [0, 4)) the
// factor omega_R^(m k1) splits into omega_64^(a k1) = +-2^K, applied at once, and omega_R^(m' k1), which does not depend on a,
// commutes with the 4-point transform over a and merges with the next layer: ONE table twiddle omega_R^(m' (k1 + 16 k_a)) per
// pass instead of two.  `a` is the top of the lane's row group: wave-uniform whenever B * C >= 64 (cfg_r4 below), so the shift
// amounts are selected by ONE scalar branch among compile-time mul_2exp bodies.  Measured trade: a table layer costs 26.3
// issue slots per coefficient, the shift layer 12.0 (profiles/r05_arith_variants.txt).  Goldilocks only (the Montgomery
// roots are no powers of two); not with the half-size LDS image, not with FEAT_KEEP.
constexpr int FEAT_IN_VALID = 1, FEAT_IN2 = 2, FEAT_OUT_VALID = 4, FEAT_KEEP = 8;
template <int LOGC_, int KIND_, bool LDSTW_ = false, bool HALF_ = false, int FEAT_ = 0, bool R4_ = false>
struct TileCfg {
  static constexpr int LOGC = LOGC_;
  static constexpr int KIND = KIND_;
  static constexpr bool LDSTW = LDSTW_;
  static constexpr bool HALF = HALF_;
  static constexpr int FEAT = FEAT_;
  static constexpr bool R4 = R4_;
};
// which specialised shapes can run the R4 round structure: the 64-point block's row count B = R/64 times the tile width
// must cover a wavefront, so that the digit a = m / B is wave-uniform
constexpr bool cfg_r4(int logr, int logc, int kind) {
  return (logr == 9 || logr == 10) && kind >= 1 && kind <= 3 && logc >= 0 && (logr - 6 + logc) >= 6;
}
// a value every lane of the wavefront agrees on, as a scalar (device: an SGPR)
RONK_HD u32 wave_uniform(u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}
// x[i] *= omega_64^(A * brev4(i)) (forward) or its inverse: register i of a 16-point DIF round holds output k1 = brev4(i)
template <int A, bool INV, int I = 0>
RONK_HD void shift_layer64(u64* x) {
  if constexpr (I < 16) {
    constexpr int E = root_exp(64, (A * brev(I, 4)) % 64, INV);
    if constexpr (E >= 96) x[I] = gl64::mul_2exp_neg<E - 96>(x[I]);
    else if constexpr (E > 0) x[I] = gl64::mul_2exp<E>(x[I]);
    shift_layer64<A, INV, I + 1>(x);
  }
}
// which specialised instantiations stage their round twiddles in LDS (launcher, emulator and kernel agree through this).
// MEASURED AND SWITCHED OFF (round 2, 2^22, 2^11 x 8 tiles, the only shape where 16 KiB fit beside the image without
// costing a resident workgroup): pass 1 / pass 2 33.8 / 24.4 us with the staged table against 31.1 / 22.9 us with the
// L1/L2-resident table read through wave-uniform bases (same box).  The fill, the extra barrier and
// 30 more ds_read_b64 per lane on the LDS pipe cost more than the global gathers they replace (those cost ~1 us per
// pass: ablation "twiddle values without table loads").  RONK_LDS_TWIDDLES=1 at compile time re-enables it.
#ifndef RONK_LDS_TWIDDLES
#define RONK_LDS_TWIDDLES 0
#endif
constexpr bool cfg_ldstw(int logr, int logc, int kind) {
  return RONK_LDS_TWIDDLES && logr == 11 && logc == 3 && (kind == 1 || kind == 2);
}

// features a launch carries (see FEAT_* above)
inline int tile_features(const TileArgs& a) {
  return (a.in_valid != ~(u64)0 || a.in_valid1 != ~(u64)0 ? FEAT_IN_VALID : 0) | (a.in2 ? FEAT_IN2 : 0) |
         (a.out_valid != ~(u64)0 ? FEAT_OUT_VALID : 0);
}

inline bool tile_cfg_matches(const TileArgs& a, int logr, int logc, int kind, int feat = 0) {
  const u64 C = (u64)1 << a.logc, R = (u64)1 << logr;
  if (logc >= 0 && a.logc != (u32)logc) return false;
  if (kind == 0) return true;
  if (tile_features(a) != feat) return false;
  if (kind == 5)   // whole polynomials side by side; any scale, ragged last tile allowed
    return !a.stage_io && !a.tw_log && !a.tw_full && a.js_log == 31 && a.in_sj == 1 && a.out_sk == 1 && a.in_sc == (i64)R &&
           a.out_sc == (i64)R && a.nb1 == 1 && a.nb2 == 1 && a.tiles == (a.ncols + C - 1) / C && R * C < ((u64)1 << 28);
  const bool common = !a.stage_io && a.scale == 1 && a.ncols % C == 0 && a.tiles == a.ncols / C && !a.xb1 && !a.yb1 &&
                      (kind == 4 || (!a.xb2 && !a.yb2));
  if (!common) return false;
  // NARROW addressing: the largest lane offset on either side (elements, relative to the tile's base) stays below 2^29
  const auto mag = [](i64 v) { return (u64)(v < 0 ? -v : v); };
  const u64 in_span = a.js_log < 31 ? (R >> a.js_log) * mag(a.in_sj_hi) + (((u64)1 << a.js_log) - 1) * mag(a.in_sj) + (C - 1) * mag(a.in_sc)
                                    : (R - 1) * mag(a.in_sj) + (C - 1) * mag(a.in_sc);
  const u64 out_span = (R - 1) * mag(a.out_sk) + (C - 1) * mag(a.out_sc);
  const u64 lim = (u64)1 << 29;
  if (in_span >= lim || out_span >= lim) return false;
  // the padding / truncation limits are compared in bytes against 64-bit sums: any size; (b2, tile) bases are 64-bit scalars
  const bool colpass = a.js_log == 31 && a.in_sc == 1 && a.out_sc == 1 && a.tw_log > 0 && a.tw_log <= 29 && a.xc == 1 &&
                       a.yk == 1 && !a.x0 && !a.y0 && a.in_sj > 0 && a.out_sk > 0;
  if (kind == 5) return false;   // (matched before `common`: see tile_cfg_matches_whole)
  if (kind == 4)
    return !a.tw_full && a.in_sc == 1 && a.out_sc == 1 && a.tw_log > 0 && a.tw_log <= 29 && a.in_sj > 0 && a.out_sk > 0 &&
           (a.js_log == 31 || a.in_sj_hi > 0);
  if (kind == 1) return colpass && !a.tw_full;
  if (kind == 3) return colpass && a.tw_full && a.tf_sc == 1 && a.tf_sb2 == 0 && a.tf_sk == a.ncols && R * a.ncols < lim;
  if (kind == 2)
    return !a.tw_full && (a.js_log == 31 || (R / 16) >= ((u64)1 << a.js_log)) && a.in_sj == 1 && a.out_sc == 1 &&
           a.tw_log == 0 && a.in_sc > 0 && a.out_sk > 0;
  return false;
}

// global access at a 32-bit offset: bytes (NARROW) or elements
template <bool NARROW>
RONK_HD u64 ld_g(const u64* base, u32 off) {
  if constexpr (NARROW) return *reinterpret_cast<const u64*>(reinterpret_cast<const char*>(base) + off);
  else return base[off];
}
template <bool NARROW>
RONK_HD void st_g(u64* base, u32 off, u64 v) {
  if constexpr (NARROW) st_out(reinterpret_cast<u64*>(reinterpret_cast<char*>(base) + off), v);
  else st_out(base + off, v);
}

// ---- the tile body --------------------------------------------------------------------
//
// LOGR = 4*(Q-1) + LOGLAST, Q rounds; radices 16,..,16,2^LOGLAST.
// ABL: ablation mask for tools/ubench only (wrong results by design; 0 in the product):
//   1 no inter-pass twiddle   2 no round twiddles   4 no butterflies   8 no LDS exchange
//   16 no global loads        32 no global stores        64 twiddle values without table loads
// What a tile's lanes know about themselves: the launch arguments with everything the instantiation knows folded in,
// the lane's coordinates and the wave-uniform bases.  Built by tile_ctx() for a (work-item, block) pair; the load
// half (tile_load) and the arithmetic / store half (tile_compute) of a tile both start from it, so that a persistent
// kernel can issue the loads of its NEXT tile before it computes the current one (tile_kernel_def.h, *_pipe).
struct TileCtx {
  TileArgs a;
  u32 logc, C, c, m, t, b1, b2, col0, col;
  const u64* in;
  u64* out;
  u32 in_sj, out_sk, in_lane, out_lane;
  bool live;
};

template <int LOGR, class CFG>
RONK_HD TileCtx tile_ctx(const TileArgs& a_in, u32 tid, u32 bid) {
  constexpr int KIND = CFG::KIND;
  constexpr bool NARROW = KIND != 0;
  constexpr int SH = NARROW ? 3 : 0;             // lane offsets in bytes (NARROW) or elements
  TileCtx x;
  TileArgs& a = x.a;
  a = a_in;                                      // what the instantiation knows replaces what the launch says
  if constexpr (CFG::LOGC >= 0) a.logc = CFG::LOGC;
  if constexpr (KIND != 0) {
    a.stage_io = 0;
    if constexpr (KIND != 5) a.scale = 1;
    if constexpr (!(CFG::FEAT & FEAT_IN2)) a.in2 = nullptr;
    if constexpr (!(CFG::FEAT & FEAT_IN_VALID)) a.in_valid = a.in_valid1 = ~(u64)0;
    if constexpr (!(CFG::FEAT & FEAT_OUT_VALID)) a.out_valid = ~(u64)0;
    if constexpr (KIND != 3) a.tw_full = nullptr;
    if constexpr (KIND != 5) a.out_sc = 1;       // (nb2 / in_sb2 / out_sb2 stay run-time: the middle and last pass of a three-pass plan)
    a.xb1 = a.yb1 = 0;
    if constexpr (KIND != 4) a.xb2 = a.x0 = a.yb2 = a.y0 = 0;
    if constexpr (KIND != 5) a.ncols = (u64)a.tiles << a.logc;
  }
  if constexpr (KIND == 5) {
    a.js_log = 31; a.in_sj = 1; a.out_sk = 1; a.in_sc = a.out_sc = (i64)1 << LOGR; a.tw_log = 0; a.tw_full = nullptr;
    a.nb1 = a.nb2 = 1; a.in_sb1 = a.in_sb2 = a.out_sb1 = a.out_sb2 = 0;
    a.in_st = a.out_st = ((i64)1 << LOGR) << a.logc;
  }
  if constexpr (KIND == 1 || KIND == 3) { a.js_log = 31; a.in_sc = 1; a.xc = 1; a.yk = 1; }
  if constexpr (KIND == 4) a.in_sc = 1;
  if constexpr (KIND == 3) { a.tf_sc = 1; a.tf_sb2 = 0; a.tf_sk = (u32)a.ncols; }
  if constexpr (KIND == 2) { a.in_sj = 1; a.tw_log = 0; }

  x.logc = a.logc;
  x.C = 1u << x.logc;
  x.c = tid & (x.C - 1);
  x.m = tid >> x.logc;  // [0, M)

  // block -> (tile, b1, b2).  Everything that depends only on the block is wave-uniform (SGPRs);
  // per-lane addressing is a 32-bit offset from that base (a sub-problem has < 2^32 elements; NARROW: bytes).
  x.t = bid % a.tiles;
  const u32 bb = bid / a.tiles;
  x.b1 = bb % a.nb1; x.b2 = bb / a.nb1;
  x.col0 = x.t << x.logc;
  x.col = x.col0 + x.c;
  x.in = a.in + (i64)x.b1 * a.in_sb1 + (i64)x.b2 * a.in_sb2 + (i64)x.t * a.in_st;
  x.out = a.out + (i64)x.b1 * a.out_sb1 + (i64)x.b2 * a.out_sb2 + (i64)x.t * a.out_st;
  x.in_sj = (u32)a.in_sj << SH; x.out_sk = (u32)a.out_sk << SH;
  x.in_lane = x.c * ((u32)a.in_sc << SH); x.out_lane = x.c * ((u32)a.out_sc << SH);
  x.live = (KIND != 0 && KIND != 5) ? true : x.col < a.ncols;  // ragged last tile: dead columns compute on zeros
  return x;
}

// ---- the tile body --------------------------------------------------------------------
//
// LOGR = 4*(Q-1) + LOGLAST, Q rounds; radices 16,..,16,2^LOGLAST.
// ABL: ablation mask for tools/ubench only (wrong results by design; 0 in the product):
//   1 no inter-pass twiddle   2 no round twiddles   4 no butterflies   8 no LDS exchange
//   16 no global loads        32 no global stores        64 twiddle values without table loads

// first half: the 16 coefficients of this lane, rows j = i*M + m, straight from HBM (or through the staged image)
template <int LOGR, bool INV, int ABL = 0, class CFG = TileCfg<-1, 0>, class FLD = GlField, class Barrier>
RONK_HD void tile_load(const TileCtx& cx, u64* lds, u32 tid, u64 (&x)[16], Barrier&& barrier) {
  constexpr int R = 1 << LOGR;
  constexpr int M = R / 16;                      // threads per column
  constexpr int KIND = CFG::KIND;
  constexpr bool NARROW = KIND != 0;
  const TileArgs& a = cx.a;
  const u32 logc = cx.logc, c = cx.c, m = cx.m, t = cx.t, b1 = cx.b1, b2 = cx.b2;
  const u64* in = cx.in;
  const u32 in_sj = cx.in_sj, in_lane = cx.in_lane;
  const bool live = cx.live;
  constexpr int SH = NARROW ? 3 : 0;
  const FLD f(a.fc);
  (void)c; (void)tid; (void)lds;

  // ---- round 1: j = j1*M + m, straight from HBM
  u32 joff[16];
  if (a.js_log < 31) {  // blocked rows (tiled scratch of a two-pass plan, multi-GPU receive buffer)
    const u32 jmask = (1u << a.js_log) - 1, hi = (u32)a.in_sj_hi << SH;
    if ((u32)M >= (1u << a.js_log) || KIND == 2) {
      // M = 2^(LOGR-4) rows between a lane's consecutive loads: whole blocks when M >= 2^js_log (always true for the
      // two-pass plans: js_log <= 4 <= LOGR - 4), so the block index advances by a constant and the in-block row is fixed
      const u32 j0 = in_lane + (m >> a.js_log) * hi + (m & jmask) * in_sj, step = ((u32)M >> a.js_log) * hi;
#pragma unroll
      for (int i = 0; i < 16; i++) joff[i] = j0 + i * step;
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const u32 j = i * M + m;
        joff[i] = in_lane + (j >> a.js_log) * hi + (j & jmask) * in_sj;
      }
    }
  } else {
    const u32 j0 = in_lane + m * in_sj, step = M * in_sj;
#pragma unroll
    for (int i = 0; i < 16; i++) joff[i] = j0 + i * step;
  }
  const u32 T = (u32)(R / 16) << logc;                     // work-items of the tile
  const u64 tile_cols = a.ncols - cx.col0 < (u64)cx.C ? a.ncols - cx.col0 : (u64)cx.C;
  const u32 stage_valid = (u32)tile_cols * (u32)R;         // elements of the tile that exist (ragged last tile)
  if (a.stage_io && !(ABL & 16)) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 e = tid + T * i;
      lds[e + (e >> 4)] = e < stage_valid ? in[e] : 0;
    }
    barrier();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 e = c * (u32)R + (u32)(i * M) + m;
      x[i] = lds[e + (e >> 4)];
    }
    barrier();                                             // the image is read before round 1 parks anything
  } else if (ABL & 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (u64)(tid * 16 + i);
  } else if (live && (a.in_valid != ~(u64)0 || a.in_valid1 != ~(u64)0)) {
    // joff is in bytes when NARROW: compare in that unit (an unlimited polynomial -- ~0 -- stays unlimited)
    const u64 lin0 = ((u64)b2 * (u64)a.in_sb2 + (u64)t * (u64)a.in_st) << SH;
    const u64 valid_e = (b1 && a.in_valid1 != ~(u64)0) ? a.in_valid1 : a.in_valid;
    const u64 valid = valid_e >= ((u64)1 << 60) ? ~(u64)0 : valid_e << SH;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (lin0 + joff[i] < valid) ? ld_g<NARROW>(in, joff[i]) : 0;
  } else if (live) {
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = ld_g<NARROW>(in, joff[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = 0;
  }
  if (a.in2 && live) {
    const u64* in2 = a.in2 + (i64)b1 * a.in_sb1 + (i64)b2 * a.in_sb2 + (i64)t * a.in_st;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = f.mul_plain(x[i], ld_g<NARROW>(in2, joff[i]));
  }
}

// second half: the register rounds, the LDS exchanges between them, the output twiddle and the stores
template <int LOGR, bool INV, int ABL = 0, class CFG = TileCfg<-1, 0>, class FLD = GlField, class Barrier>
RONK_HD void tile_compute(const TileCtx& cx, u64* lds, u32 tid, u64 (&x)[16], Barrier&& barrier) {
  constexpr int R = 1 << LOGR;
  // R4: rounds of radix 16, 4, RLAST = R/64 (TileCfg::R4); otherwise radices 16, .., 16, 2^LOGLAST
  constexpr bool R4 = CFG::R4 && (LOGR == 9 || LOGR == 10) && !FLD::MONT && !CFG::HALF && !(CFG::FEAT & FEAT_KEEP);
  constexpr int Q = (LOGR + 3) / 4;              // rounds
  constexpr int LOGLAST = R4 ? LOGR - 6 : LOGR - 4 * (Q - 1);    // 1..4
  constexpr int RLAST = 1 << LOGLAST;
  constexpr int M = R / 16;                      // threads per column
  constexpr int KIND = CFG::KIND;
  constexpr bool NARROW = KIND != 0;
  constexpr int SH = NARROW ? 3 : 0;             // lane offsets in bytes (NARROW) or elements
  static_assert(LOGR >= 4 && LOGR <= 12, "pass size");
  static_assert(!R4 || (CFG::LOGC >= 0 && LOGLAST + CFG::LOGC >= 6), "R4: the digit a = m / RLAST must be wave-uniform");
  const TileArgs& a = cx.a;
  const u32 logc = cx.logc, C = cx.C, c = cx.c, m = cx.m, t = cx.t, b1 = cx.b1, b2 = cx.b2, col0 = cx.col0, col = cx.col;
  u64* out = cx.out;
  const u32 out_sk = cx.out_sk, out_lane = cx.out_lane;
  const bool live = cx.live;
  const FLD f(a.fc);
  (void)C; (void)col0; (void)SH;

  // LDS-staged round twiddles: table behind the image; filled now (its loads are in flight together with the tile's),
  // visible after the barrier that follows the first register round
  constexpr bool LDSTW = CFG::LDSTW && Q > 1;
  constexpr bool HALF = CFG::HALF && Q > 1;      // two-phase 32-bit exchanges (TileCfg); never with staged I/O or LDSTW
  static_assert(!(HALF && LDSTW), "HALF and LDSTW exclude each other");
  u32* const l32 = reinterpret_cast<u32*>(lds);  // the image as 4-byte cells (HALF)
  const u32 IMG = (u32)(R + R / 16) << logc;               // elements of the tile image
  if constexpr (LDSTW) {
    const u32 Tn = (u32)(R / 16) << logc;
    for (u32 e = tid; e < (u32)R; e += Tn) lds[IMG + e] = a.wr[e];
  }
  const u32 T = (u32)(R / 16) << logc;                     // work-items of the tile
  const u64 tile_cols = a.ncols - col0 < (u64)C ? a.ncols - col0 : (u64)C;
  const u32 stage_valid = (u32)tile_cols * (u32)R;         // elements of the tile that exist (ragged last tile)
  (void)T; (void)stage_valid;

  // rounds that are followed by a table twiddle on every output but X[0] take the lazy last stage
  // (R4: a wavefront with a == 0 parks its round-1 results without any twiddle, so they stay canonical)
  if (!(ABL & 4)) { if (Q > 1 && !R4) Dif<16, INV, true, FLD>::run(x, f); else Dif<16, INV, false, FLD>::run(x, f); }

  if (Q > 1) {
    u64* const lc = lds + c;
    // twiddle omega_R^{m*k1}, then park at row k1*M + m.  Two-round tiles (M == RLAST): the row block of k1 is the
    // last round's group with natural output index kl = k1; it is parked at group slot (k1 mod M)*G + k1 div M so that
    // lane m' of the last round (rows 16m' .. 16m'+15) owns the groups kl = m' + g*M -- adjacent output rows come from
    // adjacent lanes of one store instruction (same reasoning as the parking of round 2 below).
    // Table byte offsets (m*k1)*8 for k1 = 1..15 come from an add chain (full-rate v_add_u32) instead of 15 multiplies.
    u32 tb[16];
    tb[0] = 0; tb[1] = m << 3;
#pragma unroll
    for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
    const char* const twl = reinterpret_cast<const char*>(lds + IMG);   // the staged table (LDSTW)
    if constexpr (LDSTW) barrier();
    // LDS element index of (row, c) = (swz_row(row) << logc) + c.  Every access below is written as ONE per-lane base
    // plus a compile-time row constant (<< logc), so the address costs no VALU beyond the base (immediate offsets when
    // the tile width is a compile-time constant).  Q == 3: row k1*M + m, M a multiple of 16: swz = k1*(M + M/16) + m + m/16.
    const u32 park1 = ((m + (m >> 4)) << logc) + c;
    auto p1cell = [&](int i) -> u32 {   // LDS cell that register i of round 1 is parked in
      const u32 k1 = brev(i, 4);
      if (Q == 3 || R4) return park1 + ((u32)(k1 * (M + M / 16)) << logc);
      const u32 blk = (k1 & (M - 1)) * (16 / RLAST) + k1 / M;
      return (swz_row(blk * M + m) << logc) + c;
    };
    (void)lc;
    if constexpr (R4) {
      // omega_R^(m k1) = omega_64^(a k1) * omega_R^(m' k1), m = a*RLAST + m': the first factor is +-2^K with a wave-uniform a
      // (one scalar branch, compile-time shifts), the second waits for the table layer after the 4-point round
      if (!(ABL & 2)) {
        switch (wave_uniform(m >> LOGLAST)) {
          case 0: break;
          case 1: shift_layer64<1, INV>(x); break;
          case 2: shift_layer64<2, INV>(x); break;
          default: shift_layer64<3, INV>(x); break;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (!(ABL & 8)) lds[p1cell(i)] = x[i];
    } else {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 k1 = brev(i, 4);
      if (k1 && !(ABL & 2))
        x[i] = f.mul(x[i], ((ABL & 64) ? ((u64)(m * k1) * 0x9E3779B97F4A7C15ull >> 1)
                                           : LDSTW ? *reinterpret_cast<const u64*>(twl + tb[k1]) : ld_tabb(a.wr, tb[k1])));
      if (!(ABL & 8)) {
        const u32 cell = p1cell(i);
        if (HALF) l32[cell] = (u32)x[i]; else lds[cell] = x[i];
      }
    }
    }
    if (!(ABL & 8)) barrier();

    // round-2 lane coordinates (Q == 3): thread (d1, d3) = (m / RLAST, m % RLAST); its results are parked at p2cell(i)
    const u32 d1 = m >> LOGLAST, d3 = m & (RLAST - 1);
    const u32 park2 = ((17 * d1 + d3) << logc) + c;
    auto p2cell = [&](int i) -> u32 {
      const u32 k2 = brev(i, 4);
      return park2 + ((u32)(272 * (k2 % RLAST) + RLAST * (k2 / RLAST)) << logc);
    };
    if constexpr (R4) {
      // ---- round 2 (R4): lane (g, m') = (m / RLAST, m % RLAST) takes the four 4-point transforms over a for k1 = g + 4t,
      // t = 0..3: register 4t + a <- row k1*M + a*RLAST + m'  (swz: k1*(M + M/16) + a*RLAST + (a*RLAST)/16 + m', since
      // (a*RLAST) mod 16 + m' < 16 for RLAST = 8, 16)
      const u32 rd2 = ((d1 * (u32)(M + M / 16) + d3) << logc) + c;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int tt = i >> 2, aa = i & 3;
        if (!(ABL & 8)) x[i] = lds[rd2 + ((u32)(4 * tt * (M + M / 16) + aa * RLAST + (aa * RLAST) / 16) << logc)];
      }
      if (!(ABL & 8)) barrier();   // not in place: everything is read before anything is parked
      if (!(ABL & 4)) {
#pragma unroll
        for (int tt = 0; tt < 4; tt++) Dif<4, INV, true, FLD>::run(x + 4 * tt, f, false);   // every output meets the table twiddle next
      }
      // omega_R^(m' (k1 + 16 k_a)), k1 = g + 4t, k_a = brev2(a'): byte offsets e0 + (t + 4 k_a) * (32 m') by an add chain
      u32 ch[16];
      ch[0] = (d1 * d3) << 3;
      const u32 st4 = d3 << 5;
#pragma unroll
      for (int j = 1; j < 16; j++) ch[j] = ch[j - 1] + st4;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int tt = i >> 2, ka = brev(i & 3, 2);
        if (!(ABL & 2)) x[i] = f.mul(x[i], (ABL & 64) ? ((u64)ch[tt + 4 * ka] * 0x9E3779B97F4A7C15ull >> 1) : ld_tabb(a.wr, ch[tt + 4 * ka]));
        // park at the last round's group slot of kl = k1 + 16 k_a (same rule as the three-round tiles below):
        // row = 16 (kl mod M) + RLAST (kl div M) + m'  ->  swz = 17 (g + 4t + 16 (k_a mod (M/16))) + RLAST (k_a div (M/16)) + m'
        constexpr int KM = M / 16;   // 4 (RLAST 16) or 2 (RLAST 8)
        if (!(ABL & 8)) lds[park2 + ((u32)(17 * (4 * tt + 16 * (ka % KM)) + RLAST * (ka / KM)) << logc)] = x[i];
      }
      if (!(ABL & 8)) barrier();
    } else if (Q == 3) {
      // ---- round 2: thread (d1, d3) = (m / RLAST, m % RLAST), register digit d2
      // rows d1*16*RLAST + d3 + i*RLAST: swz = (17*RLAST*d1 + d3) + (i*RLAST + i*RLAST/16)     (d3 + (i*RLAST mod 16) < 16)
      const u32 rd2 = ((d1 * (17 * RLAST) + d3) << logc) + c;
      if (HALF && !(ABL & 8)) {
        // low words are in the image: read them, then (barrier) park the high words in the same cells and read those
        u32 lo[16];
#pragma unroll
        for (int i = 0; i < 16; i++) lo[i] = l32[rd2 + ((u32)(i * RLAST + (i * RLAST) / 16) << logc)];
        barrier();
#pragma unroll
        for (int i = 0; i < 16; i++) l32[p1cell(i)] = (u32)(x[i] >> 32);
        barrier();
#pragma unroll
        for (int i = 0; i < 16; i++)
          x[i] = ((u64)l32[rd2 + ((u32)(i * RLAST + (i * RLAST) / 16) << logc)] << 32) | lo[i];
      } else {
#pragma unroll
        for (int i = 0; i < 16; i++)
          if (!(ABL & 8)) x[i] = lds[rd2 + ((u32)(i * RLAST + (i * RLAST) / 16) << logc)];
      }
      // The results are NOT parked back at the rows just read.  The sub-transform with natural output index
      // kl = k1 + 16*k2 (k1 = d1) goes to group slot s = (kl mod M)*G + kl div M, G = 16/RLAST groups per lane, i.e. rows
      // s*RLAST .. s*RLAST + RLAST-1.  Lane m, which owns rows 16m .. 16m+15 in the last round, then holds the groups
      // kl = m + g*M: consecutive lanes finish with consecutive output rows in every store instruction (full lines;
      // with kl = (v >> 4) + 16*(v & 15) they were 16 rows apart, with kl = 2m + g a lane wrote the two halves of a
      // 128-byte line at different times: +37 % write traffic in pass 1).  Not in place: read everything, then write.
      if (!(ABL & 8)) barrier();
      if (!(ABL & 4)) Dif<16, INV, true, FLD>::run(x, f);
      // omega_{R/16}^{d3*k2} = omega_R^{16*d3*k2}: byte offsets (16*d3*k2)*8 by the same add chain
      tb[1] = d3 << 7;
#pragma unroll
      for (int k = 2; k < 16; k++) tb[k] = tb[k - 1] + tb[1];
      const u32 tstep = 16 * d3;
      // slot row = 256*(k2 mod RLAST) + 16*d1 + RLAST*(k2 div RLAST) + d3 (M = 16*RLAST, G*RLAST = 16), and
      // RLAST*(k2 div RLAST) + d3 < 16: swz = (17*d1 + d3) + (272*(k2 mod RLAST) + RLAST*(k2 div RLAST))
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const u32 k2 = brev(i, 4);
        if (k2 && !(ABL & 2))
          x[i] = f.mul(x[i], ((ABL & 64) ? ((u64)(tstep * k2) * 0x9E3779B97F4A7C15ull >> 1)
                                             : LDSTW ? *reinterpret_cast<const u64*>(twl + tb[k2]) : ld_tabb(a.wr, tb[k2])));
        if (!(ABL & 8)) {
          const u32 cell = p2cell(i);
          if (HALF) l32[cell] = (u32)x[i]; else lds[cell] = x[i];
        }
      }
      if (!(ABL & 8)) barrier();
    }

    // ---- last round: thread m owns rows 16m .. 16m+15; row = v*RLAST + d_last
    const u32 rd3 = ((17 * m) << logc) + c;   // swz_row(16m + i) = 17m + i
    if (HALF && !(ABL & 8)) {
      u32 lo[16];
#pragma unroll
      for (int i = 0; i < 16; i++) lo[i] = l32[rd3 + ((u32)i << logc)];
      barrier();
#pragma unroll
      for (int i = 0; i < 16; i++) l32[Q == 3 ? p2cell(i) : p1cell(i)] = (u32)(x[i] >> 32);
      barrier();
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = ((u64)l32[rd3 + ((u32)i << logc)] << 32) | lo[i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (!(ABL & 8)) x[i] = lds[rd3 + ((u32)i << logc)];
    }
  }

  // ---- last sub-DFTs + output, one group of GSZ registers at a time (sub-DFT -> inter-pass twiddle -> scale ->
  // store): the stores of a group are in flight while the next group is still being computed, instead of all
  // 16 stores of every wave landing together at the very end of the workgroup.
  constexpr int GSZ = (Q == 1) ? 16 : RLAST;
  u64* __restrict__ const outp = out;
  const u32 twX = (u32)a.xc * col + (u32)a.xb1 * b1 + (u32)a.xb2 * b2 + (u32)a.x0;
  const u32 twYb = (u32)a.yb1 * b1 + (u32)a.yb2 * b2 + (u32)a.y0;
  const u32 twyk = (u32)a.yk;
  const u32 nmask = a.tw_log >= 32 ? 0xFFFFFFFFu : ((1u << a.tw_log) - 1);
  const u32 lmask = (1u << a.tw_lo_bits) - 1;
  const u64* const tf = (KIND == 3 || a.tw_full) ? a.tw_full : nullptr;
  const u32 tf_lane = (col * a.tf_sc + b2 * a.tf_sb2) << SH, tf_sk = a.tf_sk << SH;
  const u64 lin_out0 = ((u64)b2 * (u64)a.out_sb2 + (u64)t * (u64)a.out_st) << SH;   // in the unit of the lane offsets
  const u64 out_valid = a.out_valid >= ((u64)1 << 60) ? ~(u64)0 : a.out_valid << SH;
  u64 keep = 0;
  if (a.stage_io && Q > 1) barrier();                      // every lane has read its last-round rows: LDS is free
  // Twiddles one group ahead (specialised twiddled passes with more than one group: KIND 1 / 3 / 4).  gfx950 counts vector
  // loads AND stores on one in-order counter (vmcnt): a twiddle load issued after a group's stores can only be waited for
  // together with those stores, i.e. with their full trip to memory -- seven times per tile for a 2^9-row pass (eight groups
  // of two: 105 us for the first pass of a 2^24 transform whose arithmetic takes 65).  So the twiddles of group g+1 are
  // fetched (matrix: loaded; two-level tables: gathered and multiplied together) BEFORE the stores of group g are issued.
  constexpr int NG = 16 / GSZ;
  // (the half-image kernels are built for 64 VGPRs: there only the two-element groups of the 2^9-row passes can afford it)
  constexpr bool TWPIPE = (KIND == 1 || KIND == 3 || KIND == 4) && NG > 1 && !(ABL & 1) && (!CFG::HALF || GSZ <= 2);
  auto fetch_w = [&](int g, u64* w) {   // inter-pass twiddles of group g, in register order (TWPIPE only)
    const u32 kl = m + (u32)g * M;
    if constexpr (KIND == 3) {
      const u32 tbase = tf_lane + kl * tf_sk;
#pragma unroll
      for (int i = 0; i < GSZ; i++) w[i] = ld_g<NARROW>(tf, tbase + (u32)((R / RLAST) * brev(i, LOGLAST)) * tf_sk);
    } else {
      constexpr int ES = NARROW ? 3 : 0;
      u32 ej[GSZ];
      ej[0] = (twX * (twyk * kl + twYb)) << ES;
      const u32 estep = (twX * (twyk * (u32)(R / RLAST))) << ES;
#pragma unroll
      for (int j = 1; j < GSZ; j++) ej[j] = ej[j - 1] + estep;
      const u32 lmask8 = lmask << 3, hmask8 = (nmask >> a.tw_lo_bits) << 3;
#pragma unroll
      for (int i = 0; i < GSZ; i++) {
        const u32 ee = ej[brev(i, LOGLAST)];
        w[i] = f.mul(ld_tabb(a.tw_lo, ee & lmask8), ld_tabb(a.tw_hi, (ee >> a.tw_lo_bits) & hmask8));
      }
    }
  };
  u64 wq[2][GSZ];   // twiddles of the current / the next group (TWPIPE)
  if constexpr (TWPIPE) fetch_w(0, wq[0]);
#pragma unroll
  for (int g = 0; g < 16 / GSZ; g++) {
    u64* xg = x + g * GSZ;
    if constexpr ((CFG::FEAT & FEAT_KEEP) != 0) {   // the sub-transform only: results stay in x (keep_row_digit)
      static_assert(!(CFG::FEAT & FEAT_KEEP) || Q == 3, "FEAT_KEEP: three-round passes");
      if (!(ABL & 4)) Dif<RLAST, INV, false, FLD>::run(xg, f, false);
      continue;
    }
    u32 kg[GSZ];  // natural output row of each register of the group
    // group g of lane m is the sub-transform with natural index kl = m + g*M (see the parking of rounds 1 / 2)
    const u32 kl = (Q == 1) ? 0 : m + (u32)g * M;
    if (Q == 1) {
#pragma unroll
      for (int i = 0; i < GSZ; i++) kg[i] = brev(i, 4);
    } else {
      // column passes of known shape multiply EVERY output by the inter-pass twiddle next: lazy last stage throughout
      if (!(ABL & 4)) Dif<RLAST, INV, (KIND == 1 || KIND == 3) && !(ABL & 1), FLD>::run(xg, f, false);
#pragma unroll
      for (int i = 0; i < GSZ; i++) kg[i] = kl + (R / RLAST) * brev(i, LOGLAST);
    }
    if constexpr (TWPIPE) {
      if (g + 1 < NG) fetch_w(g + 1, wq[(g + 1) & 1]);
#pragma unroll
      for (int i = 0; i < GSZ; i++) xg[i] = f.mul(xg[i], wq[g & 1][i]);
    } else if ((KIND == 3 || tf) && !(ABL & 1)) {   // KIND 3: the matrix is there by construction (tile_cfg_matches): no second variant in the binary
      if (live) {  // dead columns of a ragged tile hold zeros anyway
        // HALF kernels are built for 64 VGPRs: at most 8 table entries in flight at a time
        constexpr int WCH = (CFG::HALF && GSZ > 8) ? 8 : GSZ;
        const u32 tbase = tf_lane + kl * tf_sk;   // like the stores: per-lane base + wave-uniform constants
#pragma unroll
        for (int i0 = 0; i0 < GSZ; i0 += WCH) {
          u64 w[WCH];
#pragma unroll
          for (int i = 0; i < WCH; i++) {
            const u32 ci = (Q == 1) ? (u32)brev(i0 + i, 4) : (u32)((R / RLAST) * brev(i0 + i, LOGLAST));
            w[i] = ld_g<NARROW>(tf, tbase + ci * tf_sk);
          }
#pragma unroll
          for (int i = 0; i < WCH; i++) xg[i0 + i] = f.mul(xg[i0 + i], w[i]);
        }
      }
    } else if (a.tw_log && !(ABL & 1)) {
      // exponent (X*Y) mod 2^tw_log with tw_log <= 32 (the low 32 bits of a 32-bit product suffice); Y is affine in the
      // output row, so the exponents of a group are Ebase + j*Estep, j = natural position inside the group: one
      // multiply per group and an add chain instead of one multiply per coefficient
      constexpr int CSTEP = (Q == 1) ? 1 : (R / RLAST);
      // NARROW (tw_log <= 29 there): the chain runs on exponents pre-scaled by 8, so the two table byte offsets are
      // and / shift+and of full-rate 32-bit ops
      constexpr int ES = NARROW ? 3 : 0;
      u32 ej[GSZ];
      ej[0] = (twX * (twyk * kl + twYb)) << ES;
      const u32 estep = (twX * (twyk * (u32)CSTEP)) << ES;
#pragma unroll
      for (int j = 1; j < GSZ; j++) ej[j] = ej[j - 1] + estep;
      const u32 lmask8 = lmask << 3, hmask8 = (nmask >> a.tw_lo_bits) << 3;
#pragma unroll
      for (int i = 0; i < GSZ; i++) {
        const u32 ee = ej[brev(i, (Q == 1) ? 4 : LOGLAST)];
        u64 wl, wh;
        if constexpr (NARROW) {
          wl = ld_tabb(a.tw_lo, ee & lmask8);
          wh = ld_tabb(a.tw_hi, (ee >> a.tw_lo_bits) & hmask8);
        } else {
          const u32 e = ee & nmask;
          wl = ld_tab(a.tw_lo, e & lmask);
          wh = ld_tab(a.tw_hi, e >> a.tw_lo_bits);
        }
        const u64 w = (ABL & 64) ? f.mul((u64)ee * 0x9E3779B97F4A7C15ull >> 1, (u64)(ee >> 3) * 0xC2B2AE3D27D4EB4Full >> 1)
                                : f.mul(wl, wh);
        xg[i] = f.mul(xg[i], w);
      }
    }
    if (a.scale != 1) {
#pragma unroll
      for (int i = 0; i < GSZ; i++) xg[i] = f.mul(xg[i], a.scale);
    }
    if (a.stage_io && !(ABL & 32)) {
#pragma unroll
      for (int i = 0; i < GSZ; i++) {
        const u32 e = c * (u32)R + kg[i];
        lds[e + (e >> 4)] = xg[i];
      }
    } else if (ABL & 32) {
#pragma unroll
      for (int i = 0; i < GSZ; i++) keep ^= xg[i];
    } else if (live) {
      // offset = out_lane + kg*out_sk = (out_lane + kl*out_sk) + const_i*out_sk: one per-lane base per group, the rest is
      // wave-uniform (SALU)
      const u32 obase = out_lane + kl * out_sk;
#pragma unroll
      for (int i = 0; i < GSZ; i++) {
        const u32 ci = (Q == 1) ? (u32)brev(i, 4) : (u32)((R / RLAST) * brev(i, LOGLAST));
        const u32 off = obase + ci * out_sk;
        if (out_valid == ~(u64)0 || lin_out0 + off < out_valid) st_g<NARROW>(outp, off, xg[i]);
      }
    }
  }
  if (a.stage_io && !(ABL & 32)) {
    barrier();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 e = tid + T * i;
      if (e < stage_valid) st_out(outp + e, lds[e + (e >> 4)]);
    }
  }
  if ((ABL & 32) && keep == 0x123456789ull) outp[0] = keep;  // keeps x live, never true in practice
}

// one tile, start to finish
template <int LOGR, bool INV, int ABL = 0, class CFG = TileCfg<-1, 0>, class FLD = GlField, class Barrier>
RONK_HD void tile_body(const TileArgs& a_in, u64* lds, u32 tid, u32 bid, Barrier&& barrier) {
  const TileCtx cx = tile_ctx<LOGR, CFG>(a_in, tid, bid);
  u64 x[16];
  tile_load<LOGR, INV, ABL, CFG, FLD>(cx, lds, tid, x, barrier);
  tile_compute<LOGR, INV, ABL, CFG, FLD>(cx, lds, tid, x, barrier);
}

}  // namespace ronk
