// plan.h -- host-side decomposition of an n-point NTT (n = 2^k) into tile passes.
//
// Pure C++ (no HIP calls): produces pass descriptors and the twiddle tables they index.
// The same code is used by the library (ronk_plan.hip uploads the tables and launches one
// ntt_tile kernel per pass) and by the host kernel emulator under tests/emu.
//
// Decomposition (natural order in, natural order out; reference semantics
// src/polynomial/mod.rs:273-323 / :430-484, omega = g^((p-1)/n)):
//   k <= 12          one pass, the polynomial batch is the tile's column axis
//   13 <= k <= 24    n = A*B:   pass 1  A-point NTTs down the B columns of [A][B], * omega_n^{b*ka}
//                               pass 2  B-point NTTs along rows, output index ka + A*kb
//   25 <= k <= 36    n = A*B*C: the B*C-point row transforms of pass 2 are split the same way
// Every pass is the same kernel with different strides; exactly one pass transposes
// (rows in, columns out), so it reads its input from a scratch buffer.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include <map>
#include <vector>

#include "gl64.h"
#include "ntt_tile.h"

namespace ronk {

enum Buf { BUF_IN = 0, BUF_OUT = 1, BUF_TMP = 2 };

struct PassDesc {
  int logr;
  TileArgs args;      // pointers left null; filled at launch from the ids below
  Buf in_buf, out_buf;
  int wr_id;          // index into PlanDesc::wr (round twiddles for 2^logr)
  int tw_id;          // index into PlanDesc::tw, or -1
  int twf_id;         // index into PlanDesc::twf (full twiddle matrix), or -1
  u32 grid, block;
  size_t lds_bytes;
  bool small = false;   // run ntt_small.h's latency form (4 coefficients per work-item) instead of the tile kernel
};

struct TwTable {
  int log_n, lo_bits;
  u64 fold = 1;       // canonical factor folded into every entry of `lo` (the inverse's n^-1)
  std::vector<u64> lo, hi;
};

struct PlanDesc {
  int log2n = 0;
  u64 batch = 1;
  bool inverse = false;
  std::vector<PassDesc> passes;
  std::vector<std::vector<u64>> wr;  // one table per distinct logr
  std::vector<int> wr_logr;
  std::vector<TwTable> tw;
  std::vector<std::vector<u64>> twf;  // full inter-pass twiddle matrices (one coalesced load per coefficient)
  bool needs_tmp = false;
};

// The field a plan is built for (host side): Goldilocks with the reference-convention generator 7, or any odd prime
// p < 2^64 with a caller-supplied primitive element g (PrimeField<P>::PRIMITIVE_ELEMENT, prime/mod.rs:87-90).  Host
// arithmetic is canonical; `tab()` converts a twiddle to the form the kernels' tables hold (field_policy.h: canonical for
// Goldilocks, w * 2^64 mod p for Montgomery primes).
struct HostField {
  u64 p = gl64::P, g = gl64::GENERATOR;
  bool mont = false;
  mont64::Field mf{};
  static HostField goldilocks() { return HostField(); }
  static HostField montgomery(u64 p_, u64 g_) {
    HostField h;
    h.p = p_; h.g = g_ % p_; h.mont = true; h.mf = mont64::make_field(p_);
    return h;
  }
  u64 mul(u64 a, u64 b) const { return mont ? (u64)(((unsigned __int128)a * b) % p) : gl64::mul(a, b); }
  u64 pow(u64 a, u64 e) const {
    u64 r = 1 % p;
    while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; }
    return r;
  }
  u64 inv(u64 a) const { return pow(a, p - 2); }
  u64 tab(u64 w) const { return mont ? (u64)((((unsigned __int128)w) << 64) % p) : w; }
  // field/mod.rs:70-75: w = g^((p-1)/n), n = 2^log_n (the caller has checked n | p - 1)
  u64 root_pow2(int log_n, bool inverse) const {
    const u64 w = pow(g, (p - 1) >> log_n);
    return inverse ? inv(w) : w;
  }
  // what the kernels need besides the tables (TileArgs::fc); p == 0 there means Goldilocks
  FieldConst consts(bool inverse) const {
    FieldConst c{};
    if (!mont) return c;
    c.p = p; c.pinv = mf.pinv; c.r2 = mf.r2;
    const u64 w16 = root_pow2(4, inverse);
    u64 x = 1;
    for (int j = 0; j < 8; j++) { c.w16[j] = tab(x); x = mul(x, w16); }
    return c;
  }
};

inline std::vector<u64> power_table(const HostField& hf, u64 w, size_t count, u64 first = 1) {
  std::vector<u64> t(count);
  u64 x = first;
  for (size_t i = 0; i < count; i++) { t[i] = hf.tab(x); x = hf.mul(x, w); }
  return t;
}

struct PlanBuilder {
  PlanDesc d;
  HostField hf;

  int wr_table(int logr) {
    for (size_t i = 0; i < d.wr_logr.size(); i++)
      if (d.wr_logr[i] == logr) return (int)i;
    d.wr_logr.push_back(logr);
    d.wr.push_back(power_table(hf, hf.root_pow2(logr, d.inverse), (size_t)1 << logr));
    if (logr >= 10 && logr <= 12) {
      // behind the R-entry table, the copy ntt_tile_wl.h's column pass reads: entry R + (w * 16 + k1) * 16 + l =
      // omega_R^{(RL l + w) k1}, RL = R / 256 wavefronts -- the first-round twiddles of wavefront w, 16 lanes of a column side by side
      std::vector<u64>& t = d.wr.back();
      const size_t R = (size_t)1 << logr;
      const u32 RL = (u32)(R >> 8);
      t.resize(2 * R);
      for (u32 w = 0; w < RL; w++)
        for (u32 k1 = 0; k1 < 16; k1++)
          for (u32 l = 0; l < 16; l++) t[R + (w * 16 + k1) * 16 + l] = t[((RL * l + w) * k1) & (R - 1)];
    }
    return (int)d.wr.size() - 1;
  }
  // `fold` != 1 pre-multiplies the low-level table: every coefficient receives exactly one inter-pass twiddle,
  // so the inverse transform's n^-1 (F::from(D).inverse(), polynomial/mod.rs:442) rides along for free.
  int tw_table(int log_n, u64 fold = 1) {
    for (size_t i = 0; i < d.tw.size(); i++)
      if (d.tw[i].log_n == log_n && d.tw[i].fold == fold) return (int)i;
    TwTable t;
    t.log_n = log_n;
    t.lo_bits = (log_n + 1) / 2;
    t.fold = fold;
    u64 w = hf.root_pow2(log_n, d.inverse);
    t.lo = power_table(hf, w, (size_t)1 << t.lo_bits, fold);   // fold * w^i
    t.hi = power_table(hf, hf.pow(w, (u64)1 << t.lo_bits), (size_t)1 << (log_n - t.lo_bits));
    d.tw.push_back(t);
    return (int)d.tw.size() - 1;
  }

  // columns per tile for an R-row pass over `ncols` columns
  // wg_floor_log: log2 of the smallest tile in coefficients (12 = 256 work-items; single-pass plans use 10 = one wave)
  static u32 pick_logc(int logr, u64 ncols, int max_logc, int wg_floor_log = 12) {
    int lc = 14 - logr;                 // R*C <= 16384 coefficients (128 KiB LDS, 1024 threads)
    int want = max_logc > wg_floor_log - logr ? max_logc : wg_floor_log - logr;  // small R: widen the workgroup
    if (lc > want) lc = want;
    if (lc < 0) lc = 0;
    while (lc > 0 && ((u64)1 << lc) > ncols) lc--;
    return (u32)lc;
  }

  int multi_pass_floor_log = 12;   // smallest multi-pass tile, log2 coefficients (planner knob, RONK_WG_FLOOR_LOG)
  PassDesc& add_pass(int logr, u64 ncols, int max_logc, int wg_floor_log = 0) {
    if (wg_floor_log == 0) wg_floor_log = multi_pass_floor_log;
    PassDesc p;
    p.logr = logr;
    p.args = TileArgs();
    p.args.in = p.args.in2 = nullptr;
    p.args.out = nullptr;
    p.args.logc = pick_logc(logr, ncols, max_logc, wg_floor_log);
    u64 C = (u64)1 << p.args.logc;
    p.args.tiles = (u32)((ncols + C - 1) / C);
    p.args.ncols = ncols;
    p.args.nb1 = p.args.nb2 = 1;
    p.args.in_sb1 = p.args.in_sb2 = p.args.out_sb1 = p.args.out_sb2 = 0;
    p.args.in_sj_hi = 0; p.args.js_log = 31;
    p.args.in_st = p.args.out_st = 0;
    p.args.tw_log = 0; p.args.tw_lo_bits = 0;
    p.args.tw_lo = p.args.tw_hi = nullptr;
    p.args.xc = p.args.xb1 = p.args.xb2 = p.args.x0 = 0;
    p.args.yk = p.args.yb1 = p.args.yb2 = p.args.y0 = 0;
    p.args.scale = 1;
    p.args.in_valid = p.args.out_valid = p.args.in_valid1 = ~(u64)0;
    p.args.stage_io = 0;
    p.args.fc = hf.consts(d.inverse);
    p.wr_id = wr_table(logr);
    p.args.wr = nullptr;
    p.tw_id = -1;
    p.twf_id = -1;
    p.args.tw_full = nullptr; p.args.tf_sk = p.args.tf_sc = p.args.tf_sb2 = 0;
    p.in_buf = BUF_IN; p.out_buf = BUF_OUT;
    p.block = (u32)((((u64)1 << logr) * C) / 16);
    p.lds_bytes = logr > 4 ? ((((size_t)1 << logr) + ((size_t)1 << logr) / 16) * C * 8) : 0;  // +1 dummy row per 16
    // (the staged-I/O image of a single-pass plan, e + e/16 over R*C elements, has the same size; build_plan sets it for R = 16)
    d.passes.push_back(p);
    return d.passes.back();
  }
  // switch a pass to the latency form (ntt_small.h): 256 work-items x 4 coefficients = R x C with C = 1024/R columns
  static void make_small(PassDesc& p) {
    int lc = 10 - p.logr;
    if (lc < 0) lc = 0;
    while (lc > 0 && ((u64)1 << lc) > p.args.ncols) lc--;
    const u64 C = (u64)1 << lc, R = (u64)1 << p.logr;
    p.args.logc = (u32)lc;
    p.args.tiles = (u32)((p.args.ncols + C - 1) / C);
    p.block = (u32)(R * C / 4);
    p.lds_bytes = (R + R / 4) * C * 8;
    p.small = true;
  }
  int twf_max_log = 0;  // build the full twiddle matrix of a pass when it has at most 2^twf_max_log entries
  // Lay the matrix of a 2^11-row x 4-column column pass out TRANSPOSED, [col][k] instead of [k][col] (round 6): the lanes of
  // ntt_tile_wl.h's last round hold 16 consecutive k per column, so a load instruction reads four whole 128-byte lines instead
  // of sixteen 32-byte pieces of lines that four neighbouring tiles share.  Only ntt_tile_wl.h reads this layout at full speed
  // (tile_wl_matches); everything else falls back to the generic body, so the planner asks for it only where those kernels run.
  bool twf_transposed = false;
  // TileArgs::scale in table form; 1 stays the "no scale" sentinel.  (Montgomery: should s * 2^64 mod p come out as 1 for a real
  // scale s != 1, the representative p + 1 is used -- the product accepts it, p <= 2^64 - 59, and it is not the sentinel.)
  u64 tab_scale(u64 s) const {
    if (s == 1) return 1;
    const u64 t = hf.tab(s);
    return (hf.mont && t == 1) ? hf.p + 1 : t;
  }
  // product of two TABLE-form entries, in table form (Montgomery: (aR)(bR)/R = abR -- what the kernel's mul computes)
  u64 tab_mul(u64 a, u64 b) const { return hf.mont ? mont64::mmul(hf.mf, a, b) : gl64::mul(a, b); }

  // Full matrix of the pass's output twiddle, T[k*tf_sk + col*tf_sc + b2*tf_sb2] = omega_N^{X*Y}, strides chosen
  // like the pass's own output so the load is the same coalesced tile pattern.  Only the indices the
  // exponent depends on get a stride (a batch index it ignores shares the table).
  void maybe_full_table(PassDesc& p) {
    const TileArgs& a = p.args;
    if (!a.tw_log || a.xb1 || a.yb1) return;
    const u64 R = (u64)1 << p.logr;
    const bool dep_c = a.xc != 0, dep_b2 = a.xb2 != 0 || a.yb2 != 0;
    const u64 nc = dep_c ? a.ncols : 1, nb2 = dep_b2 ? a.nb2 : 1;
    const u64 entries = R * nc * nb2;
    if (twf_max_log <= 0 || entries > ((u64)1 << twf_max_log)) return;
    // dense layout [k][b2][col] restricted to the dependent indices -- or [col][k] (twf_transposed, above)
    const bool tr = twf_transposed && p.logr == 11 && a.logc == 2 && dep_c && !dep_b2 && !p.small;
    const u32 sc = tr ? (u32)R : dep_c ? 1 : 0, sb2 = dep_b2 ? (u32)nc : 0, sk = tr ? 1 : (u32)(nc * nb2);
    const TwTable& t = d.tw[p.tw_id];
    const u64 nmask = a.tw_log >= 64 ? ~(u64)0 : (((u64)1 << a.tw_log) - 1), lmask = ((u64)1 << t.lo_bits) - 1;
    std::vector<u64> T(entries);
    for (u64 k = 0; k < R; k++)
      for (u64 b2 = 0; b2 < nb2; b2++)
        for (u64 c = 0; c < nc; c++) {
          const u64 X = a.xc * c + a.xb2 * b2 + a.x0, Y = a.yk * k + a.yb2 * b2 + a.y0;
          const u64 e = (X * Y) & nmask;
          T[k * sk + b2 * sb2 + c * sc] = tab_mul(t.lo[e & lmask], t.hi[e >> t.lo_bits]);
        }
    d.twf.push_back(std::move(T));
    p.twf_id = (int)d.twf.size() - 1;
    p.args.tf_sk = sk; p.args.tf_sc = sc; p.args.tf_sb2 = sb2;
  }
  void finish(PassDesc& p) {
    const i64 C = (i64)1 << p.args.logc;
    if (p.args.in_st == 0) p.args.in_st = C * p.args.in_sc;     // plain matrix: tile t starts at column t*C
    if (p.args.out_st == 0) p.args.out_st = C * p.args.out_sc;
    p.grid = p.args.tiles * p.args.nb1 * p.args.nb2;
    maybe_full_table(p);
  }
};

// max_logc: widest tile (log2 columns) the builder may pick; 4 = 128-byte segments.
// three_pass_from: smallest log2n that is split in three passes (25 = only when two do not fit).
// auto_tiles: the caller left the tile width to the planner -- apply the measured per-pass preferences (below).
// split_ka: rows (log2) of the first pass of a two-pass plan, 0 = the planner's choice (balanced; the fused multiply asks for the
// other split of an odd log2n so that its inverse's column pass has the rows of the forward row pass, ntt_mul.h).
inline PlanDesc build_plan(int log2n, u64 batch, bool inverse, int max_logc = 4, int twf_max_log = 0,
                           int three_pass_from = 25, bool auto_tiles = false, int split_ka = 0,
                           const HostField& hf = HostField(), bool twf_transposed = false) {
  PlanBuilder b;
  b.hf = hf;
  b.twf_max_log = twf_max_log;
  b.twf_transposed = twf_transposed;
  if (const char* e = getenv("RONK_WG_FLOOR_LOG")) { int v = atoi(e); if (v >= 10 && v <= 14) b.multi_pass_floor_log = v; }
  b.d.log2n = log2n; b.d.batch = batch; b.d.inverse = inverse;
  const u64 n = (u64)1 << log2n;
  const u64 scale = inverse ? hf.inv(n % hf.p) : 1;  // F::from(D).inverse(), mod.rs:442
  if (log2n <= 12) {
    // the batch is the column axis: column c = polynomial c, rows contiguous.  Neighbouring columns are n elements
    // apart, so narrow tiles (down to one wave) keep each wave on long contiguous runs of every polynomial.
    PassDesc& p = b.add_pass(log2n, batch, max_logc, 10);
    p.args.in_sj = 1; p.args.in_sc = (i64)n;
    p.args.out_sk = 1; p.args.out_sc = (i64)n;
    p.args.scale = b.tab_scale(scale);
    if (log2n <= 5) {   // n = 16, 32: HBM <-> LDS copies of the contiguous tile (TileArgs::stage_io); n = 16: 112 -> 58 us per
                        // 2^24 coefficients, n = 32: 0.428 -> 0.338 ms per 2^26; n = 64 is better without (0.301 vs 0.346 ms)
      p.args.stage_io = 1;
      const size_t E = (size_t)n << p.args.logc;
      p.lds_bytes = (E + E / 16) * 8;
    }
    b.finish(p);
  } else if (log2n <= 24 && log2n < three_pass_from) {
    // Balanced split, except 2^18: a 2^9-row pass costs three radix rounds (16*16*2), so (10, 8) -- five rounds -- beats
    // (9, 9) -- six -- by 8 % (0.790 -> 0.724 ms per 256 transforms); the other sizes are within 2 % of balanced.
    // RONK_SPLIT_KA overrides (planner experiments).
    int ka = log2n == 18 ? 10 : (log2n + 1) / 2;
    if (const char* e = getenv("RONK_SPLIT_KA")) { int v = atoi(e); if (v >= 4 && v <= 12 && log2n - v >= 4 && log2n - v <= 12) ka = v; }
    if (split_ka >= 4 && split_ka <= 12 && log2n - split_ka >= 4 && log2n - split_ka <= 12) ka = split_ka;
    const int kb = log2n - ka;
    const u64 A = (u64)1 << ka, B = (u64)1 << kb;
    // The scratch buffer between the two passes is stored TILE BY TILE: element (ka, b) lives at
    //   (b / Cp)*(A*Cp) + ka*Cp + (b % Cp),     Cp = columns per pass-1 tile.
    // Pass 1 then writes one contiguous A*Cp block per workgroup (adjacent 64-byte rows pair up into full
    // 128-byte lines inside one workgroup), and pass 2 -- lanes over (ka, 8 consecutive b) -- reads fully
    // contiguous 512-byte runs.  Only the two passes see this layout.
    // Measured tile preference, applied when the tile width is left to the planner: when a pass has
    // many tiles per CU (>= 4 of the largest size), tiles of 8192 coefficients -- two resident workgroups per CU, in
    // different phases -- beat 16384 for 2^10-row passes (2^20 x 64: 0.961 -> 0.826 ms); with one tile per CU
    // (a single 2^22 transform) the large tile stays better, and for 2^8 / 2^9-row passes C = 16 stays best.
    const bool many_tiles = auto_tiles && (double)batch * (double)n / 16384.0 >= 1024.0;
    // Latency regime (ntt_small.h): the whole batch is at most 2^19 .. 2^20 coefficients -- with 16 coefficients per work-item that
    // is at most 512 waves on 1024 SIMDs and the time is one wave's instruction stream.  Measured (forward + inverse, same
    // box): 2^13 34.9 -> 23.3 us, 2^16 35.1 -> 25.9, 2^17 43.8 -> 32.0, 2^18 56.6 -> 38.2, 4 x 2^16 forward 18.1 -> 13.3;
    // one forward 2^19 37.3 -> 26.7, 64 x 2^13 19.1 -> 13.7; at 2^20 coefficients it is a draw or worse (single 47.2 -> 49.3,
    // 16 x 2^16 19.4 -> 22.7).  RONK_SMALL = 0 / 1 forces it.
    // Round 3 (late), both forms as they are now (the latency kernel issues all its loads up front, the tile kernels are
    // specialised; profiles/r03_small_kernel_loads.txt, one transform / batch at a time): up to 2^18 coefficients always
    // (2^18: 15.7 us against 20.8); at 2^19 for n <= 2^17 (4 x 2^17: 20.0 against 22.6; but 2 x 2^18: 23.8 against 22.7, one
    // 2^19: 27.1 against 24.5); at 2^20 for n <= 2^14 (128 x 2^13: 18.6 against 20.9, 64 x 2^14: 19.4 against 21.0; 32 x 2^15 a
    // draw, 16 x 2^16: 21.2 against 19.8).
    const u64 total = (u64)batch * n;
    bool small = auto_tiles && ka <= 10 && kb <= 10 &&
                 (total <= ((u64)1 << 18) || (total <= ((u64)1 << 19) && log2n <= 17) || (total <= ((u64)1 << 20) && log2n <= 14));
    if (const char* e = getenv("RONK_SMALL")) small = atoi(e) != 0 && ka <= 10 && kb <= 10;
    int lc1 = max_logc, lc2 = max_logc;
    // Round 3, re-measured with the specialised kernels and HBM-cold buffers (bench.py --mode batch --rotate 8, same box,
    // profiles/r03_planner_sweep.txt): 2^10-row passes keep the 8192-coefficient tiles (2^20 x 64: 92.3 k NTT/s against
    // 78.4 k with 16 columns), but 2^11-row passes are better off with the full 16384-coefficient tile (2^22 x 16: 20.9 k
    // against 19.2 k with 4 columns; 2^21 x 32: 47.5 k / 47.3 k) -- a 4-column tile reads 32-byte row segments.
    if (many_tiles && ka == 10 && lc1 > 3) lc1 = 3;
    if (many_tiles && kb == 10 && lc2 > 3) lc2 = 3;
    // Few tiles (one transform of 2^20 / 2^21, small batches of them): narrower tiles until every CU has one -- a pass of 64
    // workgroups leaves three quarters of the chip idle (2^20: 47.1 -> 29.8 us, 2^21: 50.5 -> 41.1 us with 4-column tiles).
    if (auto_tiles && !small) {
      auto narrow = [&](int logr, u64 ncols, int lc) {
        int eff = (int)PlanBuilder::pick_logc(logr, ncols, lc, b.multi_pass_floor_log);
        while (eff > 2 && (double)batch * (double)(ncols >> eff) < 256.0) eff--;
        return eff < lc ? eff : lc;
      };
      lc1 = narrow(ka, B, lc1);
      lc2 = narrow(kb, A, lc2);
    }
    u32 logcp;
    {
      PassDesc& p = b.add_pass(ka, B, lc1);  // [A][B]: columns b, rows a
      if (small) PlanBuilder::make_small(p);
      logcp = p.args.logc;
      const i64 Cp = (i64)1 << logcp;
      p.args.in_sj = (i64)B; p.args.in_sc = 1;
      p.args.out_sk = Cp; p.args.out_sc = 1; p.args.out_st = (i64)A * Cp;
      p.args.nb1 = (u32)batch; p.args.in_sb1 = p.args.out_sb1 = (i64)n;
      p.tw_id = b.tw_table(log2n, scale);        // n^-1 of the inverse folded in
      p.args.tw_log = log2n; p.args.tw_lo_bits = b.d.tw[p.tw_id].lo_bits;
      p.args.xc = 1; p.args.yk = 1;              // omega_n^{b * ka}
      p.in_buf = BUF_IN; p.out_buf = BUF_TMP;
      b.finish(p);
    }
    {
      PassDesc& p = b.add_pass(kb, A, lc2);  // rows ka are the columns of this pass
      if (small) PlanBuilder::make_small(p);
      const i64 Cp = (i64)1 << logcp, C2 = (i64)1 << p.args.logc;
      // j = b: (b >> logcp) selects the pass-1 tile, (b & (Cp-1)) the column inside it
      p.args.in_sj = 1; p.args.js_log = logcp; p.args.in_sj_hi = (i64)A * Cp;
      p.args.in_sc = Cp; p.args.in_st = C2 * Cp;
      p.args.out_sk = (i64)A; p.args.out_sc = 1;
      p.args.nb1 = (u32)batch; p.args.in_sb1 = p.args.out_sb1 = (i64)n;
      p.in_buf = BUF_TMP; p.out_buf = BUF_OUT;
      b.finish(p);
    }
    b.d.needs_tmp = true;
  } else {
    int ka = (log2n + 2) / 3, kb = (log2n - ka + 1) / 2;
    // 2^24: (9, 8, 7) instead of the balanced (8, 8, 8) -- the first pass (row stride B*C elements on both sides) is the slow
    // one of every three-pass plan, and 2^9-row tiles make it 109 instead of 125 us (0.226 -> 0.210 ms; round 3 sweep with the
    // specialised kernels, profiles/r03_three_pass_splits.txt; 2^25 / 2^26: the balanced split stays best)
    if (log2n == 24) { ka = 9; kb = 8; }
    if (const char* e = getenv("RONK_SPLIT3")) {   // "ka,kb" (planner experiments)
      int va = 0, vb = 0;
      if (sscanf(e, "%d,%d", &va, &vb) == 2 && va >= 4 && va <= 12 && vb >= 4 && vb <= 12 && log2n - va - vb >= 4 &&
          log2n - va - vb <= 12) { ka = va; kb = vb; }
    }
    const int kc = log2n - ka - kb;
    const u64 A = (u64)1 << ka, B = (u64)1 << kb, C = (u64)1 << kc, BC = B * C;
    {
      PassDesc& p = b.add_pass(ka, BC, max_logc);  // [A][B*C]
      p.args.in_sj = (i64)BC; p.args.in_sc = 1; p.args.out_sk = (i64)BC; p.args.out_sc = 1;
      p.args.nb1 = (u32)batch; p.args.in_sb1 = p.args.out_sb1 = (i64)n;
      p.tw_id = b.tw_table(log2n, scale);        // n^-1 of the inverse folded in
      p.args.tw_log = log2n; p.args.tw_lo_bits = b.d.tw[p.tw_id].lo_bits;
      p.args.xc = 1; p.args.yk = 1;
      p.in_buf = BUF_IN; p.out_buf = BUF_TMP;
      b.finish(p);
    }
    {
      PassDesc& p = b.add_pass(kb, C, max_logc);  // inside row ka: [B][C], in place
      p.args.in_sj = (i64)C; p.args.in_sc = 1; p.args.out_sk = (i64)C; p.args.out_sc = 1;
      p.args.nb1 = (u32)batch; p.args.in_sb1 = p.args.out_sb1 = (i64)n;
      p.args.nb2 = (u32)A; p.args.in_sb2 = p.args.out_sb2 = (i64)BC;
      p.tw_id = b.tw_table(kb + kc);
      p.args.tw_log = kb + kc; p.args.tw_lo_bits = b.d.tw[p.tw_id].lo_bits;
      p.args.xc = 1; p.args.yk = 1;              // omega_{BC}^{c * kb}
      p.in_buf = BUF_TMP; p.out_buf = BUF_TMP;
      b.finish(p);
    }
    {
      PassDesc& p = b.add_pass(kc, A, max_logc);  // columns ka, batch kb; k = ka + A*(kb + B*kc)
      p.args.in_sj = 1; p.args.in_sc = (i64)BC; p.args.out_sk = (i64)(A * B); p.args.out_sc = 1;
      p.args.nb1 = (u32)batch; p.args.in_sb1 = p.args.out_sb1 = (i64)n;
      p.args.nb2 = (u32)B; p.args.in_sb2 = (i64)C; p.args.out_sb2 = (i64)A;
      p.in_buf = BUF_TMP; p.out_buf = BUF_OUT;
      b.finish(p);
    }
    b.d.needs_tmp = true;
  }
  return b.d;
}

// ---- multi-GPU four-step (one process per GPU; the exchange between the two phases is an
// all-to-all over xGMI issued by the host side).  n = R*C, R = 2^(k - k/2), C = 2^(k/2);
// input index i = r*C + c, output index k = k1 + R*k2.  Rank g of W owns columns
// [g*C/W, (g+1)*C/W) as [R][C/W]; phase 1 = R-point NTTs down those columns times
// omega_n^{c*k1}, written as [R][C/W] (= W consecutive send blocks of R/W rows);
// after the exchange rank h holds W blocks [R/W][C/W] (block g from rank g) and phase 2 =
// C-point NTTs along each of its R/W rows, written as out[k2*(R/W) + k1_local].
struct DistShape {
  int log2n, logR, logC, logW;
  u64 n, R, C, W, Rw, Cw;
};
inline bool dist_shape(int log2n, int world, DistShape* s) {
  int lw = 0;
  while ((1 << lw) < world) lw++;
  if ((1 << lw) != world) return false;
  s->log2n = log2n; s->logC = log2n / 2; s->logR = log2n - s->logC; s->logW = lw;
  s->n = (u64)1 << log2n; s->R = (u64)1 << s->logR; s->C = (u64)1 << s->logC; s->W = (u64)world;
  if (s->logC < lw + 4 || s->logR > 24 || s->logC > 24) return false;  // >= 16 columns and rows per rank
  // two-pass phase 2 (logC > 12) splits the received row index at logC - logW - kb bits, kb = floor(logC/2)
  if (s->logC > 12 && s->logC - lw < s->logC / 2) return false;
  s->Rw = s->R / s->W; s->Cw = s->C / s->W;
  return true;
}

// Column chunks (exchange overlap): a rank's C/W columns are split in `chunks` groups of Cwc = C/W/chunks
// columns; phase 1 of chunk j reads columns [j*Cwc, (j+1)*Cwc) of the local [R][C/W] input and writes a CONTIGUOUS
// [R][Cwc] piece of the send buffer at offset j*R*Cwc, i.e. W blocks [R/W][Cwc], block h for rank h -- so chunk j can
// be shipped (W contiguous blocks) while chunk j+1 is still being computed.  The receiver stores block (g, j) at
// (g*chunks + j)*(R/W)*Cwc: with b = c / Cwc the global column c = g*C/W + j*Cwc + cc lives at b*(R/W)*Cwc + k1*Cwc + cc,
// which is the two-level row addressing of TileArgs (js_log = log2 Cwc).  chunks == 1 is the plain layout.
inline bool dist_chunks_ok(const DistShape& sh, int chunks) {
  if (chunks < 1 || (chunks & (chunks - 1))) return false;
  if (sh.Cw % (u64)chunks) return false;
  const u64 cwc = sh.Cw / (u64)chunks;
  if (cwc < 16) return false;
  if (sh.logC > 12) {   // two-pass phase 2: the received row index splits at log2(cwc) - kb bits
    int lcwc = 0; while (((u64)1 << lcwc) < cwc) lcwc++;
    if (lcwc < sh.logC / 2) return false;
  }
  return true;
}

inline PlanDesc build_dist_phase1(int log2n, bool inverse, int rank, int world, int max_logc = 4, int twf_max_log = 0,
                                  int chunk = 0, int chunks = 1, const HostField& hf = HostField()) {
  DistShape sh;
  PlanBuilder b;
  b.hf = hf;
  b.twf_max_log = twf_max_log;
  b.d.log2n = log2n; b.d.inverse = inverse;
  if (!dist_shape(log2n, world, &sh) || !dist_chunks_ok(sh, chunks) || chunk < 0 || chunk >= chunks) return b.d;
  const u64 Cw = sh.Cw, Cwc = Cw / (u64)chunks;       // input row stride / columns of this chunk = output row stride
  const u64 g0 = (u64)rank * Cw + (u64)chunk * Cwc;   // global index of the chunk's first column
  const u64 scale = inverse ? hf.inv(sh.n % hf.p) : 1;   // folded into the global twiddle of phase 1
  // the launcher adds chunk*Cwc to the input pointer and chunk*R*Cwc to the output pointer (ronk_dist.hip)
  if (sh.logR <= 12) {
    PassDesc& p = b.add_pass(sh.logR, Cwc, max_logc);
    p.args.in_sj = (i64)Cw; p.args.in_sc = 1; p.args.out_sk = (i64)Cwc; p.args.out_sc = 1;
    p.tw_id = b.tw_table(log2n, scale);
    p.args.tw_log = log2n; p.args.tw_lo_bits = b.d.tw[p.tw_id].lo_bits;
    p.args.xc = 1; p.args.x0 = g0; p.args.yk = 1;            // omega_n^{(g0 + cl) * k1}
    p.in_buf = BUF_IN; p.out_buf = BUF_OUT;
    b.finish(p);
  } else {
    const int ka = (sh.logR + 1) / 2, kb = sh.logR - ka;
    const u64 A = (u64)1 << ka, B = (u64)1 << kb;
    {
      PassDesc& p = b.add_pass(ka, Cwc, max_logc);            // r = a*B + b: A-point over a, batch b; tmp is [R][Cwc]
      p.args.in_sj = (i64)(B * Cw); p.args.in_sc = 1; p.args.out_sk = (i64)(B * Cwc); p.args.out_sc = 1;
      p.args.nb2 = (u32)B; p.args.in_sb2 = (i64)Cw; p.args.out_sb2 = (i64)Cwc;
      p.tw_id = b.tw_table(sh.logR);
      p.args.tw_log = sh.logR; p.args.tw_lo_bits = b.d.tw[p.tw_id].lo_bits;
      p.args.xb2 = 1; p.args.yk = 1;                          // omega_R^{b * ka}
      p.in_buf = BUF_IN; p.out_buf = BUF_TMP;
      b.finish(p);
    }
    {
      PassDesc& p = b.add_pass(kb, Cwc, max_logc);            // B-point over b, batch ka; k1 = ka + A*kb
      p.args.in_sj = (i64)Cwc; p.args.in_sc = 1; p.args.out_sk = (i64)(A * Cwc); p.args.out_sc = 1;
      p.args.nb2 = (u32)A; p.args.in_sb2 = (i64)(B * Cwc); p.args.out_sb2 = (i64)Cwc;
      p.tw_id = b.tw_table(log2n, scale);
      p.args.tw_log = log2n; p.args.tw_lo_bits = b.d.tw[p.tw_id].lo_bits;
      p.args.xc = 1; p.args.x0 = g0; p.args.yk = A; p.args.yb2 = 1;   // omega_n^{c * (ka + A*kb)}
      p.in_buf = BUF_TMP; p.out_buf = BUF_OUT;
      b.finish(p);
    }
    b.d.needs_tmp = true;
  }
  return b.d;
}

inline PlanDesc build_dist_phase2(int log2n, bool inverse, int rank, int world, int max_logc = 4, int twf_max_log = 0,
                                  int chunks = 1, const HostField& hf = HostField()) {
  (void)rank;
  DistShape sh;
  PlanBuilder b;
  b.hf = hf;
  b.twf_max_log = twf_max_log;
  b.d.log2n = log2n; b.d.inverse = inverse;
  if (!dist_shape(log2n, world, &sh) || !dist_chunks_ok(sh, chunks)) return b.d;
  const u64 Cw = sh.Cw, Rw = sh.Rw, C = sh.C, Cwc = Cw / (u64)chunks;
  int lcwc = 0; while (((u64)1 << lcwc) < Cwc) lcwc++;
  const u64 scale = 1;  // the inverse's n^-1 is applied by phase 1 (folded into its global twiddle)
  if (sh.logC <= 12) {
    PassDesc& p = b.add_pass(sh.logC, Rw, max_logc);          // columns = local rows k1, rows j = c (blocked)
    p.args.in_sc = (i64)Cwc; p.args.in_sj = 1;
    p.args.js_log = (u32)lcwc; p.args.in_sj_hi = (i64)(Rw * Cwc);
    p.args.out_sk = (i64)Rw; p.args.out_sc = 1;
    p.args.scale = scale;
    p.in_buf = BUF_IN; p.out_buf = BUF_OUT;
    b.finish(p);
  } else {
    const int ka = (sh.logC + 1) / 2, kb = sh.logC - ka;
    const u64 A2 = (u64)1 << ka, B2 = (u64)1 << kb;
    {
      PassDesc& p = b.add_pass(ka, B2, max_logc);             // c = a2*B2 + b2: A2-point over a2; batch k1
      p.args.in_sc = 1; p.args.in_sj = (i64)B2;
      p.args.js_log = (u32)(lcwc - kb); p.args.in_sj_hi = (i64)(Rw * Cwc);
      p.args.nb2 = (u32)Rw; p.args.in_sb2 = (i64)Cwc;
      p.args.out_sb2 = (i64)C; p.args.out_sk = (i64)B2; p.args.out_sc = 1;   // tmp: natural [k1][a2][b2]
      p.tw_id = b.tw_table(sh.logC);
      p.args.tw_log = sh.logC; p.args.tw_lo_bits = b.d.tw[p.tw_id].lo_bits;
      p.args.xc = 1; p.args.yk = 1;                           // omega_C^{b2 * ka2}
      p.in_buf = BUF_IN; p.out_buf = BUF_TMP;
      b.finish(p);
    }
    {
      PassDesc& p = b.add_pass(kb, Rw, max_logc);             // B2-point over b2; columns k1; batch ka2
      p.args.in_sc = (i64)C; p.args.in_sj = 1;
      p.args.nb2 = (u32)A2; p.args.in_sb2 = (i64)B2;
      p.args.out_sc = 1; p.args.out_sb2 = (i64)Rw; p.args.out_sk = (i64)(A2 * Rw);   // k2 = ka2 + A2*kb2
      p.args.scale = scale;
      p.in_buf = BUF_TMP; p.out_buf = BUF_OUT;
      b.finish(p);
    }
    b.d.needs_tmp = true;
  }
  return b.d;
}

}  // namespace ronk
