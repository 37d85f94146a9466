// emu_tile.cpp -- HOST EMULATOR of the HIP tile kernel (TEST INFRASTRUCTURE ONLY).
//
// Runs the very same ronk::tile_body<> template that ronk_plan.hip launches on the GPU,
// but on ucontext fibers (one fiber per work-item, barrier = yield), so the plan algebra
// (strides, digit maps, twiddle exponents, LDS swizzle) can be checked against the oracle
// in the CPU-only container.  It is built and run by tests/test_emu_kernel.py only; the
// product library never contains or calls it.
//
// usage: emu_tile <log2n> <batch> <inverse 0|1> <max_logc> [twf_max_log] [three_pass_from] [in_valid] [out_valid] [auto_tiles] [in_valid1]
//        (prints OK or the first mismatch)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <vector>

#include "../../oracle/ronk_oracle.h"
#include "../../ronkathon_amd/csrc/plan.h"
#include "../../ronkathon_amd/csrc/ntt_small.h"
#include "../../ronkathon_amd/csrc/ntt_mul.h"
#include "../../ronkathon_amd/csrc/tile_cfg_table.h"
#include "../../ronkathon_amd/csrc/ntt_tile_wl.h"

using namespace ronk;

static ucontext_t g_sched;
static std::vector<ucontext_t> g_ctx;
static std::vector<char> g_stacks;
static std::vector<char> g_done;
static int g_cur;

struct FiberArgs {
  const TileArgs* a; u64* lds; u32 bid; int logr; bool inv; bool small;
};
static FiberArgs g_fa;

static void fiber_barrier() { swapcontext(&g_ctx[g_cur], &g_sched); }

// RONK_EMU_P / RONK_EMU_G: run the Montgomery instantiations (field_policy.h MontField) for this odd prime and primitive
// element instead of the Goldilocks ones -- the same plan builder, tables in Montgomery form, the oracle called with (p, g)
static u64 g_p = gl64::P, g_g = gl64::GENERATOR;
static bool g_mont = false;
static HostField g_hf;
static u64 rnd_elem(u64& s);

template <int LOGR, bool INV, class FLD>
static void run_body(u32 tid) { tile_body<LOGR, INV, 0, TileCfg<-1, 0>, FLD>(*g_fa.a, g_fa.lds, tid, g_fa.bid, fiber_barrier); }

// the compile-time-specialised instantiations the library launches for recognised pass shapes (tile_kernels.hip):
// same selection rule, same table
static int g_cfg_used = 0;
template <bool INV, class FLD>
static bool dispatch_cfg(int logr, u32 tid) {
  const TileArgs& a = *g_fa.a;
  static const bool half = getenv("RONK_HALF_LDS") && atoi(getenv("RONK_HALF_LDS")) == 1;   // TileCfg::HALF instantiations
  // the wave-local bodies of the 2^11-row x 4-column passes (ntt_tile_wl.h), selected like the library does: RONK_WL = 0 off, 1
  // (default) both passes, 2 column pass only, 3 row pass only; RONK_WL_HALF=1 the half-image form (Goldilocks)
  {
    static const int wl = getenv("RONK_WL") ? atoi(getenv("RONK_WL")) : 1;
    static const bool wl_half = getenv("RONK_WL_HALF") && atoi(getenv("RONK_WL_HALF")) != 0;
    static const bool r4_first = getenv("RONK_R4MID") && atoi(getenv("RONK_R4MID")) != 0;   // the opt-in below wins at 2^10 rows
    if (wl && wl_logr_ok(logr) && (int)a.logc == WL_LOGC && !(r4_first && logr == 10)) {
      for (int kind : {1, 2, 3}) {
        if (!tile_wl_matches(a, logr, kind)) continue;
        if (kind == 2 ? wl == 2 : wl == 3) continue;
        u32* l32 = reinterpret_cast<u32*>(g_fa.lds);
#define EMU_WL_RUN(LR, FULLIMG)                                                                                                  \
  do {                                                                                                                           \
    if (kind == 1) tile_body_wl_col<LR, INV, 1, FULLIMG, FLD>(a, l32, tid, g_fa.bid, fiber_barrier, fiber_barrier);              \
    else if (kind == 3) tile_body_wl_col<LR, INV, 3, FULLIMG, FLD>(a, l32, tid, g_fa.bid, fiber_barrier, fiber_barrier);         \
    else tile_body_wl_row<LR, INV, FULLIMG, FLD>(a, l32, tid, g_fa.bid, fiber_barrier, fiber_barrier);                           \
  } while (0)
        if (wl_half && !FLD::MONT && logr == 11) EMU_WL_RUN(11, false);
        else if (logr == 10) EMU_WL_RUN(10, true);
        else if (logr == 11) EMU_WL_RUN(11, true);
        else EMU_WL_RUN(12, true);
#undef EMU_WL_RUN
        g_cfg_used = 30 + kind;
        return true;
      }
    }
  }
  if constexpr (FLD::MONT) {   // the shapes the library instantiates for Montgomery primes (tile_kernels_mont.hip): the plain table
    if (const int feat = tile_features(a)) {   // ... and the feature shapes in the direction they occur in (tile_kernels_mont_feat.hip)
#define EMU_MONT_FEAT_CASE(LR, LC, KD, FT)                                                       \
  if (logr == LR && (int)a.logc == LC && feat == FT && INV == (FT != 1) && tile_cfg_matches(a, LR, LC, KD, FT)) { \
    tile_body<LR, (FT != 1), 0, TileCfg<LC, KD, false, false, FT>, FLD>(a, g_fa.lds, tid, g_fa.bid, fiber_barrier); \
    g_cfg_used = KD + 100 * FT;                                                                  \
    return true;                                                                                 \
  }
      RONK_CFG_TABLE_FEAT(EMU_MONT_FEAT_CASE)
#undef EMU_MONT_FEAT_CASE
      return false;
    }
#define EMU_MONT_CASE(LR, LC, KD)                                                                \
  if (logr == LR && (int)a.logc == LC && tile_cfg_matches(a, LR, LC, KD)) {                      \
    tile_body<LR, INV, 0, TileCfg<LC, KD>, FLD>(a, g_fa.lds, tid, g_fa.bid, fiber_barrier);      \
    g_cfg_used = KD;                                                                             \
    return true;                                                                                 \
  }
    RONK_CFG_TABLE(EMU_MONT_CASE)
#undef EMU_MONT_CASE
    return false;
  } else {
  if (half) {
#define EMU_HALF_CASE(LR, LC, KD)                                                                \
  if (logr == LR && (int)a.logc == LC && tile_cfg_matches(a, LR, LC, KD)) {                      \
    tile_body<LR, INV, 0, TileCfg<LC, KD, false, true>>(a, g_fa.lds, tid, g_fa.bid, fiber_barrier); \
    g_cfg_used = KD + 10;                                                                        \
    return true;                                                                                 \
  }
    RONK_CFG_TABLE(EMU_HALF_CASE)
#undef EMU_HALF_CASE
  }
  {
    const int feat = tile_features(a);
#define EMU_FEAT_CASE(LR, LC, KD, FT)                                                            \
  if (logr == LR && (int)a.logc == LC && feat == FT && tile_cfg_matches(a, LR, LC, KD, FT)) {    \
    tile_body<LR, INV, 0, TileCfg<LC, KD, false, false, FT>>(a, g_fa.lds, tid, g_fa.bid, fiber_barrier); \
    g_cfg_used = KD + 100 * FT;                                                                  \
    return true;                                                                                 \
  }
    RONK_CFG_TABLE_FEAT(EMU_FEAT_CASE)
#undef EMU_FEAT_CASE
    if (feat) return false;
  }
  // the R4 round structure of the 2^9 / 2^10-row shapes (tile_kernels_r4.hip; opt-in with RONK_R4MID=1 like the library)
  static const bool r4_on = getenv("RONK_R4MID") && atoi(getenv("RONK_R4MID")) != 0;
  if (r4_on && (logr == 9 || logr == 10)) {
#define EMU_R4_CASE(LR, LC, KD)                                                                  \
  if constexpr (cfg_r4(LR, LC, KD)) {                                                            \
    if (logr == LR && (int)a.logc == LC && KD < 4 && tile_cfg_matches(a, LR, LC, KD)) {          \
      tile_body<LR, INV, 0, TileCfg<LC, KD, false, false, 0, true>>(a, g_fa.lds, tid, g_fa.bid, fiber_barrier); \
      g_cfg_used = KD + 20;                                                                      \
      return true;                                                                               \
    }                                                                                            \
  }
    RONK_CFG_TABLE(EMU_R4_CASE)
#undef EMU_R4_CASE
  }
#define EMU_CFG_CASE(LR, LC, KD)                                                                 \
  if (logr == LR && (int)a.logc == LC && tile_cfg_matches(a, LR, LC, KD)) {                      \
    tile_body<LR, INV, 0, TileCfg<LC, KD, cfg_ldstw(LR, LC, KD)>>(a, g_fa.lds, tid, g_fa.bid, fiber_barrier); \
    g_cfg_used = KD;                                                                             \
    return true;                                                                                 \
  }
  RONK_CFG_TABLE(EMU_CFG_CASE)
  RONK_CFG_TABLE_DIST(EMU_CFG_CASE)
#undef EMU_CFG_CASE
  return false;
  }
}

template <bool INV, class FLD>
static void dispatch_small(int logr, u32 tid) {
  switch (logr) {
#define EMU_SMALL_CASE(LR) case LR: small_body<LR, INV, FLD>(*g_fa.a, g_fa.lds, tid, g_fa.bid, fiber_barrier); break;
    EMU_SMALL_CASE(4) EMU_SMALL_CASE(5) EMU_SMALL_CASE(6) EMU_SMALL_CASE(7) EMU_SMALL_CASE(8) EMU_SMALL_CASE(9) EMU_SMALL_CASE(10)
#undef EMU_SMALL_CASE
    default: abort();
  }
}

template <bool INV, class FLD>
static void dispatch(int logr, u32 tid) {
  if (g_fa.small) { dispatch_small<INV, FLD>(logr, tid); return; }
  if (!getenv("RONK_NO_CFG_KERNELS") && dispatch_cfg<INV, FLD>(logr, tid)) return;
  switch (logr) {
    case 4: run_body<4, INV, FLD>(tid); break;
    case 5: run_body<5, INV, FLD>(tid); break;
    case 6: run_body<6, INV, FLD>(tid); break;
    case 7: run_body<7, INV, FLD>(tid); break;
    case 8: run_body<8, INV, FLD>(tid); break;
    case 9: run_body<9, INV, FLD>(tid); break;
    case 10: run_body<10, INV, FLD>(tid); break;
    case 11: run_body<11, INV, FLD>(tid); break;
    case 12: run_body<12, INV, FLD>(tid); break;
    default: abort();
  }
}

static void fiber_main(int tid) {
  if (g_mont) { if (g_fa.inv) dispatch<true, MontField>(g_fa.logr, (u32)tid); else dispatch<false, MontField>(g_fa.logr, (u32)tid); }
  else if (g_fa.inv) dispatch<true, GlField>(g_fa.logr, (u32)tid); else dispatch<false, GlField>(g_fa.logr, (u32)tid);
  g_done[tid] = 1;
  swapcontext(&g_ctx[tid], &g_sched);
}

static void run_block(u32 T) {
  const size_t STK = 64 * 1024;
  if (g_ctx.size() < T) { g_ctx.resize(T); g_stacks.resize((size_t)T * STK); g_done.resize(T); }
  for (u32 t = 0; t < T; t++) {
    getcontext(&g_ctx[t]);
    g_ctx[t].uc_stack.ss_sp = &g_stacks[(size_t)t * STK];
    g_ctx[t].uc_stack.ss_size = STK;
    g_ctx[t].uc_link = &g_sched;
    makecontext(&g_ctx[t], (void (*)())fiber_main, 1, (int)t);
    g_done[t] = 0;
  }
  for (;;) {
    bool any = false;
    for (u32 t = 0; t < T; t++) {
      if (g_done[t]) continue;
      any = true;
      g_cur = (int)t;
      swapcontext(&g_sched, &g_ctx[t]);
    }
    if (!any) break;
  }
}

static u64 splitmix(u64& s) {
  s += 0x9E3779B97F4A7C15ull;
  u64 z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static u64 rnd_elem(u64& s) {   // uniform by rejection for primes near 2^64, by remainder for small ones
  if (g_p < ((u64)1 << 63)) return splitmix(s) % g_p;
  u64 v; do v = splitmix(s); while (v >= g_p); return v;
}

static int g_dist_cfg = 0, g_dist_generic = 0;   // passes of the dist mode that ran a specialised / the generic body
static void run_plan(const PlanDesc& pd, bool inv, const u64* in, u64* out, u64* tmp) {
  std::vector<u64> lds;
  for (auto& p : pd.passes) {
    TileArgs a = p.args;
    const u64* bufs_in[3] = {in, out, tmp};
    u64* bufs_out[3] = {nullptr, out, tmp};
    a.in = bufs_in[p.in_buf];
    a.out = bufs_out[p.out_buf];
    a.wr = pd.wr[p.wr_id].data();
    if (p.tw_id >= 0) { a.tw_lo = pd.tw[p.tw_id].lo.data(); a.tw_hi = pd.tw[p.tw_id].hi.data(); }
    if (p.twf_id >= 0) a.tw_full = pd.twf[p.twf_id].data();
    lds.assign(p.lds_bytes / 8 + 1 + ((size_t)1 << p.logr), 0);   // + room for the LDS-staged round twiddles
    g_cfg_used = 0;
    for (u32 bid = 0; bid < p.grid; bid++) {
      g_fa.a = &a; g_fa.lds = lds.data(); g_fa.bid = bid; g_fa.logr = p.logr; g_fa.inv = inv; g_fa.small = p.small;
      run_block(p.block);
    }
    (g_cfg_used ? g_dist_cfg : g_dist_generic)++;
  }
}

// dist mode: simulate all W ranks of the four-step in one process (the all-to-all is a memcpy); `chunks` > 1 = the
// column-chunked exchange layout (plan.h: chunk j of rank g is shipped as W contiguous blocks [R/W][Cwc])
static int g_twf = 0;
static int dist_main(int log2n, int world, bool inv, int chunks) {
  DistShape sh;
  if (!dist_shape(log2n, world, &sh) || !dist_chunks_ok(sh, chunks)) { printf("bad dist shape\n"); return 2; }
  const u64 n = sh.n, per = n / sh.W, Cwc = sh.Cw / (u64)chunks;
  std::vector<u64> x(n), ref(n), got(n);
  u64 s = 0x5EED0005ull + log2n;
  for (auto& v : x) v = rnd_elem(s);
  std::vector<std::vector<u64>> loc(world), snd(world), rcv(world), res(world), tmp(world);
  for (int g = 0; g < world; g++) {
    loc[g].resize(per); snd[g].assign(per, 1); rcv[g].resize(per); res[g].assign(per, 2); tmp[g].assign(per, 3);
    for (u64 r = 0; r < sh.R; r++)
      for (u64 cl = 0; cl < sh.Cw; cl++) loc[g][r * sh.Cw + cl] = x[r * sh.C + g * sh.Cw + cl];
    for (int j = 0; j < chunks; j++) {
      PlanDesc p1 = build_dist_phase1(log2n, inv, g, world, 4, g_twf, j, chunks, g_hf);
      if (p1.passes.empty()) { printf("no phase-1 plan\n"); return 2; }
      run_plan(p1, inv, loc[g].data() + (u64)j * Cwc, snd[g].data() + (u64)j * sh.R * Cwc, tmp[g].data());
    }
  }
  const u64 blk = sh.Rw * Cwc;   // one (source rank, chunk) block on the receiver
  for (int g = 0; g < world; g++)
    for (int j = 0; j < chunks; j++)
      for (int h = 0; h < world; h++)
        memcpy(&rcv[h][((u64)g * chunks + j) * blk], &snd[g][(u64)j * sh.R * Cwc + (u64)h * blk], blk * 8);
  for (int h = 0; h < world; h++) {
    PlanDesc p2 = build_dist_phase2(log2n, inv, h, world, 4, g_twf, chunks, g_hf);
    if (p2.passes.empty()) { printf("no phase-2 plan\n"); return 2; }
    run_plan(p2, inv, rcv[h].data(), res[h].data(), tmp[h].data());
    for (u64 k2 = 0; k2 < sh.C; k2++)
      for (u64 k1l = 0; k1l < sh.Rw; k1l++) got[(h * sh.Rw + k1l) + sh.R * k2] = res[h][k2 * sh.Rw + k1l];
  }
  int rc = inv ? orc_ifft(g_p, g_g, x.data(), ref.data(), n) : orc_fft(g_p, g_g, x.data(), ref.data(), n);
  if (rc) return 1;
  for (u64 i = 0; i < n; i++)
    if (got[i] != ref[i]) { printf("DIST MISMATCH at %llu\n", (unsigned long long)i); return 1; }
  printf("passes: specialised=%d generic=%d\n", g_dist_cfg, g_dist_generic);
  printf("OK dist log2n=%d world=%d inv=%d chunks=%d\n", log2n, world, (int)inv, chunks);
  return 0;
}

// mul mode: the fused multiply (ntt_mul.h) the way ronk_plan.hip conv_dev runs it -- column pass of both operands (zero
// padding implicit), mul_mid_body per tile (forward row pass of a and b, product, inverse column pass), inverse row pass with
// truncated output -- against the oracle's schoolbook Mul on the host
struct MidArgs { const TileArgs *fa, *ia; u64* lds; u32 bid; int logr, logc, kindi; };
static MidArgs g_mid;
static void mid_fiber(int tid) {
  const MidArgs& m = g_mid;
#define EMU_MID_CASE(LR, LC, KD) \
  if (m.logr == LR && m.logc == LC && m.kindi == KD) { \
    if (g_mont) mul_mid_body<LR, LC, KD, MontField>(*m.fa, *m.ia, m.lds, (u32)tid, m.bid, fiber_barrier); \
    else mul_mid_body<LR, LC, KD>(*m.fa, *m.ia, m.lds, (u32)tid, m.bid, fiber_barrier); \
  }
  EMU_MID_CASE(10, 2, 1) EMU_MID_CASE(10, 2, 3) EMU_MID_CASE(11, 2, 1) EMU_MID_CASE(11, 2, 3) EMU_MID_CASE(10, 3, 1) EMU_MID_CASE(11, 3, 1)
#undef EMU_MID_CASE
  g_done[tid] = 1;
  swapcontext(&g_ctx[tid], &g_sched);
}
static void run_mid_block(u32 T) {
  const size_t STK = 64 * 1024;
  if (g_ctx.size() < T) { g_ctx.resize(T); g_stacks.resize((size_t)T * STK); g_done.resize(T); }
  for (u32 t = 0; t < T; t++) {
    getcontext(&g_ctx[t]);
    g_ctx[t].uc_stack.ss_sp = &g_stacks[(size_t)t * STK];
    g_ctx[t].uc_stack.ss_size = STK;
    g_ctx[t].uc_link = &g_sched;
    makecontext(&g_ctx[t], (void (*)())mid_fiber, 1, (int)t);
    g_done[t] = 0;
  }
  for (;;) {
    bool any = false;
    for (u32 t = 0; t < T; t++) {
      if (g_done[t]) continue;
      any = true;
      g_cur = (int)t;
      swapcontext(&g_sched, &g_ctx[t]);
    }
    if (!any) break;
  }
}
static TileArgs bind_pass(const PlanDesc& pd, size_t idx, const u64* in, u64* out, u64* tmp) {
  const PassDesc& p = pd.passes[idx];
  TileArgs a = p.args;
  const u64* bufs_in[3] = {in, out, tmp};
  u64* bufs_out[3] = {nullptr, out, tmp};
  a.in = bufs_in[p.in_buf];
  a.out = bufs_out[p.out_buf];
  a.wr = pd.wr[p.wr_id].data();
  if (p.tw_id >= 0) { a.tw_lo = pd.tw[p.tw_id].lo.data(); a.tw_hi = pd.tw[p.tw_id].hi.data(); }
  if (p.twf_id >= 0) a.tw_full = pd.twf[p.twf_id].data();
  return a;
}
static int mul_main(int log2n, u64 d, u64 d2, int logc, int inv_twf) {
  const u64 n = (u64)1 << log2n, m = d + d2 - 1;
  if (m > n) { printf("operands too long\n"); return 2; }
  PlanDesc F = build_plan(log2n, 2, false, logc, 18, 25, false, 0, g_hf), I = build_plan(log2n, 1, true, logc, inv_twf, 25, false, log2n / 2, g_hf);
  std::vector<u64> ab(2 * n, 0x1111), ftmp(2 * n, 0xDEADBEEFull), itmp(n, 0xDEADBEEFull), out(n, 0xDEADBEEFull), ref(m);
  u64 s = 0x5EED0C00ull + log2n;
  for (u64 i = 0; i < d; i++) ab[i] = rnd_elem(s);
  for (u64 i = 0; i < d2; i++) ab[n + i] = rnd_elem(s);
  ab[0] = g_p - 1; ab[n + d2 - 1] = g_p - 1;
  std::vector<u64> lds;
  {   // F1: column pass of the batch of two, padding limits d / d2
    TileArgs a = bind_pass(F, 0, ab.data(), nullptr, ftmp.data());
    a.in_valid = d; a.in_valid1 = d2;
    const PassDesc& p = F.passes[0];
    lds.assign(p.lds_bytes / 8 + 1 + ((size_t)1 << p.logr), 0);
    for (u32 bid = 0; bid < p.grid; bid++) {
      g_fa.a = &a; g_fa.lds = lds.data(); g_fa.bid = bid; g_fa.logr = p.logr; g_fa.inv = false; g_fa.small = p.small;
      run_block(p.block);
    }
  }
  TileArgs fa = bind_pass(F, 1, nullptr, nullptr, ftmp.data()), ia = bind_pass(I, 0, nullptr, nullptr, itmp.data());
  const int kindi = ia.tw_full ? 3 : 1, logr = F.passes[1].logr;
  if (!mul_mid_matches(fa, ia, logr, (int)fa.logc, kindi)) { printf("passes do not fuse\n"); return 2; }
  lds.assign(F.passes[1].lds_bytes / 8 + 1, 0);
  for (u32 bid = 0; bid < fa.tiles; bid++) {
    g_mid = MidArgs{&fa, &ia, lds.data(), bid, logr, (int)fa.logc, kindi};
    run_mid_block(F.passes[1].block);
  }
  {   // I2: the inverse's row pass, output truncated to d + d2 - 1 coefficients
    TileArgs a = bind_pass(I, 1, nullptr, out.data(), itmp.data());
    a.out_valid = m;
    const PassDesc& p = I.passes[1];
    lds.assign(p.lds_bytes / 8 + 1 + ((size_t)1 << p.logr), 0);
    for (u32 bid = 0; bid < p.grid; bid++) {
      g_fa.a = &a; g_fa.lds = lds.data(); g_fa.bid = bid; g_fa.logr = p.logr; g_fa.inv = true; g_fa.small = p.small;
      run_block(p.block);
    }
  }
  // the oracle: NTT product of the zero-padded operands (the schoolbook Mul is O(d * d2); at these sizes the two agree by
  // tests/test_oracle_golden.py) -- three oracle transforms
  std::vector<u64> pa(n, 0), pb(n, 0), fa_(n), fb_(n), pr(n);
  memcpy(pa.data(), ab.data(), d * 8); memcpy(pb.data(), ab.data() + n, d2 * 8);
  if (orc_fft(g_p, g_g, pa.data(), fa_.data(), n) || orc_fft(g_p, g_g, pb.data(), fb_.data(), n)) return 1;
  for (u64 i = 0; i < n; i++) fa_[i] = orc_mul(g_p, fa_[i], fb_[i]);
  if (orc_ifft(g_p, g_g, fa_.data(), pr.data(), n)) return 1;
  for (u64 i = 0; i < n; i++) {
    if (i >= m) { if (out[i] != 0xDEADBEEFull) { printf("store beyond the product at %llu\n", (unsigned long long)i); return 1; } continue; }
    if (out[i] != pr[i]) { printf("MUL MISMATCH at %llu: got %llu want %llu\n", (unsigned long long)i, (unsigned long long)out[i], (unsigned long long)pr[i]); return 1; }
  }
  printf("OK mul log2n=%d d=%llu d2=%llu logc=%d inverse twiddles=%s\n", log2n, (unsigned long long)d, (unsigned long long)d2, logc,
         kindi == 3 ? "matrix" : "two-level");
  return 0;
}

int main(int argc, char** argv) {
  if (const char* e = getenv("RONK_EMU_P")) {
    g_p = strtoull(e, 0, 0);
    g_g = getenv("RONK_EMU_G") ? strtoull(getenv("RONK_EMU_G"), 0, 0) : 0;
    if (!g_g && orc_find_primitive_element(g_p, &g_g)) { printf("no generator\n"); return 2; }
    g_mont = true;
    g_hf = HostField::montgomery(g_p, g_g);
  }
  if (argc >= 6 && !strcmp(argv[1], "mul"))   // emu_tile mul <log2n> <d> <d2> <logc> [inverse twf_max_log]
    return mul_main(atoi(argv[2]), strtoull(argv[3], 0, 10), strtoull(argv[4], 0, 10), atoi(argv[5]), argc > 6 ? atoi(argv[6]) : 18);
  if (argc >= 6 && !strcmp(argv[1], "dist")) g_twf = atoi(argv[5]);
  if (argc >= 5 && !strcmp(argv[1], "dist"))   // emu_tile dist <log2n> <world> <inverse> [twf_max_log] [chunks]
    return dist_main(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]) != 0, argc >= 7 ? atoi(argv[6]) : 1);
  if (argc < 5) { fprintf(stderr, "usage: emu_tile log2n batch inverse max_logc\n"); return 2; }
  int log2n = atoi(argv[1]);
  u64 batch = strtoull(argv[2], 0, 10);
  bool inv = atoi(argv[3]) != 0;
  int max_logc = atoi(argv[4]);
  u64 n = (u64)1 << log2n;
  int twf = argc > 5 ? atoi(argv[5]) : 0;
  int three_from = argc > 6 ? atoi(argv[6]) : 25;
  u64 in_valid = argc > 7 ? strtoull(argv[7], 0, 10) : 0, out_valid = argc > 8 ? strtoull(argv[8], 0, 10) : 0;
  bool auto_tiles = argc > 9 && atoi(argv[9]) != 0;   // the planner's own per-pass tile rules (ronk_plan_create's default)
  u64 in_valid1 = argc > 10 ? strtoull(argv[10], 0, 10) : 0;   // TileArgs::in_valid1: the limit of batch entries >= 1 (paired multiply operands)
  const bool with_in2 = argc > 11 && atoi(argv[11]) != 0;      // TileArgs::in2: a second operand multiplied in on load (fused pointwise product)
  const bool twf_t = getenv("RONK_TWF_T") && atoi(getenv("RONK_TWF_T")) != 0;   // the full twiddle matrix transposed (plan.h)
  PlanDesc pd = build_plan(log2n, batch, inv, max_logc, twf, three_from, auto_tiles, 0, g_hf, twf_t);

  std::vector<u64> in(n * batch), out(n * batch, 0xDEADBEEFull), tmp(n * batch, 0xDEADBEEFull), ref(n * batch);
  u64 s = 0x5EED0000ull + log2n;
  for (auto& v : in) v = rnd_elem(s);
  // adversarial corners
  in[0] = g_p - 1; if (n > 1) in[n - 1] = g_p - 1; if (n > 2) in[1] = 0;

  std::vector<u64> in2;
  if (with_in2) {
    in2.resize(n * batch);
    for (auto& v : in2) v = rnd_elem(s);
    in2[0] = g_p - 1; in2[n - 1] = 0;
  }
  std::vector<u64> lds;
  for (auto& p : pd.passes) {
    TileArgs a = p.args;
    const u64* bufs_in[3] = {in.data(), out.data(), tmp.data()};
    u64* bufs_out[3] = {nullptr, out.data(), tmp.data()};
    a.in = bufs_in[p.in_buf];
    a.out = bufs_out[p.out_buf];
    a.wr = pd.wr[p.wr_id].data();
    if (p.tw_id >= 0) { a.tw_lo = pd.tw[p.tw_id].lo.data(); a.tw_hi = pd.tw[p.tw_id].hi.data(); }
    if (p.twf_id >= 0) a.tw_full = pd.twf[p.twf_id].data();
    if (with_in2 && p.in_buf == BUF_IN) a.in2 = in2.data();
    if (in_valid && p.in_buf == BUF_IN) a.in_valid = in_valid;
    if (in_valid1 && p.in_buf == BUF_IN) a.in_valid1 = in_valid1;
    if (out_valid && p.out_buf == BUF_OUT) a.out_valid = out_valid;
    lds.assign(p.lds_bytes / 8 + 1 + ((size_t)1 << p.logr), 0);   // + room for the LDS-staged round twiddles
    g_cfg_used = 0;
    for (u32 bid = 0; bid < p.grid; bid++) {
      g_fa.a = &a; g_fa.lds = lds.data(); g_fa.bid = bid; g_fa.logr = p.logr; g_fa.inv = inv; g_fa.small = p.small;
      run_block(p.block);
    }
    printf("pass logr=%d logc=%u tiles=%u nb1=%u nb2=%u grid=%u block=%u lds=%zu kernel=%s\n", p.logr, a.logc, a.tiles,
           a.nb1, a.nb2, p.grid, p.block, p.lds_bytes, g_cfg_used == 1 ? "cfg:column/two-level" : g_cfg_used == 3 ? "cfg:column/matrix" :
           g_cfg_used == 2 ? "cfg:row" : g_cfg_used == 5 ? "cfg:whole" : g_cfg_used == 4 ? "cfg:general" : g_cfg_used == 11 ? "half:column/two-level" : g_cfg_used == 13 ? "half:column/matrix" :
           g_cfg_used == 12 ? "half:row" : g_cfg_used == 31 ? "wl:column/two-level" : g_cfg_used == 33 ? "wl:column/matrix" : g_cfg_used == 32 ? "wl:row" : g_cfg_used == 21 ? "r4:column/two-level" : g_cfg_used == 23 ? "r4:column/matrix" : g_cfg_used == 22 ? "r4:row" : g_cfg_used >= 100 ? (g_cfg_used % 100 == 2 ? "feat:row" : g_cfg_used % 100 == 3 ? "feat:column/matrix" : "feat:column/two-level") :
           p.small ? "small" : "generic");
  }
  if (in_valid) for (u64 b = 0; b < batch; b++) for (u64 i = (b && in_valid1) ? in_valid1 : in_valid; i < n; i++) in[b * n + i] = 0;  // what the kernel must have seen
  if (with_in2) for (u64 i = 0; i < n * batch; i++) in[i] = orc_mul(g_p, in[i], in2[i]);
  for (u64 b = 0; b < batch; b++) {
    int rc = inv ? orc_ifft(g_p, g_g, &in[b * n], &ref[b * n], n) : orc_fft(g_p, g_g, &in[b * n], &ref[b * n], n);
    if (rc) { printf("oracle rc %d\n", rc); return 1; }
  }
  for (u64 i = 0; i < n * batch; i++)
    if (out_valid && (i % n) >= out_valid) {
      if (out[i] != 0xDEADBEEFull) { printf("store beyond out_valid at %llu\n", (unsigned long long)i); return 1; }
    } else if (out[i] != ref[i]) {
      printf("MISMATCH at %llu (poly %llu, k %llu): got %llu want %llu\n", (unsigned long long)i,
             (unsigned long long)(i / n), (unsigned long long)(i % n), (unsigned long long)out[i],
             (unsigned long long)ref[i]);
      return 1;
    }
  printf("OK log2n=%d batch=%llu inv=%d\n", log2n, (unsigned long long)batch, (int)inv);
  return 0;
}
