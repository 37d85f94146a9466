// emu_scan.cpp -- HOST EMULATOR of the linear-division kernels (TEST INFRASTRUCTURE ONLY).
//
// Runs ronk::lindiv_scan_body / lindiv_apply_body (ronkathon_amd/csrc/lindiv_kernels.h: the very code the two launches of
// ronk_poly_div_linear_dev execute) on ucontext fibers -- one fiber per work-item, barrier = yield, the cross-lane
// shifts through an exchange array -- and compares quotient and remainder with the recurrence written out with the
// oracle's field operations (and, for small d, with oracle/ orc_poly_divrem itself).  Built and run by
// tests/test_emu_kernel.py only; the product library never contains or calls it.
//
// usage: emu_scan <p> <d> <z> <b1> <mode> [seed]      (prints OK or the first mismatch)
//   mode: bit 0 = 16-byte direct loads, bit 1 = the one-launch form (lindiv_one_body), bit 2 = ... with every look-back wait failing
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <vector>

#include "../../oracle/ronk_oracle.h"
#include "../../ronkathon_amd/csrc/lindiv_kernels.h"

using namespace ronk;

static ucontext_t g_sched;
static std::vector<ucontext_t> g_ctx;
static std::vector<char> g_stacks;
static std::vector<char> g_done;
static int g_cur;
static u64 g_xch[1024];

struct GlHostOps {
  u64 add(u64 a, u64 b) const { return gl64::add(a, b); }
  u64 mul(u64 a, u64 b) const { return gl64::mul(a, b); }
};
struct ModOps {
  u64 p;
  u64 add(u64 a, u64 b) const { return (u64)(((unsigned __int128)a + b) % p); }
  u64 mul(u64 a, u64 b) const { return (u64)(((unsigned __int128)a * b) % p); }
};

struct FiberCtx {
  u32 t, b;
  u64* l;
  u32 tid() const { return t; }
  u32 bid() const { return b; }
  u32 wave() const { return t >> 6; }
  u64* lds() const { return l; }
  void barrier() const { swapcontext(&g_ctx[g_cur], &g_sched); }
  u64 shfl_down(u64 v, u32 off) const {                    // __shfl_down(v, off, 64): own value when the source is outside
    g_xch[t] = v; barrier();
    const u64 r = (t & 63) + off < 64 ? g_xch[t + off] : v;
    barrier();
    return r;
  }
  u64 shfl_xor(u64 v, u32 mask) const {
    g_xch[t] = v; barrier();
    const u64 r = g_xch[(t & ~63u) | ((t ^ mask) & 63)];
    barrier();
    return r;
  }
  u64 shfl(u64 v, u32 src) const {                         // __shfl(v, src, 64)
    g_xch[t] = v; barrier();
    const u64 r = g_xch[(t & ~63u) | (src & 63)];
    barrier();
    return r;
  }
  // the one-launch form's look-back array: the emulator runs the workgroups in dispatch order, one after the other, so an entry is
  // either there or never will be (the body then recomputes it: the path RONK_LB_TEST_FLAGS=1 forces on the device)
  void lb_store(u64* p, u64 v) const { *p = v; }
  bool lb_wait(const u64* p, u64* v, u64 test_flags) const {
    if ((test_flags & 1) || *p == LINDIV_LB_EMPTY) return false;
    *v = *p;
    return true;
  }
  void ld2(const u64* p, u64& a, u64& c) const {
    if ((uintptr_t)p & 15) { printf("FAIL: unaligned 16-byte load\n"); exit(1); }
    a = p[0]; c = p[1];
  }
  void st2(u64* p, u64 a, u64 c) const {
    if ((uintptr_t)p & 15) { printf("FAIL: unaligned 16-byte store\n"); exit(1); }
    p[0] = a; p[1] = c;
  }
};

struct Job {
  int phase, pl, direct; bool gl; u64 p;
  const u64* c; size_t d; const LinDivTab* tab; u64 *W, *H; u32 nch; u64 *quot, *rem; u64* lds; u32 bid;
  const LinDiv1Tab* tab1; u64 *lb_cur, *lb_next; u32 lb_words; int pl1;   // phase 2: the one-launch form
};
static Job g_job;

template <int MODE, class Ops>
static void run_item(const Ops& ops, u32 tid) {
  FiberCtx cx{tid, g_job.bid, g_job.lds};
  if (g_job.phase == 2) {
    if (g_job.pl1 == 4) lindiv_one_body<MODE, 4>(ops, g_job.c, g_job.d, *g_job.tab1, g_job.lb_cur, g_job.lb_next, g_job.lb_words, g_job.nch, g_job.quot, g_job.rem, cx);
    else lindiv_one_body<MODE, 8>(ops, g_job.c, g_job.d, *g_job.tab1, g_job.lb_cur, g_job.lb_next, g_job.lb_words, g_job.nch, g_job.quot, g_job.rem, cx);
  }
  else if (g_job.phase == 0) lindiv_scan_body<MODE>(ops, g_job.c, g_job.d, *g_job.tab, g_job.W, g_job.H, cx);
  else lindiv_apply_body<MODE>(ops, g_job.c, g_job.d, *g_job.tab, g_job.W, g_job.H, g_job.nch, g_job.quot, g_job.rem, cx);
}
template <class Ops>
static void run_item_ops(const Ops& ops, u32 tid) {
  if (g_job.direct) run_item<LINDIV_DLOAD>(ops, tid); else run_item<0>(ops, tid);
}
static void fiber_main(int tid) {
  if (g_job.gl) run_item_ops(GlHostOps(), (u32)tid); else run_item_ops(ModOps{g_job.p}, (u32)tid);
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

int main(int argc, char** argv) {
  if (argc < 6) { printf("usage: emu_scan p d z b1 direct [seed]\n"); return 2; }
  const u64 p = strtoull(argv[1], 0, 0);
  const size_t d = (size_t)strtoull(argv[2], 0, 0);
  const u64 z = strtoull(argv[3], 0, 0) % p, b1 = strtoull(argv[4], 0, 0) % p;
  const int pl = LINDIV_PL, mode = atoi(argv[5]), direct = mode & 1;
  // the one-launch form (z != 0: like the library, which keeps two launches for a divisor b1 x); ... with every look-back wait failing
  const bool one = (mode & 2) != 0 && z != 0, lb_fail = (mode & 4) != 0;
  u64 seed = argc > 6 ? strtoull(argv[6], 0, 0) : 1;
  if (d == 0 || b1 == 0 || mode < 0 || mode > 7) { printf("bad arguments\n"); return 2; }
  std::vector<u64> cbuf(d + 2), qbuf(d + 2, 0x5555555555555555ull);
  u64* c = cbuf.data();
  u64* quot = qbuf.data();
  if ((uintptr_t)c & 15) c++;                              // 16-byte aligned, as the library demands of the direct form
  if ((uintptr_t)quot & 15) quot++;
  for (size_t i = 0; i < d; i++) c[i] = splitmix(seed) % p;
  if (d > 3) { c[d - 1] = p - 1; c[0] = 0; }               // extremes
  u64 b1inv;
  if (orc_inverse(p, b1, &b1inv)) { printf("no inverse\n"); return 2; }
  LinDivTab tab;
  lindiv_build_tab(p, z, b1inv, &tab);
  const size_t ch = (size_t)256 * pl, nch = (d + ch - 1) / ch;
  std::vector<u64> H(nch, 0xAAAAAAAAAAAAAAAAull), W(nch * 256, 0xAAAAAAAAAAAAAAAAull), lds(8192);
  u64 rem = ~0ull;
  g_job = Job{0, pl, direct, p == 0xFFFFFFFF00000001ull, p, c, d, &tab, W.data(), H.data(), (u32)nch, quot, &rem, lds.data(), 0};
  LinDiv1Tab tab1;
  // coefficients per lane of the one-launch form: the library's rule (4 below 1.5 M coefficients, else 8) unless EMU_LINDIV1_PL says
  const int pl1 = getenv("EMU_LINDIV1_PL") ? atoi(getenv("EMU_LINDIV1_PL")) : d < ((size_t)3 << 19) ? 4 : 8;
  if (pl1 != 4 && pl1 != 8) { printf("bad EMU_LINDIV1_PL\n"); return 2; }
  if (z != 0) lindiv1_build_tab(p, z, b1inv, lb_fail ? 1 : 0, pl1, &tab1);
  const size_t ch1 = (size_t)LINDIV1_NL * pl1, nch1 = (d + ch1 - 1) / ch1;
  const u32 lbw = (u32)nch1 + 37;                          // (the library's arrays hold LB_WORDS entries; any length >= chunks works)
  std::vector<u64> lbc(lbw, LINDIV_LB_EMPTY), lbn(lbw, 0x1111111111111111ull), lds1(lindiv1_lds_words(8));
  if (one) {
    if (nch1 > LINDIV1_MAX_CHUNKS) { printf("bad arguments: more than %u chunks\n", LINDIV1_MAX_CHUNKS); return 2; }
    g_job.phase = 2; g_job.nch = (u32)nch1; g_job.tab1 = &tab1; g_job.lb_cur = lbc.data(); g_job.lb_next = lbn.data(); g_job.lb_words = lbw; g_job.pl1 = pl1;
    g_job.lds = lds1.data();
    for (u32 bid = 0; bid < nch1; bid++) {                 // dispatch order: workgroup i takes chunk nch1-1-i
      g_job.bid = bid;
      for (auto& w : lds1) w = 0xDEADBEEFDEADBEEFull;
      run_block(LINDIV1_NL);
    }
    for (u32 i = 0; i < lbw; i++)
      if (lbn[i] != LINDIV_LB_EMPTY) { printf("FAIL: look-back array of the next call not cleared at %u\n", i); return 1; }
  }
  for (int phase = 0; phase < 2 && !one; phase++) {
    g_job.phase = phase;
    for (u32 b = 0; b < nch; b++) {
      const u32 bid = phase ? (u32)(nch - 1 - b) : b;      // (any order: the workgroups of a launch are independent)
      g_job.bid = bid;
      for (auto& w : lds) w = 0xDEADBEEFDEADBEEFull;
      run_block(256);
    }
  }
  // expected: S(x) = c_x + z S(x+1);  quot[j] = S(j+1) / b1;  remainder = S(0)
  std::vector<u64> want(d);
  u64 S = 0;
  for (size_t j = d; j-- > 0;) { want[j] = orc_mul(p, S, b1inv); S = orc_add(p, orc_mul(p, S, z), c[j]); }
  for (size_t j = 0; j < d; j++)
    if (quot[j] != want[j]) { printf("FAIL quot[%zu] = %llu, want %llu\n", j, (unsigned long long)quot[j], (unsigned long long)want[j]); return 1; }
  if (rem != S) { printf("FAIL remainder %llu, want %llu\n", (unsigned long long)rem, (unsigned long long)S); return 1; }
  if (quot[d] != 0x5555555555555555ull && quot + d < qbuf.data() + qbuf.size()) { printf("FAIL: wrote beyond d\n"); return 1; }
  if (d <= 4096) {                                         // the oracle's long division itself (polynomial/mod.rs:170-225)
    std::vector<u64> oq(d), orr(d);
    const u64 div[2] = {orc_mul(p, orc_neg(p, z), b1), b1};   // b0 = -z b1
    if (orc_poly_divrem(p, c, d, div, 2, oq.data(), orr.data())) { printf("FAIL: oracle divrem\n"); return 1; }
    for (size_t j = 0; j < d; j++)
      if (oq[j] != quot[j]) { printf("FAIL vs orc_poly_divrem at %zu\n", j); return 1; }
    if (orr[0] != rem) { printf("FAIL remainder vs orc_poly_divrem\n"); return 1; }
  }
  printf("OK p=%llu d=%zu pl=%d direct=%d %s chunks=%zu\n", (unsigned long long)p, d, one ? pl1 : pl, direct, one ? (lb_fail ? "one-launch/recompute" : "one-launch") : "two-launch",
         one ? nch1 : nch);
  return 0;
}
