// emu_longdiv.cpp -- HOST EMULATOR of the long-division kernel (TEST INFRASTRUCTURE ONLY).
//
// Runs ronk::poly_divrem_body (ronkathon_amd/csrc/longdiv_kernel.h: the very code ronk_poly_divrem(_dev)'s one-workgroup launch
// executes) on ucontext fibers -- one fiber per work-item, barrier = yield, the shared max as a plain max -- and compares quotient,
// remainder and the status word with oracle/ orc_poly_divrem (the statement-by-statement restatement of
// src/polynomial/mod.rs:170-225) on seeded random shapes: every divisor length up to and beyond the dividend's, trailing zeros in
// either operand (the reference's early stop / panic), zero operands, the dividend in place (a == rem) and out of place.
// Built and run by tests/test_emu_kernel.py only; the product library never contains or calls it.
//
// usage: emu_longdiv <p> <cases> <max d> <work-items> [seed]      (prints OK or the first mismatch)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <vector>

#include "../../oracle/ronk_oracle.h"
#include "../../ronkathon_amd/csrc/gl64.h"
#include "../../ronkathon_amd/csrc/longdiv_kernel.h"

using namespace ronk;

static ucontext_t g_sched;
static std::vector<ucontext_t> g_ctx;
static std::vector<char> g_stacks;
static std::vector<char> g_done;
static int g_cur;

struct GlHostOps {
  u64 sub(u64 a, u64 b) const { return gl64::sub(a, b); }
  u64 mul(u64 a, u64 b) const { return gl64::mul(a, b); }
  u64 pow(u64 a, u64 e) const { return gl64::pow(a, e); }
  u64 order() const { return gl64::P; }
};
struct ModOps {
  u64 p;
  u64 sub(u64 a, u64 b) const { return a >= b ? a - b : a + (p - b); }
  u64 mul(u64 a, u64 b) const { return (u64)(((unsigned __int128)a * b) % p); }
  u64 pow(u64 a, u64 e) const { u64 r = 1 % p; for (; e; e >>= 1, a = mul(a, a)) if (e & 1) r = mul(r, a); return r; }
  u64 order() const { return p; }
};

struct FiberCtx {
  size_t t, T;
  unsigned long long* top_;
  u64* lead_;
  size_t tid() const { return t; }
  size_t nthreads() const { return T; }
  void barrier() const { swapcontext(&g_ctx[g_cur], &g_sched); }
  unsigned long long* top() const { return top_; }
  u64* lead() const { return lead_; }
  void raise(unsigned long long* w, unsigned long long v) const { if (*w < v) *w = v; }
};

struct Job {
  bool gl; u64 p; size_t T;
  const u64* a; u64* rem; size_t d; const u64* b; size_t d2; u64* quot; int* status;
  unsigned long long top; u64 lead;
};
static Job g_job;

static void fiber_main(int tid) {
  FiberCtx cx{(size_t)tid, g_job.T, &g_job.top, &g_job.lead};
  if (g_job.gl) poly_divrem_body(GlHostOps(), g_job.a, g_job.rem, g_job.d, g_job.b, g_job.d2, g_job.quot, g_job.status, cx);
  else poly_divrem_body(ModOps{g_job.p}, g_job.a, g_job.rem, g_job.d, g_job.b, g_job.d2, g_job.quot, g_job.status, cx);
  g_done[tid] = 1;
  swapcontext(&g_ctx[tid], &g_sched);
}
static void run_block(size_t T) {
  const size_t STK = 64 * 1024;
  if (g_ctx.size() < T) { g_ctx.resize(T); g_stacks.resize(T * STK); g_done.resize(T); }
  for (size_t t = 0; t < T; t++) {
    getcontext(&g_ctx[t]);
    g_ctx[t].uc_stack.ss_sp = &g_stacks[t * STK];
    g_ctx[t].uc_stack.ss_size = STK;
    g_ctx[t].uc_link = &g_sched;
    makecontext(&g_ctx[t], (void (*)())fiber_main, 1, (int)t);
    g_done[t] = 0;
  }
  for (;;) {
    bool any = false;
    for (size_t t = 0; t < T; t++) {
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
  if (argc < 5) { printf("usage: emu_longdiv p cases maxd workitems [seed]\n"); return 2; }
  const u64 p = strtoull(argv[1], 0, 0);
  const int cases = atoi(argv[2]);
  const size_t maxd = (size_t)strtoull(argv[3], 0, 0), T = (size_t)strtoull(argv[4], 0, 0);
  u64 seed = argc > 5 ? strtoull(argv[5], 0, 0) : 1;
  if (p < 2 || cases <= 0 || maxd == 0 || T == 0 || T > 1024) { printf("bad arguments\n"); return 2; }
  int panics = 0, in_place = 0, ragged = 0;
  for (int it = 0; it < cases; it++) {
    const size_t d = 1 + (size_t)(splitmix(seed) % maxd);
    size_t d2 = 1 + (size_t)(splitmix(seed) % (d + 3));          // up to two entries longer than the dividend
    if (it % 7 == 0) d2 = 1 + (size_t)(splitmix(seed) % 3);       // short divisors: long quotients
    std::vector<u64> a(d), b(d2), q(d, 0x5555555555555555ull), r(d, 0x3333333333333333ull), oq(d), orr(d);
    for (auto& v : a) v = splitmix(seed) % p;
    for (auto& v : b) v = splitmix(seed) % p;
    const u64 shape = splitmix(seed) % 16;
    if (shape < 4) { const size_t k = 1 + (size_t)(splitmix(seed) % d2); for (size_t i = d2 - k; i < d2; i++) b[i] = 0; ragged++; }   // trailing zeros (k == d2: the zero divisor)
    else if (shape < 6) { const size_t k = 1 + (size_t)(splitmix(seed) % d); for (size_t i = d - k; i < d; i++) a[i] = 0; }
    else if (shape == 6) { for (auto& v : a) v = 0; }
    else if (shape == 7 && p > 2) { for (auto& v : a) v = p - 1; for (auto& v : b) v = p - 1; }
    const int orc = orc_poly_divrem(p, a.data(), d, b.data(), d2, oq.data(), orr.data());
    const bool alias = (splitmix(seed) & 1) != 0;
    int status = 0x7777;
    if (alias) { r = a; in_place++; }
    g_job = Job{p == gl64::P, p, T, alias ? r.data() : a.data(), r.data(), d, b.data(), d2, q.data(), &status, 0, 0};
    run_block(T);
    if (orc != 0) {
      panics++;
      if (status != orc) { printf("FAIL case %d (d %zu d2 %zu): status %d, oracle %d\n", it, d, d2, status, orc); return 1; }
      continue;
    }
    if (status != 0) { printf("FAIL case %d (d %zu d2 %zu): status %d, oracle accepts\n", it, d, d2, status); return 1; }
    for (size_t j = 0; j < d; j++) {
      if (q[j] != oq[j]) { printf("FAIL case %d (d %zu d2 %zu alias %d): quotient at %zu\n", it, d, d2, (int)alias, j); return 1; }
      if (r[j] != orr[j]) { printf("FAIL case %d (d %zu d2 %zu alias %d): remainder at %zu\n", it, d, d2, (int)alias, j); return 1; }
    }
  }
  printf("OK %d cases (%d reference panics, %d ragged divisors, %d in place)\n", cases, panics, ragged, in_place);
  return 0;
}
