// lindiv_kernels.h -- division by a linear divisor (kzg::open), the two-launch form with per-lane partial scans.
//
// Reference semantics restated: poly / (b0 + b1 x), src/polynomial/mod.rs:170-225 via src/kzg/setup.rs:63-78 (divisor
// [-z, 1]).  With z = -b0/b1 and S(x) = sum_{i >= x} c_i z^(i-x) (the value of the coefficient suffix that starts at x),
//   quot[j] = S(j+1) / b1,   remainder = S(0) = c(z).
// scan_kernels.h (lindiv_fused_kernel) pays three field products per coefficient: one for the chunk sums of the first
// launch, two in the second launch (a Horner pass up every lane's run for the lane sums, then the recurrence down the same
// run) plus a 256-lane scan of the lane sums -- 108 VALU instructions per coefficient in the second launch (PMC), which
// made it issue-bound (16.7 us for 2^22 coefficients, the memory floor of its 64 MiB is ~10 us).  Here the FIRST launch
// does the Horner pass in the lanes' own runs and the scan, and leaves both results in HBM:
//   1  lindiv_scan_body    lane t of chunk b owns PL contiguous coefficients; U_t = their value at z; suffix scan over
//                          the chunk's 256 lanes (6 doubling steps inside a wavefront through the cross-lane network,
//                          then the four wavefront sums through LDS): W[256 b + t] = S restricted to chunk b, from lane t's first coefficient;  H[b] = W[256 b]
//   2  lindiv_apply_body   carry of the chunk from the H array (as lindiv_fused_kernel did), then per lane
//                          S(first coefficient of lane t+1) = W[256 b + t + 1] + z^(PL (255 - t)) * carry and the
//                          recurrence down the lane's run: ONE product per coefficient.
// Two products per coefficient in total; HBM traffic 8 B read + 8/PL B written, then 8 + 8/PL read + 8 written.
// MODE LINDIV_DLOAD: a lane READS its run with 16-byte accesses straight from global memory (needs a 16-byte aligned
// dividend); otherwise, and for the quotient always, the chunk goes through a padded LDS image with coalesced 8-byte
// accesses (any alignment).  Measured forms -- 16 coefficients per lane, 16-byte stores of the runs, fewer resident
// workgroups -- and the split of the 22.8 us per 2^22 coefficients over the two launches: profiles/r03_lindiv_forms.txt.
//
// The bodies are written against a small context (work-item ids, barrier, cross-lane shift, 16-byte access) so that the
// CPU suite runs the very same code on fibers (tests/emu/emu_scan.cpp); the kernels are at the end of the file.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "gl64.h"

namespace ronk {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr int LINDIV_PL = 8;                   // coefficients per lane
constexpr int LINDIV_CHUNK = 256 * LINDIV_PL;  // coefficients per workgroup

struct LinDivTab {
  u64 z;
  u64 scale;      // 1/b1 (1 for monic divisors)
  u64 zs[7];      // z^(PL 2^s): multipliers of the lane scan inside a wavefront (s < 6); zs[6] = z^(64 PL): one wavefront
  u64 zp[256];    // z^(PL k)
  u64 Y;          // z^(256 PL): one chunk
  u64 Y256;       // Y^256
  u64 YA[16];     // Y^i
  u64 YB[16];     // Y^(16 i):      Y^t = YA[t & 15] * YB[t >> 4], t < 256
};

// host: the table for divisor root z, modulus p (canonical z < p), PL coefficients per lane
inline void lindiv_build_tab(u64 p, u64 z, u64 scale, LinDivTab* t) {
  const int pl = LINDIV_PL;
  auto mulm = [p](u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % p); };
  auto powm = [&](u64 a, u64 e) { u64 r = 1 % p; while (e) { if (e & 1) r = mulm(r, a); a = mulm(a, a); e >>= 1; } return r; };
  t->z = z % p;
  t->scale = scale;
  const u64 zpl = powm(t->z, (u64)pl);
  u64 y = 1 % p;
  for (int k = 0; k < 256; k++) { t->zp[k] = y; y = mulm(y, zpl); }
  t->Y = y;                                               // z^(256 pl)
  for (int s = 0; s < 7; s++) t->zs[s] = t->zp[1 << s];
  const u64 Y16 = powm(t->Y, 16);
  t->Y256 = powm(Y16, 16);
  u64 cc = 1 % p, dd = 1 % p;
  for (int i = 0; i < 16; i++) {
    t->YA[i] = cc; t->YB[i] = dd;
    cc = mulm(cc, t->Y); dd = mulm(dd, Y16);
  }
}

// LDS words (u64) a workgroup of the two bodies needs
constexpr int LINDIV_DLOAD = 1;
template <int MODE, bool APPLY>
constexpr int lindiv_lds_words() { return 8 + ((MODE & LINDIV_DLOAD) && !APPLY ? 0 : LINDIV_CHUNK + 256); }

// the lane's run e[m] = c[base + PL tid + m] (ZERO beyond d); NL lanes per workgroup
template <int MODE, int NL = 256, int PL = LINDIV_PL, class Ctx>
RONK_HD void lindiv_load_run(const u64* __restrict__ c, size_t d, size_t base, u32 tid, u64* buf, u64 (&e)[PL], Ctx& cx) {
  const bool full = base + NL * PL <= d;
  if constexpr ((MODE & LINDIV_DLOAD) != 0) {
    const size_t i0 = base + (size_t)PL * tid;
    if (full) {
#pragma unroll
      for (int m = 0; m < PL; m += 2) cx.ld2(c + i0 + m, e[m], e[m + 1]);
    } else {
#pragma unroll
      for (int m = 0; m < PL; m++) e[m] = i0 + m < d ? c[i0 + m] : 0;
    }
  } else {
    // coalesced, lane-strided fill; one pad word per run so that the PL-contiguous reads are conflict free
#pragma unroll
    for (int r = 0; r < PL; r++) {
      const u32 k = tid + NL * r;
      const size_t i = base + k;
      buf[k + k / PL] = (full || i < d) ? c[i] : 0;
    }
    cx.barrier();
#pragma unroll
    for (int m = 0; m < PL; m++) e[m] = buf[(PL + 1) * tid + m];
  }
}

// launch 1: W[256 b + t] and H[b]
template <int MODE, class Ops, class Ctx>
RONK_HD void lindiv_scan_body(const Ops& ops, const u64* __restrict__ c, size_t d, const LinDivTab& tab, u64* __restrict__ W,
                              u64* __restrict__ H, Ctx& cx) {
  const u32 tid = cx.tid();
  const size_t b = cx.bid();
  u64* sc = cx.lds();
  constexpr int PL = LINDIV_PL;
  u64 e[PL];
  lindiv_load_run<MODE>(c, d, b * (256 * PL), tid, sc + 8, e, cx);
  const u32 lane = tid & 63;
  const u64 zup = tab.zp[64 - lane];   // (a per-lane table load: issued with the coefficients, not behind the barrier below)
  const u64 z = tab.z;
  u64 U = e[PL - 1];
#pragma unroll
  for (int m = PL - 2; m >= 0; m--) U = ops.add(ops.mul(U, z), e[m]);
  // W_t = U_t + z^PL W_(t+1): doubling steps, inside the wavefront first
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const u32 off = 1u << s;
    const u64 up = cx.shfl_down(U, off);
    if (lane + off < 64) U = ops.add(U, ops.mul(tab.zs[s], up));
  }
  // across the four wavefronts: C = value of the chunk's suffix that starts at the end of this wavefront (wavefront sums
  // T_w = lane 0's U; Horner in z^(64 PL) from the top: 0..3 products, the same number for every lane of a wavefront)
  const u32 w = cx.wave();
  if (lane == 0) sc[w] = U;
  cx.barrier();
  if (w < 3) {
    u64 C = sc[3];
    for (u32 ww = 2; ww > w; ww--) C = ops.add(ops.mul(C, tab.zs[6]), sc[ww]);
    U = ops.add(U, ops.mul(zup, C));
  }
  W[b * 256 + tid] = U;
  if (tid == 0) H[b] = U;
}

// launch 2: quot[base + k] = scale * S(base + k + 1); chunk 0 also writes the remainder c(z) = H_0 + Y G_1
template <int MODE, class Ops, class Ctx>
RONK_HD void lindiv_apply_body(const Ops& ops, const u64* __restrict__ c, size_t d, const LinDivTab& tab,
                               const u64* __restrict__ W, const u64* __restrict__ H, u32 nchunks, u64* __restrict__ quot,
                               u64* __restrict__ rem, Ctx& cx) {
  const u32 tid = cx.tid();
  const u32 b = cx.bid();
  u64* sc = cx.lds();
  u64* buf = sc + 8;
  constexpr int PL = LINDIV_PL;
  const size_t base = (size_t)b * (256 * PL);
  const bool full = base + 256 * PL <= d;
  u64 e[PL];
  lindiv_load_run<MODE>(c, d, base, tid, buf, e, cx);
  // Every other load of the launch is issued here too, unconditionally (a lane that has no use for one reads a safe entry
  // and discards it): taken where they are used -- behind `tid < 255`, inside the carry loop, after the barrier -- they were
  // five dependent memory round trips in a launch of 14 us.
  const u64 wn_raw = W[(size_t)b * 256 + (tid < 255 ? tid + 1 : 255)];
  const u64 zpk = tab.zp[255 - tid];
  const u64 ya = tab.YA[tid & 15], yb = tab.YB[tid >> 4];
  // incoming carry G_(b+1) = sum_{j > b} H_j Y^(j-b-1): lane t takes j = b+1+t+256q (Horner in Y^256 over q), times Y^t
  // (eight sums are fetched at once, then folded; the counts of a workgroup's lanes differ by at most one, so skipped terms
  // cost a branch)
  const u32 first = b + 1 + tid;
  const u32 cnt = first < nchunks ? (nchunks - first + 255) / 256 : 0;
  u64 cpart = 0;
  for (u32 qb = (((nchunks - b + 254) / 256) + 7) & ~7u; qb > 0; qb -= 8) {   // (workgroup-uniform trip count: lane 0's)
    u64 h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = H[qb - 8 + i < cnt ? first + 256 * (qb - 8 + i) : 0];
#pragma unroll
    for (int i = 7; i >= 0; i--)
      if (qb - 8 + i < cnt) cpart = ops.add(ops.mul(cpart, tab.Y256), h[i]);
  }
  if (cnt) cpart = ops.mul(cpart, ops.mul(ya, yb));
  const u64 wn = tid < 255 ? wn_raw : 0;
  // sum over the workgroup: inside the wavefront through the cross-lane network, the four wavefront sums through LDS
#pragma unroll
  for (int s = 0; s < 6; s++) cpart = ops.add(cpart, cx.shfl_xor(cpart, 1u << s));
  if ((tid & 63) == 0) sc[tid >> 6] = cpart;
  cx.barrier();
  const u64 cin = ops.add(ops.add(sc[0], sc[1]), ops.add(sc[2], sc[3]));
  if (b == 0 && tid == 0 && rem) *rem = ops.add(H[0], ops.mul(tab.Y, cin));
  // S(first coefficient of the next lane), then down the run
  u64 r = ops.add(wn, ops.mul(zpk, cin));
  const u64 z = tab.z;
  u64 o[PL];
#pragma unroll
  for (int m = PL - 1; m >= 0; m--) { o[m] = r; r = ops.add(ops.mul(r, z), e[m]); }
  if (tab.scale != 1) {
#pragma unroll
    for (int m = 0; m < PL; m++) o[m] = ops.mul(o[m], tab.scale);
  }
  {
    // (the image is free: with LDS loads a lane has read nothing but its own run's words since the fill's barrier)
#pragma unroll
    for (int m = 0; m < PL; m++) buf[(PL + 1) * tid + m] = o[m];
    cx.barrier();
#pragma unroll
    for (int rr = 0; rr < PL; rr++) {
      const u32 kk = tid + 256 * rr;
      const size_t i = base + kk;
      if (full || i < d) quot[i] = buf[kk + kk / PL];
    }
  }
}

// ---- ONE launch (round 6): 16 bytes of HBM traffic per coefficient ---------------------------------------------------------------
// The two launches above read the dividend twice (24 + 2/PL bytes per coefficient against 16 algorithmic).  Here a workgroup keeps
// its chunk in registers while the chunk sums travel: workgroups of NL = 1024 lanes x PL = 4 or 8 coefficients (chunks of 4096 / 8192;
// two workgroups per CU, 512 resident at once; at most LINDIV1_MAX_CHUNKS = 1024 per call = one look-back entry per lane = 2^23
// coefficients at 8 per lane -- beyond 512 chunks the workgroups enter in two rounds), and
//   1  the lanes' Horner values U_t, weighted by z^(PL t), and their suffix SUMS over the wavefront (cross-lane network) and over the
//      16 wavefront sums (one barrier): additions only (lindiv1_chunk_sums) -- H_b published with ONE agent-scope write-through
//      store into the call's look-back array, before any lane has un-weighted its own sum (one product per lane, afterwards);
//   2  the carry G_(b+1) = sum_{j > b} H_j Y^(j-b-1): lane t polls entry b+1+t (agent-scope loads; "not there yet" = 2^64 - 1, no
//      canonical residue) -- ONE entry per lane, 512 x 8 polled wavefront requests per round against the 64 k of the round-2
//      one-launch form (scan_kernels.h lindiv_onepass_kernel: 2048 workgroups x 8 gathers; its waits were what made it slower);
//   3  the recurrence down the lane's run and the coalesced store through the LDS image, as in lindiv_apply_body.
// Measured (round 6, profiles/r06_lindiv_one.txt; two launches -> one, at every size): 2^16 9.5 -> 7.8 us, 2^20 11.2 -> 8.9, 2^21
// 14.6 -> 11.8, 2^22 22.4 -> 19.6 (0.43 of the roofline), 2^23 40.2 -> 34.9 us (0.48: two rounds stream).  Steps: the first version
// scanned the unweighted values (a field product per scan step: ten dependent products on the way to H_b) -- 21.6 us at 2^22; the
// weighted additive scans -- 20.2 us at 2^22 but SLOWER below (2^20: 11.6 -> 12.7 us) until the per-lane table entries that are not
// on the way to H_b were requested after H_b is out: per-lane reads of the kernel-argument segment are the slowest loads of the
// kernel, and five of them ahead of the first product cost more than the products saved; 4 coefficients per lane below 1.5 M
// coefficients (2^20: 10.9 -> 8.9 us: every CU gets a workgroup).  Variants that skip phases (first version, 2^22): load + all
// arithmetic 13.3 us, + stores 18.6, + the wait 21.5: with every chunk resident the whole device loads, computes and stores in
// lock-step (ONE round: nothing overlaps the arithmetic or the hand-off), where the second of two launches streams.
// Nothing can deadlock: workgroup i takes chunk nchunks-1-i and waits for HIGHER chunks only, i.e. for workgroups dispatched before
// it; every wait is bounded (the context's lb_wait gives up after 50 ms) and a workgroup whose wait ran out recomputes the chunk
// sums above it from the coefficients (slow, correct; with more chunks than resident workgroups the argument is the same: the lowest
// unfinished workgroup never waits for one that is not resident) -- which is why the entry point keeps the two-launch form for a quotient
// written over the dividend.  The array of the NEXT call is cleared here (two arrays per workspace slot, used alternately).
// Coefficients per lane PL (a template parameter; the entry point picks it by size, profiles/r06_lindiv_one.txt): 8 from 1.5 M
// coefficients up (2^22: 19.6 us; 4 per lane -- two rounds of resident workgroups -- 23.8 us), 4 below (2^20: 8.9 us against 10.9 us
// with 8 per lane, which leaves half the CUs without a workgroup; two launches: 11.1 us).  16 per lane (one workgroup per CU, half the
// per-lane overhead per coefficient) is no faster at 2^22 -- 21.4 us with direct loads, 24.0 us with the runs filled through the LDS
// image -- and much slower below: the instruction count is not what binds a one-round launch.
#ifndef RONK_LINDIV1_DIRECT
#define RONK_LINDIV1_DIRECT 1
#endif
constexpr int LINDIV1_NL = 1024;
constexpr int LINDIV1_NW = LINDIV1_NL / 64;
constexpr u32 LINDIV1_RESIDENT = 512;                              // workgroups resident at once (two per CU)
constexpr u32 LINDIV1_MAX_CHUNKS = LINDIV1_NL;                     // one look-back entry per lane: 2^23 coefficients at 8 per lane
constexpr bool LINDIV1_DIRECT = RONK_LINDIV1_DIRECT != 0;          // 16-byte loads of the lanes' runs instead of the LDS image
constexpr u64 LINDIV_LB_EMPTY = ~(u64)0;
constexpr int LINDIV1_STREAM = 2;   // lindiv1_chunk_scan: MODE of the recompute path
constexpr int LINDIV1_SC = 40;   // LDS words ahead of the image: 16 wavefront sums, 16 carry partials, flag, broadcast word
constexpr int lindiv1_lds_words(int pl) { return LINDIV1_SC + LINDIV1_NL * pl + LINDIV1_NL; }

struct LinDiv1Tab {
  u64 z;
  u64 scale;      // 1/b1
  u64 zp[65];     // z^(PL k), k <= 64
  u64 zw[16];     // z^(64 PL k)
  u64 zpinv[65];  // z^(-PL k), k <= 64
  u64 zwinv[16];  // z^(-64 PL k)
  u64 Y;          // z^(NL PL): one chunk
  u64 YA[16];     // Y^i
  u64 YB[16];     // Y^(16 i)
  u64 YC[4];      // Y^(256 i):   Y^t = YA[t & 15] YB[(t >> 4) & 15] YC[t >> 8], t < 1024
  u64 test_flags; // bit 0: every wait fails at once (the recompute path under test)
};

// p prime, z != 0 (mod p): the scans below run on values weighted by powers of z and need the inverse powers
inline void lindiv1_build_tab(u64 p, u64 z, u64 scale, u64 test_flags, int pl, LinDiv1Tab* t) {
  auto mulm = [p](u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % p); };
  auto powm = [&](u64 a, u64 e) { u64 r = 1 % p; while (e) { if (e & 1) r = mulm(r, a); a = mulm(a, a); e >>= 1; } return r; };
  t->z = z % p;
  t->scale = scale;
  t->test_flags = test_flags;
  const u64 zpl = powm(t->z, (u64)pl), zpli = powm(zpl, p - 2);
  u64 y = 1 % p, yi = 1 % p;
  for (int k = 0; k <= 64; k++) { t->zp[k] = y; t->zpinv[k] = yi; y = mulm(y, zpl); yi = mulm(yi, zpli); }
  const u64 zwave = t->zp[64], zwavei = t->zpinv[64];     // z^(+-64 pl)
  y = 1 % p; yi = 1 % p;
  for (int k = 0; k < 16; k++) { t->zw[k] = y; t->zwinv[k] = yi; y = mulm(y, zwave); yi = mulm(yi, zwavei); }
  t->Y = y;                                               // z^(1024 pl)
  const u64 Y16 = powm(t->Y, 16);
  u64 cc = 1 % p, dd = 1 % p;
  for (int i = 0; i < 16; i++) {
    t->YA[i] = cc; t->YB[i] = dd;
    cc = mulm(cc, t->Y); dd = mulm(dd, Y16);
  }
  t->YC[0] = 1 % p; t->YC[1] = powm(Y16, 16);
  t->YC[2] = mulm(t->YC[1], t->YC[1]); t->YC[3] = mulm(t->YC[2], t->YC[1]);
}

// Step 1 of the list above for chunk b, as ADDITIVE scans of weighted values: with V_t = U_t z^(PL t) (t = the lane's index in the
// chunk, U_t = the value of its run at z) the suffix sums F_t = sum_{s >= t} V_s need additions only -- six cross-lane steps inside
// the wavefront, the 16 wavefront sums through LDS, four more steps -- where the scan of the unweighted values pays a field product
// per step; F_0 = H_b is there BEFORE any lane has un-weighted anything (it is what the other workgroups wait for), and a lane needs
// one product, W_(t+1) = z^(-PL (t+1)) F_(t+1), instead of ten.  `wgt` = z^(PL t).  Returns F_(t+1) (ZERO behind the last lane) and
// H_b (every lane).  The lane's run e[] is the caller's (MODE != LINDIV1_STREAM).  One barrier; the caller must not touch
// sc[0 .. NW) before the next barrier.
template <int MODE, int PL, class Ops, class Ctx>
RONK_HD void lindiv1_chunk_sums(const Ops& ops, const u64* __restrict__ c, size_t d, u32 b, const LinDiv1Tab& tab, u64 wgt, u64* sc,
                                u64 (&e)[PL], u64* Fn_out, u64* H_out, Ctx& cx) {
  constexpr int NL = LINDIV1_NL, NW = LINDIV1_NW;
  const u32 tid = cx.tid(), lane = tid & 63, w = cx.wave();
  const u64 z = tab.z;
  u64 U;
  if constexpr (MODE == LINDIV1_STREAM) {
    // (the recompute path: the coefficients one at a time, nothing kept)
    const size_t i0 = (size_t)b * (NL * PL) + (size_t)PL * tid;
    U = 0;
    for (int m = PL - 1; m >= 0; m--) U = ops.add(ops.mul(U, z), i0 + m < d ? c[i0 + m] : 0);
  } else {
    // (the run was loaded by the caller: lindiv_one_body issues those loads before anything else)
    U = e[PL - 1];
#pragma unroll
    for (int m = PL - 2; m >= 0; m--) U = ops.add(ops.mul(U, z), e[m]);
  }
  u64 A = ops.mul(U, wgt);
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const u32 off = 1u << s;
    const u64 up = cx.shfl_down(A, off);
    if (lane + off < 64) A = ops.add(A, up);
  }
  if (lane == 0) sc[w] = A;
  cx.barrier();
  // S_k = sum of the wavefront sums k, k+1, ... (every wavefront computes all 16)
  u64 S = lane < (u32)NW ? sc[lane] : 0;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const u32 off = 1u << s;
    const u64 up = cx.shfl_down(S, off);
    if (lane + off < (u32)NW) S = ops.add(S, up);
  }
  const u64 Sn_raw = cx.shfl(S, w + 1 < (u32)NW ? w + 1 : 0);
  const u64 Sn = w + 1 < (u32)NW ? Sn_raw : 0;            // everything behind this wavefront
  *H_out = cx.shfl(S, 0);
  u64 An = cx.shfl_down(A, 1);
  if (lane == 63) An = 0;
  *Fn_out = ops.add(An, Sn);
}

template <int MODE, int PL, class Ops, class Ctx>
RONK_HD void lindiv_one_body(const Ops& ops, const u64* __restrict__ c, size_t d, const LinDiv1Tab& tab, u64* lb_cur, u64* lb_next,
                             u32 lb_words, u32 nchunks, u64* __restrict__ quot, u64* __restrict__ rem, Ctx& cx) {
  constexpr int NL = LINDIV1_NL, NW = LINDIV1_NW;
  const u32 tid = cx.tid(), lane = tid & 63, w = cx.wave();
  const u32 b = nchunks - 1 - cx.bid();
  u64* sc = cx.lds();
  u64* buf = sc + LINDIV1_SC;
  const size_t base = (size_t)b * (NL * PL);
  const bool full = base + NL * PL <= d;
  for (u32 i = cx.bid() * NL + tid; i < lb_words; i += nchunks * NL) lb_next[i] = LINDIV_LB_EMPTY;
  // Order of the loads (one in-order counter for all of them; per-lane table entries come out of the kernel-argument segment, the
  // slowest memory a kernel reads): only the ONE entry the path to H_b needs is requested ahead of the run; the four that serve the
  // carry and the way back are requested once H_b is out, their trip passes under the look-back wait.  (All five requested up
  // front: 2^20 coefficients 11.6 -> 12.7 us, 2^21 12.6 -> 14.2 us.)
  const u64 t_zp = tab.zp[lane];
  u64 e[PL], Fn, H;
  lindiv_load_run<MODE, NL, PL>(c, d, base, tid, buf, e, cx);
  const u64 wgt = ops.mul(t_zp, tab.zw[w]);                         // z^(PL t)
  lindiv1_chunk_sums<MODE, PL>(ops, c, d, b, tab, wgt, sc, e, &Fn, &H, cx);
  if (tid == 0) cx.lb_store(&lb_cur[b], H);
  const u64 unw = ops.mul(tab.zpinv[lane + 1], tab.zwinv[w]);       // z^(-PL (t + 1))
  const u64 yt = ops.mul(ops.mul(tab.YA[tid & 15], tab.YB[(tid >> 4) & 15]), tab.YC[(tid >> 8) & 3]);
  // carry: lane t takes H_(b+1+t) Y^t
  const u32 j = b + 1 + tid;
  u64 cpart = 0;
  bool late = false;
  if (j < nchunks) {
    u64 v = 0;
    if (cx.lb_wait(&lb_cur[j], &v, tab.test_flags)) cpart = ops.mul(v, yt);
    else late = true;
  }
  if (tid == 0) sc[2 * NW] = 0;
  cx.barrier();                                           // (also: every wavefront is done with sc[0 .. NW))
  if (late) sc[2 * NW] = 1;
#pragma unroll
  for (int s = 0; s < 6; s++) cpart = ops.add(cpart, cx.shfl_xor(cpart, 1u << s));
  if (lane == 0) sc[NW + w] = cpart;
  cx.barrier();
  u64 cin = 0;
  if (sc[2 * NW] == 0) {
#pragma unroll
    for (int k = 0; k < NW; k++) cin = ops.add(cin, sc[NW + k]);
  } else {
    // somebody never showed up: Horner over the chunk sums above b, every one recomputed here from the coefficients
    for (u32 jj = nchunks - 1; jj > b; jj--) {
      u64 F2, H2;
      cx.barrier();
      lindiv1_chunk_sums<LINDIV1_STREAM, PL>(ops, c, d, jj, tab, wgt, sc, e, &F2, &H2, cx);
      cin = ops.add(ops.mul(cin, tab.Y), H2);
    }
    cx.barrier();
  }
  const u64 ycin = ops.mul(tab.Y, cin);
  if (b == 0 && tid == 0 && rem) *rem = ops.add(H, ycin);            // c(z) = H_0 + Y G_1
  // S(first coefficient of lane t+1) = W_(t+1) + z^(PL (NL-1-t)) cin = z^(-PL (t+1)) (F_(t+1) + Y cin)
  u64 r = ops.mul(unw, ops.add(Fn, ycin));
  const u64 z = tab.z;
  // straight into the LDS image (it is free: since its fill's barrier a lane has read nothing but its own run's words, and the
  // recompute path ends on a barrier) -- eight results held in registers beside the run do not fit 64 VGPRs
  const bool scaled = tab.scale != 1;
#pragma unroll
  for (int m = PL - 1; m >= 0; m--) {
    buf[(PL + 1) * tid + m] = scaled ? ops.mul(r, tab.scale) : r;
    r = ops.add(ops.mul(r, z), e[m]);
  }
  cx.barrier();
#pragma unroll
  for (int rr = 0; rr < PL; rr++) {
    const u32 kk = tid + NL * rr;
    const size_t i = base + kk;
    if (full || i < d) quot[i] = buf[kk + kk / PL];
  }
}

}  // namespace ronk

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>

namespace ronk {

struct LinDivDevCtx {
  u64* lds_;
  __device__ __forceinline__ u32 tid() const { return threadIdx.x; }
  __device__ __forceinline__ u32 bid() const { return blockIdx.x; }
  __device__ __forceinline__ u32 wave() const { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
  __device__ __forceinline__ u64* lds() const { return lds_; }
  __device__ __forceinline__ void barrier() const { __syncthreads(); }
  __device__ __forceinline__ u64 shfl_down(u64 v, u32 off) const { return __shfl_down((unsigned long long)v, off, 64); }
  __device__ __forceinline__ u64 shfl_xor(u64 v, u32 mask) const { return __shfl_xor((unsigned long long)v, (int)mask, 64); }
  __device__ __forceinline__ u64 shfl(u64 v, u32 src) const { return __shfl((unsigned long long)v, (int)src, 64); }
  // the look-back array of the one-launch form: agent-scope write-through store / L2-bypassing load, no fence (the entry is the
  // whole message); a wait gives up after 50 ms of the 100 MHz wall clock
  __device__ __forceinline__ void lb_store(u64* p, u64 v) const { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ __forceinline__ bool lb_wait(const u64* p, u64* v, u64 test_flags) const {
    if (test_flags & 1) return false;
    u64 x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (x != LINDIV_LB_EMPTY) { *v = x; return true; }
    const u64 t0 = wall_clock64();
    for (;;) {
      __builtin_amdgcn_s_sleep(8);
      x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (x != LINDIV_LB_EMPTY) { *v = x; return true; }
      if (wall_clock64() - t0 > 5000000) return false;
    }
  }
  typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
  __device__ __forceinline__ void ld2(const u64* p, u64& a, u64& b) const {
    const v2 v = *reinterpret_cast<const v2*>(p);
    a = v.x; b = v.y;
  }
  __device__ __forceinline__ void st2(u64* p, u64 a, u64 b) const {
    v2 v; v.x = a; v.y = b;
    *reinterpret_cast<v2*>(p) = v;
  }
};

template <int MODE, class Ops>
__global__ void __launch_bounds__(256) lindiv_scan_kernel(Ops ops, const u64* __restrict__ c, size_t d, LinDivTab tab,
                                                           u64* __restrict__ W, u64* __restrict__ H) {
  __shared__ __attribute__((aligned(16))) u64 lds[lindiv_lds_words<MODE, false>()];
  LinDivDevCtx cx{lds};
  lindiv_scan_body<MODE>(ops, c, d, tab, W, H, cx);
}

template <int MODE, class Ops>
__global__ void __launch_bounds__(256) lindiv_apply_kernel2(Ops ops, const u64* __restrict__ c, size_t d, LinDivTab tab,
                                                             const u64* __restrict__ W, const u64* __restrict__ H,
                                                             u64* __restrict__ quot, u64* __restrict__ rem) {
  __shared__ __attribute__((aligned(16))) u64 lds[lindiv_lds_words<MODE, true>()];
  LinDivDevCtx cx{lds};
  lindiv_apply_body<MODE>(ops, c, d, tab, W, H, gridDim.x, quot, rem, cx);
}

// 8 wavefronts per SIMD: two workgroups per CU
template <int MODE, int PL, class Ops>
__global__ void __launch_bounds__(LINDIV1_NL, 8) lindiv_one_kernel(Ops ops, const u64* __restrict__ c, size_t d, LinDiv1Tab tab,
                                                                   u64* lb_cur, u64* lb_next, u32 lb_words,
                                                                   u64* __restrict__ quot, u64* __restrict__ rem) {
  __shared__ __attribute__((aligned(16))) u64 lds[lindiv1_lds_words(PL)];
  LinDivDevCtx cx{lds};
  lindiv_one_body<MODE, PL>(ops, c, d, tab, lb_cur, lb_next, lb_words, gridDim.x, quot, rem, cx);
}

}  // namespace ronk
#endif
