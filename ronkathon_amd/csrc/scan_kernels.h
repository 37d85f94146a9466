// scan_kernels.h -- Horner-type scans: Polynomial::evaluate and division by a linear divisor (kzg::open).
//
// Reference semantics restated:
//   evaluate            src/polynomial/mod.rs:133-139     sum_i c_i x^i
//   poly / (b0 + b1 x)  src/polynomial/mod.rs:170-225 via src/kzg/setup.rs:63-78 (divisor [-z, 1])
// For a divisor b1*(x - z), z = -b0/b1, long division is the recurrence q_(d-1) = 0,
// q_j = (c_(j+1) + z*q_(j+1)) ... scaled by 1/b1, remainder = c(z): an affine suffix scan whose maps all
// share the multiplier z, so composing k of them only needs z^k.
//
// Both are HBM-bound streaming jobs (8 B per coefficient read; the division also writes 8 B), organised
// in chunks of CHUNK = 4096 coefficients = one 256-lane workgroup x 16:
//   A  chunk_horner_kernel   H_b = sum_k c[base_b + k] z^k            (coalesced, lane-strided Horner in z^256)
//   S  chunk_carry_kernel    G_b = H_b + z^4096 G_(b+1), G_nchunks = 0  (one workgroup of 256; G_0 = c(z) = evaluate)
//   B  lindiv_apply_kernel   q inside a chunk from the carry G_(b+1)   (LDS transpose, 16 contiguous per lane)
// Powers of z come from the host in a kernarg table (HornerTab): no per-lane pow().
#pragma once
#include <hip/hip_runtime.h>

#include "field_kernels.h"

namespace ronk {

constexpr int HCHUNK = 4096;  // coefficients per workgroup: 256 work-items x 16

struct HornerTab {
  u64 zt[256];    // z^t, t < 256
  u64 z256;       // z^256
  u64 z16p[8];    // z^(16 * 2^s), s < 8
  u64 Zp[10];     // (z^4096)^(2^s), s < 10
  u64 z;
  u64 scale;      // 1/b1 (1 for evaluate and for monic divisors)
};

// sum over the 256 work-items of a workgroup, valid in every lane: inside a wavefront through the cross-lane network, the
// four wavefront sums through red[0..3] (ONE barrier; round 2 had an eight-step LDS tree with nine).  A caller that uses
// `red` again synchronises first.
template <class Ops>
__device__ __forceinline__ u64 block_sum_256(const Ops& ops, u64 v, u64* red) {
#pragma unroll
  for (int s = 0; s < 6; s++) v = ops.add(v, (u64)__shfl_xor((unsigned long long)v, 1 << s, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ops.add(ops.add(red[0], red[1]), ops.add(red[2], red[3]));
}

// A: H[b] = sum_{k < 4096} c[base + k] z^k  (entries beyond d read as ZERO)
template <class Ops>
__global__ void __launch_bounds__(256) chunk_horner_kernel(Ops ops, const u64* __restrict__ c, size_t d, HornerTab tab,
                                                            u64* __restrict__ H) {
  __shared__ u64 red[256];
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * HCHUNK;
  u64 e[16];
  // loads issued in the order the recurrence consumes them (e[15] first), so it can start on the first arrivals
  if (base + HCHUNK <= d) {
#pragma unroll
    for (int r = 15; r >= 0; r--) e[r] = c[base + tid + 256 * r];
  } else {
#pragma unroll
    for (int r = 15; r >= 0; r--) { const size_t i = base + tid + 256 * r; e[r] = i < d ? c[i] : 0; }
  }
  // lane t holds k = t + 256 r: Horner in z^256 over r, then the lane's own offset z^t
  u64 acc = e[15];
#pragma unroll
  for (int r = 14; r >= 0; r--) acc = ops.add(ops.mul(acc, tab.z256), e[r]);
  acc = ops.mul(acc, tab.zt[tid]);
  const u64 tot = block_sum_256(ops, acc, red);
  if (tid == 0) H[blockIdx.x] = tot;
}

// S: carry[b] = G_(b+1) with G_b = H_b + Z G_(b+1), G_nchunks = 0 (Z = z^4096); *total = G_0.
// One workgroup of 256; segments of 1024 chunks from the top down.  Inside a segment each lane owns 4 consecutive
// chunks (sequential Horner), the lanes' results are scanned with the powers Z^(4*2^s) (Hillis-Steele, 8 steps),
// then each lane re-runs its 4 entries from the value just above them.
template <class Ops>
__global__ void __launch_bounds__(256) chunk_carry_kernel(Ops ops, const u64* __restrict__ H, size_t nchunks, HornerTab tab,
                                                           u64* __restrict__ carry, u64* __restrict__ total) {
  __shared__ u64 buf[256];
  __shared__ u64 s_in;
  const int tid = threadIdx.x;
  const u64 Z = tab.Zp[0];
  const size_t nseg = (nchunks + 1023) / 1024;
  u64 incoming = 0;  // G at the first chunk above this segment
  for (size_t seg = nseg; seg-- > 0;) {
    const size_t i0 = seg * 1024 + 4 * (size_t)tid;
    u64 h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = i0 + k < nchunks ? H[i0 + k] : 0;
    if (tid == 255) h[3] = ops.add(h[3], ops.mul(Z, incoming));
    // lane value: sum_k h[k] Z^k
    u64 v = h[3];
#pragma unroll
    for (int k = 2; k >= 0; k--) v = ops.add(ops.mul(v, Z), h[k]);
    buf[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; s++) {
      const int off = 1 << s;
      u64 w = v;
      if (tid + off < 256) w = ops.add(v, ops.mul(tab.Zp[s + 2], buf[tid + off]));   // (Z^4)^(2^s)
      __syncthreads();
      buf[tid] = v = w;
      __syncthreads();
    }
    // v == G at chunk i0; the value just above this lane's 4 entries is the next lane's G (0 above the top)
    u64 g = tid < 255 ? buf[tid + 1] : 0;
    if (tid == 255) g = 0;   // incoming is already folded into h[3]
#pragma unroll
    for (int k = 3; k >= 0; k--) {
      g = ops.add(h[k], ops.mul(Z, g));                     // G at chunk i0 + k
      const size_t idx = i0 + k;
      if (idx >= 1 && idx <= nchunks) carry[idx - 1] = idx < nchunks ? g : 0;
    }
    if (tid == 0) s_in = g;                                  // == v: G at the first chunk of the segment
    __syncthreads();
    incoming = s_in;
    __syncthreads();
  }
  if (tid == 0) {
    if (nchunks >= 1 && (nchunks % 1024) == 0) carry[nchunks - 1] = 0;  // top chunk when the last segment is full
    if (total) *total = incoming;
  }
}

// B: quot[base + k] = scale * sum_{i > base+k} c_i z^(i-base-k-1), using carry[b] = G_(b+1).
// LDS image padded by one entry per 16 (index k + k/16): the coalesced lane-strided fill and the
// 16-contiguous-per-lane reads are both conflict-free.
template <class Ops>
__global__ void __launch_bounds__(256) lindiv_apply_kernel(Ops ops, const u64* __restrict__ c, size_t d, HornerTab tab,
                                                            const u64* __restrict__ carry, u64* __restrict__ quot) {
  __shared__ u64 buf[HCHUNK + HCHUNK / 16];
  __shared__ u64 sc[256];
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * HCHUNK;
  const bool full = base + HCHUNK <= d;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int k = tid + 256 * r;
    const size_t i = base + k;
    buf[k + (k >> 4)] = (full || i < d) ? c[i] : 0;
  }
  const u64 cin = carry[blockIdx.x];
  __syncthreads();
  u64 e[16];
#pragma unroll
  for (int m = 0; m < 16; m++) e[m] = buf[17 * tid + m];
  const u64 z = tab.z;
  // U_t = sum_m e[m] z^m; the top lane also absorbs the chunk's incoming carry
  u64 U = e[15];
#pragma unroll
  for (int m = 14; m >= 0; m--) U = ops.add(ops.mul(U, z), e[m]);
  if (tid == 255) U = ops.add(U, ops.mul(tab.z16p[0], cin));
  // W_t = U_t + z^16 W_(t+1): suffix scan with doubling powers of z^16
  sc[tid] = U;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const int off = 1 << s;
    u64 w = U;
    if (tid + off < 256) w = ops.add(U, ops.mul(tab.z16p[s], sc[tid + off]));
    __syncthreads();
    sc[tid] = U = w;
    __syncthreads();
  }
  // run the recurrence down this lane's 16 entries from the value just above them
  u64 r = tid < 255 ? sc[tid + 1] : cin;
  u64 o[16];
#pragma unroll
  for (int m = 15; m >= 0; m--) { o[m] = r; r = ops.add(ops.mul(r, z), e[m]); }
  if (tab.scale != 1) {
#pragma unroll
    for (int m = 0; m < 16; m++) o[m] = ops.mul(o[m], tab.scale);
  }
#pragma unroll
  for (int m = 0; m < 16; m++) buf[17 * tid + m] = o[m];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 16; rr++) {
    const int k = tid + 256 * rr;
    const size_t i = base + k;
    if (full || i < d) quot[i] = buf[k + (k >> 4)];
  }
}


// ---- fused forms (the default up to 2^23 coefficients): no separate carry kernel --------------------------------------
// Chunks of FCH = 2048 coefficients (256 work-items x 8: half the dependent Horner chain of the 4096 form).
//   evaluate   weighted_chunk_sum8_kernel -- every workgroup writes H_b * Y^b (Y = z^2048, the power from the bits of b) --
//              and partial_sum_kernel, one workgroup adding them up (a plain sum instead of the serial carry scan).
//   division   TWO launches: chunk_sum8_kernel (H_b) and lindiv_fused_kernel, whose workgroup b derives its own incoming
//              carry G_(b+1) = sum_{j>b} H_j Y^(j-b-1) from the H array (<= 4096 entries, L2 resident: 16 values per lane
//              and one block reduction) instead of waiting for a serial scan kernel: reads 8 + 8, writes 8 B/coefficient.
constexpr int FCH = 2048;

struct HornerTab2 {
  u64 zt[256];    // z^t, t < 256
  u64 z256;       // z^256
  u64 z8p[8];     // z^(8 * 2^s), s < 8
  u64 Yp[20];     // (z^2048)^(2^s), s < 20
  u64 z;
  u64 scale;      // 1/b1 (1 for evaluate and for monic divisors)
  u64 YA[16], YB[16], YC[16];   // Y^i, Y^(16 i), Y^(256 i):  Y^e = YA[e & 15] * YB[(e >> 4) & 15] * YC[e >> 8], e < 4096  (one-launch forms)
  u64 z8A[16], z8B[16];  // z^(8 i), z^(128 i):   z^(8 t) = z8A[t & 15] * z8B[t >> 4], t < 256
  u64 test_flags;        // bit 0: every wait of the one-launch forms gives up at once (tests of the recompute path)
};

// Y^e, e < 2^20 (wave-uniform or per lane): the low twelve bits from the three 16-entry tables (two products; round 2 spent
// up to twelve predicated products here, on every lane of every workgroup of an evaluate), the rest from the binary expansion
template <class Ops>
__device__ __forceinline__ u64 ypow(const Ops& ops, const HornerTab2& tab, u32 e) {
  u64 r = ops.mul(ops.mul(tab.YA[e & 15], tab.YB[(e >> 4) & 15]), tab.YC[(e >> 8) & 15]);
  if (e >> 12) {
#pragma unroll
    for (int s = 12; s < 20; s++)
      if (e & (1u << s)) r = ops.mul(r, tab.Yp[s]);
  }
  return r;
}

// H_b = sum_{k < 2048} c[base + k] z^k for the calling workgroup (entries beyond d read as ZERO); result valid in every lane
template <class Ops>
__device__ __forceinline__ u64 chunk_sum8_at(const Ops& ops, const u64* __restrict__ c, size_t d, const HornerTab2& tab, u64* red,
                                             size_t chunk) {
  const int tid = threadIdx.x;
  const size_t base = chunk * FCH;
  u64 e[8];
  if (base + FCH <= d) {
#pragma unroll
    for (int r = 7; r >= 0; r--) e[r] = c[base + tid + 256 * r];
  } else {
#pragma unroll
    for (int r = 7; r >= 0; r--) { const size_t i = base + tid + 256 * r; e[r] = i < d ? c[i] : 0; }
  }
  // sum_r e[r] z^(256 r) as two Horner chains in z^512 (even and odd r): the same seven products, four deep instead of seven
  const u64 z512 = ops.mul(tab.z256, tab.z256);
  u64 ae = e[6], ao = e[7];
#pragma unroll
  for (int r = 4; r >= 0; r -= 2) { ae = ops.add(ops.mul(ae, z512), e[r]); ao = ops.add(ops.mul(ao, z512), e[r + 1]); }
  u64 acc = ops.add(ops.mul(ao, tab.z256), ae);
  acc = ops.mul(acc, tab.zt[tid]);
  return block_sum_256(ops, acc, red);
}
template <class Ops>
__device__ __forceinline__ u64 chunk_sum8(const Ops& ops, const u64* __restrict__ c, size_t d, const HornerTab2& tab, u64* red) {
  return chunk_sum8_at(ops, c, d, tab, red, (size_t)blockIdx.x);
}

template <class Ops>
__global__ void __launch_bounds__(256) chunk_sum8_kernel(Ops ops, const u64* __restrict__ c, size_t d, HornerTab2 tab,
                                                          u64* __restrict__ H) {
  __shared__ u64 red[256];
  const u64 tot = chunk_sum8(ops, c, d, tab, red);
  if (threadIdx.x == 0) H[blockIdx.x] = tot;
}

// evaluate: weighted chunk sums H_b * Y^b, then a plain field sum of the nchunks values by one workgroup.
// (A single-launch form -- last workgroup to arrive on a device-scope counter does the sum -- was measured and dropped:
// with release/acquire on the counter every workgroup flushes and invalidates its XCD's L2, 52 us for 2^22 coefficients;
// with relaxed atomics the 2048 same-address arrivals serialise at the memory side, 34 us; two launches: 11 us.)
template <class Ops>
__global__ void __launch_bounds__(256) weighted_chunk_sum8_kernel(Ops ops, const u64* __restrict__ c, size_t d, HornerTab2 tab,
                                                                   u64* __restrict__ partial) {
  __shared__ u64 red[256];
  const u64 tot = chunk_sum8(ops, c, d, tab, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = ops.mul(tot, ypow(ops, tab, blockIdx.x));
}
template <class Ops>
__global__ void __launch_bounds__(256) partial_sum_kernel(Ops ops, const u64* __restrict__ partial, size_t n, u64* __restrict__ total) {
  __shared__ u64 red[256];
  u64 acc = 0;
  for (size_t i = threadIdx.x; i < n; i += 256) acc = ops.add(acc, partial[i]);
  const u64 sum = block_sum_256(ops, acc, red);
  if (threadIdx.x == 0) *total = sum;
}

// division by (x - z), chunk b: incoming carry from the H array, then the in-chunk suffix recurrence.
// quot[base + k] = scale * sum_{i > base+k} c_i z^(i-base-k-1); block 0 also writes the remainder c(z) = H_0 + Y*G_1.
template <class Ops>
__global__ void __launch_bounds__(256) lindiv_fused_kernel(Ops ops, const u64* __restrict__ c, size_t d, HornerTab2 tab,
                                                            const u64* __restrict__ H, u64* __restrict__ quot,
                                                            u64* __restrict__ rem) {
  __shared__ u64 buf[FCH + FCH / 8];
  __shared__ u64 sc[256];
  const int tid = threadIdx.x;
  const u32 b = blockIdx.x, nchunks = gridDim.x;
  const size_t base = (size_t)b * FCH;
  const bool full = base + FCH <= d;
  // chunk -> LDS (coalesced, lane-strided; one pad entry per 8 so that the 8-contiguous-per-lane reads are conflict free)
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int k = tid + 256 * r;
    const size_t i = base + k;
    buf[k + (k >> 3)] = (full || i < d) ? c[i] : 0;
  }
  // incoming carry G_(b+1) = sum_{j > b} H_j Y^(j-b-1): lane t takes j = b+1+t+256q (Horner in Y^256 over q), times Y^t
  u64 cpart = 0;
  {
    const u32 first = b + 1 + tid;
    if (first < nchunks) {
      const u32 cnt = (nchunks - first + 255) / 256;
      const u64 Y256 = tab.Yp[8];
      for (u32 q = cnt; q-- > 0;) cpart = ops.add(ops.mul(cpart, Y256), H[first + 256 * q]);
      cpart = ops.mul(cpart, ops.mul(tab.YA[tid & 15], tab.YB[tid >> 4]));   // Y^tid from two 16-entry tables (was: 8 predicated products)
    }
  }
  const u64 cin = block_sum_256(ops, cpart, sc);            // (its barriers also publish buf)
  __syncthreads();
  if (b == 0 && tid == 0 && rem) *rem = ops.add(H[0], ops.mul(tab.Yp[0], cin));
  u64 e[8];
#pragma unroll
  for (int m = 0; m < 8; m++) e[m] = buf[9 * tid + m];
  const u64 z = tab.z;
  // U_t = sum_m e[m] z^m; the top lane also absorbs the chunk's incoming carry
  u64 U = e[7];
#pragma unroll
  for (int m = 6; m >= 0; m--) U = ops.add(ops.mul(U, z), e[m]);
  if (tid == 255) U = ops.add(U, ops.mul(tab.z8p[0], cin));
  // W_t = U_t + z^8 W_(t+1): suffix scan with doubling powers of z^8
  sc[tid] = U;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const int off = 1 << s;
    u64 w = U;
    if (tid + off < 256) w = ops.add(U, ops.mul(tab.z8p[s], sc[tid + off]));
    __syncthreads();
    sc[tid] = U = w;
    __syncthreads();
  }
  // run the recurrence down this lane's 8 entries from the value just above them
  u64 r = tid < 255 ? sc[tid + 1] : cin;
  u64 o[8];
#pragma unroll
  for (int m = 7; m >= 0; m--) { o[m] = r; r = ops.add(ops.mul(r, z), e[m]); }
  if (tab.scale != 1) {
#pragma unroll
    for (int m = 0; m < 8; m++) o[m] = ops.mul(o[m], tab.scale);
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; m++) buf[9 * tid + m] = o[m];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 8; rr++) {
    const int k = tid + 256 * rr;
    const size_t i = base + k;
    if (full || i < d) quot[i] = buf[k + (k >> 3)];
  }
}

// ---- one-launch forms: evaluate (the default up to LB_MAX chunks outside stream capture); division (RONK_ONEPASS_DIV=1) ----
// Measured at 2^22 coefficients (same box, device time): evaluate 14.2 -> 12.8 us; division SLOWER, 28.4 -> 32.3 us with
// every workgroup polling every higher chunk sum and 35.5 us with the two-level form below, so the division keeps its two
// launches by default.  Per-workgroup timestamps (wall_clock64 at each phase) show why: all 2048 workgroups have
// published their sum 7.4 .. 13.1 us after the launch, but the carries then arrive 10 .. 28 us after it, later the more
// entries a workgroup has to poll -- 2048 workgroups polling the same handful of cache lines with L2-bypassing loads
// are served one after the other by the memory channel that owns the line (~1 ns each), where the second launch of the
// two-launch form reads the same array through its XCD's L2.
// The workgroups of ONE launch hand their chunk sums to each other through an array in global memory that is written
// with agent-scope write-through stores (global_store sc1) and polled with agent-scope loads (global_load sc1): the only
// cross-XCD traffic is 8 bytes per chunk, no fence (a release / acquire pair writes back and invalidates a whole XCD L2:
// 52 us for the first one-launch evaluate), no same-address atomics (2048 arrivals on one counter serialise: 34 us).
//   * "not there yet" is the value LB_EMPTY = 2^64 - 1, which no canonical residue of any modulus p < 2^64 equals;
//   * the array of the NEXT call is re-initialised by this one (two arrays per workspace slot, used alternately; every
//     workgroup clears a slice), so no memset launch sits between calls.  Not capturable in a hipGraph (the parity is
//     host state): the entry points use the two-launch forms while the stream is capturing;
//   * nothing can deadlock: evaluate -- only workgroup 0 waits, and it waits for workgroups that wait for nobody;
//     division -- workgroup i handles chunk nchunks-1-i and waits for the chunk sums of HIGHER chunks only, i.e. of
//     workgroups with LOWER ids (dispatched before it).  Every wait is bounded (LB_TIMEOUT of the 100 MHz wall clock);
//     a workgroup that runs out of patience recomputes what it needs from the coefficients themselves (slow, correct).
constexpr u64 LB_EMPTY = ~(u64)0;
constexpr u32 LB_MAX = 8192;                 // chunks per call (2^24 coefficients)
constexpr u32 LB_GS = 32;                    // chunks per group of the division's two-level look-back
constexpr u32 LB_DIV_MAX = 4096;             // chunks per division call: 31 peers + 127 group sums fit the 256 lanes
constexpr u32 LB_WORDS = LB_MAX + LB_MAX / LB_GS;   // one look-back array: chunk sums, then group sums
constexpr u64 LB_TIMEOUT = 5000000;          // 50 ms of wall_clock64() ticks

__device__ __forceinline__ u64 lb_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// polls until the entry is there; false (and *v undefined) after LB_TIMEOUT
__device__ __forceinline__ bool lb_wait(const u64* p, u64* v, u64 test_flags) {
  if (test_flags & 1) return false;
  u64 x = lb_load(p);
  if (x != LB_EMPTY) { *v = x; return true; }
  const u64 t0 = wall_clock64();
  for (;;) {
    __builtin_amdgcn_s_sleep(4);
    x = lb_load(p);
    if (x != LB_EMPTY) { *v = x; return true; }
    if (wall_clock64() - t0 > LB_TIMEOUT) return false;
  }
}
// up to 8 entries p[0], p[stride], ... at once: all loads are issued before the first one is looked at (one memory round
// trip instead of eight), entries that are not there yet are waited for one by one.  false after a timeout.
__device__ __forceinline__ bool lb_gather8(const u64* p, u32 stride, u32 cnt, u64* v, u64 test_flags) {
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = (u32)i < cnt ? lb_load(p + (size_t)i * stride) : 0;
  bool ok = !(test_flags & 1);
#pragma unroll
  for (int i = 0; i < 8; i++)
    if ((u32)i < cnt && v[i] == LB_EMPTY && ok) ok = lb_wait(p + (size_t)i * stride, &v[i], test_flags);
  return ok;
}
__device__ __forceinline__ void lb_clear_next(u64* lb_next) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < LB_WORDS; i += gridDim.x * 256) lb_next[i] = LB_EMPTY;
}
// true if any lane of the workgroup passes true (LDS word + barriers)
__device__ __forceinline__ bool block_any(bool v, u32* flag) {
  if (threadIdx.x == 0) *flag = 0;
  __syncthreads();
  if (v) *flag = 1;
  __syncthreads();
  const bool r = *flag != 0;
  __syncthreads();
  return r;
}

// evaluate in one launch: workgroup b publishes H_b * Y^b; workgroup 0 adds them up as they arrive.
template <class Ops>
__global__ void __launch_bounds__(256, 8) eval_onepass_kernel(Ops ops, const u64* __restrict__ c, size_t d, HornerTab2 tab,
                                                            u64* lb_cur, u64* lb_next, u64* __restrict__ total) {
  __shared__ u64 red[256];
  __shared__ u32 flag;
  const int tid = threadIdx.x;
  const u32 b = blockIdx.x, nch = gridDim.x;
  lb_clear_next(lb_next);
  const u64 tot = chunk_sum8(ops, c, d, tab, red);
  const u64 mine = ops.mul(tot, ypow(ops, tab, b));
  if (b != 0) {
    if (tid == 0) lb_store(&lb_cur[b], mine);
    return;
  }
  u64 acc = tid == 0 ? mine : 0;
  bool late = false;
  for (u32 i0 = tid ? tid : 256; i0 < nch && !late; i0 += 8 * 256) {
    const u32 cnt = (nch - i0 + 255) / 256;
    u64 v[8];
    if (!lb_gather8(&lb_cur[i0], 256, cnt < 8 ? cnt : 8, v, tab.test_flags)) { late = true; break; }
#pragma unroll
    for (int k = 0; k < 8; k++)
      if ((u32)k < cnt) acc = ops.add(acc, v[k]);
  }
  if (block_any(late, &flag)) {
    // somebody never showed up: Horner over the chunk sums, every one recomputed here
    u64 g = 0;
    for (u32 j = nch; j-- > 0;) {
      __syncthreads();
      const u64 h = chunk_sum8_at(ops, c, d, tab, red, (size_t)j);
      g = ops.add(ops.mul(g, tab.Yp[0]), h);
    }
    if (tid == 0) *total = g;
    return;
  }
  __syncthreads();
  const u64 sum = block_sum_256(ops, acc, red);
  if (tid == 0) *total = sum;
}

// division by (x - z) in one launch.  Workgroup i handles chunk b = nchunks-1-i: local Horner sums and their suffix scan
// (no carry needed), publishes H_b, collects the carry G_(b+1) = sum_{j>b} H_j Y^(j-b-1), then runs the recurrence down
// its lanes' 8 entries.  Reads 8, writes 8 bytes per coefficient.
// The carry is collected on two levels (every workgroup polling every higher chunk sum is 2 M uncached 8-byte loads at
// 2^22 coefficients -- measured: 32 us, slower than two launches): chunks form groups of LB_GS = 32; the workgroup of a
// group's LOWEST chunk, which collects the group's other 31 sums anyway, also publishes the group sum
// S_g = sum_l H_(32g+l) Y^l.  A workgroup then needs the <= 31 chunk sums above it in its own group and the <= 127 group
// sums above that: one entry per lane, ~100 polled entries per workgroup instead of ~1000.
template <class Ops>
__global__ void __launch_bounds__(256, 8) lindiv_onepass_kernel(Ops ops, const u64* __restrict__ c, size_t d, HornerTab2 tab,
                                                              u64* lb_cur, u64* lb_next, u64* __restrict__ quot,
                                                              u64* __restrict__ rem) {
  __shared__ u64 buf[FCH + FCH / 8];   // 20 KiB with sc: eight workgroups per CU, i.e. 2^22 coefficients resident at once
  __shared__ u64 sc[256];
  u32* const flag = reinterpret_cast<u32*>(&buf[FCH + FCH / 8 - 1]);   // the image uses k + k/8 <= FCH + FCH/8 - 2
  const int tid = threadIdx.x;
  const u32 nchunks = gridDim.x, b = nchunks - 1 - blockIdx.x;
  const size_t base = (size_t)b * FCH;
  const bool full = base + FCH <= d;
  lb_clear_next(lb_next);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int k = tid + 256 * r;
    const size_t i = base + k;
    buf[k + (k >> 3)] = (full || i < d) ? c[i] : 0;
  }
  __syncthreads();
  const u64 z = tab.z;
  u64 U;
  {
    u64 e[8];
#pragma unroll
    for (int m = 0; m < 8; m++) e[m] = buf[9 * tid + m];
    U = e[7];
#pragma unroll
    for (int m = 6; m >= 0; m--) U = ops.add(ops.mul(U, z), e[m]);
  }
  // W_t = U_t + z^8 W_(t+1) inside the chunk
  sc[tid] = U;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const int off = 1 << s;
    u64 w = U;
    if (tid + off < 256) w = ops.add(U, ops.mul(tab.z8p[s], sc[tid + off]));
    __syncthreads();
    sc[tid] = U = w;
    __syncthreads();
  }
  if (tid == 0) lb_store(&lb_cur[b], U);                    // H_b = W_0
  const u64 wup = tid < 255 ? sc[tid + 1] : 0;              // W_(t+1)
  // carry, level 1: lanes 0..30 take the chunk sums above b inside its group, H_(b+1+t) Y^t
  const u32 g = b / LB_GS, l = b % LB_GS, ngroups = (nchunks + LB_GS - 1) / LB_GS;
  const u64* const lbS = lb_cur + LB_MAX;                  // group sums
  u64 cpart = 0, v = 0;
  bool late = false;
  {
    const u32 j = b + 1 + tid;
    if (tid < LB_GS - 1 && j < (g + 1) * LB_GS && j < nchunks) {
      if (lb_wait(&lb_cur[j], &v, tab.test_flags)) cpart = ops.mul(v, tab.YA[tid & 15]);
      else late = true;
      if (tid >= 16) cpart = ops.mul(cpart, tab.YB[1]);
    }
  }
  if (l == 0 && ngroups > 1) {
    // this workgroup owns the group sum: S_g = H_b + Y * (level-1 sum); publish before looking at other groups
    const bool late1 = block_any(late, flag);
    const u64 L = block_sum_256(ops, cpart, sc);
    if (!late1 && tid == 0) lb_store(&lb_cur[LB_MAX + g], ops.add(U, ops.mul(tab.Yp[0], L)));
    __syncthreads();
    late = late1;
  }
  // level 2: lanes 32.. take the group sums above, S_(g+1+u) Y^(32 (g+1+u) - b - 1)
  if (tid >= 32 && !late) {
    const u32 u = tid - 32, gg = g + 1 + u;
    if (gg < ngroups) {
      if (lb_wait(&lbS[gg], &v, tab.test_flags)) {
        const u32 e = (LB_GS - 1 - l) + LB_GS * u;          // < 4096
        cpart = ops.mul(v, ops.mul(ops.mul(tab.YA[e & 15], tab.YB[(e >> 4) & 15]), tab.YC[e >> 8]));
      } else {
        late = true;
      }
    }
  }
  u64 cin;
  if (block_any(late, flag)) {
    cin = 0;
    for (u32 j = nchunks - 1; j > b; j--) {
      __syncthreads();
      const u64 h = chunk_sum8_at(ops, c, d, tab, sc, (size_t)j);
      cin = ops.add(ops.mul(cin, tab.Yp[0]), h);
    }
    __syncthreads();
  } else {
    cin = block_sum_256(ops, cpart, sc);
  }
  if (b == 0 && tid == 0 && rem) *rem = ops.add(U, ops.mul(tab.Yp[0], cin));   // c(z) = H_0 + Y G_1
  // value above this lane's entries: W_(t+1) + z^(8(255-t)) * cin
  const u32 tp = 255 - tid;
  u64 r = ops.add(wup, ops.mul(ops.mul(tab.z8A[tp & 15], tab.z8B[tp >> 4]), cin));
  // (the lane's coefficients are read from the LDS image again instead of being held in 16 registers across the gather)
  u64 o[8];
#pragma unroll
  for (int m = 7; m >= 0; m--) { o[m] = r; r = ops.add(ops.mul(r, z), buf[9 * tid + m]); }
  if (tab.scale != 1) {
#pragma unroll
    for (int m = 0; m < 8; m++) o[m] = ops.mul(o[m], tab.scale);
  }
#pragma unroll
  for (int m = 0; m < 8; m++) buf[9 * tid + m] = o[m];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 8; rr++) {
    const int k = tid + 256 * rr;
    const size_t i = base + k;
    if (full || i < d) quot[i] = buf[k + (k >> 3)];
  }
}

}  // namespace ronk
