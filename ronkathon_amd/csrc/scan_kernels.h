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

template <class Ops>
__device__ __forceinline__ u64 block_sum_256(const Ops& ops, u64 v, u64* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
#pragma unroll
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = ops.add(red[tid], red[tid + s]);
    __syncthreads();
  }
  return red[0];
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
};

// Y^e from the binary expansion of e (wave-uniform or per lane)
template <class Ops>
__device__ __forceinline__ u64 ypow(const Ops& ops, const HornerTab2& tab, u32 e) {
  u64 r = ops.one();
#pragma unroll
  for (int s = 0; s < 20; s++)
    if (e & (1u << s)) r = ops.mul(r, tab.Yp[s]);
  return r;
}

// H_b = sum_{k < 2048} c[base + k] z^k for the calling workgroup (entries beyond d read as ZERO); result valid in every lane
template <class Ops>
__device__ __forceinline__ u64 chunk_sum8(const Ops& ops, const u64* __restrict__ c, size_t d, const HornerTab2& tab, u64* red) {
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * FCH;
  u64 e[8];
  if (base + FCH <= d) {
#pragma unroll
    for (int r = 7; r >= 0; r--) e[r] = c[base + tid + 256 * r];
  } else {
#pragma unroll
    for (int r = 7; r >= 0; r--) { const size_t i = base + tid + 256 * r; e[r] = i < d ? c[i] : 0; }
  }
  u64 acc = e[7];
#pragma unroll
  for (int r = 6; r >= 0; r--) acc = ops.add(ops.mul(acc, tab.z256), e[r]);
  acc = ops.mul(acc, tab.zt[tid]);
  return block_sum_256(ops, acc, red);
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
      cpart = ops.mul(cpart, ypow(ops, tab, (u32)tid));
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

}  // namespace ronk
