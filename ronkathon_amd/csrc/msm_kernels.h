// msm_kernels.h -- bucket-method (Pippenger) multi-scalar multiplication over BN254 G1: sum_i k_i * P_i.
//
// kzg::commit on a production-size curve (reference: src/kzg/setup.rs:48-60 folds g1_srs[i] * coeff[i]
// with AffinePoint's Mul/Add, src/curve/mod.rs:157-211).  Scalars are plain 256-bit integers (any value; the caller's
// field elements mod r), split in W signed digits of c bits, digit in [-2^(c-1), 2^(c-1)]:
//
//   prepare   points -> Montgomery form, on-curve check (AffinePoint::new's assert, src/curve/mod.rs:79)
//   count     histogram of (window, |digit|) over all scalars                        } counting sort of the
//   scan      exclusive prefix sum of the histogram                                  } n*W (point, sign)
//   scatter   point indices (sign in bit 31) into their bucket's run                 } entries by bucket
//   accumulate  one lane per bucket: XYZZ sum of its run (mixed additions, 8M + 2S each): the dominant kernel
//   reduce    sum_b (b+1) B_b per window as bit planes: Q_k = sum of the buckets whose weight has bit k set (plain sums, no
//             serial running-sum chain: a lane adds at most 16 points, then three to five 8-to-1 stages), the rest
//             (sum_k 2^k Q_k, then Horner over the windows: ~270 dependent doublings) runs on the host, where one
//             doubling takes 0.5 us instead of 14 us on a single lane.
// Everything here is VALU-bound 32-bit limb arithmetic (bn254.h); no MFMA (carry chains), HBM traffic is the gather of
// 64-byte points (n*W of them) and is small beside ~3 400 instructions per mixed addition.
#pragma once
#include <hip/hip_runtime.h>

#include "bn254.h"

namespace ronk {

using bn254::Affine;
using bn254::Fp;
using bn254::Xyzz;
typedef uint64_t u64;
typedef uint32_t u32;

struct MsmShape {
  u32 n;        // points
  u32 c;        // window bits
  u32 W;        // windows, W*c >= 257
  u32 NB;       // buckets per window = 2^(c-1), weights 1 .. NB
};

// signed digit w of a 256-bit scalar (4 x u64 little endian) given the carry from the digit below; updates the carry
__device__ __forceinline__ int msm_digit(const u64* k, u32 w, u32 c, u32* carry) {
  const u32 bit = w * c;
  u32 raw = 0;
  if (bit < 256) {
    const u32 word = bit >> 6, off = bit & 63;
    u64 v = k[word] >> off;
    if (off + c > 64 && word + 1 < 4) v |= k[word + 1] << (64 - off);
    raw = (u32)(v & ((1u << c) - 1));
  }
  raw += *carry;
  const u32 half = 1u << (c - 1);
  if (raw > half) { *carry = 1; return (int)raw - (int)(1u << c); }
  *carry = 0;
  return (int)raw;
}

__global__ void __launch_bounds__(256) msm_prepare_kernel(const u64* __restrict__ pts, u32 n, Affine* __restrict__ out,
                                                           int* __restrict__ status) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Affine p;
  p.x = bn254::fp_load(pts + (size_t)i * 8);
  p.y = bn254::fp_load(pts + (size_t)i * 8 + 4);
  bool ok = !bn254::fp_geq_p(p.x) && !bn254::fp_geq_p(p.y);
  if (ok && !(bn254::fp_is_zero(p.x) && bn254::fp_is_zero(p.y))) {
    p.x = bn254::fp_to_mont(p.x);
    p.y = bn254::fp_to_mont(p.y);
    ok = bn254::affine_on_curve(p);
  }
  if (!ok) { atomicOr(status, 1); p.x = bn254::fp_zero(); p.y = bn254::fp_zero(); }
  out[i] = p;
}

// pass 1: histogram; pass 2 (SCATTER): entries into the runs
template <bool SCATTER>
__global__ void __launch_bounds__(256) msm_digits_kernel(const u64* __restrict__ scalars, MsmShape sh, u32* __restrict__ counts,
                                                          u32* __restrict__ cursor, u32* __restrict__ entries) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= sh.n) return;
  u64 k[4];
#pragma unroll
  for (int j = 0; j < 4; j++) k[j] = scalars[(size_t)i * 4 + j];
  u32 carry = 0;
  for (u32 w = 0; w < sh.W; w++) {
    const int d = msm_digit(k, w, sh.c, &carry);
    if (d == 0) continue;
    const u32 mag = (u32)(d < 0 ? -d : d);
    const u32 key = w * sh.NB + (mag - 1);
    if (SCATTER) {
      const u32 pos = atomicAdd(&cursor[key], 1u);
      entries[pos] = i | (d < 0 ? 0x80000000u : 0u);
    } else {
      atomicAdd(&counts[key], 1u);
    }
  }
}

// exclusive prefix sum of `m` counts by ONE workgroup of 1024 (m <= ~10^6): offsets[0..m], offsets[m] = total; also
// copies the offsets into `cursor` for the scatter pass
__global__ void __launch_bounds__(1024) msm_scan_kernel(const u32* __restrict__ counts, u32 m, u32* __restrict__ offsets,
                                                         u32* __restrict__ cursor) {
  __shared__ u32 part[1024];
  const u32 tid = threadIdx.x;
  const u32 per = (m + 1023) / 1024;
  const u32 lo = tid * per, hi = lo + per < m ? lo + per : m;
  u32 s = 0;
  for (u32 i = lo; i < hi; i++) s += counts[i];
  part[tid] = s;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    const u32 v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  u32 run = tid ? part[tid - 1] : 0;
  for (u32 i = lo; i < hi; i++) { offsets[i] = run; cursor[i] = run; run += counts[i]; }
  if (tid == 1023) offsets[m] = part[1023];
}

// one lane per bucket: sum of its run
__global__ void __launch_bounds__(256) msm_accumulate_kernel(const Affine* __restrict__ pts, const u32* __restrict__ offsets,
                                                              const u32* __restrict__ entries, u32 nbuckets,
                                                              Xyzz* __restrict__ buckets) {
  const u32 b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nbuckets) return;
  const u32 lo = offsets[b], hi = offsets[b + 1];
  Xyzz acc = bn254::xyzz_inf();
  for (u32 e = lo; e < hi; e++) {
    const u32 ent = entries[e];
    const Affine p = pts[ent & 0x7FFFFFFFu];
    bn254::xyzz_madd(acc, p, (ent >> 31) != 0);
  }
  buckets[b] = acc;
}

// bit planes, first stage: out[(w*K + k)*G + g] = sum of buckets b in [16g, 16g+16) of window w whose weight (b+1) has
// bit k set; K = c bit planes (weights go up to NB = 2^(c-1), so bits 0 .. c-1), G = NB/16 groups
__global__ void __launch_bounds__(256) msm_bitplane_kernel(const Xyzz* __restrict__ buckets, MsmShape sh, Xyzz* __restrict__ out) {
  const u32 G = sh.NB / 16, K = sh.c;
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= sh.W * K * G) return;
  const u32 g = t % G, k = (t / G) % K, w = t / (G * K);
  Xyzz acc = bn254::xyzz_inf();
  for (u32 j = 0; j < 16; j++) {
    const u32 b = 16 * g + j;
    if (((b + 1) >> k) & 1) acc = bn254::xyzz_add(acc, buckets[(size_t)w * sh.NB + b]);
  }
  out[t] = acc;
}

// 8-to-1 stage over rows: in[row][cnt] -> out[row][ceil(cnt/8)]
__global__ void __launch_bounds__(256) msm_sum8_kernel(const Xyzz* __restrict__ in, u32 rows, u32 cnt, Xyzz* __restrict__ out) {
  const u32 ocnt = (cnt + 7) / 8;
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * ocnt) return;
  const u32 row = t / ocnt, o = t % ocnt;
  Xyzz acc = bn254::xyzz_inf();
  for (u32 j = 0; j < 8; j++) {
    const u32 i = 8 * o + j;
    if (i < cnt) acc = bn254::xyzz_add(acc, in[(size_t)row * cnt + i]);
  }
  out[t] = acc;
}

}  // namespace ronk
