// msm_kernels.h -- bucket-method (Pippenger) multi-scalar multiplication over BN254 G1: sum_i k_i * P_i.
//
// kzg::commit on a production-size curve (reference: src/kzg/setup.rs:48-60 folds g1_srs[i] * coeff[i]
// with AffinePoint's Mul/Add, src/curve/mod.rs:157-211).  Scalars are plain 256-bit integers (any value; the caller's
// field elements mod r), split in W signed digits of c bits, digit in [-2^(c-1), 2^(c-1)]:
//
//   prepare   points -> Montgomery form, on-curve check (AffinePoint::new's assert, src/curve/mod.rs:79)
//   carries / hist / scan / place   counting sort of the n*W (point, sign) entries by (window, bucket) with the counters of
//             a window in LDS (no global atomics)
//   accumulate  one lane per TASK (<= CH entries of one bucket's run): XYZZ sum by mixed additions (8M + 2S each), the
//             dominant kernel; collect / heavy fold the tasks of a bucket
//   reduce    sum_b (b+1) B_b per window as bit planes: Q_k = sum of the buckets whose weight has bit k set (plain sums, no
//             serial running-sum chain: a lane adds at most 4 points, then 4-to-1 stages), the rest
//             (sum_k 2^k Q_k, then Horner over the windows: ~270 dependent doublings) runs on the host, where one
//             doubling takes 0.5 us instead of 14 us on a single lane.
// Everything here is VALU-bound 32-bit limb arithmetic (bn254.h); no MFMA (carry chains), HBM traffic is the gather of
// 64-byte points (n*W of them) and is small beside ~3 400 instructions per mixed addition.
#pragma once
#include <hip/hip_runtime.h>

#include "bn254.h"
#include "msm_common.h"

namespace ronk {

using bn254::Affine;
using bn254::Fp;
using bn254::Xyzz;
typedef uint64_t u64;
typedef uint32_t u32;

__global__ void __launch_bounds__(256) msm_prepare_kernel(const u64* __restrict__ pts, u32 n, Affine* __restrict__ out,
                                                           int* __restrict__ status) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Affine p;
  p.x = bn254::fp_load(pts + (size_t)i * 8);
  p.y = bn254::fp_load(pts + (size_t)i * 8 + 4);
  bool ok = !bn254::fp_geq_p(p.x) && !bn254::fp_geq_p(p.y);
  if (ok && !(bn254::fp_is_zero_exact(p.x) && bn254::fp_is_zero_exact(p.y))) {
    p.x = bn254::fp_to_mont(p.x);
    p.y = bn254::fp_to_mont(p.y);
    ok = bn254::affine_on_curve(p);
  }
  if (!ok) { atomicOr(status, 1); p.x = bn254::fp_zero(); p.y = bn254::fp_zero(); }
  out[i] = p;
}

// ---- counting sort of the n*W (point, sign) entries by (window, bucket), without global atomics ----------------------------
// (First version: one global atomicAdd per entry in a count pass and a returning one in a scatter pass: 0.64 + 1.56 ms
// of a 7.7 ms MSM at 2^20 points, and 7 ms each when the scalars share a digit -- same-address atomics are served one
// after the other.)  Now a workgroup owns (window w, chunk of the scalars) and keeps that window's 2^(c-1) counters in
// LDS:
//   carries   per scalar, the W carry bits of the signed-digit recoding (bit w = carry INTO window w), so that any window's
//             digit can be computed on its own
//   hist      LDS histogram of the chunk's digits for window w -> hist[key][chunk]        (key-major: a scan over the flat
//   scan      array yields, for every key, the start of each chunk's share of its run, and offsets[key] = pos[key][0])
//   place     LDS cursors initialised from pos[key][chunk]; ds_add_rtn hands out the slots; entries[slot] = point | sign
constexpr u32 MSM_SORT_WG = 1024;

__global__ void __launch_bounds__(256) msm_carries_kernel(const u64* __restrict__ scalars, MsmShape sh, u64* __restrict__ carries) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= sh.n) return;
  u64 k[4];
#pragma unroll
  for (int j = 0; j < 4; j++) k[j] = scalars[(size_t)i * 4 + j];
  u32 carry = 0;
  u64 mask = 0;   // W <= 52 windows (c >= 5)
  for (u32 w = 0; w < sh.W; w++) {
    mask |= (u64)carry << w;
    (void)msm_digit(k, w, sh.c, &carry);
  }
  carries[i] = mask;
}

// PLACE = false: hist[(w*NB + b)*chunks + chunk] = number of entries; PLACE = true: entries placed from pos[...]
template <bool PLACE>
__global__ void __launch_bounds__(MSM_SORT_WG) msm_sort_kernel(const u64* __restrict__ scalars, const u64* __restrict__ carries,
                                                                MsmShape sh, u32 chunk_size, u32 chunks, u32* __restrict__ hist,
                                                                const u32* __restrict__ pos, u32* __restrict__ entries) {
  extern __shared__ u32 lds_cnt[];   // NB counters / cursors
  const u32 tid = threadIdx.x, chunk = blockIdx.x, w = blockIdx.y;
  const size_t kbase = (size_t)w * sh.NB;
  for (u32 b = tid; b < sh.NB; b += MSM_SORT_WG) lds_cnt[b] = PLACE ? pos[(kbase + b) * chunks + chunk] : 0;
  __syncthreads();
  const u32 i0 = chunk * chunk_size, i1 = i0 + chunk_size < sh.n ? i0 + chunk_size : sh.n;
  for (u32 i = i0 + tid; i < i1; i += MSM_SORT_WG) {
    // only the words the window touches are needed, but the four loads are one 32-byte segment anyway
    u64 k[4];
#pragma unroll
    for (int j = 0; j < 4; j++) k[j] = scalars[(size_t)i * 4 + j];
    u32 carry = (u32)(carries[i] >> w) & 1;
    const int d = msm_digit(k, w, sh.c, &carry);
    if (d == 0) continue;
    const u32 b = (u32)(d < 0 ? -d : d) - 1;
    if (PLACE) {
      const u32 slot = atomicAdd(&lds_cnt[b], 1u);
      entries[slot] = i | (d < 0 ? 0x80000000u : 0u);
    } else {
      atomicAdd(&lds_cnt[b], 1u);
    }
  }
  if (!PLACE) {
    __syncthreads();
    for (u32 b = tid; b < sh.NB; b += MSM_SORT_WG) hist[(kbase + b) * chunks + chunk] = lds_cnt[b];
  }
}

// offsets[key] = pos[key*chunks] (start of the key's run), offsets[keys] = total; counts / ntasks from neighbours
__global__ void __launch_bounds__(256) msm_offsets_kernel(const u32* __restrict__ pos, u32 keys, u32 chunks, u32 ch,
                                                           u32* __restrict__ offsets, u32* __restrict__ ntasks) {
  const u32 k = blockIdx.x * 256 + threadIdx.x;
  if (k > keys) return;
  const u32 o = pos[(size_t)k * chunks];          // k == keys: pos[keys*chunks] = total
  offsets[k] = o;
  if (k < keys) {
    const u32 cnt = pos[(size_t)(k + 1) * chunks] - o;
    ntasks[k] = (cnt + ch - 1) / ch;
  }
}

// exclusive prefix sum of `m` counts, three launches: per-block totals (256 lanes x `per` consecutive entries per
// workgroup; the host picks `per` so that there are at most 1024 blocks), one workgroup scanning the block totals, then
// every workgroup scanning its own block from its base.  offsets[m] = total.
__global__ void __launch_bounds__(256) msm_scan_totals_kernel(const u32* __restrict__ counts, u32 m, u32 per, u32* __restrict__ totals) {
  __shared__ u32 red[256];
  const u32 tid = threadIdx.x;
  const size_t base = ((size_t)blockIdx.x * 256 + tid) * per;
  u32 s = 0;
  for (u32 j = 0; j < per; j++) s += base + j < m ? counts[base + j] : 0;
  red[tid] = s;
  __syncthreads();
  for (u32 k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
  if (tid == 0) totals[blockIdx.x] = red[0];
}
// in-place exclusive scan of nb <= 1024 block totals; totals[nb] = grand total
__global__ void __launch_bounds__(1024) msm_scan_mid_kernel(u32* __restrict__ totals, u32 nb) {
  __shared__ u32 part[1024];
  const u32 tid = threadIdx.x;
  const u32 v0 = tid < nb ? totals[tid] : 0;
  part[tid] = v0;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    const u32 v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  if (tid < nb) totals[tid] = part[tid] - v0;
  if (tid == 1023) totals[nb] = part[1023];
}
__global__ void __launch_bounds__(256) msm_scan_apply_kernel(const u32* __restrict__ counts, u32 m, u32 per,
                                                              const u32* __restrict__ totals, u32 nb, u32* __restrict__ offsets) {
  __shared__ u32 part[256];
  const u32 tid = threadIdx.x;
  const size_t base = ((size_t)blockIdx.x * 256 + tid) * per;
  u32 s = 0;
  for (u32 j = 0; j < per; j++) s += base + j < m ? counts[base + j] : 0;
  part[tid] = s;
  __syncthreads();
  for (u32 off = 1; off < 256; off <<= 1) {
    const u32 v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  u32 run = totals[blockIdx.x] + part[tid] - s;
  for (u32 j = 0; j < per; j++) {
    if (base + j < m) { const u32 c = counts[base + j]; offsets[base + j] = run; run += c; }
  }
  if (blockIdx.x == 0 && tid == 0) offsets[m] = totals[nb];
}

// Buckets are summed in TASKS of at most CH entries, so that one heavy bucket (every scalar shares its top digit, or all
// scalars are equal: n entries in one run) cannot serialise the launch: ntasks[key] = ceil(count/CH) goes through the
// same scan as the counts; one lane per task adds its slice of the run; msm_collect_kernel then folds the tasks of a
// bucket (normally one: a copy), leaving buckets with more than 24 tasks to msm_heavy_kernel (below).
// task t -> (key, slice): largest key with toff[key] <= t
__global__ void __launch_bounds__(256) msm_accumulate_kernel(const Affine* __restrict__ pts, const u32* __restrict__ offsets,
                                                              const u32* __restrict__ entries, const u32* __restrict__ toff,
                                                              u32 keys, u32 ch, Xyzz* __restrict__ partial) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= toff[keys]) return;
  u32 lo = 0, hi = keys;            // invariant: toff[lo] <= t < toff[hi]
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (toff[mid] <= t) lo = mid; else hi = mid;
  }
  const u32 key = lo, j = t - toff[key];
  const u32 b0 = offsets[key] + j * ch, bend = offsets[key + 1];
  const u32 e1 = b0 + ch < bend ? b0 + ch : bend;
  Xyzz acc = bn254::xyzz_inf();
  for (u32 e = b0; e < e1; e++) {
    const u32 ent = entries[e];
    const Affine p = pts[ent & 0x7FFFFFFFu];
    bn254::xyzz_madd(acc, p, (ent >> 31) != 0);
  }
  partial[t] = acc;
}

__global__ void __launch_bounds__(256) msm_collect_kernel(const Xyzz* __restrict__ partial, const u32* __restrict__ toff, u32 keys,
                                                           Xyzz* __restrict__ buckets, u32* __restrict__ heavy, u32* __restrict__ nheavy) {
  const u32 k = blockIdx.x * 256 + threadIdx.x;
  if (k >= keys) return;
  const u32 t0 = toff[k], T = toff[k + 1] - t0;
  if (T > 24) { heavy[atomicAdd(nheavy, 1u)] = k; return; }
  Xyzz acc = bn254::xyzz_inf();
  for (u32 i = 0; i < T; i++) acc = i ? bn254::xyzz_add(acc, partial[t0 + i]) : partial[t0];
  buckets[k] = acc;
}

// Heavy buckets (more than 24 tasks; with uniform 254-bit scalars the top window of most window sizes has a handful of
// buckets holding n/4 entries each = thousands of tasks): MSM_HY workgroups per bucket each fold one slice of its task
// sums (strided per lane, then a tree in LDS) and leave the result in the slice's first entry; a second launch folds the
// MSM_HY slice sums.  (One workgroup per bucket was 64 dependent additions per lane + the tree: ~2 ms for a 16 000-task
// bucket.)
constexpr u32 MSM_HY = 64;
__device__ __forceinline__ Xyzz msm_block_sum(Xyzz acc, Xyzz* red) {
  const u32 tid = threadIdx.x;
  red[tid] = acc;
  __syncthreads();
  for (u32 s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = bn254::xyzz_add(red[tid], red[tid + s]);
    __syncthreads();
  }
  const Xyzz r = red[0];
  __syncthreads();
  return r;
}
// STAGE 0: slice y of every heavy bucket -> partial[t0 + y*len]; STAGE 1: the MSM_HY slice sums -> buckets[k]
template <int STAGE>
__global__ void __launch_bounds__(256) msm_heavy_kernel(Xyzz* __restrict__ partial, const u32* __restrict__ toff,
                                                         const u32* __restrict__ heavy, const u32* __restrict__ nheavy,
                                                         Xyzz* __restrict__ buckets) {
  __shared__ Xyzz red[256];
  const u32 tid = threadIdx.x;
  for (u32 h = blockIdx.x; h < *nheavy; h += gridDim.x) {
    const u32 k = heavy[h], t0 = toff[k], T = toff[k + 1] - t0;
    const u32 len = (T + MSM_HY - 1) / MSM_HY;
    Xyzz acc = bn254::xyzz_inf();
    const bool sliced = T > 1024;            // below that one workgroup folds the bucket directly (one tree instead of two)
    if (STAGE == 0) {
      if (!sliced) continue;
      const u32 y = blockIdx.y, lo = y * len, hi = lo + len < T ? lo + len : T;
      for (u32 i = lo + tid; i < hi; i += 256) acc = bn254::xyzz_add(acc, partial[t0 + i]);
      const Xyzz r = msm_block_sum(acc, red);      // (its barriers order every lane's reads before the write below)
      if (tid == 0 && lo < T) partial[t0 + lo] = r;
    } else {
      if (sliced) {
        if (tid < MSM_HY && tid * len < T) acc = partial[t0 + tid * len];
      } else {
        for (u32 i = tid; i < T; i += 256) acc = bn254::xyzz_add(acc, partial[t0 + i]);
      }
      const Xyzz r = msm_block_sum(acc, red);
      if (tid == 0) buckets[k] = r;
    }
  }
}

// bit planes, first stage: out[(w*K + k)*G + g] = sum of the buckets b in [FB*g, FB*g + FB) of window w whose weight (b+1)
// has bit k set; K = c bit planes (weights go up to NB = 2^(c-1), so bits 0 .. c-1), G = NB/FB groups.
// The reduction is bound by its DEPTH (dependent point additions, ~17 us each on a lane with a quiet SIMD), not by its
// work: fan-in 16 here and 8 per later stage was 44 dependent additions (0.58 + 0.33 ms at 2^20 points), fan-in 4 / 4 is 22.
template <u32 FB>
__global__ void __launch_bounds__(256) msm_bitplane_kernel(const Xyzz* __restrict__ buckets, MsmShape sh, Xyzz* __restrict__ out) {
  const u32 G = sh.NB / FB, K = sh.c;
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= sh.W * K * G) return;
  const u32 g = t % G, k = (t / G) % K, w = t / (G * K);
  Xyzz acc = bn254::xyzz_inf();
  for (u32 j = 0; j < FB; j++) {
    const u32 b = FB * g + j;
    if (((b + 1) >> k) & 1) acc = bn254::xyzz_add(acc, buckets[(size_t)w * sh.NB + b]);
  }
  out[t] = acc;
}

// FS-to-1 stage over rows: in[row][cnt] -> out[row][ceil(cnt/FS)]
template <u32 FS>
__global__ void __launch_bounds__(256) msm_sum_kernel(const Xyzz* __restrict__ in, u32 rows, u32 cnt, Xyzz* __restrict__ out) {
  const u32 ocnt = (cnt + FS - 1) / FS;
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * ocnt) return;
  const u32 row = t / ocnt, o = t % ocnt;
  Xyzz acc = bn254::xyzz_inf();
  for (u32 j = 0; j < FS; j++) {
    const u32 i = FS * o + j;
    if (i < cnt) acc = bn254::xyzz_add(acc, in[(size_t)row * cnt + i]);
  }
  out[t] = acc;
}

}  // namespace ronk
