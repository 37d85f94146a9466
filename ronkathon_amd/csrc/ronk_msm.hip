// ronk_msm.hip -- C ABI of libronk_ntt.so, part 5: kzg::commit on a production-size curve -- the bucket-method MSM over
// BN254 G1 (reference fold: src/kzg/setup.rs:48-60).  Kernels: msm_kernels.h; arithmetic: bn254.h.
#include "runtime.h"
#include "msm_kernels.h"
#include "fr_scan_kernels.h"

namespace {

#ifndef RONK_MSM_FB
#define RONK_MSM_FB 4
#endif
#ifndef RONK_MSM_FS
#define RONK_MSM_FS 4
#endif
constexpr u32 MSM_FB = RONK_MSM_FB;   // buckets per lane in the bit-plane stage (NB >= 16 is a multiple)
constexpr u32 MSM_FS = RONK_MSM_FS;   // fan-in of the later stages

// window size: buckets cost NB*W*c point additions in the reduction, entries cost n*W mixed additions
u32 pick_window(size_t n) {
  const char* e = getenv("RONK_MSM_C");    // read per call: tests and tuning runs sweep it inside one process
  const int forced = e ? atoi(e) : 0;
  if (forced >= 5 && forced <= 16) return (u32)forced;
  int lg = 0;
  while (((size_t)1 << lg) < n) lg++;
  // measured (tools/experiments/msm_window_sweep.sh, ms per MSM): 2^14: c=10 1.04 (c=9 1.07, c=8 1.22); 2^16: c=10 1.30
  // (c=11 1.58, c=9 1.57); 2^18: c=13 1.85 (c=12 2.32, c=14 2.36); 2^20: c=13 4.29, c=15 4.41 (c=14 4.77); 2^22: c=15 12.4
  int c = lg <= 17 ? 10 : lg - 5;
  if (lg < 10) c = lg < 6 ? 6 : lg;          // tiny inputs: about one entry per bucket
  if (c > 15) c = 15;
  return (u32)c;
}

// grow-only device buffer (the MSM workspace is cached per device: thirteen hipMalloc / hipFree pairs cost more than a
// 2^16-point MSM)
struct GrowBuf {
  void* p = nullptr;
  size_t cap = 0;
  int alloc(size_t bytes) {
    if (bytes <= cap) return RONK_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (MSM workspace)");
    cap = bytes;
    return RONK_OK;
  }
};
struct MsmWork {
  MsmShape sh;
  GrowBuf pts, carries, hist, pos, offsets, entries, buckets, st0, st1, status, ntasks, toff, partial, heavy, totals;
  u32 ch = 0, max_tasks = 0, chunk_size = 0, chunks = 0;
  size_t planes = 0;   // W * c rows of the bit-plane reduction
  std::mutex mu;       // one MSM at a time per device workspace
  int alloc(size_t n) {
    sh.n = (u32)n;
    sh.c = pick_window(n);
    sh.W = (257 + sh.c - 1) / sh.c;
    sh.NB = 1u << (sh.c - 1);
    const size_t keys = (size_t)sh.W * sh.NB;
    planes = (size_t)sh.W * sh.c;
    RCHK(pts.alloc(n * sizeof(Affine)));
    // chunks of the scalars for the LDS counting sort: 32768 scalars per workgroup, at most 64 chunks
    chunk_size = 32768;
    while ((n + chunk_size - 1) / chunk_size > 64) chunk_size *= 2;
    chunks = (u32)((n + chunk_size - 1) / chunk_size);
    RCHK(carries.alloc(n * 8));
    RCHK(hist.alloc((keys * chunks + 1) * 4));
    RCHK(pos.alloc((keys * chunks + 1) * 4));
    RCHK(offsets.alloc((keys + 1) * 4));
    RCHK(entries.alloc(n * sh.W * 4));
    RCHK(buckets.alloc(keys * sizeof(Xyzz)));
    RCHK(st0.alloc(planes * (sh.NB / MSM_FB) * sizeof(Xyzz)));
    RCHK(st1.alloc(planes * ((sh.NB / MSM_FB + MSM_FS - 1) / MSM_FS) * sizeof(Xyzz)));
    RCHK(status.alloc(8));
    // tasks of at most ch entries: a QUARTER of the mean run, at least 16.  With one task per bucket (ch = 2x the mean) a
    // 2^20-point MSM had 4.5 waves per SIMD for 4 resident ones -- one round plus a half-empty one, each wave as slow as its
    // fullest bucket: 3.65 ms; with ~5 tasks per bucket 23 waves per SIMD keep every slot busy: 1.9 ms, +0.2 ms to fold
    // the tasks (sweep ch = 8 .. 128: 4.58 / 4.51 (16) / 4.76 (32) / 5.36 (64) / 5.79 ms per MSM)
    size_t mean = n / sh.NB;
    ch = (u32)(mean / 4 < 16 ? 16 : mean / 4);
    if (const char* e = getenv("RONK_MSM_CH")) { const int v = atoi(e); if (v >= 1 && v <= 65536) ch = (u32)v; }   // tuning runs
    max_tasks = (u32)(keys + (n * sh.W) / ch + 1);
    RCHK(ntasks.alloc(keys * 4));
    RCHK(toff.alloc((keys + 1) * 4));
    RCHK(partial.alloc((size_t)max_tasks * sizeof(Xyzz)));
    RCHK(heavy.alloc(keys * 4));
    RCHK(totals.alloc((1024 + 1) * 4));
    return RONK_OK;
  }
};

// offsets[0..m] = exclusive prefix sums of counts[0..m)
void msm_scan(MsmWork& wk, const u32* counts, size_t m, u32* offsets, hipStream_t s) {
  u32 per = 4;
  while ((m + (size_t)256 * per - 1) / ((size_t)256 * per) > 1024) per *= 2;
  const u32 nb = (u32)((m + (size_t)256 * per - 1) / ((size_t)256 * per));
  u32* totals = (u32*)wk.totals.p;
  hipLaunchKernelGGL(msm_scan_totals_kernel, dim3(nb), dim3(256), 0, s, counts, (u32)m, per, totals);
  hipLaunchKernelGGL(msm_scan_mid_kernel, dim3(1), dim3(1024), 0, s, totals, nb);
  hipLaunchKernelGGL(msm_scan_apply_kernel, dim3(nb), dim3(256), 0, s, counts, (u32)m, per, (const u32*)totals, nb, offsets);
}

int msm_run(const u64* d_points, const u64* d_scalars, size_t n, u64 out[8], hipStream_t s) {
  if (n == 0) { for (int i = 0; i < 8; i++) out[i] = 0; return RONK_OK; }
  if (n >= ((size_t)1 << 31)) return RONK_ERR_UNSUPPORTED;
  static MsmWork g_work[64];
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return RONK_ERR_INVALID;
  MsmWork& wk = g_work[dev];
  std::lock_guard<std::mutex> lk(wk.mu);
  {   // entry slots and task ids are 32-bit: n * W must fit (n < 2^27 for every window size the planner picks)
    const u32 c = pick_window(n), W = (257 + c - 1) / c;
    if ((u64)n * W >= ((u64)1 << 32)) return RONK_ERR_UNSUPPORTED;
  }
  RCHK(wk.alloc(n));
  const MsmShape sh = wk.sh;
  const u32 keys = sh.W * sh.NB;
  const u32 gn = (u32)((n + 255) / 256);
  HIPCHK(hipMemsetAsync(wk.status.p, 0, 8, s));
  hipLaunchKernelGGL(msm_prepare_kernel, dim3(gn), dim3(256), 0, s, d_points, sh.n, (Affine*)wk.pts.p, (int*)wk.status.p);
  // counting sort of the entries by (window, bucket)
  {
    static bool attr_done[64] = {};
    const size_t lds = (size_t)sh.NB * 4;
    if (lds > 48 * 1024 && !attr_done[dev]) {
      HIPCHK(hipFuncSetAttribute((const void*)msm_sort_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)msm_sort_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done[dev] = true;
    }
    const size_t m = (size_t)keys * wk.chunks;
    hipLaunchKernelGGL(msm_carries_kernel, dim3(gn), dim3(256), 0, s, d_scalars, sh, (u64*)wk.carries.p);
    hipLaunchKernelGGL((msm_sort_kernel<false>), dim3(wk.chunks, sh.W), dim3(MSM_SORT_WG), lds, s, d_scalars,
                       (const u64*)wk.carries.p, sh, wk.chunk_size, wk.chunks, (u32*)wk.hist.p, (const u32*)nullptr, (u32*)nullptr);
    msm_scan(wk, (const u32*)wk.hist.p, m, (u32*)wk.pos.p, s);
    hipLaunchKernelGGL((msm_sort_kernel<true>), dim3(wk.chunks, sh.W), dim3(MSM_SORT_WG), lds, s, d_scalars,
                       (const u64*)wk.carries.p, sh, wk.chunk_size, wk.chunks, (u32*)nullptr, (const u32*)wk.pos.p, (u32*)wk.entries.p);
    hipLaunchKernelGGL(msm_offsets_kernel, dim3((keys + 256) / 256), dim3(256), 0, s, (const u32*)wk.pos.p, keys, wk.chunks, wk.ch,
                       (u32*)wk.offsets.p, (u32*)wk.ntasks.p);
  }
  msm_scan(wk, (const u32*)wk.ntasks.p, keys, (u32*)wk.toff.p, s);
  hipLaunchKernelGGL(msm_accumulate_kernel, dim3((wk.max_tasks + 255) / 256), dim3(256), 0, s, (const Affine*)wk.pts.p,
                     (const u32*)wk.offsets.p, (const u32*)wk.entries.p, (const u32*)wk.toff.p, keys, wk.ch,
                     (Xyzz*)wk.partial.p);
  hipLaunchKernelGGL(msm_collect_kernel, dim3((keys + 255) / 256), dim3(256), 0, s, (const Xyzz*)wk.partial.p,
                     (const u32*)wk.toff.p, keys, (Xyzz*)wk.buckets.p, (u32*)wk.heavy.p, (u32*)wk.status.p + 1);
  hipLaunchKernelGGL((msm_heavy_kernel<0>), dim3(64, MSM_HY), dim3(256), 0, s, (Xyzz*)wk.partial.p, (const u32*)wk.toff.p,
                     (const u32*)wk.heavy.p, (const u32*)wk.status.p + 1, (Xyzz*)wk.buckets.p);
  hipLaunchKernelGGL((msm_heavy_kernel<1>), dim3(64), dim3(256), 0, s, (Xyzz*)wk.partial.p, (const u32*)wk.toff.p,
                     (const u32*)wk.heavy.p, (const u32*)wk.status.p + 1, (Xyzz*)wk.buckets.p);
  const u32 rows = (u32)wk.planes;
  u32 cnt = sh.NB / MSM_FB;
  hipLaunchKernelGGL((msm_bitplane_kernel<MSM_FB>), dim3((rows * cnt + 255) / 256), dim3(256), 0, s, (const Xyzz*)wk.buckets.p, sh,
                     (Xyzz*)wk.st0.p);
  Xyzz* cur = (Xyzz*)wk.st0.p;
  Xyzz* nxt = (Xyzz*)wk.st1.p;
  while (cnt > 1) {
    const u32 ocnt = (cnt + MSM_FS - 1) / MSM_FS;
    hipLaunchKernelGGL((msm_sum_kernel<MSM_FS>), dim3((rows * ocnt + 255) / 256), dim3(256), 0, s, (const Xyzz*)cur, rows, cnt, nxt);
    Xyzz* t = cur; cur = nxt; nxt = t;
    cnt = ocnt;
  }
  HIPCHK(hipGetLastError());
  std::vector<Xyzz> h(rows);
  int st[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(h.data(), cur, (size_t)rows * sizeof(Xyzz), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(st, wk.status.p, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (st[0] & 1) return RONK_ERR_NOT_ON_CURVE;
  msm_host_tail(sh, h.data(), out);
  return RONK_OK;
}

}  // namespace

// sum_i scalars[i] * points[i] over BN254 G1.  points: n x 8 words (x, y as 4 x 64-bit little-endian limbs, standard
// form, < p; (0, 0) = the point at infinity); scalars: n x 4 words, any 256-bit integers; out: 8 words, same encoding.
extern "C" int ronk_msm_bn254_dev(const uint64_t* d_points, const uint64_t* d_scalars, size_t n, uint64_t* out, void* stream) {
  if (!out || (n && (!d_points || !d_scalars))) return RONK_ERR_INVALID;
  RCHK(need_device());
  return msm_run(d_points, d_scalars, n, out, (hipStream_t)stream);
}
extern "C" int ronk_msm_bn254(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t* out) {
  if (!out || (n && (!points || !scalars))) return RONK_ERR_INVALID;
  RCHK(need_device());
  if (n == 0) { for (int i = 0; i < 8; i++) out[i] = 0; return RONK_OK; }
  DevBuf dp, ds;
  RCHK(dp.alloc(n * 64)); RCHK(ds.alloc(n * 32));
  HIPCHK(hipMemcpy(dp.p, points, n * 64, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
  return msm_run(dp.u(), ds.u(), n, out, 0);
}

// ------------------------------------------------------------------------------ kzg::open over BN254 (src/kzg/setup.rs:63-78)
namespace {
// per-device workspace of the scalar-field division: the two multiplier tables, the chunk sums and carries
struct FrWork {
  GrowBuf tabs, H, G, rem;
  std::mutex mu;
};
FrWork g_fr_work[64];

int fr_div_linear_run(const uint64_t* d_coeffs, size_t n, const uint64_t* z, uint64_t* d_quot, uint64_t* d_rem, hipStream_t s) {
  using namespace bn254;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return RONK_ERR_UNSUPPORTED;
  FrWork& w = g_fr_work[dev];
  std::lock_guard<std::mutex> lk(w.mu);
  const size_t nchunks = (n + FR_CHUNK - 1) / FR_CHUNK;
  RCHK(w.tabs.alloc(2 * sizeof(FrScanTab)));
  RCHK(w.H.alloc(nchunks * 32 + 32));
  RCHK(w.G.alloc(nchunks * 32 + 32));
  RCHK(w.rem.alloc(32));
  FrScanTab tabs[2];
  const Fr zc = fr_canon(fr_load(z));                       // eval_point mod r
  fr_build_tab(zc, &tabs[0]);
  fr_build_tab(fr_from_mont_host(tabs[0].muchunk), &tabs[1]);   // ratio Y = z^1024 for the chunk sums
  // (the previous call on this device may still be reading the tables: stream-ordered copy from a pageable buffer is
  //  synchronous with respect to the host, and kernels of earlier calls on OTHER streams are waited for)
  HIPCHK(hipStreamSynchronize(s));
  // From the first enqueue on, EVERY exit path drains `s` before the per-device lock is dropped: the workspace (tables, chunk
  // sums) and the stack buffer `tabs` the copy reads are shared / short-lived, so a failed launch in the middle must not let
  // the next opener overwrite tables that queued work still reads.
  struct DrainOnExit { hipStream_t st; ~DrainOnExit() { (void)hipStreamSynchronize(st); } } drain{s};
  HIPCHK(hipMemcpyAsync(w.tabs.p, tabs, sizeof tabs, hipMemcpyHostToDevice, s));
  const FrScanTab* dt = (const FrScanTab*)w.tabs.p;
  uint64_t* rem = d_rem ? d_rem : (uint64_t*)w.rem.p;
  hipLaunchKernelGGL(fr_chunk_sum_kernel, dim3((u32)nchunks), dim3(256), 0, s, d_coeffs, n, dt, (uint64_t*)w.H.p);
  hipLaunchKernelGGL(fr_carry_kernel, dim3(1), dim3(256), 0, s, (const uint64_t*)w.H.p, nchunks, dt + 1, (uint64_t*)w.G.p, rem);
  hipLaunchKernelGGL(fr_apply_kernel, dim3((u32)nchunks), dim3(256), 0, s, d_coeffs, n, dt, (const uint64_t*)w.G.p, d_quot);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));                          // the workspace is per device, the tables per call (errors surface here)
  return RONK_OK;
}
}  // namespace

// poly / (x - z) over BN254's scalar field: d_coeffs, d_quot: n x 4 words (standard form; inputs are taken mod r); d_rem: 4 words
// on the device (may be NULL).  The quotient's top entry is ZERO (n entries like the reference's D-long quotient).
extern "C" int ronk_poly_div_linear_bn254_dev(const uint64_t* d_coeffs, size_t n, const uint64_t* z, uint64_t* d_quot,
                                              uint64_t* d_rem, void* stream) {
  if (!d_coeffs || !z || !d_quot || n == 0) return RONK_ERR_INVALID;
  if (n > ((size_t)1 << 30)) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  return fr_div_linear_run(d_coeffs, n, z, d_quot, d_rem, (hipStream_t)stream);
}

// kzg::open: the quotient committed against the SRS, and poly(z)
extern "C" int ronk_kzg_open_bn254_dev(const uint64_t* d_coeffs, size_t n, const uint64_t* z, const uint64_t* d_srs,
                                       uint64_t* d_quot, uint64_t* out_point, uint64_t* out_value, void* stream) {
  if (!d_coeffs || !z || !d_srs || !d_quot || !out_point || n == 0) return RONK_ERR_INVALID;
  if (n > ((size_t)1 << 30)) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  DevBuf drem;
  RCHK(drem.alloc(32));
  RCHK(fr_div_linear_run(d_coeffs, n, z, d_quot, drem.u(), (hipStream_t)stream));
  if (out_value) HIPCHK(hipMemcpy(out_value, drem.p, 32, hipMemcpyDeviceToHost));
  return msm_run(d_srs, d_quot, n, out_point, (hipStream_t)stream);
}
extern "C" int ronk_kzg_open_bn254(const uint64_t* coeffs, size_t n, const uint64_t* z, const uint64_t* srs, size_t n_srs,
                                   uint64_t* out_point, uint64_t* out_value) {
  if (!coeffs || !z || !srs || !out_point || n == 0) return RONK_ERR_INVALID;
  if (n_srs < n) return RONK_ERR_INDEX;                      // assert!(g1_srs.len() >= coeffs.len()), setup.rs:53
  RCHK(need_device());
  DevBuf dc, dsrs, dq;
  RCHK(dc.alloc(n * 32)); RCHK(dsrs.alloc(n * 64)); RCHK(dq.alloc(n * 32));
  HIPCHK(hipMemcpy(dc.p, coeffs, n * 32, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dsrs.p, srs, n * 64, hipMemcpyHostToDevice));
  return ronk_kzg_open_bn254_dev(dc.u(), n, z, dsrs.u(), dq.u(), out_point, out_value, 0);
}
