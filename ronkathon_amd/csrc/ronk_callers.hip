// ronk_callers.hip -- C ABI of libronk_ntt.so, part 3: the callers either side of the transform:
// evaluate, division (kzg::open), Lagrange evaluate, Reed-Solomon encode / decode, KZG commit (curve MSM).
#include "runtime.h"
#include "scan_kernels.h"
#include "interp_kernels.h"
#include "curve_kernels.h"

// ------------------------------------------------------------------------------ evaluate / divrem / RS
// Workspace pool for the scan entry points: hipMalloc'd buffers, each guarded by a completion event, so a call
// never synchronises the device and never frees memory that queued work still uses.  A slot is reused when its
// last work has completed or was queued on the same stream (stream order then protects it).  (hipMallocAsync /
// hipFreeAsync were tried first and dropped: on ROCm 7.2 / gfx950 a kernel intermittently read stale data from
// a pool block reused across calls -- 4 of 12 test runs -- while plain allocations never did.)
struct WsSlot {
  void* p = nullptr;
  size_t bytes = 0;
  int device = -1;
  hipEvent_t done = nullptr;
  hipStream_t last = nullptr;
  bool used = false;   // ever had work queued
  bool busy = false;   // leased right now
};
static std::mutex g_ws_mu;
static std::vector<WsSlot*> g_ws;
struct WsLease {
  WsSlot* slot = nullptr;
  hipStream_t s = nullptr;
  ~WsLease() {
    if (!slot) return;
    (void)hipEventRecord(slot->done, s);
    std::lock_guard<std::mutex> lk(g_ws_mu);
    slot->last = s; slot->used = true; slot->busy = false;
  }
  int acquire(size_t bytes, hipStream_t st) {
    s = st;
    size_t need = 65536;
    while (need < bytes) need <<= 1;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ws_mu);
    size_t on_dev = 0;
    WsSlot* waitable = nullptr;
    for (WsSlot* w : g_ws) {
      if (w->device != dev) continue;
      on_dev++;
      if (w->busy || w->bytes < need) continue;
      if (!w->used || w->last == st || hipEventQuery(w->done) == hipSuccess) { slot = w; break; }
      if (!waitable) waitable = w;
    }
    if (!slot && waitable && on_dev >= 32) {  // bound the pool: wait for an old slot instead of growing
      (void)hipEventSynchronize(waitable->done);
      slot = waitable;
    }
    if (!slot) {
      WsSlot* w = new WsSlot();
      hipError_t e = hipMalloc(&w->p, need);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&w->done, hipEventDisableTiming);
      if (e != hipSuccess) { if (w->p) (void)hipFree(w->p); delete w; return hip_fail(e, "workspace"); }
      w->bytes = need; w->device = dev;
      g_ws.push_back(w);
      slot = w;
    }
    slot->busy = true;
    return RONK_OK;
  }
  u64* u() const { return (u64*)slot->p; }
};
static void make_horner_tab(u64 p, u64 z, u64 scale, HornerTab* t) {
  u64 x = 1 % p;
  for (int i = 0; i < 256; i++) { t->zt[i] = x; x = h_mulmod(x, z, p); }
  t->z256 = x;
  u64 y = t->zt[16];
  for (int s = 0; s < 8; s++) { t->z16p[s] = y; y = h_mulmod(y, y, p); }
  y = h_powmod(z, HCHUNK, p);
  for (int s = 0; s < 10; s++) { t->Zp[s] = y; y = h_mulmod(y, y, p); }
  t->z = z % p;
  t->scale = scale;
}
// chunk sums + carry scan shared by evaluate and the linear division: ws = [H: nchunks][carry: nchunks];
// total (may be null) receives c(z) directly from the scan kernel
static int horner_reduce_dev(const FieldCtx& f, const u64* d_c, size_t d, const HornerTab& tab, u64* ws, size_t nchunks,
                             u64* total, hipStream_t s) {
  u64* H = ws; u64* carry = ws + nchunks;
  FIELD_DISPATCH(f, {
    hipLaunchKernelGGL((chunk_horner_kernel<decltype(ops)>), dim3((u32)nchunks), dim3(256), 0, s, ops, d_c, d, tab, H);
    hipLaunchKernelGGL((chunk_carry_kernel<decltype(ops)>), dim3(1), dim3(256), 0, s, ops, H, nchunks, tab, carry, total);
  });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
static const size_t HORNER_MAX = (size_t)HCHUNK << 31;  // grid limit

extern "C" int ronk_poly_eval_dev(uint64_t p, const uint64_t* d_c, size_t d, uint64_t x, uint64_t* d_out, void* stream) {
  if (!d_out || (!d_c && d)) return RONK_ERR_INVALID;
  if (d > HORNER_MAX) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  hipStream_t s = (hipStream_t)stream;
  if (d == 0) { HIPCHK(hipMemsetAsync(d_out, 0, 8, s)); return RONK_OK; }
  const size_t nchunks = (d + HCHUNK - 1) / HCHUNK;
  HornerTab tab;
  make_horner_tab(p, x % p, 1, &tab);
  WsLease ws;
  RCHK(ws.acquire(2 * nchunks * 8, s));
  RCHK(horner_reduce_dev(f, d_c, d, tab, ws.u(), nchunks, d_out, s));
  return RONK_OK;
}
extern "C" int ronk_poly_eval(uint64_t p, const uint64_t* c, size_t d, uint64_t x, uint64_t* out) {
  if (!c || !out) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  if (d == 0) { *out = 0; return RONK_OK; }
  DevBuf dc, dres;
  RCHK(dc.alloc(d * 8)); RCHK(dres.alloc(8));
  HIPCHK(hipMemcpy(dc.p, c, d * 8, hipMemcpyHostToDevice));
  RCHK(ronk_poly_eval_dev(p, dc.u(), d, x, dres.u(), 0));
  HIPCHK(hipMemcpy(out, dres.p, 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

// poly / (b0 + b1 x), b1 != 0: the kzg::open shape (src/kzg/setup.rs:63-78).  d_quot: d coefficients (the top one
// is ZERO, as in the reference's D-long quotient); d_rem (optional): ONE element, the remainder's constant
// coefficient c(-b0/b1) -- its other d-1 coefficients are ZERO.
extern "C" int ronk_poly_div_linear_dev(uint64_t p, const uint64_t* d_c, size_t d, uint64_t b0, uint64_t b1,
                                        uint64_t* d_quot, uint64_t* d_rem, void* stream) {
  if (!d_c || !d_quot || d == 0) return RONK_ERR_INVALID;
  if (d > HORNER_MAX) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  b0 %= p; b1 %= p;
  if (b1 == 0) return RONK_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const u64 b1inv = h_powmod(b1, p - 2, p);
  const u64 z = h_mulmod((p - b0) % p, b1inv, p);        // -b0 / b1
  const size_t nchunks = (d + HCHUNK - 1) / HCHUNK;
  HornerTab tab;
  make_horner_tab(p, z, b1inv, &tab);
  WsLease ws;
  RCHK(ws.acquire(2 * nchunks * 8, s));
  RCHK(horner_reduce_dev(f, d_c, d, tab, ws.u(), nchunks, d_rem, s));
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((lindiv_apply_kernel<decltype(ops)>), dim3((u32)nchunks), dim3(256), 0, s, ops, d_c, d,
                                        tab, ws.u() + nchunks, d_quot); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}

// Polynomial::<Lagrange<F>,F,D>::evaluate (polynomial/mod.rs:382-415)
extern "C" int ronk_lagrange_eval(uint64_t p, const uint64_t* c, const uint64_t* nodes, size_t n, uint64_t x, uint64_t* out) {
  if (!c || !nodes || !out || n == 0) return RONK_ERR_INVALID;
  if (n > ((size_t)1 << 16)) return RONK_ERR_UNSUPPORTED;  // O(n^2) weights, as in the reference
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  const u32 blocks = (u32)((n + 255) / 256);
  DevBuf dc, dn, ds, dp, dres, dflag;
  RCHK(dc.alloc(n * 8)); RCHK(dn.alloc(n * 8)); RCHK(ds.alloc(blocks * 8)); RCHK(dp.alloc(blocks * 8));
  RCHK(dres.alloc(8)); RCHK(dflag.alloc(4));
  HIPCHK(hipMemcpy(dc.p, c, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dn.p, nodes, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 4));
  FIELD_DISPATCH(f, {
    hipLaunchKernelGGL((lagrange_terms_kernel<decltype(ops)>), dim3(blocks), dim3(256), 0, 0, ops, dc.u(), dn.u(), n, x % p,
                       ds.u(), dp.u(), (int*)dflag.p);
    hipLaunchKernelGGL((lagrange_finish_kernel<decltype(ops)>), dim3(1), dim3(256), 0, 0, ops, ds.u(), dp.u(), (size_t)blocks,
                       dres.u());
  });
  HIPCHK(hipGetLastError());
  int hflag = 0;
  HIPCHK(hipMemcpy(&hflag, dflag.p, 4, hipMemcpyDeviceToHost));
  if (hflag) return RONK_ERR_ZERO_INVERSE;  // coincident nodes: F::ONE.div(ZERO) -> unwrap on None
  HIPCHK(hipMemcpy(out, dres.p, 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

extern "C" int ronk_poly_divrem(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* quot,
                                uint64_t* rem) {
  if (!a || !b || !quot || !rem || d == 0 || d2 == 0) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  // kzg::open shape (src/kzg/setup.rs:63-78): a linear divisor b0 + b1*x with b1 != 0 -> Horner scan
  if (d2 == 2 && b[1] % p != 0 && d <= HORNER_MAX) {
    DevBuf dc, dqq, dr;
    RCHK(dc.alloc(d * 8)); RCHK(dqq.alloc(d * 8)); RCHK(dr.alloc(8));
    HIPCHK(hipMemcpy(dc.p, a, d * 8, hipMemcpyHostToDevice));
    RCHK(ronk_poly_div_linear_dev(p, dc.u(), d, b[0], b[1], dqq.u(), dr.u(), 0));
    HIPCHK(hipMemcpy(quot, dqq.p, d * 8, hipMemcpyDeviceToHost));
    memset(rem, 0, d * 8);
    HIPCHK(hipMemcpy(rem, dr.p, 8, hipMemcpyDeviceToHost));
    return RONK_OK;
  }
  DevBuf drem, db, dq, dst;
  RCHK(drem.alloc(d * 8)); RCHK(db.alloc(d2 * 8)); RCHK(dq.alloc(d * 8)); RCHK(dst.alloc(4));
  HIPCHK(hipMemcpy(drem.p, a, d * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db.p, b, d2 * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dst.p, 0, 4));
  const u32 T = d2 >= 1024 ? 1024 : d2 > 256 ? 512 : 256;
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((poly_divrem_kernel<decltype(ops)>), dim3(1), dim3(T), 0, 0, ops, drem.u(), d,
                                        db.u(), d2, dq.u(), (int*)dst.p); });
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpy(&status, dst.p, 4, hipMemcpyDeviceToHost));
  if (status) return status;
  HIPCHK(hipMemcpy(quot, dq.p, d * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(rem, drem.p, d * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

// Message::encode::<N> (codes/reed_solomon.rs:42-52) == nodes + size-N DFT of the zero-padded message
extern "C" int ronk_rs_encode(uint64_t p, uint64_t g, const uint64_t* msg, size_t k, size_t n, uint64_t* xs,
                              uint64_t* ys) {
  if (!msg || !xs || !ys || k == 0 || n == 0) return RONK_ERR_INVALID;
  if (n < k) return RONK_ERR_INDEX;  // assert_ge::<N, K>()
  RCHK(ronk_lagrange_nodes(p, g, xs, n));
  std::vector<u64> padded(n, 0);
  memcpy(padded.data(), msg, k * 8);
  return ronk_dft(p, g, padded.data(), ys, n);
}

// Message::decode (codes/reed_solomon.rs:54-106): the first k coordinates -> the k message coefficients
extern "C" int ronk_rs_decode(uint64_t p, const uint64_t* xs, const uint64_t* ys, size_t k, uint64_t* out) {
  if (k == 0) return RONK_OK;
  if (!xs || !ys || !out) return RONK_ERR_INVALID;
  if (k > RS_DECODE_MAX_K) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  const u32 nblk = (u32)((k + 255) / 256);
  std::vector<u64> hx(k), hy(k);
  for (size_t i = 0; i < k; i++) { hx[i] = xs[i] % p; hy[i] = ys[i] % p; }
  DevBuf dx, dy, dw, dm, dpart, dout, dflag;
  RCHK(dx.alloc(k * 8)); RCHK(dy.alloc(k * 8)); RCHK(dw.alloc(k * 8)); RCHK(dm.alloc((k + 1) * 8));
  RCHK(dpart.alloc((size_t)nblk * k * 8)); RCHK(dout.alloc(k * 8)); RCHK(dflag.alloc(4));
  HIPCHK(hipMemcpy(dx.p, hx.data(), k * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dy.p, hy.data(), k * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 4));
  FIELD_DISPATCH(f, {
    hipLaunchKernelGGL((rs_weights_kernel<decltype(ops)>), dim3(nblk), dim3(256), 0, 0, ops, dx.u(), dy.u(), k, dw.u(),
                       (int*)dflag.p);
    hipLaunchKernelGGL((master_poly_kernel<decltype(ops)>), dim3(1), dim3(1024), 0, 0, ops, dx.u(), k, dm.u());
    hipLaunchKernelGGL((rs_accumulate_kernel<decltype(ops)>), dim3(nblk), dim3(256), 0, 0, ops, dx.u(), dw.u(), dm.u(), k,
                       dpart.u());
    hipLaunchKernelGGL((rs_finish_kernel<decltype(ops)>), dim3(nblk), dim3(256), 0, 0, ops, dpart.u(), (size_t)nblk, k,
                       dout.u());
  });
  HIPCHK(hipGetLastError());
  int hflag = 0;
  HIPCHK(hipMemcpy(&hflag, dflag.p, 4, hipMemcpyDeviceToHost));
  if (hflag) return RONK_ERR_ZERO_INVERSE;  // coincident nodes: numerator / ZERO
  HIPCHK(hipMemcpy(out, dout.p, k * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

// kzg::commit (src/kzg/setup.rs:45-60): sum_i points[i] * scalars[i] on y^2 = x^3 + a x + b over F_p[u]/(u^2 - nr)
extern "C" int ronk_curve_msm(const ronk_curve* cv, const uint64_t* points, size_t n_points, const uint64_t* scalars,
                              size_t n, uint64_t out[5]) {
  if (!cv || !out || (n && (!points || !scalars))) return RONK_ERR_INVALID;
  if (cv->p < 3 || cv->p >= ((u64)1 << 32)) return RONK_ERR_UNSUPPORTED;   // products of residues must fit 64 bits
  RCHK(ronk_check_prime(cv->p));
  if (n_points < n) return RONK_ERR_INDEX;          // assert!(g1_srs.len() >= coeffs.len())
  if (n == 0) { out[0] = out[1] = out[2] = out[3] = 0; out[4] = 1; return RONK_OK; }   // empty sum -> Infinity
  RCHK(need_device());
  const u64 p = cv->p;
  CurveCtx c{p, cv->nr % p, cv->a % p, cv->b % p};
  std::vector<u64> hp(5 * n), hs(n);
  for (size_t i = 0; i < n; i++) {
    for (int w = 0; w < 4; w++) hp[5 * i + w] = points[5 * i + w] % p;
    hp[5 * i + 4] = points[5 * i + 4] ? 1 : 0;
    hs[i] = scalars[i];
  }
  const u32 nblk = (u32)((n + 255) / 256);
  DevBuf dp, ds, dpart, dout, dflag;
  RCHK(dp.alloc(5 * n * 8)); RCHK(ds.alloc(n * 8)); RCHK(dpart.alloc((size_t)nblk * 5 * 8)); RCHK(dout.alloc(5 * 8));
  RCHK(dflag.alloc(4));
  HIPCHK(hipMemcpy(dp.p, hp.data(), 5 * n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds.p, hs.data(), n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 4));
  hipLaunchKernelGGL(msm_terms_kernel, dim3(nblk), dim3(256), 0, 0, c, dp.u(), ds.u(), n, dpart.u(), (int*)dflag.p);
  hipLaunchKernelGGL(msm_reduce_kernel, dim3(1), dim3(256), 0, 0, c, dpart.u(), (size_t)nblk, dout.u(), (int*)dflag.p);
  HIPCHK(hipGetLastError());
  int hflag = 0;
  HIPCHK(hipMemcpy(&hflag, dflag.p, 4, hipMemcpyDeviceToHost));
  if (hflag & CURVE_ERR_NOT_ON_CURVE) return RONK_ERR_NOT_ON_CURVE;   // AffinePoint::new: "Point is not on curve"
  if (hflag & CURVE_ERR_INVERSE) return RONK_ERR_ZERO_INVERSE;        // Div: expect("invalid inverse")
  HIPCHK(hipMemcpy(out, dout.p, 5 * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

