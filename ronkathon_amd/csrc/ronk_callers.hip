// ronk_callers.hip -- C ABI of libronk_ntt.so, part 3: the callers either side of the transform:
// evaluate, division (kzg::open), Lagrange evaluate, Reed-Solomon encode / decode, KZG commit (curve MSM).
#include "runtime.h"
#include "scan_kernels.h"
#include "lindiv_kernels.h"
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
  u32* ctl = nullptr;   // 64 bytes of control words, ZERO whenever the slot is not in use (fused scan kernels)
  u64* lb = nullptr;    // two look-back arrays of LB_WORDS entries (scan_kernels.h, one-launch forms): the one a call uses
  u32 lb_calls = 0;     // was set to LB_EMPTY by the call before it (at creation: both), parity = lb_calls & 1
  size_t bytes = 0;
  int device = -1;
  hipEvent_t done = nullptr;
  hipStream_t last = nullptr;
  bool used = false;   // ever had work queued
  bool busy = false;   // leased right now
  bool ev_valid = false;   // `done` was recorded behind the slot's last work
  bool oneoff = false;     // larger than WS_CACHE_MAX: never pooled, freed once its work has completed
};
static std::mutex g_ws_mu;
static std::vector<WsSlot*> g_ws;
// Retention bound: the pool keeps what the scans, the decode and ordinary divisions need (64 KiB .. 256 MiB per slot, a power
// of two each).  Anything larger -- the ~10 x Lp x 8 bytes of a 2^27 Newton division, 16 GiB after rounding -- is a ONE-OFF
// allocation of exactly the requested size: on release it goes to g_ws_grave with an event behind its last work and is freed
// by the next acquire / ronk_trim_workspace() that finds the event complete (never while queued work still uses it).
static constexpr size_t WS_CACHE_MAX = (size_t)256 << 20;
static std::vector<WsSlot*> g_ws_grave;
static void ws_free_slot(WsSlot* w) {
  if (w->p) (void)hipFree(w->p);
  if (w->ctl) (void)hipFree(w->ctl);
  if (w->lb) (void)hipFree(w->lb);
  if (w->done) (void)hipEventDestroy(w->done);
  delete w;
}
// g_ws_mu held.  wait = false: free what has completed; true: wait for the rest (ronk_trim_workspace).
static void ws_reap(bool wait) {
  for (size_t i = 0; i < g_ws_grave.size();) {
    WsSlot* w = g_ws_grave[i];
    bool done = !w->ev_valid || hipEventQuery(w->done) == hipSuccess;
    if (!done && wait) done = hipEventSynchronize(w->done) == hipSuccess;
    if (!done) { (void)hipGetLastError(); i++; continue; }
    ws_free_slot(w);
    g_ws_grave[i] = g_ws_grave.back();
    g_ws_grave.pop_back();
  }
}
// Completion events are recorded only once the pool has been seen from a second stream: while every call comes from ONE
// stream, stream order protects a reused slot and the hipEventRecord per call (~1.5 us of the 12.5 us an evaluate of 2^22
// coefficients takes end to end) buys nothing.  The stream that triggers the switch simply gets a fresh slot.
static bool g_ws_multi = false;
struct WsLease {
  WsSlot* slot = nullptr;
  hipStream_t s = nullptr;
  ~WsLease() {
    if (!slot) return;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (slot->oneoff) {   // never pooled: an event behind its work, freed by whoever finds it complete
      // A capturing stream: an event recorded inside a capture never completes for hipEventQuery / hipEventSynchronize, so the
      // buffer would sit in the grave for ever (through ronk_trim_workspace as well).  A captured graph may be replayed long
      // after this call returned, so its one-off workspace cannot be freed at all while the graph lives: acquire() refuses
      // one-off workspaces under capture (RONK_ERR_UNSUPPORTED); should one get here regardless, it is kept, not leaked
      // silently: the grave entry has no event and is freed by the next reap.
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
      if (cap != hipStreamCaptureStatusNone) {
        slot->ev_valid = false;
      } else {
        slot->ev_valid = hipEventRecord(slot->done, s) == hipSuccess;
        if (!slot->ev_valid) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); }
      }
      slot->busy = false;
      g_ws_grave.push_back(slot);
      ws_reap(false);
      return;
    }
    // (a capturing stream leaves no event: one recorded inside a capture never completes for the pool's queries; the slot stays
    //  tied to the stream it was last used on -- the captured graph's stream -- and other streams wait for the device instead)
    hipStreamCaptureStatus capr = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capr) != hipSuccess) { (void)hipGetLastError(); capr = hipStreamCaptureStatusNone; }
    slot->ev_valid = capr == hipStreamCaptureStatusNone && g_ws_multi && hipEventRecord(slot->done, s) == hipSuccess;
    slot->last = s; slot->used = true; slot->busy = false;
  }
  int acquire(size_t bytes, hipStream_t st) {
    s = st;
    size_t need = 65536;
    while (need < bytes) need <<= 1;
    const bool oneoff = need > WS_CACHE_MAX;
    if (oneoff) need = (bytes + 255) & ~(size_t)255;   // exactly what was asked for, not the next power of two
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (oneoff) {   // (see ~WsLease: a captured graph would use the buffer after it has been freed)
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
      if (cap != hipStreamCaptureStatusNone) return RONK_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(g_ws_mu);
    ws_reap(false);
    size_t on_dev = 0;
    WsSlot* waitable = nullptr;
    for (WsSlot* w : g_ws) {
      if (oneoff) break;
      if (w->device != dev) continue;
      on_dev++;
      if (w->busy || w->bytes < need) continue;
      if (!w->used || w->last == st || (w->ev_valid && hipEventQuery(w->done) == hipSuccess)) { slot = w; break; }
      g_ws_multi = true;                        // a slot last used by ANOTHER stream: from now on every release leaves an event
      if (!waitable) waitable = w;
    }
    if (!slot && waitable && on_dev >= 32) {  // bound the pool: wait for an old slot instead of growing
      // A slot released before the pool went multi-stream has no event, and its stream handle cannot be trusted any more (a
      // destroyed hipStream_t crashes hipEventRecord / hipStreamSynchronize -- tests/test_gpu_parity.py destroys one on
      // purpose), so the one thing left to wait on is the device: the slot's device IS the current one (filter above).
      // At most once per process and device: from the switch on every release records its event.
      if (waitable->ev_valid) (void)hipEventSynchronize(waitable->done);
      else (void)hipDeviceSynchronize();
      slot = waitable;
    }
    if (!slot) {
      // A capturing stream may not be handed a FRESH slot: creating one needs hipMalloc and the look-back fills, which either
      // invalidate a global-mode capture (legacy null-stream calls) or would become nodes of the caller's graph.  The caller
      // warms the pool with one call of the same shape outside the capture (include/ronk_ntt.h); until then the captured call
      // is refused, never half-initialised.
      hipStreamCaptureStatus capn = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &capn) != hipSuccess) { (void)hipGetLastError(); capn = hipStreamCaptureStatusNone; }
      if (capn != hipStreamCaptureStatusNone) {
        (void)hip_fail(hipErrorStreamCaptureUnsupported, "workspace: no warm pool slot for a capturing stream (call once outside the capture first)");
        return RONK_ERR_UNSUPPORTED;
      }
      WsSlot* w = new WsSlot();
      w->oneoff = oneoff;
      hipError_t e = hipMalloc(&w->p, need);
      if (e == hipErrorOutOfMemory && !g_ws_grave.empty()) {
        // two back-to-back large divisions on one stream: the first one's one-off buffer (gigabytes) is still in the grave
        // behind its event -- wait for the grave instead of reporting out-of-memory, then try once more
        (void)hipGetLastError();
        ws_reap(true);
        e = hipMalloc(&w->p, need);
      }
      // The initial fills are ordered on the ACQUIRING stream: hipMemset runs on the null stream and returns before it has
      // executed, so a kernel launched right afterwards on a NON-BLOCKING stream could read the look-back arrays before they
      // were filled -- and take garbage for a published chunk sum (found by the four-thread fuzz sweep, round 5: one evaluate
      // in ~5 000 threaded cases, always on a caller-created stream and a fresh slot).  A later user on another stream gets
      // the slot only behind its completion event (or the device drain of the first cross-stream use), i.e. behind the fills.
      // (A capturing stream never gets here: refused above.)
      const bool on_st = true;
      if (e == hipSuccess) e = hipMalloc((void**)&w->ctl, 64);
      if (e == hipSuccess) e = on_st ? hipMemsetAsync(w->ctl, 0, 64, st) : hipMemset(w->ctl, 0, 64);
      if (e == hipSuccess) e = hipMalloc((void**)&w->lb, (size_t)2 * LB_WORDS * 8);
      if (e == hipSuccess) e = on_st ? hipMemsetAsync(w->lb, 0xFF, (size_t)2 * LB_WORDS * 8, st)
                                     : hipMemset(w->lb, 0xFF, (size_t)2 * LB_WORDS * 8);   // LB_EMPTY everywhere
      if (e == hipSuccess) e = hipEventCreateWithFlags(&w->done, hipEventDisableTiming);
      if (e != hipSuccess) { if (w->p) (void)hipFree(w->p); if (w->ctl) (void)hipFree(w->ctl); if (w->lb) (void)hipFree(w->lb); delete w; return hip_fail(e, "workspace"); }
      w->bytes = need; w->device = dev;
      if (!oneoff) g_ws.push_back(w);
      slot = w;
    }
    slot->busy = true;
    return RONK_OK;
  }
  u64* u() const { return (u64*)slot->p; }
  u32* ctl() const { return slot->ctl; }
  // look-back arrays of a one-launch scan: *cur is all LB_EMPTY now, *next is cleared by the kernel for the call after
  // The parity advances only through lb_commit(), AFTER the launch was accepted: a failed launch never ran the kernel that
  // clears *next, so the same (still clean) *cur must serve the following call.
  void lb_arrays(u64** cur, u64** next) {
    const u32 par = slot->lb_calls & 1;
    *cur = slot->lb + (size_t)par * LB_WORDS;
    *next = slot->lb + (size_t)(par ^ 1) * LB_WORDS;
  }
  void lb_commit() { slot->lb_calls++; }
};
// Releases every idle pool slot and every finished one-off buffer (include/ronk_ntt.h).  Waits for the work behind them
// (events; a slot released in single-stream mode has none: its device is synchronised).
extern "C" int ronk_trim_workspace(void) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  ws_reap(true);
  int cur = -1;
  (void)hipGetDevice(&cur);
  for (size_t i = 0; i < g_ws.size();) {
    WsSlot* w = g_ws[i];
    if (w->busy) { i++; continue; }
    (void)hipSetDevice(w->device);
    if (w->used) {   // (no event = released in single-stream mode; its stream handle may be gone: wait for the device)
      hipError_t e = w->ev_valid ? hipEventSynchronize(w->done) : hipDeviceSynchronize();
      if (e != hipSuccess) (void)hipGetLastError();
    }
    ws_free_slot(w);
    g_ws[i] = g_ws.back();
    g_ws.pop_back();
  }
  if (cur >= 0) (void)hipSetDevice(cur);
  return RONK_OK;
}
// the one-launch scans keep host state per call (which look-back array is clean): not for a capturing stream
static bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}
static void build_horner_tab2(u64 p, u64 z, u64 scale, HornerTab2* t) {
  u64 x = 1 % p;
  for (int i = 0; i < 256; i++) { t->zt[i] = x; x = h_mulmod(x, z, p); }
  t->z256 = x;
  u64 y = t->zt[8];
  for (int s = 0; s < 8; s++) { t->z8p[s] = y; y = h_mulmod(y, y, p); }
  y = h_powmod(z, FCH, p);
  for (int s = 0; s < 20; s++) { t->Yp[s] = y; y = h_mulmod(y, y, p); }
  t->z = z % p;
  t->scale = scale;
  static const u64 test_flags = [] { const char* e = getenv("RONK_LB_TEST_FLAGS"); return e ? (u64)atoi(e) : (u64)0; }();
  t->test_flags = test_flags;
  const u64 Y = t->Yp[0], Y16 = t->Yp[4], Y256 = t->Yp[8], z8 = t->zt[8], z128 = t->zt[128];
  u64 a = 1 % p, b = 1 % p, cc = 1 % p, c8 = 1 % p, d8 = 1 % p;
  for (int i = 0; i < 16; i++) {
    t->YA[i] = a; t->YB[i] = b; t->YC[i] = cc; t->z8A[i] = c8; t->z8B[i] = d8;
    a = h_mulmod(a, Y, p); b = h_mulmod(b, Y16, p); cc = h_mulmod(cc, Y256, p); c8 = h_mulmod(c8, z8, p); d8 = h_mulmod(d8, z128, p);
  }
}
// ~330 modular products on the host (~6 us): more than the launch itself, and callers open or evaluate many polynomials at
// ONE point (kzg batch openings), so the last table is kept
static void make_horner_tab2(u64 p, u64 z, u64 scale, HornerTab2* t) {
  static std::mutex mu;
  static HornerTab2 last;
  static u64 lp = 0, lz = 0, lscale = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (lp != p || lz != z || lscale != scale) { build_horner_tab2(p, z, scale, &last); lp = p; lz = z; lscale = scale; }
  *t = last;
}
// fused paths (scan_kernels.h): evaluate in one launch up to 2^20 chunks; division in two launches up to 4096 chunks
static const size_t FUSED_EVAL_MAX = (size_t)FCH << 20, FUSED_DIV_MAX = (size_t)FCH * 4096;
static const bool g_no_fused_scans = getenv("RONK_NO_FUSED_SCANS") != nullptr;   // experiments / A-B
static const bool g_no_onepass_scans = getenv("RONK_NO_ONEPASS_SCANS") != nullptr;
static const bool g_onepass_div = getenv("RONK_ONEPASS_DIV") != nullptr;   // one-launch division: built, parity-tested, slower (scan_kernels.h)
// ---- division by a linear divisor, lindiv_kernels.h: the default up to 4096 chunks ---------------------------------------
// RONK_LINDIV = "0": scan_kernels.h forms only; "l": the chunk goes through the LDS image both ways even when the operands
// are 16-byte aligned (A-B runs; profiles/r03_lindiv_forms.txt)
static const int g_lindiv = [] {
  const char* e = getenv("RONK_LINDIV");
  if (!e || !*e) return 2;
  return e[0] == '0' ? 0 : e[strlen(e) - 1] == 'l' ? 1 : 2;
}();
static void make_lindiv_tab(u64 p, u64 z, u64 scale, LinDivTab* t) {   // (kept for the next call: see make_horner_tab2)
  static std::mutex mu;
  static LinDivTab last;
  static u64 lp = 0, lz = 0, lscale = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (lp != p || lz != z || lscale != scale) { lindiv_build_tab(p, z, scale, &last); lp = p; lz = z; lscale = scale; }
  *t = last;
}
// the one-launch form (lindiv_kernels.h lindiv_one_kernel): RONK_LINDIV_ONE = 0 never, else (default) whenever it applies
static const int g_lindiv_one = [] { const char* e = getenv("RONK_LINDIV_ONE"); return e ? atoi(e) : 1; }();
// RONK_LINDIV_ONE_MAXCH: most chunks of 8192 coefficients it is used for (default: one look-back entry per lane, 1024; 512 = only
// while every chunk is resident at once)
static const u32 g_lindiv_one_maxch = [] {
  const char* e = getenv("RONK_LINDIV_ONE_MAXCH");
  const long v = e ? atol(e) : (long)LINDIV1_MAX_CHUNKS;
  return (u32)(v < 1 ? 1 : v > (long)LINDIV1_MAX_CHUNKS ? (long)LINDIV1_MAX_CHUNKS : v);
}();
static void make_lindiv1_tab(u64 p, u64 z, u64 scale, int pl, LinDiv1Tab* t) {   // (kept for the next call: see make_horner_tab2)
  static std::mutex mu;
  static LinDiv1Tab last;
  static u64 lp = 0, lz = 0, lscale = 0;
  static int lpl = 0;
  static const u64 test_flags = [] { const char* e = getenv("RONK_LB_TEST_FLAGS"); return e ? (u64)atoi(e) : (u64)0; }();
  std::lock_guard<std::mutex> lk(mu);
  if (lp != p || lz != z || lscale != scale || lpl != pl) { lindiv1_build_tab(p, z, scale, test_flags, pl, &last); lp = p; lz = z; lscale = scale; lpl = pl; }
  *t = last;
}
static_assert(LINDIV_LB_EMPTY == LB_EMPTY && LINDIV1_MAX_CHUNKS <= LB_WORDS, "one look-back array convention for every one-launch scan");
template <int MODE, int PL>
static int lindiv1_launch(const FieldCtx& f, const u64* d_c, size_t d, const LinDiv1Tab& tab, u64* cur, u64* next, u32 nch,
                          u64* d_quot, u64* d_rem, hipStream_t s) {
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((lindiv_one_kernel<MODE, PL, decltype(ops)>), dim3(nch), dim3(LINDIV1_NL), 0, s, ops, d_c, d,
                                        tab, cur, next, (u32)LB_WORDS, d_quot, d_rem); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
template <int MODE>
static int lindiv2_launch(const FieldCtx& f, const u64* d_c, size_t d, const LinDivTab& tab, u64* W, u64* H, u32 nch, u64* d_quot,
                          u64* d_rem, hipStream_t s) {
  FIELD_DISPATCH(f, {
    hipLaunchKernelGGL((lindiv_scan_kernel<MODE, decltype(ops)>), dim3(nch), dim3(256), 0, s, ops, d_c, d, tab, W, H);
    hipLaunchKernelGGL((lindiv_apply_kernel2<MODE, decltype(ops)>), dim3(nch), dim3(256), 0, s, ops, d_c, d, tab, W, H, d_quot, d_rem);
  });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
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
  if (d <= FUSED_EVAL_MAX && !g_no_fused_scans) {   // weighted chunk sums + a plain sum
    const size_t nch = (d + FCH - 1) / FCH;
    HornerTab2 tab2;
    make_horner_tab2(p, x % p, 1, &tab2);
    WsLease ws;
    RCHK(ws.acquire(nch * 8, s));
    if (nch <= LB_MAX && !g_no_onepass_scans && !stream_is_capturing(s)) {   // one launch (scan_kernels.h)
      u64 *cur, *next;
      ws.lb_arrays(&cur, &next);
      FIELD_DISPATCH(f, { hipLaunchKernelGGL((eval_onepass_kernel<decltype(ops)>), dim3((u32)nch), dim3(256), 0, s, ops, d_c, d,
                                            tab2, cur, next, d_out); });
      HIPCHK(hipGetLastError());
      ws.lb_commit();
      return RONK_OK;
    }
    FIELD_DISPATCH(f, {
      hipLaunchKernelGGL((weighted_chunk_sum8_kernel<decltype(ops)>), dim3((u32)nch), dim3(256), 0, s, ops, d_c, d, tab2, ws.u());
      hipLaunchKernelGGL((partial_sum_kernel<decltype(ops)>), dim3(1), dim3(256), 0, s, ops, ws.u(), nch, d_out);
    });
    HIPCHK(hipGetLastError());
    return RONK_OK;
  }
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
  if (g_lindiv && d <= (size_t)LINDIV_CHUNK * 4096 && !g_no_fused_scans && !g_onepass_div) {   // lindiv_kernels.h
    const bool direct = g_lindiv == 2 && ((uintptr_t)d_c & 15) == 0;
    // ONE launch, 16 bytes of traffic per coefficient (lindiv_kernels.h lindiv_one_kernel): 4 coefficients per lane below 1.5 M
    // coefficients (chunks of 4096: every CU has a workgroup from 2^20 coefficients on), 8 from there to 2^23 (chunks of 8192, at most
    // one look-back entry per lane; beyond 2^22 the workgroups enter in two rounds, each waiting for earlier ones only).  Measured
    // against the two launches at every size from 2^16 to 2^23: 7.8 / 9.5 us ... 8.9 / 11.1 (2^20) ... 19.6 / 22.9 (2^22) ... 34.9 /
    // 41.7 us (profiles/r06_lindiv_one.txt).  Not for a quotient written over the dividend (a workgroup whose wait runs out
    // recomputes chunk sums from the coefficients, which other workgroups may have overwritten by then), not under stream capture
    // (the look-back parity is host state), not for an unaligned dividend (the 16-byte run loads; the LDS-image fill does not fit
    // 64 VGPRs), not for z = 0 (its scans run on values weighted by powers of z; division by b1 x is a shift: the two launches do
    // it).  RONK_LINDIV_ONE=0: never; RONK_LINDIV_ONE_PL=4 / 8: that many per lane at every size (A/B).
    static const int pl_forced = [] { const char* e = getenv("RONK_LINDIV_ONE_PL"); const int v = e ? atoi(e) : 0; return v == 4 || v == 8 ? v : 0; }();
    const int pl1 = pl_forced ? pl_forced : d < ((size_t)3 << 19) ? 4 : 8;
    const size_t nch1 = (d + (size_t)LINDIV1_NL * pl1 - 1) / ((size_t)LINDIV1_NL * pl1);
    const bool overlap1 = d_quot < d_c + d && d_c < d_quot + d;
    if (g_lindiv_one && z != 0 && (direct || !LINDIV1_DIRECT) && nch1 <= g_lindiv_one_maxch && !overlap1 && !g_no_onepass_scans &&
        !stream_is_capturing(s)) {
      LinDiv1Tab tab1;
      make_lindiv1_tab(p, z, b1inv, pl1, &tab1);
      WsLease ws;
      RCHK(ws.acquire(64, s));
      u64 *cur, *next;
      ws.lb_arrays(&cur, &next);
      constexpr int M1 = LINDIV1_DIRECT ? LINDIV_DLOAD : 0;
      int rc1;
      if (pl1 == 4) rc1 = lindiv1_launch<M1, 4>(f, d_c, d, tab1, cur, next, (u32)nch1, d_quot, d_rem, s);
      else rc1 = lindiv1_launch<M1, 8>(f, d_c, d, tab1, cur, next, (u32)nch1, d_quot, d_rem, s);
      RCHK(rc1);
      ws.lb_commit();
      return RONK_OK;
    }
    const size_t nch = (d + LINDIV_CHUNK - 1) / LINDIV_CHUNK;
    LinDivTab tab;
    make_lindiv_tab(p, z, b1inv, &tab);
    WsLease ws;
    RCHK(ws.acquire(nch * 257 * 8, s));
    u64* H = ws.u();
    u64* W = ws.u() + nch;
    return direct ? lindiv2_launch<LINDIV_DLOAD>(f, d_c, d, tab, W, H, (u32)nch, d_quot, d_rem, s)
                  : lindiv2_launch<0>(f, d_c, d, tab, W, H, (u32)nch, d_quot, d_rem, s);
  }
  if (d <= FUSED_DIV_MAX && !g_no_fused_scans) {   // two launches: chunk sums, then carry + recurrence per chunk
    const size_t nch = (d + FCH - 1) / FCH;
    HornerTab2 tab2;
    make_horner_tab2(p, z, b1inv, &tab2);
    WsLease ws;
    RCHK(ws.acquire(nch * 8, s));
    // one launch (scan_kernels.h); not for a quotient written over the dividend: a workgroup whose wait runs out recomputes
    // the chunk sums above it from the coefficients, which other workgroups may have overwritten by then
    const bool overlap = d_quot < d_c + d && d_c < d_quot + d;
    if (g_onepass_div && !overlap && nch <= LB_DIV_MAX && !g_no_onepass_scans && !stream_is_capturing(s)) {
      u64 *cur, *next;
      ws.lb_arrays(&cur, &next);
      FIELD_DISPATCH(f, { hipLaunchKernelGGL((lindiv_onepass_kernel<decltype(ops)>), dim3((u32)nch), dim3(256), 0, s, ops, d_c, d,
                                            tab2, cur, next, d_quot, d_rem); });
      HIPCHK(hipGetLastError());
      ws.lb_commit();
      return RONK_OK;
    }
    FIELD_DISPATCH(f, {
      hipLaunchKernelGGL((chunk_sum8_kernel<decltype(ops)>), dim3((u32)nch), dim3(256), 0, s, ops, d_c, d, tab2, ws.u());
      hipLaunchKernelGGL((lindiv_fused_kernel<decltype(ops)>), dim3((u32)nch), dim3(256), 0, s, ops, d_c, d, tab2, ws.u(), d_quot,
                         d_rem);
    });
    HIPCHK(hipGetLastError());
    return RONK_OK;
  }
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

// Polynomial::<Lagrange<F>,F,D>::evaluate (polynomial/mod.rs:382-415), device resident.  d_status (may be NULL): set
// non-zero when two nodes coincide (the reference's F::ONE.div(ZERO) -> unwrap on None = RONK_ERR_ZERO_INVERSE).
extern "C" int ronk_lagrange_eval_dev(uint64_t p, const uint64_t* d_c, const uint64_t* d_nodes, size_t n, uint64_t x,
                                      uint64_t* d_out, int* d_status, void* stream) {
  if (!d_c || !d_nodes || !d_out || n == 0) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  hipStream_t s = (hipStream_t)stream;
  static const size_t fast_min = [] { const char* e = getenv("RONK_LAGRANGE_FAST_MIN"); return e ? (size_t)atol(e) : ((size_t)1 << 16) + 1; }();
  if (n >= fast_min) {
    // O(n) form for nodes = consecutive powers of an order-n element (what Lagrange::new builds); *d_status bit 2 = the
    // nodes are something else (the general O(n^2) formula is limited to n <= 2^16)
    if (!d_status || n > ((size_t)1 << 32)) return RONK_ERR_UNSUPPORTED;
    LagPrimes pr;
    pr.count = 0;
    { size_t m = n; for (size_t q = 2; q * q <= m; q++) if (m % q == 0) { pr.q[pr.count++] = q; while (m % q == 0) m /= q; }
      if (m > 1) pr.q[pr.count++] = m; }
    const u32 gb = grid_for(n);
    WsLease ws;
    RCHK(ws.acquire((size_t)gb * 8 + 64, s));
    FIELD_DISPATCH(f, {
      hipLaunchKernelGGL((lagrange_check_kernel<decltype(ops)>), dim3((u32)((n + 255) / 256)), dim3(256), 0, s, ops, d_nodes, n, pr, d_status);
      hipLaunchKernelGGL((lagrange_fast_terms_kernel<decltype(ops)>), dim3(gb), dim3(256), 0, s, ops, d_c, d_nodes, n, x % p, ws.u());
      hipLaunchKernelGGL((lagrange_fast_finish_kernel<decltype(ops)>), dim3(1), dim3(256), 0, s, ops, ws.u(), (size_t)gb, n, x % p, d_out,
                         (const int*)nullptr);
    });
    HIPCHK(hipGetLastError());
    return RONK_OK;
  }
  if (n > ((size_t)1 << 16)) return RONK_ERR_UNSUPPORTED;  // O(n^2) weights, as in the reference
  const u32 blocks = (u32)((n + 255) / 256);
  WsLease ws;
  RCHK(ws.acquire((size_t)blocks * 24 + 128, s));
  u64* ds = ws.u(); u64* dp = ws.u() + blocks;
  int* flag = d_status ? d_status : (int*)(ws.u() + 2 * (size_t)blocks);   // a scratch word when the caller does not ask
  // From 256 nodes on the O(n) form is tried first ON THE DEVICE: lagrange_check_kernel leaves `sel` at 0 for an omega^i table
  // (what Lagrange::new builds), the O(n) kernels then write the value and the general O(n^2) kernels return at once;
  // any other table sets `sel` and it is the other way round (2^16 nodes: ~5 ms -> ~0.05 ms for the usual table).
  int* sel = nullptr;
  if (n >= 256) {
    sel = (int*)(ws.u() + 2 * (size_t)blocks + 1);
    u64* fs = ws.u() + 2 * (size_t)blocks + 2;
    HIPCHK(hipMemsetAsync(sel, 0, 4, s));
    LagPrimes pr;
    pr.count = 0;
    { size_t m = n; for (size_t q = 2; q * q <= m; q++) if (m % q == 0) { pr.q[pr.count++] = q; while (m % q == 0) m /= q; }
      if (m > 1) pr.q[pr.count++] = m; }
    FIELD_DISPATCH(f, {
      hipLaunchKernelGGL((lagrange_check_kernel<decltype(ops)>), dim3(blocks), dim3(256), 0, s, ops, d_nodes, n, pr, sel);
      hipLaunchKernelGGL((lagrange_fast_terms_kernel<decltype(ops)>), dim3(blocks), dim3(256), 0, s, ops, d_c, d_nodes, n, x % p, fs);
      hipLaunchKernelGGL((lagrange_fast_finish_kernel<decltype(ops)>), dim3(1), dim3(256), 0, s, ops, fs, (size_t)blocks, n, x % p, d_out,
                         (const int*)sel);
    });
  }
  FIELD_DISPATCH(f, {
    hipLaunchKernelGGL((lagrange_terms_kernel<decltype(ops)>), dim3(blocks), dim3(256), 0, s, ops, d_c, d_nodes, n, x % p,
                       ds, dp, flag, (const int*)sel);
    hipLaunchKernelGGL((lagrange_finish_kernel<decltype(ops)>), dim3(1), dim3(256), 0, s, ops, ds, dp, (size_t)blocks, d_out,
                       (const int*)sel);
  });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_lagrange_eval(uint64_t p, const uint64_t* c, const uint64_t* nodes, size_t n, uint64_t x, uint64_t* out) {
  if (!c || !nodes || !out || n == 0) return RONK_ERR_INVALID;
  RCHK(need_device());
  DevBuf dc, dn, dres, dflag;
  RCHK(dc.alloc(n * 8)); RCHK(dn.alloc(n * 8)); RCHK(dres.alloc(8)); RCHK(dflag.alloc(4));
  HIPCHK(hipMemcpy(dc.p, c, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dn.p, nodes, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 4));
  RCHK(ronk_lagrange_eval_dev(p, dc.u(), dn.u(), n, x, dres.u(), (int*)dflag.p, 0));
  int hflag = 0;
  HIPCHK(hipMemcpy(&hflag, dflag.p, 4, hipMemcpyDeviceToHost));
  if (hflag & 4) return RONK_ERR_UNSUPPORTED;   // more than 2^16 nodes that are not the powers of an order-n element
  if (hflag) return RONK_ERR_ZERO_INVERSE;      // coincident nodes: F::ONE.div(ZERO) -> unwrap on None
  HIPCHK(hipMemcpy(out, dres.p, 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

// ------------------------------------------------------------------------------ fast general division (any NTT-friendly prime)
// quotient_and_remainder (polynomial/mod.rs:170-225) in O(n log n) on the NTT path, for divisors that are not linear:
//   Q = rev( rev(a) * inv(rev(b)) mod x^L ),  L = deg a - deg b + 1,   inv by Newton iteration g <- g + g*(1 - f*g),
// every product through ronk_poly_mul_dev (cached plans).  Used only when the divisor's last coefficient is non-zero
// (D2 = deg b + 1): then the reference's loop -- which compares the remainder's trimmed length with the divisor's
// UNTRIMMED length (mod.rs:184-186) and indexes p_coeffs[diff + i] over all D2 divisor entries -- is plain Euclidean
// division.  A divisor with trailing zero coefficients makes the reference stop early or panic (index out of bounds in
// its second iteration); those inputs stay on poly_divrem_kernel, which follows the loop statement by statement.
__global__ void __launch_bounds__(256) dv_reverse_kernel(const u64* __restrict__ in, size_t top, u64* __restrict__ out, size_t len) {
  // out[i] = in[top - i] for i < len (coefficients above the source read as ZERO: top - i < 0 never happens, len <= top + 1)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) out[i] = in[top - i];
}
template <class Ops>
__global__ void __launch_bounds__(256) dv_neg_kernel(Ops ops, const u64* __restrict__ in, u64* __restrict__ out, size_t len) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) out[i] = ops.neg(in[i]);
}
// out = [1 / lead, 0, 0, ...]: the precision-1 start of the Newton iteration, rhs.leading_coefficient().inverse().unwrap()
// (mod.rs:181, :196), computed HERE so that the call needs nothing from the host.  `lead` / `top_a` point at b[m] / a[n]: when the
// caller only PROMISED full-length operands (ronk_poly_divrem_full_dev), a zero there is reported through *status.
template <class Ops>
__global__ void __launch_bounds__(256) dv_fill_inv_kernel(Ops ops, u64* __restrict__ out, size_t len, const u64* __restrict__ lead,
                                                          const u64* __restrict__ top_a, int* status) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
    if (i == 0) {
      const u64 l = *lead;
      if (status) *status = (l == 0 || *top_a == 0) ? RONK_ERR_INVALID : 0;
      out[0] = l ? ops.pow(l, ops.order() - 2) : 0;
    } else {
      out[i] = 0;
    }
  }
}
// out[i] = i < n_src ? in[i] : 0 for i < len.  The Newton ladder copies and clears with KERNELS, not with hipMemcpyAsync /
// hipMemsetAsync: captured in a hipGraph, copy nodes of more than 16 KiB replay wrongly from the second replay on (ROCm 7.0.2;
// profiles/r05_capture_division.txt found it for the long division, tests/test_gpu_mont.py::test_full_length_division_is_capturable
// for this path), and the full-length form is meant to be captured.
__global__ void __launch_bounds__(256) dv_copy_kernel(const u64* __restrict__ in, size_t n_src, u64* __restrict__ out, size_t len) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) out[i] = i < n_src ? in[i] : 0;
}
// quot[i] = (i >= t && i < L) ? qrev[L - 1 - i] : 0   for i < d     (reverse back, clear below t)
__global__ void __launch_bounds__(256) dv_quot_kernel(const u64* __restrict__ qrev, size_t L, size_t t, u64* __restrict__ quot, size_t d) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < d; i += (size_t)gridDim.x * blockDim.x)
    quot[i] = (i >= t && i < L) ? qrev[L - 1 - i] : 0;
}
// rem[i] = a[i] - prod[i] (prod has plen entries)
template <class Ops>
__global__ void __launch_bounds__(256) dv_rem_kernel(Ops ops, const u64* __restrict__ a, const u64* __restrict__ prod, size_t plen, u64* __restrict__ rem, size_t d) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < d; i += (size_t)gridDim.x * blockDim.x)
    rem[i] = i < plen ? ops.sub(a[i], prod[i]) : a[i];
}

// Which fields the O(n log n) form serves, and with which root: the products inside it run on the NTT path (ronk_poly_mul_dev),
// whose result does not depend on WHICH primitive root the transform uses -- so for a prime other than Goldilocks any quadratic
// non-residue z serves (omega_{2^k} = z^((p-1)/2^k) has order exactly 2^k), found here like FieldExt::sqrt finds one
// (prime/mod.rs:197-201), and the caller needs no generator.  The prime must have the 2-adicity of the largest product
// (2 * 2^ceil(log2 d) points).  Round 5 knew only Goldilocks here: every other field took the one-workgroup long division,
// minutes at 2^22 by 2^21.
static bool newton_field(const FieldCtx& f, size_t d, u64* g) {
  if (d > ((size_t)1 << 27)) return false;
  if (f.kind == F_GL) { *g = RONK_GOLDILOCKS_G; return true; }
  if (f.kind != F_MONT || ronk_check_prime(f.p) != RONK_OK) return false;
  int need = 1; while (((size_t)1 << need) < d) need++;
  need += 1;
  if (need > 30 || (f.p - 1) % ((u64)1 << need) != 0) return false;
  u64 z = 2 % f.p;
  for (int tries = 0; tries < 1000 && h_powmod(z, (f.p - 1) / 2, f.p) == 1; tries++) z = (z + 1) % f.p;
  if (h_powmod(z, (f.p - 1) / 2, f.p) != f.p - 1) return false;
  *g = z;
  return true;
}

// a: d coefficients with degree n (a[n] != 0), b: degree m (b[m] != 0), n >= m.  d_quot / d_rem: d coefficients each.
// full_status: the device word that receives RONK_ERR_INVALID when a[n] or b[m] turns out to be ZERO (nullptr: the caller has
// looked at the operands itself)
static int newton_divrem_dev(const FieldCtx& fld, u64 G, const u64* d_a, size_t d, size_t n, const u64* d_b, size_t d2, size_t m,
                             int* full_status, u64* d_quot, u64* d_rem, hipStream_t s) {
  const u64 P = fld.p;
  const size_t L = n - m + 1;                       // coefficients of the true quotient
  size_t Lp = 1; while (Lp < L) Lp <<= 1;           // Newton runs to a power-of-two precision
  // temporaries from the event-guarded workspace pool (one lease, carved up): nothing is allocated, freed or waited for
  // per call, so the entry point stays asynchronous on `s`
  struct Span { u64* p; u64* u() const { return p; } };
  const size_t n_ar = (L > d ? L : d) + 1, n_qr = 2 * L, n_prod = L + m + 1;
  WsLease ws;
  RCHK(ws.acquire((Lp + 2 * Lp + 4 * Lp + Lp + 2 * Lp + n_ar + n_qr + n_prod) * 8, s));
  u64* cur = ws.u();
  auto take = [&](size_t cnt) { Span sp{cur}; cur += cnt; return sp; };
  const Span f = take(Lp), g = take(2 * Lp), e = take(4 * Lp), h = take(Lp), t1 = take(2 * Lp), ar = take(n_ar), qr = take(n_qr),
             prod = take(n_prod);
  // f = rev(b) mod x^Lp: f[i] = b[m - i] for i <= min(m, Lp - 1), ZERO above
  const size_t flen = (m + 1 < Lp) ? m + 1 : Lp;
  if (flen < Lp) hipLaunchKernelGGL(dv_copy_kernel, dim3(grid_for(Lp - flen)), dim3(256), 0, s, (const u64*)nullptr, (size_t)0, f.u() + flen, Lp - flen);
  hipLaunchKernelGGL(dv_reverse_kernel, dim3(grid_for(flen)), dim3(256), 0, s, d_b, m, f.u(), flen);
  // g = 1 / f[0] = 1 / lead(b)   (precision 1)
  FIELD_DISPATCH(fld, { hipLaunchKernelGGL((dv_fill_inv_kernel<decltype(ops)>), dim3(grid_for(2 * Lp)), dim3(256), 0, s, ops, g.u(), 2 * Lp,
                                          d_b + m, d_a + n, full_status); });
  for (size_t k = 1; k < Lp; k <<= 1) {
    // e = f[0:2k] * g[0:k]: coefficients [k, 2k) are the error term (the low k are [1, 0, ..])
    RCHK(ronk_poly_mul_dev(P, G, f.u(), 2 * k, g.u(), k, e.u(), s));
    FIELD_DISPATCH(fld, { hipLaunchKernelGGL((dv_neg_kernel<decltype(ops)>), dim3(grid_for(k)), dim3(256), 0, s, ops, e.u() + k, h.u(), k); });
    // g[k:2k] = (g[0:k] * h)[0:k]
    RCHK(ronk_poly_mul_dev(P, G, g.u(), k, h.u(), k, t1.u(), s));
    hipLaunchKernelGGL(dv_copy_kernel, dim3(grid_for(k)), dim3(256), 0, s, (const u64*)t1.u(), k, g.u() + k, k);
  }
  // qrev = (rev(a)[0:L] * g[0:L])[0:L]
  hipLaunchKernelGGL(dv_reverse_kernel, dim3(grid_for(L)), dim3(256), 0, s, d_a, n, ar.u(), L);
  RCHK(ronk_poly_mul_dev(P, G, ar.u(), L, g.u(), L, qr.u(), s));
  const size_t t = 0;                               // d2 == m + 1: every quotient coefficient is produced
  (void)d2;
  hipLaunchKernelGGL(dv_quot_kernel, dim3(grid_for(d)), dim3(256), 0, s, qr.u(), L, t, d_quot, d);
  // rem = a - quot[0:L] * b[0:m+1]
  RCHK(ronk_poly_mul_dev(P, G, d_quot, L, d_b, m + 1, prod.u(), s));
  FIELD_DISPATCH(fld, { hipLaunchKernelGGL((dv_rem_kernel<decltype(ops)>), dim3(grid_for(d)), dim3(256), 0, s, ops, d_a, prod.u(), L + m, d_rem, d); });
  HIPCHK(hipGetLastError());
  return RONK_OK;                                   // the lease's destructor leaves an event behind the last kernel
}

// number of significant coefficients (degree + 1, 0 for the zero polynomial) of two coefficient vectors, and the divisor's
// leading coefficient: out[0] = n(a), out[1] = n(b) (both zeroed by the caller), then out[2] = b[n(b) - 1]
__global__ void __launch_bounds__(256) dv_degree_kernel(const u64* __restrict__ a, size_t d, const u64* __restrict__ b, size_t d2,
                                                        unsigned long long* __restrict__ out) {
  unsigned long long na = 0, nb = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < d; i += (size_t)gridDim.x * blockDim.x) {
    if (a[i] != 0) na = i + 1;
    if (i < d2 && b[i] != 0) nb = i + 1;
  }
  if (na) atomicMax(&out[0], na);
  if (nb) atomicMax(&out[1], nb);
}
__global__ void dv_lead_kernel(const u64* __restrict__ b, unsigned long long* __restrict__ out) {
  out[2] = out[1] ? b[out[1] - 1] : 0;
}

// quotient_and_remainder on device-resident operands.  d_quot / d_rem receive d coefficients each (d_rem may alias d_a);
// *d_status (device int, required) receives 0 or the RONK_ERR_* code of the reference's panic.
// Any prime, any divisor: the long-division kernel that follows the reference's loop, asynchronous on `stream`.
// Goldilocks with a divisor of >= 64 coefficients and a quotient of >= 2048 (where the single-block long division would run
// for seconds to minutes): the operands' degrees and the divisor's leading coefficient are read back first -- ONE stream
// synchronisation, 24 bytes -- and the O(n log n) Newton form on the NTT path runs, as behind ronk_poly_divrem.  A capturing
// stream cannot be synchronised, so under capture the long-division kernel is kept.
extern "C" int ronk_poly_divrem_dev(uint64_t p, const uint64_t* d_a, size_t d, const uint64_t* d_b, size_t d2,
                                    uint64_t* d_quot, uint64_t* d_rem, int* d_status, void* stream) {
  if (!d_a || !d_b || !d_quot || !d_rem || !d_status || d == 0 || d2 == 0) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  hipStream_t s = (hipStream_t)stream;
  u64 gz = 0;
  if (d >= d2 && d2 >= 64 && d - d2 + 1 >= 2048 && newton_field(f, d, &gz)) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    // A capturing stream cannot be synchronised for the degree probe.  Where the one-workgroup long division is still a
    // matter of milliseconds it is captured instead; beyond that (d2 * (d - d2 + 1) field operations on ONE workgroup: seconds to
    // minutes) the call is refused rather than silently captured as a kernel nobody can afford -- ronk_poly_divrem_full_dev is
    // the capturable O(n log n) form.
    if (cap != hipStreamCaptureStatusNone && (double)d2 * (double)(d - d2 + 1) > 1.0e9) return RONK_ERR_UNSUPPORTED;
    if (cap == hipStreamCaptureStatusNone) {
      unsigned long long probe[3] = {0, 0, 0};
      {
        WsLease ws;
        RCHK(ws.acquire(64, s));
        unsigned long long* dp = (unsigned long long*)ws.u();
        HIPCHK(hipMemsetAsync(dp, 0, 24, s));
        hipLaunchKernelGGL(dv_degree_kernel, dim3(grid_for(d)), dim3(256), 0, s, d_a, d, d_b, d2, dp);
        hipLaunchKernelGGL(dv_lead_kernel, dim3(1), dim3(1), 0, s, d_b, dp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(probe, dp, 24, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
      }
      const size_t n = (size_t)probe[0], m = (size_t)probe[1];
      // the same window as the host-pointer form: a full-length divisor (a ragged one is the reference's panic, reported by
      // the long-division kernel), a quotient long enough for the O(n log n) form to win
      if (n > 0 && m == d2 && n >= m && (n - m + 1) >= 2048 && m >= 64) {
        HIPCHK(hipMemsetAsync(d_status, 0, 4, s));
        return newton_divrem_dev(f, gz, d_a, d, n - 1, d_b, d2, m - 1, nullptr, d_quot, d_rem, s);
      }
    }
  }
  const u32 T = d2 >= 1024 ? 1024 : d2 > 256 ? 512 : 256;
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((poly_divrem_kernel<decltype(ops)>), dim3(1), dim3(T), 0, s, ops, d_a, d_rem, d, d_b, d2,
                                        d_quot, d_status); });   // (copies the dividend and clears the status itself)
  HIPCHK(hipGetLastError());
  return RONK_OK;
}

// quotient_and_remainder for FULL-LENGTH operands (a[d - 1] != 0, b[d2 - 1] != 0 -- the common case: the caller knows its degrees),
// d >= d2: the O(n log n) form with NOTHING read back -- the leading coefficient is inverted on the device, the promise is
// checked there (*d_status = RONK_ERR_INVALID when a top coefficient is ZERO; the outputs are then meaningless) -- so the call
// is asynchronous on `stream` and can be captured in a hipGraph whatever the sizes.  Fields: Goldilocks, and every odd prime
// whose p - 1 has the 2-adicity of the product sizes (2^(ceil(log2 d) + 1) | p - 1); otherwise RONK_ERR_UNSUPPORTED (use
// ronk_poly_divrem_dev).  For full-length operands the reference's loop is plain Euclidean division (mod.rs:170-225).
extern "C" int ronk_poly_divrem_full_dev(uint64_t p, const uint64_t* d_a, size_t d, const uint64_t* d_b, size_t d2,
                                         uint64_t* d_quot, uint64_t* d_rem, int* d_status, void* stream) {
  if (!d_a || !d_b || !d_quot || !d_rem || !d_status || d == 0 || d2 == 0 || d < d2) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  u64 gz = 0;
  if (!newton_field(f, d, &gz)) return RONK_ERR_UNSUPPORTED;
  return newton_divrem_dev(f, gz, d_a, d, d - 1, d_b, d2, d2 - 1, d_status, d_quot, d_rem, (hipStream_t)stream);
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
  // general divisor over Goldilocks, large enough for the O(n log n) form to win: Newton inversion on the NTT path.
  // Zero operands, short dividends and the reference's panics keep going through the long-division kernel below,
  // which follows the reference loop statement by statement.
  u64 gz = 0;
  if (d >= d2 && d2 >= 64 && d - d2 + 1 >= 2048 && newton_field(f, d, &gz)) {
    size_t n = d, m = d2;
    while (n > 0 && a[n - 1] % p == 0) n--;        // n = degree + 1 of the dividend (0: zero polynomial)
    while (m > 0 && b[m - 1] % p == 0) m--;
    if (n > 0 && m == d2 && n >= m && (n - m + 1) >= 2048 && m >= 64) {
      const size_t dn = n - 1, dm = m - 1;
      DevBuf da, db2, dq2, dr2;
      RCHK(da.alloc(d * 8)); RCHK(db2.alloc(d2 * 8)); RCHK(dq2.alloc(d * 8)); RCHK(dr2.alloc(d * 8));
      HIPCHK(hipMemcpy(da.p, a, d * 8, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(db2.p, b, d2 * 8, hipMemcpyHostToDevice));
      RCHK(newton_divrem_dev(f, gz, da.u(), d, dn, db2.u(), d2, dm, nullptr, dq2.u(), dr2.u(), 0));
      HIPCHK(hipMemcpy(quot, dq2.p, d * 8, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(rem, dr2.p, d * 8, hipMemcpyDeviceToHost));
      return RONK_OK;
    }
  }
  DevBuf drem, db, dq, dst;
  RCHK(drem.alloc(d * 8)); RCHK(db.alloc(d2 * 8)); RCHK(dq.alloc(d * 8)); RCHK(dst.alloc(4));
  HIPCHK(hipMemcpy(drem.p, a, d * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db.p, b, d2 * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dst.p, 0, 4));
  const u32 T = d2 >= 1024 ? 1024 : d2 > 256 ? 512 : 256;
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((poly_divrem_kernel<decltype(ops)>), dim3(1), dim3(T), 0, 0, ops, (const u64*)drem.u(), drem.u(), d,
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

// Message::decode (codes/reed_solomon.rs:54-106): the first k coordinates -> the k message coefficients.
// Device resident: d_xs / d_ys hold CANONICAL residues; d_status (may be NULL) is set non-zero for coincident nodes.
// Message::decode for x_j = q^j in O(K log K) (interp_kernels.h): two linear convolutions on the NTT path.  `sel` (device word,
// zeroed by the caller) gets bit 2 when the nodes are not such a sequence; the result is written only when it stays 0.
// Asynchronous on `s`: the temporaries are one lease of the event-guarded workspace pool.
static int rs_decode_fast_dev(const u64* d_xs, const u64* d_ys, size_t k, u64* d_out, int* sel, hipStream_t s) {
  const u64 P = RONK_GOLDILOCKS_P, G = RONK_GOLDILOCKS_G;
  struct Span { u64* p; u64* u() const { return p; } };
  WsLease ws;
  RCHK(ws.acquire(((k + 1) + k + (2 * k - 1) + (3 * k - 2) + (k + 1) + k + 2 * k + 1025) * 8, s));
  u64* cur = ws.u();
  auto take = [&](size_t cnt) { Span sp{cur}; cur += cnt; return sp; };
  const Span B = take(k + 1), arev = take(k), b = take(2 * k - 1), conv = take(3 * k - 2), M = take(k + 1), srev = take(k),
             conv2 = take(2 * k), tot = take(1025);
  const size_t m = k + 1;
  const u32 nb = (u32)((m + 256 * RSF_PER - 1) / (256 * RSF_PER));
  hipLaunchKernelGGL(rsf_check_kernel, dim3((u32)((k + 255) / 256)), dim3(256), 0, s, d_xs, k, sel);
  hipLaunchKernelGGL(rsf_factors_kernel, dim3(grid_for(m)), dim3(256), 0, s, d_xs, k, B.u());
  hipLaunchKernelGGL(rsf_scan_totals_kernel, dim3(nb), dim3(256), 0, s, (const u64*)B.u(), m, tot.u());
  hipLaunchKernelGGL(rsf_scan_mid_kernel, dim3(1), dim3(1024), 0, s, tot.u(), nb);
  hipLaunchKernelGGL(rsf_scan_apply_kernel, dim3(nb), dim3(256), 0, s, B.u(), m, (const u64*)tot.u());
  hipLaunchKernelGGL(rsf_prepare_kernel, dim3(grid_for(2 * k - 1)), dim3(256), 0, s, d_xs, d_ys, (const u64*)B.u(), k, arev.u(), b.u(),
                     M.u());
  HIPCHK(hipGetLastError());
  RCHK(ronk_poly_mul_dev(P, G, arev.u(), k, b.u(), 2 * k - 1, conv.u(), s));
  hipLaunchKernelGGL(rsf_unchirp_kernel, dim3(grid_for(k)), dim3(256), 0, s, d_xs, (const u64*)conv.u(), k, srev.u());
  HIPCHK(hipGetLastError());
  RCHK(ronk_poly_mul_dev(P, G, srev.u(), k, M.u(), k + 1, conv2.u(), s));
  hipLaunchKernelGGL(rsf_extract_kernel, dim3(grid_for(k)), dim3(256), 0, s, (const u64*)conv2.u(), k, (const int*)sel, d_out);
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
static const size_t RS_FAST_MAX_K = (size_t)1 << 21;

extern "C" int ronk_rs_decode_dev(uint64_t p, const uint64_t* d_xs, const uint64_t* d_ys, size_t k, uint64_t* d_out,
                                  int* d_status, void* stream) {
  if (k == 0) return RONK_OK;
  if (!d_xs || !d_ys || !d_out) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  hipStream_t s = (hipStream_t)stream;
  static const size_t fast_min = [] { const char* e = getenv("RONK_RS_FAST_MIN"); return e ? (size_t)atol(e) : (size_t)1024; }();
  const bool try_fast = f.kind == F_GL && k >= fast_min && k >= 2 && k <= RS_FAST_MAX_K;
  if (k > RS_DECODE_MAX_K) {
    // beyond the O(K^2) kernels only the geometric node sequences of Message::encode are covered: bit 2 of *d_status otherwise
    if (!try_fast || !d_status) return RONK_ERR_UNSUPPORTED;
    return rs_decode_fast_dev(d_xs, d_ys, k, d_out, d_status, s);
  }
  const u32 nblk = (u32)((k + 255) / 256);
  WsLease ws;
  RCHK(ws.acquire((k + (k + 1) + (size_t)nblk * k + 16) * 8, s));
  u64* dw = ws.u(); u64* dm = dw + k; u64* dpart = dm + (k + 1);
  int* flag = d_status ? d_status : (int*)(dpart + (size_t)nblk * k);
  int* sel = nullptr;
  if (try_fast) {   // the device decides: `sel` stays 0 for x_j = q^j (fast kernels write the result, the others return at once)
    sel = (int*)(dpart + (size_t)nblk * k + 1);
    HIPCHK(hipMemsetAsync(sel, 0, 4, s));
    RCHK(rs_decode_fast_dev(d_xs, d_ys, k, d_out, sel, s));
  }
  FIELD_DISPATCH(f, {
    hipLaunchKernelGGL((rs_weights_kernel<decltype(ops)>), dim3(nblk), dim3(256), 0, s, ops, d_xs, d_ys, k, dw, flag, (const int*)sel);
    hipLaunchKernelGGL((master_poly_kernel<decltype(ops)>), dim3(1), dim3(1024), 0, s, ops, d_xs, k, dm, (const int*)sel);
    hipLaunchKernelGGL((rs_accumulate_kernel<decltype(ops)>), dim3(nblk), dim3(256), 0, s, ops, d_xs, dw, dm, k, dpart, (const int*)sel);
    hipLaunchKernelGGL((rs_finish_kernel<decltype(ops)>), dim3(nblk), dim3(256), 0, s, ops, dpart, (size_t)nblk, k, d_out, (const int*)sel);
  });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_rs_decode(uint64_t p, const uint64_t* xs, const uint64_t* ys, size_t k, uint64_t* out) {
  if (k == 0) return RONK_OK;
  if (!xs || !ys || !out) return RONK_ERR_INVALID;
  if (k > RS_FAST_MAX_K) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  std::vector<u64> hx(k), hy(k);
  for (size_t i = 0; i < k; i++) { hx[i] = xs[i] % p; hy[i] = ys[i] % p; }
  DevBuf dx, dy, dout, dflag;
  RCHK(dx.alloc(k * 8)); RCHK(dy.alloc(k * 8)); RCHK(dout.alloc(k * 8)); RCHK(dflag.alloc(4));
  HIPCHK(hipMemcpy(dx.p, hx.data(), k * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dy.p, hy.data(), k * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 4));
  RCHK(ronk_rs_decode_dev(p, dx.u(), dy.u(), k, dout.u(), (int*)dflag.p, 0));
  int hflag = 0;
  HIPCHK(hipMemcpy(&hflag, dflag.p, 4, hipMemcpyDeviceToHost));
  if (hflag & 4) return RONK_ERR_UNSUPPORTED;  // more than 2^14 nodes that are not q^j
  if (hflag) return RONK_ERR_ZERO_INVERSE;     // coincident nodes: numerator / ZERO
  HIPCHK(hipMemcpy(out, dout.p, k * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

// kzg::commit (src/kzg/setup.rs:45-60): sum_i points[i] * scalars[i] on y^2 = x^3 + a x + b over F_p[u]/(u^2 - nr).
// Device resident: d_points = n_points x 5 words (x0, x1, y0, y1, infinity flag; coordinates canonical), d_scalars = n
// words, d_out = 5 words.  *d_status (device int, required; the caller zeroes it): bit 0 = a point is not on the curve
// (AffinePoint::new's panic), bit 1 = a division by zero inside the group law.  With ronk_poly_div_linear_dev the whole of
// kzg::open (src/kzg/setup.rs:63-78: quotient, then commit) stays on the device.
extern "C" int ronk_curve_msm_dev(const ronk_curve* cv, const uint64_t* d_points, size_t n_points, const uint64_t* d_scalars,
                                  size_t n, uint64_t* d_out, int* d_status, void* stream) {
  if (!cv || !d_out || !d_status || (n && (!d_points || !d_scalars))) return RONK_ERR_INVALID;
  if (cv->p < 3 || cv->p >= ((u64)1 << 32)) return RONK_ERR_UNSUPPORTED;   // products of residues must fit 64 bits
  RCHK(ronk_check_prime(cv->p));
  if (n_points < n) return RONK_ERR_INDEX;          // assert!(g1_srs.len() >= coeffs.len())
  RCHK(need_device());
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {                                     // empty sum -> Infinity
    const u64 inf[5] = {0, 0, 0, 0, 1};
    HIPCHK(hipMemcpyAsync(d_out, inf, 40, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));                // `inf` lives on this stack frame
    return RONK_OK;
  }
  const u64 p = cv->p;
  CurveCtx c{p, cv->nr % p, cv->a % p, cv->b % p};
  const u32 nblk = (u32)((n + 255) / 256);
  WsLease ws;
  RCHK(ws.acquire((size_t)nblk * 5 * 8, s));
  hipLaunchKernelGGL(msm_terms_kernel, dim3(nblk), dim3(256), 0, s, c, d_points, d_scalars, n, ws.u(), d_status);
  hipLaunchKernelGGL(msm_reduce_kernel, dim3(1), dim3(256), 0, s, c, ws.u(), (size_t)nblk, d_out, d_status);
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_curve_msm(const ronk_curve* cv, const uint64_t* points, size_t n_points, const uint64_t* scalars,
                              size_t n, uint64_t out[5]) {
  if (!cv || !out || (n && (!points || !scalars))) return RONK_ERR_INVALID;
  if (cv->p < 3 || cv->p >= ((u64)1 << 32)) return RONK_ERR_UNSUPPORTED;
  RCHK(ronk_check_prime(cv->p));
  if (n_points < n) return RONK_ERR_INDEX;
  if (n == 0) { out[0] = out[1] = out[2] = out[3] = 0; out[4] = 1; return RONK_OK; }   // empty sum -> Infinity
  RCHK(need_device());
  const u64 p = cv->p;
  std::vector<u64> hp(5 * n), hs(n);
  for (size_t i = 0; i < n; i++) {
    for (int w = 0; w < 4; w++) hp[5 * i + w] = points[5 * i + w] % p;
    hp[5 * i + 4] = points[5 * i + 4] ? 1 : 0;
    hs[i] = scalars[i];
  }
  DevBuf dp, ds, dout, dflag;
  RCHK(dp.alloc(5 * n * 8)); RCHK(ds.alloc(n * 8)); RCHK(dout.alloc(5 * 8)); RCHK(dflag.alloc(4));
  HIPCHK(hipMemcpy(dp.p, hp.data(), 5 * n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds.p, hs.data(), n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 4));
  RCHK(ronk_curve_msm_dev(cv, dp.u(), n, ds.u(), n, dout.u(), (int*)dflag.p, 0));
  int hflag = 0;
  HIPCHK(hipMemcpy(&hflag, dflag.p, 4, hipMemcpyDeviceToHost));
  if (hflag & CURVE_ERR_NOT_ON_CURVE) return RONK_ERR_NOT_ON_CURVE;   // AffinePoint::new: "Point is not on curve"
  if (hflag & CURVE_ERR_INVERSE) return RONK_ERR_ZERO_INVERSE;        // Div: expect("invalid inverse")
  HIPCHK(hipMemcpy(out, dout.p, 5 * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}
