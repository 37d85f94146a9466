// runtime.h -- internal declarations shared by the translation units of libronk_ntt.so (not part of the C ABI).
//
//   ronk_core.hip     errors, device query, host integer logic, element-wise vector ops, device helpers
//   ronk_plan.hip     plans, transforms, plan cache, fft/ifft/dft, polynomial multiply, batched RS encode
//   ronk_callers.hip  evaluate, division, Lagrange evaluate, Reed-Solomon encode/decode, KZG commit (MSM)
//   ronk_dist.hip     multi-GPU four-step phases
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/ronk_ntt.h"
#include "field_kernels.h"
#include "plan.h"
#include "tile_launch.h"

using namespace ronk;

// ---- errors (ronk_core.hip)
int hip_fail(hipError_t e, const char* what);   // records the message for ronk_last_hip_error, returns RONK_ERR_HIP
int need_device();                              // RONK_OK or RONK_ERR_NO_DEVICE: there is no CPU compute path
#define HIPCHK(call)                                   \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #call);  \
  } while (0)
#define RCHK(call)            \
  do {                        \
    int rc_ = (call);         \
    if (rc_ != RONK_OK) return rc_; \
  } while (0)

// ---- host integer logic
typedef unsigned __int128 u128;
static inline u64 h_mulmod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }
static inline u64 h_powmod(u64 a, u64 e, u64 p) {
  u64 r = 1 % p;
  a %= p;
  while (e) { if (e & 1) r = h_mulmod(r, a, p); a = h_mulmod(a, a, p); e >>= 1; }
  return r;
}

// ---- field dispatch
enum FieldKind { F_GL, F_MONT, F_MOD2 };
struct FieldCtx {
  FieldKind kind;
  u64 p;
  MontOps mont;
};
static inline int make_field(u64 p, FieldCtx* f) {
  if (p < 2) return RONK_ERR_INVALID;
  f->p = p;
  if (p == RONK_GOLDILOCKS_P) { f->kind = F_GL; return RONK_OK; }
  if (p == 2) { f->kind = F_MOD2; return RONK_OK; }
  if (!(p & 1)) return RONK_ERR_NOT_PRIME;
  f->kind = F_MONT;
  f->mont.f = mont64::make_field(p);
  return RONK_OK;
}
// run `body(ops)` with the Ops object matching the field
#define FIELD_DISPATCH(fctx, ...)                                               \
  do {                                                                          \
    if ((fctx).kind == F_GL) { GlOps ops; __VA_ARGS__; }                        \
    else if ((fctx).kind == F_MONT) { MontOps ops = (fctx).mont; __VA_ARGS__; } \
    else { Mod2Ops ops; __VA_ARGS__; }                                          \
  } while (0)

static inline u32 grid_for(size_t n, u32 block = 256) {
  size_t g = (n + block - 1) / block;
  if (g > 8192) g = 8192;  // 256 CUs x 32: grid-stride the rest
  if (g < 1) g = 1;
  return (u32)g;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    return RONK_OK;
  }
  u64* u() const { return (u64*)p; }
};

static inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static inline int ilog2(size_t n) { int k = 0; while (((size_t)1 << k) < n) k++; return k; }

// ---- plans
static inline int upload(const std::vector<u64>& h, u64** d) {
  HIPCHK(hipMalloc((void**)d, h.size() * 8 + 8));
  HIPCHK(hipMemcpy(*d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  return RONK_OK;
}

// a PlanDesc (plan.h) with its twiddle tables resident in HBM
struct CompiledPlan {
  PlanDesc pd;
  std::vector<u64*> d_wr;
  std::vector<std::pair<u64*, u64*>> d_tw;
  std::vector<u64*> d_twf;
  int compile(const PlanDesc& desc) {
    pd = desc;
    for (auto& t : pd.twf) { u64* d = nullptr; RCHK(upload(t, &d)); d_twf.push_back(d); }
    for (auto& t : pd.twf) std::vector<u64>().swap(t);  // the host copy is not needed any more
    for (auto& t : pd.wr) { u64* d = nullptr; RCHK(upload(t, &d)); d_wr.push_back(d); }
    for (auto& t : pd.tw) {
      u64 *lo = nullptr, *hi = nullptr;
      RCHK(upload(t.lo, &lo)); RCHK(upload(t.hi, &hi));
      d_tw.push_back({lo, hi});
    }
    return RONK_OK;
  }
  void release() {
    for (auto* q : d_wr) (void)hipFree(q);
    for (auto& q : d_tw) { (void)hipFree(q.first); (void)hipFree(q.second); }
    for (auto* q : d_twf) (void)hipFree(q);
    d_wr.clear(); d_tw.clear(); d_twf.clear();
  }
  // launch pass idx: BUF_IN -> in (and in2), BUF_OUT -> out, BUF_TMP -> tmp
  // in_valid / out_valid: implicit zero padding of the input / truncation of the output (TileArgs), ~0 = none
  // in_poly_stride (multi-pass plans, 0 = n): element stride between the polynomials of a batched input
  // x0_add: added to the X offset of passes whose inter-pass twiddle depends on the column (column chunks of the
  // sharded four-step: chunk j starts j*Cwc columns further right, ronk_dist.hip)
  // sb0 / sbn: launch the pass for polynomials [sb0, sb0 + sbn) of the batch only (sbn = 0: all of them)
  // the launch arguments of pass idx with buffers and tables bound, nothing else changed (the fused multiply hands two
  // passes of two plans to one kernel: ronk_plan.hip conv_dev)
  TileArgs bound(size_t idx, const u64* in, const u64* in2, u64* out, u64* tmp) const {
    const PassDesc& ps = pd.passes[idx];
    TileArgs a = ps.args;
    const u64* bufs_in[3] = {in, out, tmp};
    u64* bufs_out[3] = {nullptr, out, tmp};
    a.in = bufs_in[ps.in_buf];
    a.in2 = (ps.in_buf == BUF_IN) ? in2 : nullptr;
    a.out = bufs_out[ps.out_buf];
    a.wr = d_wr[ps.wr_id];
    if (ps.tw_id >= 0) { a.tw_lo = d_tw[ps.tw_id].first; a.tw_hi = d_tw[ps.tw_id].second; }
    if (ps.twf_id >= 0) a.tw_full = d_twf[ps.twf_id];
    return a;
  }
  int launch(size_t idx, const u64* in, const u64* in2, u64* out, u64* tmp, hipStream_t s, u64 in_valid = ~(u64)0,
             u64 out_valid = ~(u64)0, u64 in_poly_stride = 0, u64 x0_add = 0, u32 sb0 = 0, u32 sbn = 0,
             u64 in_valid1 = ~(u64)0) const {
    const PassDesc& ps = pd.passes[idx];
    TileArgs a = ps.args;
    const u64* bufs_in[3] = {in, out, tmp};
    u64* bufs_out[3] = {nullptr, out, tmp};
    a.in = bufs_in[ps.in_buf];
    a.in2 = (ps.in_buf == BUF_IN) ? in2 : nullptr;
    a.out = bufs_out[ps.out_buf];
    if (ps.in_buf == BUF_IN) {
      a.in_valid = in_valid;
      a.in_valid1 = in_valid1;
      if (in_poly_stride) a.in_sb1 = (i64)in_poly_stride;   // nb1 is the batch axis of every multi-pass plan (plan.h)
    }
    if (ps.out_buf == BUF_OUT) a.out_valid = out_valid;
    if (x0_add && a.tw_log && a.xc) a.x0 += x0_add * a.xc;
    if (in_valid != ~(u64)0 || out_valid != ~(u64)0 || in_poly_stride) a.stage_io = 0;   // staged I/O copies whole tiles
    a.wr = d_wr[ps.wr_id];
    if (ps.tw_id >= 0) { a.tw_lo = d_tw[ps.tw_id].first; a.tw_hi = d_tw[ps.tw_id].second; }
    if (ps.twf_id >= 0) a.tw_full = d_twf[ps.twf_id];
    u32 grid = ps.grid;
    if (sbn) {   // a slice of the batch axis b1: shift the three base pointers, shrink the grid
      a.in += (i64)sb0 * a.in_sb1;
      if (a.in2) a.in2 += (i64)sb0 * a.in_sb1;
      a.out += (i64)sb0 * a.out_sb1;
      grid = ps.grid / a.nb1 * sbn;
      a.nb1 = sbn;
      // a slice that starts past polynomial 0 sees its first polynomial as b1 = 0: its padding limit is the one of b1 >= 1
      if (sb0 >= 1 && ps.in_buf == BUF_IN && in_valid1 != ~(u64)0) a.in_valid = in_valid1;
    }
    hipError_t e = ps.small ? launch_small(ps.logr, pd.inverse, a, grid, ps.block, ps.lds_bytes, s)
                            : launch_tile(ps.logr, pd.inverse, a, grid, ps.block, ps.lds_bytes, s);
    if (e != hipSuccess) return hip_fail(e, "launch_tile");
    return RONK_OK;
  }
  // Large batches of multi-pass plans run in SLICES of the batch axis, all passes of a slice before the next slice: the
  // scratch a slice writes in one pass is read back by the next pass while it is still in the 256 MB Infinity Cache,
  // instead of streaming the whole batch (512 MiB for 1024 x 2^16) through HBM between the passes.
  // RONK_SUB_BATCH_MIB: slice size in MiB of coefficients (0 = off).
  int run(const u64* in, const u64* in2, u64* out, u64* tmp, hipStream_t s, u64 in_valid = ~(u64)0,
          u64 out_valid = ~(u64)0, u64 in_poly_stride = 0, u64 x0_add = 0, u64 in_valid1 = ~(u64)0) const {
    static const long slice_mib = [] { const char* e = getenv("RONK_SUB_BATCH_MIB"); return e ? atol(e) : 0L; }();
    const u64 n = (u64)1 << pd.log2n;
    if (slice_mib > 0 && pd.passes.size() >= 2 && pd.batch > 1) {
      u64 per = ((u64)slice_mib << 20) / (n * 8);
      if (per < 1) per = 1;
      bool ok = per < pd.batch;
      for (auto& ps : pd.passes) ok = ok && ps.args.nb1 == pd.batch;
      if (ok) {
        for (u64 b0 = 0; b0 < pd.batch; b0 += per) {
          const u32 cnt = (u32)(pd.batch - b0 < per ? pd.batch - b0 : per);
          for (size_t i = 0; i < pd.passes.size(); i++)
            RCHK(launch(i, in, in2, out, tmp, s, in_valid, out_valid, in_poly_stride, x0_add, (u32)b0, cnt, in_valid1));
        }
        return RONK_OK;
      }
    }
    for (size_t i = 0; i < pd.passes.size(); i++)
      RCHK(launch(i, in, in2, out, tmp, s, in_valid, out_valid, in_poly_stride, x0_add, 0, 0, in_valid1));
    return RONK_OK;
  }
  // every pass of the plan for polynomials [sb0, sb0 + sbn) of the batch only (multi-pass plans whose batch axis is b1)
  int run_slice(u32 sb0, u32 sbn, const u64* in, const u64* in2, u64* out, u64* tmp, hipStream_t s, u64 in_valid = ~(u64)0,
                u64 out_valid = ~(u64)0, u64 in_poly_stride = 0, u64 in_valid1 = ~(u64)0) const {
    for (size_t i = 0; i < pd.passes.size(); i++)
      RCHK(launch(i, in, in2, out, tmp, s, in_valid, out_valid, in_poly_stride, 0, sb0, sbn, in_valid1));
    return RONK_OK;
  }
  bool sliceable() const {   // the batch is axis b1 of every pass (two- and three-pass plans)
    if (pd.passes.size() < 2 || pd.batch < 2) return false;
    for (auto& ps : pd.passes) if (ps.args.nb1 != pd.batch) return false;
    return true;
  }
};

struct ronk_plan {
  u64 p, g;
  u32 log2n;
  u64 n, batch;
  int device;
  FieldCtx field;
  bool fast;                // tile path (Goldilocks shift-twiddle kernels, or ...
  bool mont_tiled = false;  // ... the same kernels over Montgomery arithmetic: any other prime with a full 2-power subgroup under g)
  CompiledPlan fwd, inv;    // fast path
  u64* d_tmp = nullptr;     // scratch [batch][n]
  u64* d_stage_in = nullptr;   // staging for the host-pointer API (lazy)
  u64* d_stage_out = nullptr;
  // generic path: w^i tables (n/2 entries) for the radix-2 stages
  u64* d_wtab_f = nullptr; u64* d_wtab_i = nullptr;
  u64 w_f = 0, w_i = 0, n_inv = 1;
  std::mutex mu;
  // cross-stream guard of d_tmp (transform_dev): once a plan has been seen on more than one stream, every transform
  // that touches the scratch records `scratch_ev` at its END on its own stream, and a call arriving on another stream
  // waits for that already-recorded event -- the previous stream's handle is never used again (it may be gone)
  std::mutex stream_mu;
  hipStream_t scratch_stream = nullptr;   // identity of the last user, compared only
  hipEvent_t scratch_ev = nullptr;
  bool scratch_used = false, scratch_multi = false, scratch_ev_valid = false;
  // Library-side concurrency (ronk_plan_opts::in_flight = 2): a batched call is split in two halves of the batch axis,
  // the second half on an internal side stream (event fork / join on the caller's stream), so that the load / store
  // phases of one half's workgroups run under the arithmetic of the other's -- what two caller streams with two plans
  // do, behind ONE plan handle.  The halves use disjoint slices of d_tmp.  `d_tmp2`: a second scratch for
  // ronk_ntt_forward_many_dev (independent [batch][n] arrays round-robin over the two lanes).
  int in_flight = 1;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  u64* d_tmp2 = nullptr;
  // the three streams and per-slice events of the pipelined host-pointer transform (ronk_plan.hip transform_host)
  hipStream_t st_h2d = nullptr, st_d2h = nullptr, st_exec = nullptr;
  std::vector<hipEvent_t> st_ev;
};

// ronk_plan.hip
// in_poly_stride / in_valid1: see CompiledPlan::launch; tmp_override: a caller-owned scratch (the caller orders its uses)
int transform_dev(ronk_plan* pl, bool inverse, const u64* in, const u64* in2, u64* out, hipStream_t s,
                  u64 in_valid = ~(u64)0, u64 out_valid = ~(u64)0, u64 in_poly_stride = 0, u64 in_valid1 = ~(u64)0,
                  u64* tmp_override = nullptr);
