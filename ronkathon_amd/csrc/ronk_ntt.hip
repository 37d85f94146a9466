// ronk_ntt.hip -- C ABI of libronk_ntt.so (declared in include/ronk_ntt.h).
//
// Host-side runtime of the engine: field dispatch (Goldilocks fast path / generic Montgomery /
// p = 2), plan construction (plan.h), twiddle upload, kernel launches (tile_kernels.hip,
// field_kernels.h), staging for the host-pointer entry points.  No CPU compute path exists:
// every entry point that would compute returns RONK_ERR_NO_DEVICE without a HIP device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/ronk_ntt.h"
#include "field_kernels.h"
#include "scan_kernels.h"
#include "interp_kernels.h"
#include "curve_kernels.h"
#include "plan.h"
#include "tile_launch.h"

using namespace ronk;

// ------------------------------------------------------------------------------------ errors
static thread_local std::string g_hip_err;

static int hip_fail(hipError_t e, const char* what) {
  g_hip_err = std::string(what) + ": " + hipGetErrorString(e);
  return RONK_ERR_HIP;
}
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

extern "C" const char* ronk_strerror(int code) {
  switch (code) {
    case RONK_OK: return "ok";
    case RONK_ERR_NO_ROOT: return "n must divide p^q - 1";
    case RONK_ERR_ZERO_INVERSE: return "called `Option::unwrap()` on a `None` value (inverse of zero)";
    case RONK_ERR_NOT_POW2: return "number of coefficients is not a power of two";
    case RONK_ERR_NOT_PRIME: return "input is not a prime number";
    case RONK_ERR_NO_GENERATOR: return "generator not found";
    case RONK_ERR_INDEX: return "index out of bounds / unwrap on None";
    case RONK_ERR_INVALID: return "invalid argument";
    case RONK_ERR_HIP: return "HIP runtime error";
    case RONK_ERR_UNSUPPORTED: return "size not supported by this kernel";
    case RONK_ERR_NO_DEVICE: return "no HIP device (libronk_ntt has no CPU path)";
    case RONK_ERR_NOT_ON_CURVE: return "Point is not on curve";
    default: return "unknown error";
  }
}
extern "C" const char* ronk_last_hip_error(void) { return g_hip_err.c_str(); }

extern "C" int ronk_device_count(int* count) {
  if (!count) return RONK_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return RONK_OK; }
  *count = n;
  return RONK_OK;
}
static int need_device() {
  int n = 0;
  ronk_device_count(&n);
  return n > 0 ? RONK_OK : RONK_ERR_NO_DEVICE;
}

// --------------------------------------------------------------- host integer logic (no compute path)
typedef unsigned __int128 u128;
static u64 h_mulmod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }
static u64 h_powmod(u64 a, u64 e, u64 p) {
  u64 r = 1 % p;
  a %= p;
  while (e) { if (e & 1) r = h_mulmod(r, a, p); a = h_mulmod(a, a, p); e >>= 1; }
  return r;
}
// deterministic Miller-Rabin for 64-bit inputs; same predicate as the reference's trial division
// (prime/mod.rs:92-100), including its vacuous pass for n < 2
extern "C" int ronk_check_prime(uint64_t n) {
  static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return RONK_OK;
  for (u64 b : bases) { if (n == b) return RONK_OK; if (n % b == 0) return RONK_ERR_NOT_PRIME; }
  u64 d = n - 1; int s = 0;
  while (!(d & 1)) { d >>= 1; s++; }
  for (u64 b : bases) {
    u64 x = h_powmod(b, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int r = 1; r < s; r++) { x = h_mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
    if (comp) return RONK_ERR_NOT_PRIME;
  }
  return RONK_OK;
}

// FiniteField::PRIMITIVE_ELEMENT: prime/mod.rs:87-90, :110-123 restated literally for small p
// (g = 2 for F_101, 14 for F_17, 3 for F_127); Goldilocks carries the explicit generator 7
// because the heuristic returns the non-generator 3 there.
extern "C" int ronk_primitive_element(uint64_t p, uint64_t* g) {
  if (!g || p < 2) return RONK_ERR_INVALID;
  RCHK(ronk_check_prime(p));
  if (p == RONK_GOLDILOCKS_P) { *g = RONK_GOLDILOCKS_G; return RONK_OK; }
  if (p == 2) { *g = 1; return RONK_OK; }
  for (u128 i = 2; i * i <= p; i++) {
    u64 ii = (u64)i;
    if ((p - 1) % ii == 0) {
      if (h_powmod(ii, (p - 1) / ii, p) != 1) { *g = ii; return RONK_OK; }
      if (h_powmod(p + 1 - ii, ii, p) != 1) { *g = p + 1 - ii; return RONK_OK; }
    }
  }
  return RONK_ERR_NO_GENERATOR;
}

// field/mod.rs:70-75
extern "C" int ronk_root_of_unity(uint64_t p, uint64_t g, uint64_t n, uint64_t* out) {
  if (!out || p < 2 || n == 0) return RONK_ERR_INVALID;
  if ((p - 1) % n != 0) return RONK_ERR_NO_ROOT;
  *out = h_powmod(g, (p - 1) / n, p);
  return RONK_OK;
}

// ------------------------------------------------------------------------------ field dispatch
enum FieldKind { F_GL, F_MONT, F_MOD2 };
struct FieldCtx {
  FieldKind kind;
  u64 p;
  MontOps mont;
};
static int make_field(u64 p, FieldCtx* f) {
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

static u32 grid_for(size_t n, u32 block = 256) {
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

// ------------------------------------------------------------------------------ vector ops
template <int OP>
static int vec_binary_dev(u64 p, const u64* a, const u64* b, u64* out, size_t n, size_t nb, hipStream_t s) {
  FieldCtx f;
  RCHK(make_field(p, &f));
  if (n == 0) return RONK_OK;
  if (nb >= n && (n & 1) == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0) {
    FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_binary2_kernel<decltype(ops), OP>), dim3(grid_for(n / 2)), dim3(256), 0, s, ops,
                                          (const ulonglong2*)a, (const ulonglong2*)b, (ulonglong2*)out, n / 2); });
    HIPCHK(hipGetLastError());
    return RONK_OK;
  }
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_binary_kernel<decltype(ops), OP>), dim3(grid_for(n)), dim3(256), 0, s, ops,
                                        a, b, out, n, nb); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_vec_add_dev(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* st) {
  return vec_binary_dev<VEC_ADD>(p, a, b, out, n, n, (hipStream_t)st);
}
extern "C" int ronk_vec_sub_dev(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* st) {
  return vec_binary_dev<VEC_SUB>(p, a, b, out, n, n, (hipStream_t)st);
}
extern "C" int ronk_vec_mul_dev(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* st) {
  return vec_binary_dev<VEC_MUL>(p, a, b, out, n, n, (hipStream_t)st);
}

template <int OP>
static int vec_binary_host(u64 p, const u64* a, size_t n, const u64* b, size_t nb, u64* out) {
  if (!a || !b || !out) return RONK_ERR_INVALID;
  RCHK(need_device());
  DevBuf da, db;
  RCHK(da.alloc(n * 8)); RCHK(db.alloc(nb * 8));
  HIPCHK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db.p, b, nb * 8, hipMemcpyHostToDevice));
  RCHK((vec_binary_dev<OP>(p, da.u(), db.u(), da.u(), n, nb, 0)));
  HIPCHK(hipMemcpy(out, da.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}
extern "C" int ronk_vec_add(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  return vec_binary_host<VEC_ADD>(p, a, n, b, n, out);
}
extern "C" int ronk_vec_sub(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  return vec_binary_host<VEC_SUB>(p, a, n, b, n, out);
}
extern "C" int ronk_vec_mul(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  return vec_binary_host<VEC_MUL>(p, a, n, b, n, out);
}
// impl Add / Sub for Polynomial (arithmetic.rs:16-68): rhs zero-extended or truncated to len(lhs)
extern "C" int ronk_poly_add(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out) {
  return vec_binary_host<VEC_ADD>(p, a, d, b, d2 < d ? d2 : d, out);
}
extern "C" int ronk_poly_sub(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out) {
  return vec_binary_host<VEC_SUB>(p, a, d, b, d2 < d ? d2 : d, out);
}

extern "C" int ronk_vec_neg(uint64_t p, const uint64_t* a, uint64_t* out, size_t n) {
  if (!a || !out) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  DevBuf da;
  RCHK(da.alloc(n * 8));
  HIPCHK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice));
  if (n) FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_neg_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, 0, ops,
                                               da.u(), da.u(), n); });
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, da.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}
static int vec_pow_host(u64 p, const u64* a, u64 e, u64* out, size_t n, bool is_inverse) {
  if (!a || !out) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  DevBuf da, dflag;
  RCHK(da.alloc(n * 8)); RCHK(dflag.alloc(4));
  HIPCHK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 4));
  int* flag = is_inverse ? (int*)dflag.p : nullptr;
  if (n) FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_pow_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, 0, ops,
                                               da.u(), e, da.u(), n, flag); });
  HIPCHK(hipGetLastError());
  int hflag = 0;
  HIPCHK(hipMemcpy(&hflag, dflag.p, 4, hipMemcpyDeviceToHost));
  if (hflag) return RONK_ERR_ZERO_INVERSE;
  HIPCHK(hipMemcpy(out, da.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}
extern "C" int ronk_vec_pow(uint64_t p, const uint64_t* a, uint64_t e, uint64_t* out, size_t n) {
  return vec_pow_host(p, a, e, out, n, false);
}
extern "C" int ronk_vec_inv(uint64_t p, const uint64_t* a, uint64_t* out, size_t n) {
  if (p < 2) return RONK_ERR_INVALID;
  return vec_pow_host(p, a, p - 2, out, n, true);
}

// ------------------------------------------------------------------------------------- plans
static int upload(const std::vector<u64>& h, u64** d) {
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
  int launch(size_t idx, const u64* in, const u64* in2, u64* out, u64* tmp, hipStream_t s, u64 in_valid = ~(u64)0,
             u64 out_valid = ~(u64)0, u64 in_poly_stride = 0) const {
    const PassDesc& ps = pd.passes[idx];
    TileArgs a = ps.args;
    const u64* bufs_in[3] = {in, out, tmp};
    u64* bufs_out[3] = {nullptr, out, tmp};
    a.in = bufs_in[ps.in_buf];
    a.in2 = (ps.in_buf == BUF_IN) ? in2 : nullptr;
    a.out = bufs_out[ps.out_buf];
    if (ps.in_buf == BUF_IN) {
      a.in_valid = in_valid;
      if (in_poly_stride) a.in_sb1 = (i64)in_poly_stride;   // nb1 is the batch axis of every multi-pass plan (plan.h)
    }
    if (ps.out_buf == BUF_OUT) a.out_valid = out_valid;
    a.wr = d_wr[ps.wr_id];
    if (ps.tw_id >= 0) { a.tw_lo = d_tw[ps.tw_id].first; a.tw_hi = d_tw[ps.tw_id].second; }
    if (ps.twf_id >= 0) a.tw_full = d_twf[ps.twf_id];
    hipError_t e = launch_tile(ps.logr, pd.inverse, a, ps.grid, ps.block, ps.lds_bytes, s);
    if (e != hipSuccess) return hip_fail(e, "launch_tile");
    return RONK_OK;
  }
  int run(const u64* in, const u64* in2, u64* out, u64* tmp, hipStream_t s, u64 in_valid = ~(u64)0,
          u64 out_valid = ~(u64)0, u64 in_poly_stride = 0) const {
    for (size_t i = 0; i < pd.passes.size(); i++)
      RCHK(launch(i, in, in2, out, tmp, s, in_valid, out_valid, in_poly_stride));
    return RONK_OK;
  }
};

struct ronk_plan {
  u64 p, g;
  u32 log2n;
  u64 n, batch;
  int device;
  FieldCtx field;
  bool fast;                // Goldilocks tile path
  CompiledPlan fwd, inv;    // fast path
  u64* d_tmp = nullptr;     // scratch [batch][n]
  u64* d_stage_in = nullptr;   // staging for the host-pointer API (lazy)
  u64* d_stage_out = nullptr;
  // generic path: w^i tables (n/2 entries) for the radix-2 stages
  u64* d_wtab_f = nullptr; u64* d_wtab_i = nullptr;
  u64 w_f = 0, w_i = 0, n_inv = 1;
  std::mutex mu;
};

extern "C" int ronk_plan_destroy(ronk_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  pl->fwd.release();
  pl->inv.release();
  if (pl->d_tmp) (void)hipFree(pl->d_tmp);
  if (pl->d_stage_in) (void)hipFree(pl->d_stage_in);
  if (pl->d_stage_out) (void)hipFree(pl->d_stage_out);
  if (pl->d_wtab_f) (void)hipFree(pl->d_wtab_f);
  if (pl->d_wtab_i) (void)hipFree(pl->d_wtab_i);
  delete pl;
  return RONK_OK;
}

extern "C" int ronk_plan_create(ronk_plan** out, uint64_t p, uint64_t g, uint32_t log2n, uint64_t batch, int device) {
  return ronk_plan_create_tuned(out, p, g, log2n, batch, device, -1, -1);
}

extern "C" int ronk_plan_create_tuned(ronk_plan** out, uint64_t p, uint64_t g, uint32_t log2n, uint64_t batch, int device,
                                      int tile_log2_columns, int twiddle_matrix_log2_max) {
  if (!out || batch == 0 || log2n > 36) return RONK_ERR_INVALID;
  *out = nullptr;
  RCHK(ronk_check_prime(p));                                   // PrimeField::new -> is_prime
  if (p < 2) return RONK_ERR_INVALID;
  const u64 n = (u64)1 << log2n;
  if ((p - 1) % n != 0) return RONK_ERR_NO_ROOT;               // field/mod.rs:72, polynomial/mod.rs:361
  RCHK(need_device());
  if (device >= 0) HIPCHK(hipSetDevice(device));
  else HIPCHK(hipGetDevice(&device));
  ronk_plan* pl = new ronk_plan();
  pl->p = p; pl->g = g % p; pl->log2n = log2n; pl->n = n; pl->batch = batch; pl->device = device;
  int rc = make_field(p, &pl->field);
  if (rc) { delete pl; return rc; }
  // tile path: Goldilocks with the reference generator, 16 <= n <= 2^30 (32-bit lane offsets inside a tile,
  // grids below 2^31 workgroups); anything else takes the generic radix-2 path
  pl->fast = (p == RONK_GOLDILOCKS_P && pl->g == RONK_GOLDILOCKS_G && log2n >= 4 && log2n <= 30 &&
              batch < ((u64)1 << 31) && (double)batch * (double)n / 2048.0 < 2.0e9);
  if (pl->fast) {
    // columns per tile = 2^max_logc at most (4 = 128-byte segments, 1 workgroup per CU at 2^11 rows; 2 = 32-byte
    // segments but two workgroups per CU, better when several transforms are in flight); RONK_MAX_LOGC overrides
    int max_logc = 4;
    if (const char* e = getenv("RONK_MAX_LOGC")) { int v = atoi(e); if (v >= 0 && v <= 8) max_logc = v; }
    if (tile_log2_columns >= 0 && tile_log2_columns <= 8) max_logc = tile_log2_columns;
    // Full inter-pass twiddle matrix (one coalesced load + one multiply instead of two gathers + two
    // multiplies) while it stays L2-resident: up to 2^18 entries = 2 MiB.  Larger matrices would add an
    // n-element HBM read per transform (measured +4 % speed at 2^22 for +25 % traffic): left to RONK_TWF_MAX_LOG.
    int twf_max_log = 18;
    if (const char* e = getenv("RONK_TWF_MAX_LOG")) { int v = atoi(e); if (v >= 0 && v <= 26) twf_max_log = v; }
    if (twiddle_matrix_log2_max >= 0 && twiddle_matrix_log2_max <= 26) twf_max_log = twiddle_matrix_log2_max;
    int three_from = 25;  // RONK_THREE_PASS_FROM: split smaller sizes in three passes too (experiment knob)
    if (const char* e = getenv("RONK_THREE_PASS_FROM")) { int v = atoi(e); if (v >= 13 && v <= 25) three_from = v; }
    rc = pl->fwd.compile(build_plan((int)log2n, batch, false, max_logc, twf_max_log, three_from));
    if (!rc) rc = pl->inv.compile(build_plan((int)log2n, batch, true, max_logc, twf_max_log, three_from));
    for (auto& ps : pl->fwd.pd.passes)  // grid must fit the launch API
      if (!rc && (u64)ps.args.tiles * ps.args.nb1 * ps.args.nb2 > 0x7FFFFFFFull) rc = RONK_ERR_UNSUPPORTED;
  } else {
    pl->w_f = h_powmod(pl->g, (p - 1) / n, p);
    pl->w_i = h_powmod(pl->w_f, p - 2, p);                     // root.inverse().unwrap(), mod.rs:433
    pl->n_inv = h_powmod(n % p, p - 2, p);                     // F::from(D).inverse().unwrap(), mod.rs:442
    if (n % p == 0) { ronk_plan_destroy(pl); return RONK_ERR_ZERO_INVERSE; }
    if (log2n >= 1) {
      size_t half = n / 2;
      hipError_t e = hipMalloc((void**)&pl->d_wtab_f, half * 8);
      if (e == hipSuccess) e = hipMalloc((void**)&pl->d_wtab_i, half * 8);
      if (e != hipSuccess) { ronk_plan_destroy(pl); return hip_fail(e, "hipMalloc"); }
      FIELD_DISPATCH(pl->field, {
        hipLaunchKernelGGL((power_table_kernel<decltype(ops)>), dim3(grid_for(half)), dim3(256), 0, 0, ops, pl->w_f,
                           pl->d_wtab_f, half);
        hipLaunchKernelGGL((power_table_kernel<decltype(ops)>), dim3(grid_for(half)), dim3(256), 0, 0, ops, pl->w_i,
                           pl->d_wtab_i, half);
      });
      if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = RONK_ERR_HIP;
    }
  }
  if (!rc) {
    hipError_t e = hipMalloc((void**)&pl->d_tmp, n * batch * 8);
    if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(scratch)");
  }
  if (rc) { ronk_plan_destroy(pl); return rc; }
  *out = pl;
  return RONK_OK;
}

extern "C" int ronk_plan_num_passes(const ronk_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  return pl->fast ? (int)pl->fwd.pd.passes.size() : (int)pl->log2n + 1;
}

// generic power-of-two transform: bit-reversal copy + log2(n) radix-2 DIT stages
static int generic_transform(ronk_plan* pl, bool inverse, const u64* in, u64* out, hipStream_t s) {
  const size_t total = pl->n * pl->batch;
  u64* work = out;
  if (in == out) work = pl->d_tmp;  // bit-reversal is not in-place safe
  FIELD_DISPATCH(pl->field, {
    hipLaunchKernelGGL((bitrev_copy_kernel<decltype(ops)>), dim3(grid_for(total)), dim3(256), 0, s, ops, in, work,
                       (int)pl->log2n, total);
    for (int st = 0; st < (int)pl->log2n; st++) {
      const bool last = st == (int)pl->log2n - 1;
      hipLaunchKernelGGL((radix2_stage_kernel<decltype(ops)>), dim3(grid_for(total / 2)), dim3(256), 0, s, ops, work,
                         inverse ? pl->d_wtab_i : pl->d_wtab_f, (int)pl->log2n, st, total / 2,
                         (inverse && last) ? pl->n_inv : (u64)1);
    }
  });
  HIPCHK(hipGetLastError());
  if (work != out) HIPCHK(hipMemcpyAsync(out, work, total * 8, hipMemcpyDeviceToDevice, s));
  return RONK_OK;
}

static int transform_dev(ronk_plan* pl, bool inverse, const u64* in, const u64* in2, u64* out, hipStream_t s,
                         u64 in_valid = ~(u64)0, u64 out_valid = ~(u64)0) {
  if (!pl || !in || !out) return RONK_ERR_INVALID;
  if (pl->fast) {
    // in == out is safe: a single-pass plan rewrites exactly the tile it read; multi-pass plans
    // read BUF_IN only in pass 1 and write BUF_OUT only in the last pass.
    return (inverse ? pl->inv : pl->fwd).run(in, in2, out, pl->d_tmp, s, in_valid, out_valid);
  }
  if (in2 || in_valid != ~(u64)0 || out_valid != ~(u64)0) return RONK_ERR_UNSUPPORTED;
  if (pl->log2n == 0) {  // n = 1: fft/ifft are the identity (the recursion returns at n <= 1)
    if (in != out) HIPCHK(hipMemcpyAsync(out, in, pl->batch * 8, hipMemcpyDeviceToDevice, s));
    return RONK_OK;
  }
  return generic_transform(pl, inverse, in, out, s);
}
// Batched Message::encode::<N> on device (codes/reed_solomon.rs:42-52): the y-coordinates of `batch` codewords,
// ys[b][i] = message_b(omega_N^i), from compact messages msgs[b][0..k).  Multi-pass Goldilocks plans read the
// messages in place with implicit zero padding; other plans pad into d_ys first and transform in place.
extern "C" int ronk_rs_encode_batch_dev(ronk_plan* pl, const uint64_t* d_msgs, size_t k, uint64_t* d_ys, void* st) {
  if (!pl || !d_msgs || !d_ys || k == 0) return RONK_ERR_INVALID;
  if (k > pl->n) return RONK_ERR_INDEX;   // assert_ge::<N, K>()
  hipStream_t s = (hipStream_t)st;
  if (pl->fast && pl->fwd.pd.passes.size() > 1)
    return pl->fwd.run(d_msgs, nullptr, d_ys, pl->d_tmp, s, (u64)k, ~(u64)0, (u64)k);
  const size_t total = pl->n * pl->batch;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(total)), dim3(256), 0, s, d_msgs, k, d_ys, (size_t)pl->n, total);
  HIPCHK(hipGetLastError());
  return transform_dev(pl, false, d_ys, nullptr, d_ys, s);
}
extern "C" int ronk_ntt_forward_dev(ronk_plan* pl, const uint64_t* in, uint64_t* out, void* st) {
  return transform_dev(pl, false, in, nullptr, out, (hipStream_t)st);
}
extern "C" int ronk_ntt_inverse_dev(ronk_plan* pl, const uint64_t* in, uint64_t* out, void* st) {
  return transform_dev(pl, true, in, nullptr, out, (hipStream_t)st);
}

static int ensure_stage(ronk_plan* pl) {
  const size_t bytes = pl->n * pl->batch * 8;
  if (!pl->d_stage_in) HIPCHK(hipMalloc((void**)&pl->d_stage_in, bytes));
  if (!pl->d_stage_out) HIPCHK(hipMalloc((void**)&pl->d_stage_out, bytes));
  return RONK_OK;
}
static int transform_host(ronk_plan* pl, bool inverse, const u64* in, u64* out) {
  if (!pl || !in || !out) return RONK_ERR_INVALID;
  std::lock_guard<std::mutex> lk(pl->mu);
  HIPCHK(hipSetDevice(pl->device));
  RCHK(ensure_stage(pl));
  const size_t bytes = pl->n * pl->batch * 8;
  HIPCHK(hipMemcpy(pl->d_stage_in, in, bytes, hipMemcpyHostToDevice));
  RCHK(transform_dev(pl, inverse, pl->d_stage_in, nullptr, pl->d_stage_out, 0));
  HIPCHK(hipMemcpy(out, pl->d_stage_out, bytes, hipMemcpyDeviceToHost));
  return RONK_OK;
}

static int lagrange_nodes_dev(const FieldCtx& f, u64 w, u64* d_nodes, size_t n, hipStream_t s) {
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((power_table_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, s, ops, w,
                                        d_nodes, n); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_lagrange_nodes(uint64_t p, uint64_t g, uint64_t* nodes, size_t n) {
  if (!nodes || n == 0) return RONK_ERR_INVALID;
  u64 w;
  RCHK(ronk_root_of_unity(p, g % p, n, &w));   // assert_eq!((F::ORDER - 1) % n, 0), mod.rs:361
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  DevBuf d;
  RCHK(d.alloc(n * 8));
  RCHK(lagrange_nodes_dev(f, w, d.u(), n, 0));
  HIPCHK(hipMemcpy(nodes, d.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

extern "C" int ronk_ntt_forward(ronk_plan* pl, const uint64_t* in, uint64_t* out, uint64_t* nodes) {
  RCHK(transform_host(pl, false, in, out));
  if (nodes) RCHK(ronk_lagrange_nodes(pl->p, pl->g, nodes, pl->n));
  return RONK_OK;
}
extern "C" int ronk_ntt_inverse(ronk_plan* pl, const uint64_t* in, uint64_t* out) {
  return transform_host(pl, true, in, out);
}

extern "C" int ronk_plan_time_passes(ronk_plan* pl, const uint64_t* d_in, uint64_t* d_out, int inverse, int iters,
                                     float* ms, void* st) {
  if (!pl || !d_in || !d_out || !ms || iters < 1) return RONK_ERR_INVALID;
  if (!pl->fast) return RONK_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)st;
  const CompiledPlan& cp = inverse ? pl->inv : pl->fwd;
  const size_t np = cp.pd.passes.size();
  std::vector<hipEvent_t> ev(np + 1);
  for (auto& e : ev) HIPCHK(hipEventCreate(&e));
  std::vector<double> acc(np, 0.0);
  for (int it = 0; it < iters; it++) {
    HIPCHK(hipEventRecord(ev[0], s));
    for (size_t i = 0; i < np; i++) {
      RCHK(cp.launch(i, d_in, nullptr, d_out, pl->d_tmp, s));
      HIPCHK(hipEventRecord(ev[i + 1], s));
    }
    HIPCHK(hipEventSynchronize(ev[np]));
    for (size_t i = 0; i < np; i++) {
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      acc[i] += t;
    }
  }
  for (size_t i = 0; i < np; i++) ms[i] = (float)(acc[i] / iters);
  for (auto& e : ev) (void)hipEventDestroy(e);
  return RONK_OK;
}

// ------------------------------------------------------------------------------ plan cache
// One-shot entry points (ronk_fft/ronk_ifft/ronk_dft/ronk_poly_mul*) reuse plans -- twiddle tables and
// scratch stay resident in HBM -- through a small LRU cache guarded by one lock (the reference is
// stateless; `cargo test` calls in from many threads).  An entry also owns two padded operand
// buffers for the multiply; an event orders successive uses of an entry across streams.
static bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static int ilog2(size_t n) { int k = 0; while (((size_t)1 << k) < n) k++; return k; }

struct CacheEntry {
  ronk_plan* pl = nullptr;
  u64 *fa = nullptr, *fb = nullptr;  // poly_mul operands, n elements each (lazy)
  hipEvent_t done = nullptr;
  uint64_t stamp = 0;
};
static std::mutex g_cache_mu;
static std::vector<CacheEntry> g_cache;
static uint64_t g_cache_clock = 0;

static void cache_entry_free(CacheEntry& e) {
  if (e.pl) ronk_plan_destroy(e.pl);
  if (e.fa) (void)hipFree(e.fa);
  if (e.fb) (void)hipFree(e.fb);
  if (e.done) (void)hipEventDestroy(e.done);
  e = CacheEntry();
}
// caller holds g_cache_mu
static int cache_get(u64 p, u64 g, u32 log2n, CacheEntry** out) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  for (auto& e : g_cache)
    if (e.pl && e.pl->p == p && e.pl->g == g % p && e.pl->log2n == log2n && e.pl->batch == 1 && e.pl->device == dev) {
      e.stamp = ++g_cache_clock;
      *out = &e;
      return RONK_OK;
    }
  if (g_cache.size() >= 8) {  // evict the least recently used entry (its work must have drained)
    size_t lru = 0;
    for (size_t i = 1; i < g_cache.size(); i++) if (g_cache[i].stamp < g_cache[lru].stamp) lru = i;
    if (g_cache[lru].done) (void)hipEventSynchronize(g_cache[lru].done);
    cache_entry_free(g_cache[lru]);
    g_cache.erase(g_cache.begin() + lru);
  }
  CacheEntry e;
  RCHK(ronk_plan_create(&e.pl, p, g, log2n, 1, -1));
  hipError_t he = hipEventCreateWithFlags(&e.done, hipEventDisableTiming);
  if (he != hipSuccess) { cache_entry_free(e); return hip_fail(he, "hipEventCreate"); }
  e.stamp = ++g_cache_clock;
  g_cache.push_back(e);
  *out = &g_cache.back();
  return RONK_OK;
}

// Polynomial::fft / ifft one-shot forms (polynomial/mod.rs:273-292, :430-453) on host pointers
static int fft_oneshot(bool inverse, u64 p, u64 g, const u64* in, u64* out, u64* nodes, size_t n) {
  if (!in || !out || n == 0) return RONK_ERR_INVALID;
  if (!is_pow2(n)) return RONK_ERR_NOT_POW2;  // `[(); D.is_power_of_two() as usize - 1]:`
  RCHK(ronk_check_prime(p));
  if (p < 2) return RONK_ERR_INVALID;
  if ((p - 1) % n != 0) return RONK_ERR_NO_ROOT;
  RCHK(need_device());
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    CacheEntry* e = nullptr;
    RCHK(cache_get(p, g, (u32)ilog2(n), &e));
    RCHK(inverse ? ronk_ntt_inverse(e->pl, in, out) : ronk_ntt_forward(e->pl, in, out, nullptr));
  }
  if (nodes) RCHK(ronk_lagrange_nodes(p, g, nodes, n));
  return RONK_OK;
}
extern "C" int ronk_fft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, uint64_t* nodes, size_t n) {
  return fft_oneshot(false, p, g, in, out, nodes, n);
}
extern "C" int ronk_ifft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n) {
  return fft_oneshot(true, p, g, in, out, nullptr, n);
}

// ------------------------------------------------------------------------------ dft (any n | p-1)
static int conv_dev(u64 p, u64 g, int k, const u64* d_a, size_t d, const u64* d_b, size_t d2, u64* d_out, size_t out_len,
                    hipStream_t s);

// Bluestein's chirp-z for n that is not a power of two (Goldilocks): with C(m) = m(m-1)/2, j*k = C(j+k) - C(j) - C(k), so
//   X_k = w^-C(k) * sum_j (x_j w^-C(j)) * w^C(j+k)
// is a correlation, computed as ONE cyclic convolution of size 2^ceil(log2(2n-1)) on the fast NTT path.  Only w = omega_n
// itself is needed (no square root of it).  Same values as Polynomial::dft (polynomial/mod.rs:240-258), O(n log n).
static int bluestein_dev(u64 p, u64 g, u64 w, const u64* d_x, u64* d_out, size_t n, hipStream_t s) {
  FieldCtx f;
  RCHK(make_field(p, &f));
  const size_t lb = 2 * n - 1;
  int k = ilog2(lb);
  if (k < 4) k = 4;
  DevBuf dT, da, db, dc;
  RCHK(dT.alloc(n * 8)); RCHK(da.alloc(n * 8)); RCHK(db.alloc(lb * 8)); RCHK(dc.alloc(lb * 8));
  RCHK(lagrange_nodes_dev(f, w, dT.u(), n, s));
  hipLaunchKernelGGL(bluestein_pre_kernel, dim3(grid_for(lb)), dim3(256), 0, s, d_x, dT.u(), n, da.u(), db.u());
  HIPCHK(hipGetLastError());
  RCHK(conv_dev(p, g, k, da.u(), n, db.u(), lb, dc.u(), lb, s));   // cyclic wrap-around only reaches indices < n-1
  hipLaunchKernelGGL(bluestein_post_kernel, dim3(grid_for(n)), dim3(256), 0, s, dc.u(), dT.u(), n, d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));   // the temporaries above are freed on return
  return RONK_OK;
}

extern "C" int ronk_dft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n) {
  if (!in || !out || n == 0) return RONK_ERR_INVALID;
  u64 w;
  RCHK(ronk_root_of_unity(p, g % p, n, &w));
  RCHK(need_device());
  if (is_pow2(n) && p == RONK_GOLDILOCKS_P && g % p == RONK_GOLDILOCKS_G && n >= 16)
    return fft_oneshot(false, p, g, in, out, nullptr, n);
  const bool chirp = p == RONK_GOLDILOCKS_P && g % p == RONK_GOLDILOCKS_G && n >= 512 && n <= ((size_t)1 << 29);
  if (!chirp && n > ((size_t)1 << 16)) return RONK_ERR_UNSUPPORTED;
  FieldCtx f;
  RCHK(make_field(p, &f));
  DevBuf di, dout;
  RCHK(di.alloc(n * 8)); RCHK(dout.alloc(n * 8));
  HIPCHK(hipMemcpy(di.p, in, n * 8, hipMemcpyHostToDevice));
  if (chirp) {
    RCHK(bluestein_dev(p, g % p, w, di.u(), dout.u(), n, 0));
    HIPCHK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
    return RONK_OK;
  }
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((dft_naive_kernel<decltype(ops)>), dim3((u32)((n + 255) / 256)), dim3(256), 0, 0,
                                        ops, di.u(), dout.u(), n, w); });
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

// ------------------------------------------------------------------------------ polynomial multiply
// Goldilocks: pad both to N = 2^k >= d + d2 - 1, NTT(a), then the inverse plan's first pass loads
// NTT(a) * NTT(b) (fused pointwise product) -- 3 transforms, 48*N algorithmic bytes.
extern "C" int ronk_poly_mul_dev(uint64_t p, uint64_t g, const uint64_t* d_a, size_t d, const uint64_t* d_b, size_t d2,
                                 uint64_t* d_out, void* st) {
  if (!d_a || !d_b || !d_out || d == 0 || d2 == 0) return RONK_ERR_INVALID;
  hipStream_t s = (hipStream_t)st;
  const size_t m = d + d2 - 1;
  FieldCtx f;
  RCHK(make_field(p, &f));
  const bool fast = (p == RONK_GOLDILOCKS_P && g % p == RONK_GOLDILOCKS_G && m > 64);
  if (!fast) {
    if ((double)d * (double)d2 > 1.2e12) return RONK_ERR_UNSUPPORTED;
    FIELD_DISPATCH(f, { hipLaunchKernelGGL((poly_mul_schoolbook_kernel<decltype(ops)>), dim3(grid_for(m)), dim3(256), 0, s,
                                          ops, d_a, d, d_b, d2, d_out); });
    HIPCHK(hipGetLastError());
    return RONK_OK;
  }
  int k = ilog2(m);
  if (k < 4) k = 4;
  return conv_dev(p, g, k, d_a, d, d_b, d2, d_out, m, s);
}
// cyclic convolution of size N = 2^k of a (d entries) and b (d2 entries), the first out_len entries stored
static int conv_dev(u64 p, u64 g, int k, const u64* d_a, size_t d, const u64* d_b, size_t d2, u64* d_out, size_t out_len,
                    hipStream_t s) {
  const size_t N = (size_t)1 << k;
  const size_t m = out_len;
  RCHK(need_device());
  std::lock_guard<std::mutex> lk(g_cache_mu);
  CacheEntry* e = nullptr;
  RCHK(cache_get(p, g, (u32)k, &e));
  if (!e->fa) HIPCHK(hipMalloc((void**)&e->fa, N * 8));
  if (!e->fb) HIPCHK(hipMalloc((void**)&e->fb, N * 8));
  ronk_plan* pl = e->pl;
  HIPCHK(hipStreamWaitEvent(s, e->done, 0));                                // previous use of this entry's scratch
  // From<[F;N]> zero padding (mod.rs:503-515) is implicit: the forward transforms read the operands in place and
  // treat indices >= d (d2) as ZERO; the inverse loads NTT(a)*NTT(b) (pointwise product fused into the load) and
  // stores only the d + d2 - 1 product coefficients, straight into the caller's buffer.  No memset, no copy.
  RCHK(transform_dev(pl, false, d_a, nullptr, e->fa, s, (u64)d));
  RCHK(transform_dev(pl, false, d_b, nullptr, e->fb, s, (u64)d2));
  RCHK(transform_dev(pl, true, e->fa, e->fb, d_out, s, ~(u64)0, (u64)m));
  HIPCHK(hipEventRecord(e->done, s));
  return RONK_OK;
}
extern "C" int ronk_poly_mul(uint64_t p, uint64_t g, const uint64_t* a, size_t d, const uint64_t* b, size_t d2,
                             uint64_t* out) {
  if (!a || !b || !out || d == 0 || d2 == 0) return RONK_ERR_INVALID;
  RCHK(need_device());
  const size_t m = d + d2 - 1;
  DevBuf da, db, dout;
  RCHK(da.alloc(d * 8)); RCHK(db.alloc(d2 * 8)); RCHK(dout.alloc(m * 8));
  HIPCHK(hipMemcpy(da.p, a, d * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db.p, b, d2 * 8, hipMemcpyHostToDevice));
  RCHK(ronk_poly_mul_dev(p, g, da.u(), d, db.u(), d2, dout.u(), 0));
  HIPCHK(hipMemcpy(out, dout.p, m * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

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
    hipLaunchKernelGGL((chunk_carry_kernel<decltype(ops)>), dim3(1), dim3(1024), 0, s, ops, H, nchunks, tab, carry, total);
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

// ------------------------------------------------------------------------------ multi-GPU four-step
struct ronk_dist_plan {
  DistShape sh;
  CompiledPlan p1, p2;
  u64* d_tmp = nullptr;  // n / world elements
  int device;
};
extern "C" int ronk_dist_plan_destroy(ronk_dist_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  pl->p1.release(); pl->p2.release();
  if (pl->d_tmp) (void)hipFree(pl->d_tmp);
  delete pl;
  return RONK_OK;
}
extern "C" int ronk_dist_plan_create(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world, int device) {
  if (!out || world < 1 || rank < 0 || rank >= world) return RONK_ERR_INVALID;
  *out = nullptr;
  if (log2n > 32) return RONK_ERR_NO_ROOT;  // 2-adicity of p - 1 is 32
  DistShape sh;
  if (!dist_shape((int)log2n, world, &sh)) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  if (device >= 0) HIPCHK(hipSetDevice(device));
  else HIPCHK(hipGetDevice(&device));
  ronk_dist_plan* pl = new ronk_dist_plan();
  pl->sh = sh; pl->device = device;
  int rc = pl->p1.compile(build_dist_phase1((int)log2n, inverse != 0, rank, world));
  if (!rc) rc = pl->p2.compile(build_dist_phase2((int)log2n, inverse != 0, rank, world));
  if (!rc) {
    hipError_t e = hipMalloc((void**)&pl->d_tmp, (sh.n / sh.W) * 8);
    if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(scratch)");
  }
  if (rc) { ronk_dist_plan_destroy(pl); return rc; }
  *out = pl;
  return RONK_OK;
}
extern "C" int ronk_dist_phase1_dev(ronk_dist_plan* pl, const uint64_t* d_in, uint64_t* d_send, void* st) {
  if (!pl || !d_in || !d_send || d_in == d_send) return RONK_ERR_INVALID;
  return pl->p1.run(d_in, nullptr, d_send, pl->d_tmp, (hipStream_t)st);
}
extern "C" int ronk_dist_phase2_dev(ronk_dist_plan* pl, const uint64_t* d_recv, uint64_t* d_out, void* st) {
  if (!pl || !d_recv || !d_out || d_recv == d_out) return RONK_ERR_INVALID;
  return pl->p2.run(d_recv, nullptr, d_out, pl->d_tmp, (hipStream_t)st);
}

// ------------------------------------------------------------------------------ device helpers
extern "C" int ronk_dev_alloc(void** ptr, size_t bytes) {
  if (!ptr) return RONK_ERR_INVALID;
  RCHK(need_device());
  HIPCHK(hipMalloc(ptr, bytes ? bytes : 8));
  return RONK_OK;
}
extern "C" int ronk_dev_free(void* ptr) { HIPCHK(hipFree(ptr)); return RONK_OK; }
extern "C" int ronk_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return RONK_OK;
}
extern "C" int ronk_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return RONK_OK;
}
extern "C" int ronk_dev_sync(void) { HIPCHK(hipDeviceSynchronize()); return RONK_OK; }
