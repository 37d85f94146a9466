// ronk_core.hip -- C ABI of libronk_ntt.so (include/ronk_ntt.h), part 1: errors, device query, the host-side integer
// logic of the field traits, element-wise vector operations, raw device helpers.
// No CPU compute path exists: every entry point that would compute returns RONK_ERR_NO_DEVICE without a HIP device.
#include "runtime.h"

// ------------------------------------------------------------------------------------ errors
thread_local std::string g_hip_err;

int hip_fail(hipError_t e, const char* what) {
  g_hip_err = std::string(what) + ": " + hipGetErrorString(e);
  return RONK_ERR_HIP;
}

// an RCCL call failed (or librccl could not be loaded: code < 0, `what` is the whole message); the text goes where the HIP
// messages go (ronk_last_hip_error)
int rccl_fail(int code, const char* what) {
  g_hip_err = code < 0 ? std::string(what) : std::string(what) + ": ncclResult_t " + std::to_string(code);
  return RONK_ERR_RCCL;
}

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
    case RONK_ERR_RCCL: return "RCCL error";
    case RONK_ERR_NOT_RESIDUE: return "Element is not a quadratic residue";
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
int need_device() {
  int n = 0;
  ronk_device_count(&n);
  return n > 0 ? RONK_OK : RONK_ERR_NO_DEVICE;
}

// --------------------------------------------------------------- host integer logic (no compute path)
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

// device-resident Neg / pow / inverse over arrays; in == out allowed.  ronk_vec_inv_dev: *d_status (may be NULL) is set
// non-zero when an element is ZERO (Field::inverse returns None, prime/mod.rs:62-72; unwrap = RONK_ERR_ZERO_INVERSE).
extern "C" int ronk_vec_neg_dev(uint64_t p, const uint64_t* d_a, uint64_t* d_out, size_t n, void* st) {
  if (!d_a || !d_out) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  if (n) FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_neg_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)st,
                                               ops, d_a, d_out, n); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_vec_pow_dev(uint64_t p, const uint64_t* d_a, uint64_t e, uint64_t* d_out, size_t n, void* st) {
  if (!d_a || !d_out) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  if (n) FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_pow_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)st,
                                               ops, d_a, e, d_out, n, (int*)nullptr); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_vec_inv_dev(uint64_t p, const uint64_t* d_a, uint64_t* d_out, size_t n, int* d_status, void* st) {
  if (!d_a || !d_out || p < 2) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  if (n) FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_pow_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)st,
                                               ops, d_a, p - 2, d_out, n, d_status); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
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

// FieldExt (src/algebra/field/mod.rs:79-84; prime/mod.rs:142-226) over arrays.  The constants of the prime -- p - 1 = q 2^s and
// c0 = z^q for the first z >= 2 that fails the criterion, exactly the reference's search -- are host integers.
static int sqrt_consts(u64 p, u64* q, u32* s, u64* c0) {
  if (p == 2) return RONK_ERR_UNSUPPORTED;   // the reference's non-residue search never ends over F_2
  u64 qq = p - 1; u32 ss = 0;
  while ((qq & 1) == 0) { qq >>= 1; ss++; }
  u64 z = 2 % p;
  while (h_powmod(z, (p - 1) / 2, p) == 1) z = (z + 1) % p;
  *q = qq; *s = ss; *c0 = h_powmod(z, qq, p);
  return RONK_OK;
}
extern "C" int ronk_vec_euler_dev(uint64_t p, const uint64_t* d_a, uint64_t* d_out, size_t n, void* st) {
  if (!d_a || !d_out) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  if (n) FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_euler_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)st,
                                               ops, d_a, d_out, n); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_vec_sqrt_dev(uint64_t p, const uint64_t* d_a, uint64_t* d_r0, uint64_t* d_r1, size_t n, int* d_status, void* st) {
  if (!d_a || !d_r0 || !d_r1) return RONK_ERR_INVALID;
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  RCHK(ronk_check_prime(p));
  u64 q, c0; u32 s;
  RCHK(sqrt_consts(p, &q, &s, &c0));
  if (n) FIELD_DISPATCH(f, { hipLaunchKernelGGL((vec_sqrt_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)st,
                                               ops, d_a, d_r0, d_r1, n, q, s, c0, d_status); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_vec_euler(uint64_t p, const uint64_t* a, uint64_t* out, size_t n) {
  if (!a || !out) return RONK_ERR_INVALID;
  RCHK(need_device());
  DevBuf da;
  RCHK(da.alloc(n * 8));
  HIPCHK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice));
  RCHK(ronk_vec_euler_dev(p, da.u(), da.u(), n, nullptr));
  HIPCHK(hipMemcpy(out, da.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}
extern "C" int ronk_vec_sqrt(uint64_t p, const uint64_t* a, uint64_t* r0, uint64_t* r1, size_t n) {
  if (!a || !r0 || !r1) return RONK_ERR_INVALID;
  RCHK(need_device());
  DevBuf da, db, dflag;
  RCHK(da.alloc(n * 8)); RCHK(db.alloc(n * 8)); RCHK(dflag.alloc(4));
  HIPCHK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 4));
  RCHK(ronk_vec_sqrt_dev(p, da.u(), da.u(), db.u(), n, (int*)dflag.p, nullptr));
  int hflag = 0;
  HIPCHK(hipMemcpy(&hflag, dflag.p, 4, hipMemcpyDeviceToHost));
  if (hflag) return RONK_ERR_NOT_RESIDUE;
  HIPCHK(hipMemcpy(r0, da.p, n * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(r1, db.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
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
extern "C" int ronk_set_device(int device) {
  int n = 0;
  ronk_device_count(&n);
  if (n <= 0) return RONK_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return RONK_ERR_INVALID;
  HIPCHK(hipSetDevice(device));
  return RONK_OK;
}
extern "C" int ronk_get_device(int* device) {
  if (!device) return RONK_ERR_INVALID;
  RCHK(need_device());
  HIPCHK(hipGetDevice(device));
  return RONK_OK;
}

