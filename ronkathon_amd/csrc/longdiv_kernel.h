// longdiv_kernel.h -- quotient_and_remainder (src/polynomial/mod.rs:170-225) as ONE workgroup that follows the reference's loop
// statement by statement: the loop guard compares the remainder's TRIMMED length with the divisor's UNTRIMMED length d2, the
// update walks all d2 divisor coefficients, a zero divisor or an out-of-range update is the reference's panic (status -6), a
// zero leading inverse cannot occur for a non-zero divisor.  Any prime (the field comes in as `ops`), any divisor; d * d2
// steps -- the O(n log n) forms (ronk_callers.hip: Newton inversion, the linear-divisor scans) take the shapes where that matters,
// this kernel keeps every shape whose RESULT depends on the reference's control flow (ragged divisors, short dividends, panics).
//
// The whole call is this one launch: rem[] (d entries) is filled here from the dividend a[] (which may BE rem), quot[] and
// *status are zeroed here -- a captured hipGraph of the call holds no memset / memcpy node (a memset node in front of a memcpy
// node of more than 16 KiB came back as 0xFCFCFCFC from the second replay on, ROCm 7.0.2: profiles/r05_capture_division.txt).
//
// The body is written against a small context (work-item id, workgroup size, barrier, two shared words, a shared max) so that
// the CPU suite runs the very same code on fibers against the CPU restatement of the reference (tests/emu/emu_longdiv.cpp); the kernel
// is at the end of the file.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ronk {

typedef uint64_t u64;

// Ops: mul, sub, pow(a, e), order().  Ctx: tid(), nthreads(), barrier(), top() / lead() = the two shared words,
// raise(word, v) = word = max(word, v) across the workgroup.
template <class Ops, class Ctx>
#if defined(__HIPCC__)
__device__ __forceinline__
#else
inline
#endif
void poly_divrem_body(const Ops& ops, const u64* a, u64* rem, size_t d, const u64* b, size_t d2, u64* quot, int* status,
                      const Ctx& cx) {
  unsigned long long* const s_top = cx.top();   // 1 + highest non-zero index found by the scan (0 = none)
  u64* const s_s = cx.lead();
  const size_t T = cx.nthreads(), tid = cx.tid();
  if (tid == 0) *status = 0;                    // (the only later writer is this same lane)
  for (size_t i = tid; i < d; i += T) quot[i] = 0;
  if (a != rem)
    for (size_t i = tid; i < d; i += T) rem[i] = a[i];
  // divisor degree / leading coefficient
  if (tid == 0) *s_top = 0;
  cx.barrier();
  long long mine = -1;
  for (size_t i = tid; i < d2; i += T) if (b[i] != 0) mine = (long long)i;
  if (mine >= 0) cx.raise(s_top, (unsigned long long)(mine + 1));
  cx.barrier();
  const long long rhs_degree = (long long)*s_top - 1;
  cx.barrier();
  u64 cinv = 0;
  if (rhs_degree >= 0) cinv = ops.pow(b[rhs_degree], ops.order() - 2);
  size_t plen = d;
  for (;;) {
    // p_degree = rposition(!= 0) over the current (trimmed) remainder
    if (tid == 0) *s_top = 0;
    cx.barrier();
    mine = -1;
    for (size_t i = tid; i < plen; i += T) if (rem[i] != 0) mine = (long long)i;
    if (mine >= 0) cx.raise(s_top, (unsigned long long)(mine + 1));
    cx.barrier();
    const long long p_degree = (long long)*s_top - 1;
    cx.barrier();
    if (!(p_degree >= 0 && plen >= d2)) break;       // while nonzero-count > 0 && len >= rhs.len()
    if (rhs_degree < 0) { if (tid == 0) *status = -6; break; }  // rposition(..).unwrap() on zero divisor
    if (p_degree < rhs_degree) break;
    const size_t diff = (size_t)(p_degree - rhs_degree);
    if (diff + d2 > plen) { if (tid == 0) *status = -6; break; }  // p_coeffs[diff + i] out of bounds
    if (tid == 0) { *s_s = ops.mul(rem[p_degree], cinv); quot[diff] = *s_s; }
    cx.barrier();
    const u64 s = *s_s;
    for (size_t i = tid; i < d2; i += T) rem[diff + i] = ops.sub(rem[diff + i], ops.mul(b[i], s));
    cx.barrier();
    // trim_zeros: the new length is one past the highest non-zero entry (found by the next scan);
    // entries above it are already zero, so only the guard `plen >= d2` needs the trimmed value
    if (tid == 0) *s_top = 0;
    cx.barrier();
    mine = -1;
    for (size_t i = tid; i < plen; i += T) if (rem[i] != 0) mine = (long long)i;
    if (mine >= 0) cx.raise(s_top, (unsigned long long)(mine + 1));
    cx.barrier();
    plen = (size_t)*s_top;
    cx.barrier();
  }
}

}  // namespace ronk

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>

namespace ronk {

struct LongDivDevCtx {
  unsigned long long* top_;
  u64* lead_;
  __device__ __forceinline__ size_t tid() const { return threadIdx.x; }
  __device__ __forceinline__ size_t nthreads() const { return blockDim.x; }
  __device__ __forceinline__ void barrier() const { __syncthreads(); }
  __device__ __forceinline__ unsigned long long* top() const { return top_; }
  __device__ __forceinline__ u64* lead() const { return lead_; }
  __device__ __forceinline__ void raise(unsigned long long* w, unsigned long long v) const { atomicMax(w, v); }
};

template <class Ops>
__global__ void __launch_bounds__(1024) poly_divrem_kernel(Ops ops, const u64* a, u64* rem, size_t d, const u64* __restrict__ b,
                                                            size_t d2, u64* __restrict__ quot, int* status) {
  __shared__ unsigned long long s_top;
  __shared__ u64 s_s;
  LongDivDevCtx cx{&s_top, &s_s};
  poly_divrem_body(ops, a, rem, d, b, d2, quot, status, cx);
}

}  // namespace ronk
#endif
