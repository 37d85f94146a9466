// bn254_host.cpp -- TEST INFRASTRUCTURE: csrc/bn254.h compiled for the host behind a C interface, so that
// tests/test_bn254_host.py can check the field and group arithmetic (the same source the kernels and the MSM's host tail
// use) against oracle/bn254.py on the CPU-only container.  Never linked into the product library.
#include <vector>

#include "../../ronkathon_amd/csrc/bn254.h"
#include "../../ronkathon_amd/csrc/bn254_fr.h"
#include "../../ronkathon_amd/csrc/msm_common.h"

using namespace bn254;

extern "C" {
// standard form in / out (4 x u64 little endian)
void h_fp_mul(const u64* a, const u64* b, u64* out) {
  fp_store(out, fp_from_mont(fp_mul(fp_to_mont(fp_load(a)), fp_to_mont(fp_load(b)))));
}
void h_fp_add(const u64* a, const u64* b, u64* out) { fp_store(out, fp_canon(fp_add(fp_load(a), fp_load(b)))); }
void h_fp_sub(const u64* a, const u64* b, u64* out) { fp_store(out, fp_canon(fp_sub(fp_load(a), fp_load(b)))); }
void h_fp_inv(const u64* a, u64* out) { fp_store(out, fp_from_mont(fp_inv(fp_to_mont(fp_load(a))))); }
int h_fp_geq_p(const u64* a) { return fp_geq_p(fp_load(a)) ? 1 : 0; }

static Affine load_affine(const u64* p) {
  Affine a;
  a.x = fp_load(p); a.y = fp_load(p + 4);
  if (!(fp_is_zero_exact(a.x) && fp_is_zero_exact(a.y))) { a.x = fp_to_mont(a.x); a.y = fp_to_mont(a.y); }
  return a;
}
int h_on_curve(const u64* p) { return affine_on_curve(load_affine(p)) ? 1 : 0; }
// out = (p + q) through madd (q affine), then the general add and dbl paths: mode 0 madd, 1 add of two XYZZ with non-unit
// ZZ (both scaled first), 2 dbl of p (q ignored)
void h_point_op(const u64* p, const u64* q, int neg_q, int mode, u64* out) {
  const Affine ap = load_affine(p), aq = load_affine(q);
  Xyzz acc = xyzz_inf();
  xyzz_madd(acc, ap, false);
  if (mode == 0) {
    xyzz_madd(acc, aq, neg_q != 0);
  } else if (mode == 1) {
    // rescale both operands to non-trivial ZZ: (X l^2, Y l^3, ZZ l^2, ZZZ l^3) with l = 5 and 7
    Xyzz b = xyzz_inf();
    xyzz_madd(b, aq, neg_q != 0);
    Fp five = fp_zero(); five.l[0] = 5; five = fp_to_mont(five);
    Fp seven = fp_zero(); seven.l[0] = 7; seven = fp_to_mont(seven);
    auto scale = [](Xyzz& v, const Fp& l) {
      if (xyzz_is_inf(v)) return;
      const Fp l2 = fp_sqr(l), l3 = fp_mul(l2, l);
      v.X = fp_mul(v.X, l2); v.Y = fp_mul(v.Y, l3); v.ZZ = fp_mul(v.ZZ, l2); v.ZZZ = fp_mul(v.ZZZ, l3);
    };
    scale(acc, five); scale(b, seven);
    acc = xyzz_add(acc, b);
  } else {
    acc = xyzz_dbl(acc);
  }
  xyzz_store_affine(acc, out);
}
// k * p by double-and-add on XYZZ (k: 4 x u64)
void h_scalar_mul(const u64* p, const u64* k, u64* out) {
  const Affine ap = load_affine(p);
  Xyzz acc = xyzz_inf();
  for (int i = 255; i >= 0; i--) {
    acc = xyzz_dbl(acc);
    if ((k[i >> 6] >> (i & 63)) & 1) xyzz_madd(acc, ap, false);
  }
  xyzz_store_affine(acc, out);
}

// The bucket-method pipeline of msm_kernels.h restated sequentially on the host with the SAME shared pieces (msm_digit,
// the carry-bit mask, the bit-plane reduction, msm_host_tail): window bits c, W = ceil(257 / c) windows, 2^(c-1) buckets.
// Returns 0, or 1 if the per-window digit from the carry mask ever differs from the rippled one.
int h_msm_pipeline(const u64* points, const u64* scalars, u32 n, u32 c, u64* out) {
  ronk::MsmShape sh;
  sh.n = n; sh.c = c; sh.W = (257 + c - 1) / c; sh.NB = 1u << (c - 1);
  std::vector<Xyzz> buckets((size_t)sh.W * sh.NB, xyzz_inf());
  int bad = 0;
  for (u32 i = 0; i < n; i++) {
    const Affine pt = load_affine(points + (size_t)i * 8);
    const u64* k = scalars + (size_t)i * 4;
    u32 carry = 0;
    u64 mask = 0;
    for (u32 w = 0; w < sh.W; w++) { mask |= (u64)carry << w; (void)ronk::msm_digit(k, w, c, &carry); }
    if (carry) bad = 1;                                   // W*c >= 257: nothing is carried out of the top window
    carry = 0;
    for (u32 w = 0; w < sh.W; w++) {
      u32 cm = (u32)(mask >> w) & 1;
      const int dm = ronk::msm_digit(k, w, c, &cm);       // what the sort kernels compute (window on its own)
      const int d = ronk::msm_digit(k, w, c, &carry);     // rippled
      if (d != dm) bad = 1;
      if (d == 0) continue;
      const u32 b = (u32)(d < 0 ? -d : d) - 1;
      if (b >= sh.NB) { bad = 1; continue; }
      xyzz_madd(buckets[(size_t)w * sh.NB + b], pt, d < 0);
    }
  }
  std::vector<Xyzz> rows((size_t)sh.W * c, xyzz_inf());
  for (u32 w = 0; w < sh.W; w++)
    for (u32 kk = 0; kk < c; kk++)
      for (u32 b = 0; b < sh.NB; b++)
        if (((b + 1) >> kk) & 1) rows[(size_t)w * c + kk] = xyzz_add(rows[(size_t)w * c + kk], buckets[(size_t)w * sh.NB + b]);
  ronk::msm_host_tail(sh, rows.data(), out);
  return bad;
}
}

// ---- the scalar field (bn254_fr.h): standard form in / out; the second operand of a product goes through Montgomery form
// exactly as the kernels use it (fr_mul(x, wR) = x * w)
extern "C" {
void h_fr_mul(const u64* a, const u64* b, u64* out) { fr_store(out, fr_mul(fr_load(a), fr_to_mont(fr_canon(fr_load(b))))); }
void h_fr_add(const u64* a, const u64* b, u64* out) { fr_store(out, fr_add(fr_load(a), fr_load(b))); }
void h_fr_sub(const u64* a, const u64* b, u64* out) { fr_store(out, fr_sub(fr_load(a), fr_load(b))); }
void h_fr_canon(const u64* a, u64* out) { fr_store(out, fr_canon(fr_load(a))); }
// powers z^(4 k), k = 0 .. 255 and z^1024 as the scan's multiplier table builds them (Montgomery x Montgomery products),
// returned in standard form: out = 257 x 4 words
void h_fr_powers(const u64* z, u64* out) {
  const Fr mu = fr_to_mont(fr_canon(fr_load(z)));
  Fr m4 = fr_const_one_mont();
  for (int i = 0; i < 4; i++) m4 = fr_mul(m4, mu);
  Fr x = fr_const_one_mont(), one = fr_zero();
  one.l[0] = 1;
  for (int k = 0; k <= 256; k++) { fr_store(out + 4 * k, fr_mul(x, one)); x = fr_mul(x, m4); }
}
}
