/*
 * ronk_oracle.c -- CPU restatement of ronkathon's prime-field / polynomial hot path.
 * TEST INFRASTRUCTURE ONLY (see ronk_oracle.h).  Plain C11, gcc, unsigned __int128.
 *
 * Where the reference's literal code is asymptotically pathological (pow recursing
 * twice per level, prime/mod.rs:74-84; is_prime by trial division on every `new`,
 * prime/mod.rs:48-51) the restatement computes the SAME VALUE with the textbook
 * algorithm; where the reference's control flow decides the result (fft recursion
 * order, long-division loop conditions, the Lagrange-evaluate fold) it is followed
 * step by step so that quirks are reproduced, not fixed.
 */
#include "ronk_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ field */

/* prime/mod.rs:92-100: trial division; panics ("input is not a prime number") on a
 * proper divisor.  Note n = 0, 1 pass vacuously in the reference (loop never runs);
 * kept.  For n >= 2^40 the same predicate is decided by deterministic Miller-Rabin. */
static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)(((u128)a * b) % p); }
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t p) {
  uint64_t r = 1 % p;
  a %= p;
  while (e) {
    if (e & 1) r = mulmod(r, a, p);
    a = mulmod(a, a, p);
    e >>= 1;
  }
  return r;
}
static int miller_rabin(uint64_t n) {
  static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return 1; /* vacuous pass, as the reference */
  for (size_t i = 0; i < sizeof bases / sizeof *bases; i++) {
    if (n == bases[i]) return 1;
    if (n % bases[i] == 0) return 0;
  }
  uint64_t d = n - 1;
  int s = 0;
  while ((d & 1) == 0) { d >>= 1; s++; }
  for (size_t i = 0; i < sizeof bases / sizeof *bases; i++) {
    uint64_t x = powmod(bases[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int r = 1; r < s; r++) {
      x = mulmod(x, x, n);
      if (x == n - 1) { comp = 0; break; }
    }
    if (comp) return 0;
  }
  return 1;
}
int orc_is_prime(uint64_t n) {
  if (n >= ((uint64_t)1 << 40)) return miller_rabin(n) ? ORC_OK : ORC_PANIC_NOT_PRIME;
  for (uint64_t i = 2; i * i <= n; i++)
    if (n % i == 0) return ORC_PANIC_NOT_PRIME;
  return ORC_OK;
}

/* prime/mod.rs:48-51 */
uint64_t orc_new(uint64_t p, uint64_t v) { return v % p; }

/* prime/arithmetic.rs:3-7: (a + b) % ORDER, widened so P > 2^63 does not wrap */
uint64_t orc_add(uint64_t p, uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % p); }

/* prime/arithmetic.rs:19-28: overflowing_sub then wrapping_add(ORDER) on borrow */
uint64_t orc_sub(uint64_t p, uint64_t a, uint64_t b) {
  uint64_t diff = a - b;
  if (a < b) diff += p;
  return diff;
}

/* prime/arithmetic.rs:61-65: ZERO - self */
uint64_t orc_neg(uint64_t p, uint64_t a) { return orc_sub(p, 0, a); }

/* prime/arithmetic.rs:34-38: (a * b) % ORDER, widened */
uint64_t orc_mul(uint64_t p, uint64_t a, uint64_t b) { return mulmod(a, b, p); }

/* prime/mod.rs:74-84: value of the double recursion == a^e; pow(_,0) == ONE even for a == 0 */
uint64_t orc_pow(uint64_t p, uint64_t a, uint64_t e) {
  if (e == 0) return 1 % p;
  return powmod(a, e, p);
}

/* prime/mod.rs:62-72: None for zero, else a^(P-2) */
int orc_inverse(uint64_t p, uint64_t a, uint64_t* out) {
  if (a == 0) return ORC_PANIC_ZERO_INVERSE;
  *out = orc_pow(p, a, p - 2);
  return ORC_OK;
}

/* prime/arithmetic.rs:50-55: self * rhs.inverse().unwrap() */
int orc_div(uint64_t p, uint64_t a, uint64_t b, uint64_t* out) {
  uint64_t bi;
  int rc = orc_inverse(p, b, &bi);
  if (rc) return rc;
  *out = orc_mul(p, a, bi);
  return ORC_OK;
}

/* prime/arithmetic.rs:67-71: self - (self / rhs) * rhs */
int orc_rem(uint64_t p, uint64_t a, uint64_t b, uint64_t* out) {
  uint64_t q;
  int rc = orc_div(p, a, b, &q);
  if (rc) return rc;
  *out = orc_sub(p, a, orc_mul(p, q, b));
  return ORC_OK;
}

/* prime/mod.rs:87-90 + :110-123.  P == 2 -> ONE.  The heuristic is restated literally
 * (it is NOT a correct generator search: it returns 3 for Goldilocks, a non-generator;
 * SURVEY.md section 0.1), which is why the 64-bit field carries an explicit generator. */
int orc_find_primitive_element(uint64_t p, uint64_t* g) {
  if (p == 2) { *g = 1; return ORC_OK; }
  for (u128 i = 2; i * i <= p; i++) {
    uint64_t ii = (uint64_t)i;
    if ((p - 1) % ii == 0) {
      if (orc_pow(p, orc_new(p, ii), (p - 1) / ii) != 1) { *g = ii; return ORC_OK; }
      else if (orc_pow(p, orc_new(p, p + 1 - ii), ii) != 1) { *g = p + 1 - ii; return ORC_OK; }
    }
  }
  return ORC_PANIC_NO_GENERATOR;
}

/* field/mod.rs:70-75 */
int orc_primitive_root_of_unity(uint64_t p, uint64_t g, uint64_t n, uint64_t* out) {
  uint64_t pm1 = p - 1;
  if (n == 0) return ORC_PANIC_INDEX; /* `% 0` panics */
  if (pm1 % n != 0) return ORC_PANIC_NO_ROOT;
  *out = orc_pow(p, g, pm1 / n);
  return ORC_OK;
}

void orc_vec_add(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = orc_add(p, a[i], b[i]);
}
void orc_vec_sub(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = orc_sub(p, a[i], b[i]);
}
void orc_vec_mul(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = orc_mul(p, a[i], b[i]);
}
void orc_vec_neg(uint64_t p, const uint64_t* a, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = orc_neg(p, a[i]);
}
int orc_vec_inv(uint64_t p, const uint64_t* a, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    int rc = orc_inverse(p, a[i], &out[i]);
    if (rc) return rc;
  }
  return ORC_OK;
}
void orc_vec_pow(uint64_t p, const uint64_t* a, uint64_t e, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = orc_pow(p, a[i], e);
}

/* ------------------------------------------------------------- FieldExt (field/mod.rs:79-84) */

/* prime/mod.rs:172: self.pow((P - 1) / 2).value == 1 */
int orc_euler_criterion(uint64_t p, uint64_t a) { return orc_pow(p, a, (p - 1) / 2) == 1; }

/* prime/mod.rs:174-226, Tonelli-Shanks as written: ZERO -> (0, 0); a non-residue trips the assert; the pair is returned
 * smaller root first (`if -r < r { (-r, r) } else { (r, -r) }`).  P = 2 never leaves the reference's search for a
 * non-residue (every element passes the criterion there): reported as ORC_PANIC_NOT_RESIDUE too. */
int orc_sqrt(uint64_t p, uint64_t a, uint64_t* r0, uint64_t* r1) {
  if (a == 0) { *r0 = *r1 = 0; return ORC_OK; }
  if (p == 2 || !orc_euler_criterion(p, a)) return ORC_PANIC_NOT_RESIDUE;
  /* P - 1 = q * 2^s, q odd (the loop of lines 183-195 stops at the first power of two that does not divide P - 1) */
  uint64_t q = p - 1, s = 0;
  while ((q & 1) == 0) { q >>= 1; s++; }
  uint64_t z = orc_new(p, 2);
  while (orc_euler_criterion(p, z)) z = orc_add(p, z, 1);
  uint64_t m = s, c = orc_pow(p, z, q), t = orc_pow(p, a, q), r = orc_pow(p, a, (q + 1) / 2);
  for (;;) {
    if (t == 1) {
      const uint64_t nr = orc_neg(p, r);
      if (nr < r) { *r0 = nr; *r1 = r; } else { *r0 = r; *r1 = nr; }
      return ORC_OK;
    }
    uint64_t i = 1, t_pow = orc_mul(p, t, t);
    while (t_pow != 1) { t_pow = orc_mul(p, t_pow, t_pow); i++; }
    const uint64_t b = orc_pow(p, c, (uint64_t)1 << (m - i - 1));
    m = i;
    c = orc_mul(p, b, b);
    t = orc_mul(p, t, c);
    r = orc_mul(p, r, b);
  }
}
void orc_vec_euler(uint64_t p, const uint64_t* a, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = (uint64_t)orc_euler_criterion(p, a[i]);
}
int orc_vec_sqrt(uint64_t p, const uint64_t* a, uint64_t* r0, uint64_t* r1, size_t n) {
  for (size_t i = 0; i < n; i++) {
    int rc = orc_sqrt(p, a[i], &r0[i], &r1[i]);
    if (rc) return rc;
  }
  return ORC_OK;
}

/* ------------------------------------------------------------- polynomial */

/* polynomial/mod.rs:358-365: nodes[i] = w^i, w = primitive_root_of_unity(n) */
int orc_lagrange_nodes(uint64_t p, uint64_t g, uint64_t* nodes, size_t n) {
  uint64_t w;
  if (n == 0) return ORC_PANIC_INDEX;
  if ((p - 1) % n != 0) return ORC_PANIC_NO_ROOT; /* assert_eq!((F::ORDER - 1) % n, 0) */
  int rc = orc_primitive_root_of_unity(p, g, n, &w);
  if (rc) return rc;
  uint64_t x = 1 % p;
  for (size_t i = 0; i < n; i++) { nodes[i] = x; x = orc_mul(p, x, w); }
  return ORC_OK;
}

/* polynomial/mod.rs:240-258: out[i] = fold(ZERO, acc + c[j] * w^(i*j)), natural order */
int orc_dft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n) {
  uint64_t w;
  int rc = orc_primitive_root_of_unity(p, g, n, &w);
  if (rc) return rc;
  uint64_t* tmp = (uint64_t*)malloc(n * sizeof *tmp);
  uint64_t wi = 1 % p; /* w^i */
  for (size_t i = 0; i < n; i++) {
    uint64_t acc = 0, wij = 1 % p; /* w^(i*j) */
    for (size_t j = 0; j < n; j++) {
      acc = orc_add(p, acc, orc_mul(p, in[j], wij));
      wij = orc_mul(p, wij, wi);
    }
    tmp[i] = acc;
    wi = orc_mul(p, wi, w);
  }
  memcpy(out, tmp, n * sizeof *tmp);
  free(tmp);
  return ORC_OK;
}

/* polynomial/mod.rs:295-323 (and the identical :456-484): even/odd split into fresh
 * vectors, recurse with omega^2, then t = w*odd[i]; v[i] = even[i]+t; v[i+half] = even[i]-t */
void orc_fft_recursive(uint64_t p, uint64_t* values, size_t n, uint64_t omega) {
  if (n <= 1) return;
  size_t half = n / 2;
  uint64_t* even = (uint64_t*)malloc(half * sizeof *even);
  uint64_t* odd = (uint64_t*)malloc(half * sizeof *odd);
  for (size_t i = 0; i < half; i++) { even[i] = values[2 * i]; odd[i] = values[2 * i + 1]; }
  uint64_t omega2 = orc_mul(p, omega, omega); /* omega.pow(2) */
  orc_fft_recursive(p, even, half, omega2);
  orc_fft_recursive(p, odd, half, omega2);
  uint64_t cur = 1 % p;
  for (size_t i = 0; i < half; i++) {
    uint64_t t = orc_mul(p, cur, odd[i]);
    values[i] = orc_add(p, even[i], t);
    values[i + half] = orc_sub(p, even[i], t);
    cur = orc_mul(p, cur, omega);
  }
  free(even);
  free(odd);
}

static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

/* polynomial/mod.rs:273-292 (+ Lagrange::new's assert at :361) */
int orc_fft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n) {
  if (!is_pow2(n)) return ORC_PANIC_NOT_POW2;
  uint64_t w;
  int rc = orc_primitive_root_of_unity(p, g, n, &w);
  if (rc) return rc;
  if (out != in) memmove(out, in, n * sizeof *out);
  orc_fft_recursive(p, out, n, w);
  return ORC_OK;
}

/* polynomial/mod.rs:430-453: omega = root(D).inverse().unwrap(); recurse; scale by F::from(D).inverse().unwrap() */
int orc_ifft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n) {
  if (!is_pow2(n)) return ORC_PANIC_NOT_POW2;
  uint64_t w, wi, dinv;
  int rc = orc_primitive_root_of_unity(p, g, n, &w);
  if (rc) return rc;
  if ((rc = orc_inverse(p, w, &wi))) return rc;
  if (out != in) memmove(out, in, n * sizeof *out);
  orc_fft_recursive(p, out, n, wi);
  if ((rc = orc_inverse(p, orc_new(p, (uint64_t)n), &dinv))) return rc;
  for (size_t i = 0; i < n; i++) out[i] = orc_mul(p, out[i], dinv);
  return ORC_OK;
}

/* polynomial/arithmetic.rs:16-35: zip(lhs, rhs.chain(repeat(ZERO))).take(D) */
void orc_poly_add(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out) {
  for (size_t i = 0; i < d; i++) out[i] = orc_add(p, a[i], i < d2 ? b[i] : 0);
}
/* polynomial/arithmetic.rs:49-68 */
void orc_poly_sub(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out) {
  for (size_t i = 0; i < d; i++) out[i] = orc_sub(p, a[i], i < d2 ? b[i] : 0);
}
/* polynomial/arithmetic.rs:77-95 */
void orc_poly_neg(uint64_t p, const uint64_t* a, size_t d, uint64_t* out) {
  for (size_t i = 0; i < d; i++) out[i] = orc_neg(p, a[i]);
}
/* polynomial/arithmetic.rs:97-119: c[i+j] += a[i]*b[j]; D+D2-1 outputs */
void orc_poly_mul(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out) {
  size_t m = d + d2 - 1;
  uint64_t* c = (uint64_t*)calloc(m ? m : 1, sizeof *c);
  for (size_t i = 0; i < d; i++)
    for (size_t j = 0; j < d2; j++) c[i + j] = orc_add(p, c[i + j], orc_mul(p, a[i], b[j]));
  memcpy(out, c, m * sizeof *c);
  free(c);
}

/* polynomial/mod.rs:113-115: rposition(!= ZERO).unwrap_or(0) */
size_t orc_degree(const uint64_t* c, size_t d) {
  for (size_t i = d; i-- > 0;)
    if (c[i] != 0) return i;
  return 0;
}
/* polynomial/mod.rs:120-122 */
uint64_t orc_leading_coefficient(const uint64_t* c, size_t d) {
  for (size_t i = d; i-- > 0;)
    if (c[i] != 0) return c[i];
  return 0;
}
/* polynomial/mod.rs:503-515: zero-pad or truncate */
void orc_poly_from(const uint64_t* c, size_t n, uint64_t* out, size_t d) {
  size_t k = n < d ? n : d;
  memmove(out, c, k * sizeof *out);
  for (size_t i = k; i < d; i++) out[i] = 0;
}

/* polynomial/mod.rs:170-225, followed statement by statement.  `plen` is p_coeffs.len()
 * (shrinks through trim_zeros); the loop guard compares it with the divisor's
 * UNTRIMMED length d2, and the inner update walks all d2 divisor coefficients. */
int orc_poly_divrem(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2,
                    uint64_t* quot, uint64_t* rem) {
  uint64_t* q = (uint64_t*)calloc(d ? d : 1, sizeof *q);
  uint64_t* pc = (uint64_t*)malloc((d ? d : 1) * sizeof *pc);
  memcpy(pc, a, d * sizeof *pc);
  size_t plen = d;
  uint64_t c = orc_leading_coefficient(b, d2);
  int rc = ORC_OK;
  for (;;) {
    size_t nz = 0;
    for (size_t i = 0; i < plen; i++) nz += pc[i] != 0;
    if (!(nz > 0 && plen >= d2)) break;
    size_t p_degree = 0, rhs_degree = 0;
    int found = 0;
    for (size_t i = plen; i-- > 0;) if (pc[i] != 0) { p_degree = i; found = 1; break; }
    (void)found; /* nz > 0 guarantees Some */
    found = 0;
    for (size_t i = d2; i-- > 0;) if (b[i] != 0) { rhs_degree = i; found = 1; break; }
    if (!found) { rc = ORC_PANIC_INDEX; break; } /* rposition(..).unwrap() on a zero divisor */
    if (p_degree < rhs_degree) break;
    size_t diff = p_degree - rhs_degree;
    uint64_t cinv;
    if ((rc = orc_inverse(p, c, &cinv))) break;
    uint64_t s = orc_mul(p, pc[p_degree], cinv);
    q[diff] = s;
    for (size_t i = 0; i < d2; i++) {
      if (diff + i >= plen) { rc = ORC_PANIC_INDEX; break; } /* p_coeffs[diff + i] out of bounds */
      pc[diff + i] = orc_sub(p, pc[diff + i], orc_mul(p, b[i], s));
    }
    if (rc) break;
    while (plen > 0 && pc[plen - 1] == 0) plen--; /* trim_zeros */
  }
  if (!rc) {
    memcpy(quot, q, d * sizeof *q);
    for (size_t i = 0; i < d; i++) rem[i] = i < plen ? pc[i] : 0;
  }
  free(q);
  free(pc);
  return rc;
}

/* polynomial/mod.rs:133-139: result += c_i * x.pow(i) */
uint64_t orc_poly_eval(uint64_t p, const uint64_t* c, size_t d, uint64_t x) {
  uint64_t r = 0, xi = 1 % p;
  for (size_t i = 0; i < d; i++) {
    r = orc_add(p, r, orc_mul(p, c[i], xi));
    xi = orc_mul(p, xi, x);
  }
  return r;
}

/* polynomial/mod.rs:382-415, including its fold: when a node equals x the closure's
 * `return c` REPLACES the accumulator by c_j (it does not leave evaluate()), later
 * terms keep accumulating, and the product with l(x) == 0 then yields ZERO. */
int orc_lagrange_eval(uint64_t p, const uint64_t* c, const uint64_t* nodes, size_t n, uint64_t x, uint64_t* out) {
  uint64_t* w = (uint64_t*)malloc((n ? n : 1) * sizeof *w);
  int rc = ORC_OK;
  for (size_t idx = 0; idx < n && !rc; idx++) {
    w[idx] = 1 % p;
    for (size_t m = 0; m < n; m++) {
      if (idx == m) continue;
      uint64_t t;
      if ((rc = orc_div(p, 1 % p, orc_sub(p, nodes[idx], nodes[m]), &t))) break;
      w[idx] = orc_mul(p, w[idx], t);
    }
  }
  if (!rc) {
    uint64_t l = 1 % p;
    for (size_t i = 0; i < n; i++) l = orc_mul(p, l, orc_sub(p, x, nodes[i]));
    uint64_t acc = 0;
    for (size_t j = 0; j < n; j++) {
      if (nodes[j] == x) { acc = c[j]; continue; }
      uint64_t t;
      if ((rc = orc_div(p, orc_mul(p, c[j], w[j]), orc_sub(p, x, nodes[j]), &t))) break;
      acc = orc_add(p, acc, t);
    }
    if (!rc) *out = orc_mul(p, l, acc);
  }
  free(w);
  return rc;
}

/* polynomial/mod.rs:153-157: out[i] = i >= d2 ? c[i-d2]*coeff : 0, length d+d2 */
void orc_pow_mult(uint64_t p, const uint64_t* c, size_t d, size_t d2, uint64_t coeff, uint64_t* out) {
  for (size_t i = d + d2; i-- > 0;) out[i] = i >= d2 ? orc_mul(p, c[i - d2], coeff) : 0;
}

/* codes/reed_solomon.rs:42-52: x = root.pow(i), y = polynomial.evaluate(root.pow(i)) */
int orc_rs_encode(uint64_t p, uint64_t g, const uint64_t* msg, size_t k, size_t n, uint64_t* xs, uint64_t* ys) {
  if (n < k) return ORC_PANIC_INDEX; /* assert_ge::<N, K>() */
  uint64_t w;
  int rc = orc_primitive_root_of_unity(p, g, n, &w);
  if (rc) return rc;
  uint64_t x = 1 % p;
  for (size_t i = 0; i < n; i++) {
    xs[i] = x;
    ys[i] = orc_poly_eval(p, msg, k, x);
    x = orc_mul(p, x, w);
  }
  return ORC_OK;
}

/* codes/reed_solomon.rs:54-106, Message::decode: Lagrange interpolation through the FIRST k coordinates,
 *   data[i] += (i odd ? -1 : 1) * (sum over (k-1-i)-subsets of {x_m : m != j} of their product) * y_j
 *              / prod_{m != j} (x_m - x_j)
 * The subset sum is the elementary symmetric polynomial e_(k-1-i) of the k-1 remaining nodes; it is built
 * here with the usual recurrence e_r(v_1..v_n) = e_r(v_1..v_(n-1)) + v_n * e_(r-1)(v_1..v_(n-1)) instead of
 * enumerating `combinations` (same value, field arithmetic is exact).  `numerator / denominator` is
 * Div -> inverse().unwrap() (prime/arithmetic.rs:50-55): coincident nodes panic. */
int orc_rs_decode(uint64_t p, const uint64_t* xs, const uint64_t* ys, size_t k, uint64_t* out) {
  if (k == 0) return ORC_OK;
  uint64_t* e = (uint64_t*)malloc(k * sizeof *e); /* e[r], r = 0..k-1 */
  for (size_t i = 0; i < k; i++) out[i] = 0;
  for (size_t j = 0; j < k; j++) {
    /* elementary symmetric polynomials of the nodes without x_j */
    for (size_t r = 0; r < k; r++) e[r] = 0;
    e[0] = 1 % p;
    size_t cnt = 0;
    for (size_t m = 0; m < k; m++) {
      if (m == j) continue;
      cnt++;
      for (size_t r = cnt; r >= 1; r--) e[r] = orc_add(p, e[r], orc_mul(p, xs[m] % p, e[r - 1]));
    }
    uint64_t den = 1 % p; /* reed_solomon.rs:92-99 */
    for (size_t m = 0; m < k; m++) {
      if (m == j) continue;
      den = orc_mul(p, den, orc_sub(p, xs[m] % p, xs[j] % p));
    }
    for (size_t i = 0; i < k; i++) {
      uint64_t xc = e[k - 1 - i];
      if (i % 2 == 1) xc = orc_mul(p, orc_sub(p, 0, 1 % p), xc); /* (ZERO - ONE) * ..., :78-82 */
      uint64_t num = orc_mul(p, xc, ys[j] % p), q;
      int rc = orc_div(p, num, den, &q);
      if (rc) { free(e); return rc; }
      out[i] = orc_add(p, out[i], q);
    }
  }
  free(e);
  return ORC_OK;
}

/* kzg/setup.rs:63-78: poly.div([-z, 1]) */
int orc_kzg_open_quotient(uint64_t p, const uint64_t* coeffs, size_t d, uint64_t z, uint64_t* quot) {
  uint64_t divisor[2] = {orc_neg(p, z), 1 % p};
  uint64_t* rem = (uint64_t*)malloc((d ? d : 1) * sizeof *rem);
  int rc = orc_poly_divrem(p, coeffs, d, divisor, 2, quot, rem);
  free(rem);
  return rc;
}

/* ====================================================================================================
 * the curve arithmetic behind kzg::commit (src/kzg/setup.rs:45-60).
 * Quadratic extension F_p[u]/(u^2 - nr) (src/algebra/field/extension/gf_101_2.rs: X^2 + 2, i.e. nr = -2),
 * short Weierstrass curve y^2 = x^3 + a x + b with a, b in F_p (src/curve/pluto_curve.rs:27-51),
 * affine points with an Infinity variant (src/curve/mod.rs:66-73).  A point is 5 words: x0 x1 y0 y1 inf.
 * ==================================================================================================== */
typedef struct { uint64_t c0, c1; } fp2;
static fp2 f2(uint64_t a, uint64_t b) { fp2 r = {a, b}; return r; }
static fp2 f2_add(const orc_curve* c, fp2 a, fp2 b) { return f2(orc_add(c->p, a.c0, b.c0), orc_add(c->p, a.c1, b.c1)); }
static fp2 f2_sub(const orc_curve* c, fp2 a, fp2 b) { return f2(orc_sub(c->p, a.c0, b.c0), orc_sub(c->p, a.c1, b.c1)); }
static fp2 f2_neg(const orc_curve* c, fp2 a) { return f2(orc_neg(c->p, a.c0), orc_neg(c->p, a.c1)); }
static int f2_eq(fp2 a, fp2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
/* gf_101_2.rs:83-97: (poly_self * poly_rhs) % irreducible = (a0 b0 + nr a1 b1) + (a0 b1 + a1 b0) u */
static fp2 f2_mul(const orc_curve* c, fp2 a, fp2 b) {
  uint64_t p = c->p;
  return f2(orc_add(p, orc_mul(p, a.c0, b.c0), orc_mul(p, c->nr % p, orc_mul(p, a.c1, b.c1))),
            orc_add(p, orc_mul(p, a.c0, b.c1), orc_mul(p, a.c1, b.c0)));
}
/* gf_101_2.rs:34-47: multiply by the conjugate, scalar = (a0^2 - nr a1^2)^-1 */
static int f2_inv(const orc_curve* c, fp2 a, fp2* out) {
  uint64_t p = c->p;
  if (a.c0 == 0 && a.c1 == 0) return ORC_PANIC_ZERO_INVERSE; /* Div: rhs.inverse().expect("invalid inverse") */
  uint64_t norm = orc_sub(p, orc_mul(p, a.c0, a.c0), orc_mul(p, c->nr % p, orc_mul(p, a.c1, a.c1))), s;
  int rc = orc_inverse(p, norm, &s);
  if (rc) return rc;
  *out = f2(orc_mul(p, a.c0, s), orc_mul(p, orc_neg(p, a.c1), s));
  return ORC_OK;
}
static fp2 f2_small(const orc_curve* c, uint64_t k) { return f2(k % c->p, 0); }

/* curve/mod.rs:129-138 */
int orc_curve_is_on_curve(const orc_curve* c, const uint64_t pt[5]) {
  if (pt[4]) return 1;
  fp2 x = f2(pt[0], pt[1]), y = f2(pt[2], pt[3]);
  fp2 rhs = f2_add(c, f2_add(c, f2_mul(c, f2_mul(c, x, x), x), f2_mul(c, f2_small(c, c->a), x)), f2_small(c, c->b));
  return f2_eq(f2_mul(c, y, y), rhs);
}
/* impl Add, curve/mod.rs:176-211, case by case; AffinePoint::new at the end asserts is_on_curve (:77-81) */
int orc_curve_add(const orc_curve* c, const uint64_t p1[5], const uint64_t p2[5], uint64_t out[5]) {
  if (p1[4]) { memcpy(out, p2, 5 * sizeof *out); return ORC_OK; }
  if (p2[4]) { memcpy(out, p1, 5 * sizeof *out); return ORC_OK; }
  fp2 x1 = f2(p1[0], p1[1]), y1 = f2(p1[2], p1[3]), x2 = f2(p2[0], p2[1]), y2 = f2(p2[2], p2[3]);
  if (f2_eq(x1, x2) && f2_eq(y1, f2_neg(c, y2))) { out[0] = out[1] = out[2] = out[3] = 0; out[4] = 1; return ORC_OK; }
  fp2 lambda, den, inv;
  if (f2_eq(x1, x2) && f2_eq(y1, y2)) {
    fp2 num = f2_add(c, f2_mul(c, f2_mul(c, f2_small(c, 3), x1), x1), f2_small(c, c->a));
    den = f2_mul(c, f2_small(c, 2), y1);
    int rc = f2_inv(c, den, &inv);
    if (rc) return rc;
    lambda = f2_mul(c, num, inv);
  } else {
    den = f2_sub(c, x2, x1);
    int rc = f2_inv(c, den, &inv);
    if (rc) return rc;
    lambda = f2_mul(c, f2_sub(c, y2, y1), inv);
  }
  fp2 x = f2_sub(c, f2_sub(c, f2_mul(c, lambda, lambda), x1), x2);
  fp2 y = f2_sub(c, f2_mul(c, lambda, f2_sub(c, x1, x)), y1);
  out[0] = x.c0; out[1] = x.c1; out[2] = y.c0; out[3] = y.c1; out[4] = 0;
  return orc_curve_is_on_curve(c, out) ? ORC_OK : ORC_PANIC_NOT_ON_CURVE;
}
/* impl Mul<ScalarField>, curve/mod.rs:152-166: ZERO -> Infinity, else self added rhs-1 times */
int orc_curve_mul(const orc_curve* c, const uint64_t pt[5], uint64_t k, uint64_t out[5]) {
  if (k == 0) { out[0] = out[1] = out[2] = out[3] = 0; out[4] = 1; return ORC_OK; }
  uint64_t val[5], t[5];
  memcpy(val, pt, sizeof val);
  for (uint64_t i = 1; i < k; i++) {
    int rc = orc_curve_add(c, val, pt, t);
    if (rc) return rc;
    memcpy(val, t, sizeof val);
  }
  memcpy(out, val, sizeof val);
  return ORC_OK;
}
/* kzg::commit, kzg/setup.rs:45-60: assert!(g1_srs.len() >= coeffs.len()); zip, map(g1 * coeff), sum (reduce, :213-217) */
int orc_kzg_commit(const orc_curve* c, const uint64_t* srs, size_t n_srs, const uint64_t* coeffs, size_t n, uint64_t out[5]) {
  if (n_srs < n) return ORC_PANIC_INDEX;
  uint64_t acc[5] = {0, 0, 0, 0, 1}, term[5], t[5];
  for (size_t i = 0; i < n; i++) {
    if (!orc_curve_is_on_curve(c, srs + 5 * i)) return ORC_PANIC_NOT_ON_CURVE;
    int rc = orc_curve_mul(c, srs + 5 * i, coeffs[i], term);
    if (rc) return rc;
    if (i == 0) { memcpy(acc, term, sizeof acc); continue; }
    rc = orc_curve_add(c, acc, term, t);
    if (rc) return rc;
    memcpy(acc, t, sizeof acc);
  }
  memcpy(out, acc, sizeof acc);
  return ORC_OK;
}
