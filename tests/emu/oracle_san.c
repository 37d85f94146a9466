/* oracle_san.c -- drives the oracle (oracle/ronk_oracle.c, the CHECKER) through every entry-point family under
 * AddressSanitizer + UndefinedBehaviourSanitizer (`make sanitize`, tests/test_sanitizers.py).  TEST INFRASTRUCTURE.
 * The checks are self-consistency properties, so a memory error or UB in the restatement shows up as a sanitizer
 * report (non-zero exit), not as a wrong golden vector. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/ronk_oracle.h"

#define GP 0xFFFFFFFF00000001ull
static uint64_t st = 0x243F6A8885A308D3ull;
static uint64_t rnd(uint64_t p) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st % p; }
static int fails = 0;
#define EXPECT(c, w) do { if (!(c)) { printf("FAIL %s\n", w); fails++; } } while (0)

int main(void) {
  const uint64_t primes[] = {17, 101, 127, GP};
  for (int pi = 0; pi < 4; pi++) {
    const uint64_t p = primes[pi];
    uint64_t g = 0;
    if (p == GP) g = 7; else EXPECT(orc_find_primitive_element(p, &g) == 0, "generator");
    for (size_t n = 1; n <= 64; n *= 2) {
      if ((p - 1) % n) continue;
      uint64_t x[64], y[64], z[64], nodes[64];
      for (size_t i = 0; i < n; i++) x[i] = rnd(p);
      EXPECT(orc_fft(p, g, x, y, n) == 0 && orc_ifft(p, g, y, z, n) == 0 && !memcmp(x, z, n * 8), "fft round trip");
      EXPECT(orc_dft(p, g, x, z, n) == 0 && !memcmp(y, z, n * 8), "dft == fft");
      EXPECT(orc_lagrange_nodes(p, g, nodes, n) == 0, "nodes");
      if (n >= 2) {
        uint64_t t = rnd(p), v = 0;
        int rc = orc_lagrange_eval(p, y, nodes, n, t, &v);
        int on_node = 0;
        for (size_t i = 0; i < n; i++) on_node |= nodes[i] == t;
        if (rc == 0 && !on_node) EXPECT(v == orc_poly_eval(p, x, n, t), "barycentric == Horner");
      }
    }
    /* polynomial arithmetic incl. ragged lengths, zero operands */
    for (int it = 0; it < 50; it++) {
      size_t d = 1 + rnd(40), d2 = 1 + rnd(40);
      uint64_t a[64], b[64], prod[128], q[64], r[64], s1[64], s2[64];
      for (size_t i = 0; i < d; i++) a[i] = rnd(p);
      for (size_t i = 0; i < d2; i++) b[i] = rnd(p);
      if (it % 7 == 0) memset(b, 0, sizeof b);
      if (it % 11 == 0) a[d - 1] = 0;
      orc_poly_mul(p, a, d, b, d2, prod);
      orc_poly_add(p, a, d, b, d2, s1); orc_poly_sub(p, s1, d, b, d2, s2);
      EXPECT(!memcmp(s2, a, d * 8), "(a + b) - b == a");
      int rc = orc_poly_divrem(p, a, d, b, d2, q, r);   /* may be a reference panic code: only memory safety matters */
      if (rc == 0 && b[d2 - 1] != 0) {
        uint64_t t = rnd(p);
        uint64_t lhs = orc_poly_eval(p, a, d, t);
        uint64_t rhs = orc_add(p, orc_mul(p, orc_poly_eval(p, q, d, t), orc_poly_eval(p, b, d2, t)), orc_poly_eval(p, r, d, t));
        EXPECT(lhs == rhs, "a == q b + r");
      }
      uint64_t pm[128];
      orc_pow_mult(p, a, d, 5, rnd(p), pm);
      (void)orc_degree(a, d); (void)orc_leading_coefficient(a, d);
      orc_poly_from(a, d, s1, 64); orc_poly_from(a, d, s1, d > 3 ? d - 3 : 1);
    }
    /* Reed-Solomon round trip through interpolation */
    if ((p - 1) % 8 == 0) {
      uint64_t msg[4], xs[8], ys[8], back[4];
      for (int i = 0; i < 4; i++) msg[i] = rnd(p);
      EXPECT(orc_rs_encode(p, g, msg, 4, 8, xs, ys) == 0, "rs encode");
      EXPECT(orc_rs_decode(p, xs + 3, ys + 3, 4, back) == 0 && !memcmp(back, msg, 32), "rs decode after erasures");
    }
    uint64_t c[9], qq[9];
    for (int i = 0; i < 9; i++) c[i] = rnd(p);
    EXPECT(orc_kzg_open_quotient(p, c, 9, rnd(p), qq) == 0, "open quotient");
  }
  /* the reference's toy curve: group law closure over a few multiples */
  orc_curve cv = {101, 99, 0, 3};   /* y^2 = x^3 + 3 over GF(101^2), u^2 = -2 */
  uint64_t gen[5] = {1, 2, 0, 0, 0}, acc[5], nxt[5];
  if (orc_curve_is_on_curve(&cv, gen)) {
    memcpy(acc, gen, sizeof acc);
    for (int k = 2; k <= 20; k++) {
      EXPECT(orc_curve_add(&cv, acc, gen, nxt) == 0, "curve add");
      uint64_t viamul[5];
      EXPECT(orc_curve_mul(&cv, gen, (uint64_t)k, viamul) == 0 && !memcmp(viamul, nxt, sizeof nxt), "k*G by addition == mul");
      memcpy(acc, nxt, sizeof acc);
    }
  }
  printf(fails ? "FAILED %d\n" : "ALL OK\n", fails);
  return fails ? 1 : 0;
}
