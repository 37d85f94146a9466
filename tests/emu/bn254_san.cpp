// bn254_san.cpp -- TEST INFRASTRUCTURE (make sanitize): csrc/bn254.h under ASan + UBSan on the host: field identities and
// the group law driven through every branch (infinity operands, P + P, P - P), order of the generator.
#include <stdio.h>

#include "../../ronkathon_amd/csrc/bn254.h"

using namespace bn254;

static int fails = 0;
#define CHECK(x) do { if (!(x)) { printf("FAIL line %d: %s\n", __LINE__, #x); fails++; } } while (0)

int main() {
  // field: (a*b) * b^-1 == a, a - a == 0, a + (p - a) == 0 for a few values incl. the edges
  u64 seed = 0x9E3779B97F4A7C15ull;
  for (int it = 0; it < 50; it++) {
    Fp a, b;
    for (int i = 0; i < 8; i++) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; a.l[i] = (u32)(seed >> 32); seed = seed * 6364136223846793005ull + 1; b.l[i] = (u32)(seed >> 33); }
    a.l[7] &= 0x1FFFFFFF; b.l[7] &= 0x1FFFFFFF;   // < 2^253 < p
    if (it == 0) { a = fp_zero(); a.l[0] = 1; }
    if (it == 1) { for (int i = 0; i < 8; i++) a.l[i] = P_limb(i); a.l[0] -= 1; }   // p - 1
    if (fp_is_zero_exact(b)) b.l[0] = 7;
    const Fp am = fp_to_mont(a), bm = fp_to_mont(b);
    CHECK(fp_eq(fp_from_mont(fp_mul(fp_mul(am, bm), fp_inv(bm))), a));
    CHECK(fp_is_zero(fp_sub(am, am)));
    CHECK(fp_is_zero(fp_add(am, fp_neg(am))));
    CHECK(fp_eq(fp_sqr(am), fp_mul(am, am)));
  }
  // group: G = (1, 2); r * G == infinity, (r - 1) * G == -G, branches of madd / add / dbl
  Affine g;
  g.x = fp_zero(); g.x.l[0] = 1; g.y = fp_zero(); g.y.l[0] = 2;
  g.x = fp_to_mont(g.x); g.y = fp_to_mont(g.y);
  CHECK(affine_on_curve(g));
  const u64 r[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};   // group order
  Xyzz acc = xyzz_inf();
  for (int i = 255; i >= 0; i--) { acc = xyzz_dbl(acc); if ((r[i >> 6] >> (i & 63)) & 1) xyzz_madd(acc, g, false); }
  CHECK(xyzz_is_inf(acc));
  Xyzz two = xyzz_inf();
  xyzz_madd(two, g, false); xyzz_madd(two, g, false);            // P + P inside the mixed addition
  u64 out[8];
  xyzz_store_affine(two, out);
  CHECK(out[0] != 0 || out[4] != 0);                             // an affine point, not infinity
  Xyzz z = two;
  Affine ng = g; ng.y = fp_neg(g.y);
  xyzz_madd(z, ng, false); xyzz_madd(z, g, true);                // 2G - G - G
  CHECK(xyzz_is_inf(z));
  CHECK(xyzz_is_inf(xyzz_add(two, xyzz_inf())) == false && xyzz_is_inf(xyzz_add(xyzz_inf(), xyzz_inf())));
  Xyzz four_a = xyzz_dbl(two), four_b = xyzz_add(two, two);      // dbl vs add's equal-operand branch
  u64 oa[8], ob[8];
  xyzz_store_affine(four_a, oa); xyzz_store_affine(four_b, ob);
  for (int i = 0; i < 8; i++) CHECK(oa[i] == ob[i]);
  printf(fails ? "bn254 sanitize run: %d failure(s)\n" : "bn254 sanitize run ok\n", fails);
  return fails ? 1 : 0;
}
