/* test_rust_ffi_replay.c -- replays, in plain C through the real libronk_ntt.so, the FFI call sequence of the Rust shim
 * rust/ronk-goldilocks (src/polynomial.rs): the same entry points, argument order, buffer shapes (`[u64; D]` arrays, a
 * D-element node vector, NULL never passed where the shim passes a pointer) and the same error-code -> panic mapping.
 * Results are checked against the oracle's restatement of the reference (oracle/ronk_oracle.c, the CHECKER).
 * TEST INFRASTRUCTURE; built and run by tests/test_cpp_host_mirror.py (needs a GPU to run). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ronk_ntt.h"
#include "../../oracle/ronk_oracle.h"

#define P RONK_GOLDILOCKS_P
#define G RONK_GOLDILOCKS_G
static int fails = 0;
#define EXPECT(cond, what) do { if (!(cond)) { printf("FAIL %s (line %d)\n", what, __LINE__); fails++; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_field(void) {
  for (;;) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    if (rng_state < P) return rng_state;   /* Distribution<Goldilocks> for Standard: rejection of non-canonical draws */
  }
}
static uint64_t* fresh(size_t n) { uint64_t* v = malloc(n * 8 + 8); for (size_t i = 0; i < n; i++) v[i] = next_field(); return v; }

/* Accelerated::fft_gpu / AcceleratedLagrange::ifft_gpu / dft_gpu / evaluate_gpu for one D */
static void replay_transforms(size_t D) {
  uint64_t *c = fresh(D), *out = calloc(D, 8), *nodes = calloc(D, 8), *ref = calloc(D, 8), *back = calloc(D, 8);
  /* fft_gpu: out = [ZERO; D], nodes = vec![ZERO; D]; ronk_fft(P, G, coeffs, out, nodes, D) */
  EXPECT(ronk_fft(P, G, c, out, nodes, D) == 0, "ronk_fft");
  EXPECT(orc_fft(P, G, c, ref, D) == 0 && !memcmp(out, ref, D * 8), "fft values");
  EXPECT(orc_lagrange_nodes(P, G, ref, D) == 0 && !memcmp(nodes, ref, D * 8), "fft nodes");
  /* ifft_gpu on the Lagrange polynomial just produced */
  EXPECT(ronk_ifft(P, G, out, back, D) == 0 && !memcmp(back, c, D * 8), "ifft round trip");
  /* dft_gpu: ronk_dft then ronk_lagrange_nodes */
  if (D <= 4096) {
    memset(out, 0, D * 8); memset(nodes, 0, D * 8);
    EXPECT(ronk_dft(P, G, c, out, D) == 0, "ronk_dft");
    EXPECT(ronk_lagrange_nodes(P, G, nodes, D) == 0, "ronk_lagrange_nodes");
    EXPECT(orc_dft(P, G, c, ref, D) == 0 && !memcmp(out, ref, D * 8), "dft values");
    /* AcceleratedLagrange::evaluate_gpu(x): barycentric from (values, basis.nodes) */
    uint64_t y = 0, yr = 0, x = next_field();
    EXPECT(ronk_lagrange_eval(P, out, nodes, D, x, &y) == 0, "ronk_lagrange_eval");
    EXPECT(orc_lagrange_eval(P, out, nodes, D, x, &yr) == 0 && y == yr, "lagrange evaluate");
  }
  /* Accelerated::evaluate_gpu */
  uint64_t y = 0, x = next_field();
  EXPECT(ronk_poly_eval(P, c, D, x, &y) == 0 && y == orc_poly_eval(P, c, D, x), "evaluate");
  free(c); free(out); free(nodes); free(ref); free(back);
}

/* Accelerated::mul_gpu::<D2> and quotient_and_remainder_gpu::<D2> */
static void replay_arithmetic(size_t D, size_t D2) {
  uint64_t *a = fresh(D), *b = fresh(D2), *prod = calloc(D + D2 - 1, 8), *ref = calloc(D + D2 - 1, 8);
  EXPECT(ronk_poly_mul(P, G, a, D, b, D2, prod) == 0, "ronk_poly_mul");
  orc_poly_mul(P, a, D, b, D2, ref);
  EXPECT(!memcmp(prod, ref, (D + D2 - 1) * 8), "mul values");
  uint64_t *q = calloc(D, 8), *r = calloc(D, 8), *qr = calloc(D, 8), *rr = calloc(D, 8);
  EXPECT(ronk_poly_divrem(P, a, D, b, D2, q, r) == 0, "ronk_poly_divrem");
  EXPECT(orc_poly_divrem(P, a, D, b, D2, qr, rr) == 0 && !memcmp(q, qr, D * 8) && !memcmp(r, rr, D * 8), "divrem values");
  free(a); free(b); free(prod); free(ref); free(q); free(r); free(qr); free(rr);
}

int main(void) {
  int ndev = 0;
  if (ronk_device_count(&ndev) != 0 || ndev < 1) { printf("no device\n"); return 2; }
  /* the reference's n = 4 test polynomial [1, 2, 3, 4] (src/polynomial/tests.rs) over the 64-bit field */
  uint64_t c4[4] = {1, 2, 3, 4}, o4[4], n4[4], r4[4];
  EXPECT(ronk_fft(P, G, c4, o4, n4, 4) == 0 && orc_dft(P, G, c4, r4, 4) == 0 && !memcmp(o4, r4, 32), "fft([1,2,3,4]) == dft");
  size_t sizes[] = {1, 2, 4, 16, 1024, 4096, 65536, 1u << 20};
  for (size_t i = 0; i < sizeof(sizes) / sizeof(sizes[0]); i++) replay_transforms(sizes[i]);
  replay_arithmetic(4, 2); replay_arithmetic(5, 5); replay_arithmetic(17, 17); replay_arithmetic(1000, 3);
  replay_arithmetic(3000, 3000); replay_arithmetic(1 << 15, 2);
  /* rs_decode::<K> */
  { enum { K = 64 };
    uint64_t xs[K], *ys = fresh(K), out[K], ref[K];
    EXPECT(ronk_lagrange_nodes(P, G, xs, K) == 0, "nodes for rs_decode");
    EXPECT(ronk_rs_decode(P, xs, ys, K, out) == 0 && orc_rs_decode(P, xs, ys, K, ref) == 0 && !memcmp(out, ref, K * 8), "rs_decode");
    free(ys); }
  /* code -> panic mapping (ffi::check): the codes and texts `#[should_panic]` tests rely on */
  uint64_t c7[7] = {1, 1, 1, 1, 1, 1, 1}, o7[7];
  int rc = ronk_dft(P, G, c7, o7, 7);               /* 7 does not divide p - 1: "n must divide p^q - 1" (field/mod.rs:72) */
  EXPECT(rc == RONK_ERR_NO_ROOT && strstr(ronk_strerror(rc), "divide"), "no roots of unity -> panic text");
  uint64_t c3[3] = {1, 2, 3}, o3[3], n3[3];
  rc = ronk_fft(P, G, c3, o3, n3, 3);              /* fft's `D.is_power_of_two()` bound (polynomial/mod.rs:274) */
  EXPECT(rc == RONK_ERR_NOT_POW2 && strlen(ronk_strerror(rc)) > 0, "fft of 3 elements");
  uint64_t zero2[2] = {0, 0}, q3[3], r3[3];
  rc = ronk_poly_divrem(P, c3, 3, zero2, 2, q3, r3);   /* zero divisor: the reference's index / unwrap panic */
  EXPECT(rc == RONK_ERR_INDEX, "division by the zero polynomial");
  uint64_t xs2[2] = {5, 5}, ys2[2] = {1, 2}, o2[2];
  rc = ronk_rs_decode(P, xs2, ys2, 2, o2);             /* coincident nodes: inverse().unwrap() on ZERO */
  EXPECT(rc == RONK_ERR_ZERO_INVERSE, "coincident interpolation nodes");
  printf(fails ? "FAILED %d\n" : "ALL OK\n", fails);
  return fails ? 1 : 0;
}
