/* test_rust_ffi_replay.c -- replays, in plain C through the real libronk_ntt.so, the FFI call sequence of the Rust shim
 * rust/ronk-goldilocks (src/polynomial.rs, prime64.rs, device.rs, codes.rs, bn254.rs): the same entry points, argument order, buffer shapes (`[u64; D]` arrays, a
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

/* device.rs: DevicePoly::{from_host, fft, ifft, mul, evaluate, div_linear, div_rem, add, to_host}, Plan::{with_opts, forward,
 * inverse_in_place, forward_many}, HeapPoly through the host-pointer forms, ShardedPlan::{new, info, transform_host} -- the same
 * calls in the same order with the same buffers (device pointers from ronk_dev_alloc, the status word zeroed by a 4-byte upload) */
static uint64_t* dev_from_host(const uint64_t* h, size_t n) {
  void* d = NULL;
  EXPECT(ronk_dev_alloc(&d, n * 8) == 0, "ronk_dev_alloc");
  if (h) EXPECT(ronk_memcpy_h2d(d, h, n * 8) == 0, "ronk_memcpy_h2d");
  return (uint64_t*)d;
}
static void dev_to_host(uint64_t* h, const uint64_t* d, size_t n) {
  EXPECT(ronk_dev_sync() == 0, "ronk_dev_sync");
  EXPECT(ronk_memcpy_d2h(h, d, n * 8) == 0, "ronk_memcpy_d2h");
}
static void replay_device(unsigned log2n) {
  const size_t n = (size_t)1 << log2n;
  uint64_t *x = fresh(n), *ref = calloc(n, 8), *got = calloc(n, 8);
  /* DevicePoly::from_host(x).fft(): Plan::new -> ronk_plan_create_opts with the default options block */
  ronk_plan_opts opts = RONK_PLAN_OPTS_DEFAULT;
  ronk_plan* plan = NULL;
  EXPECT(ronk_plan_create_opts(&plan, P, G, log2n, 1, -1, &opts) == 0, "ronk_plan_create_opts");
  EXPECT(ronk_plan_in_flight(plan) == 1, "default in_flight");
  uint64_t *dx = dev_from_host(x, n), *dy = dev_from_host(NULL, n);
  EXPECT(ronk_ntt_forward_dev(plan, dx, dy, NULL) == 0, "ronk_ntt_forward_dev");
  dev_to_host(got, dy, n);
  EXPECT(orc_fft(P, G, x, ref, n) == 0 && !memcmp(got, ref, n * 8), "DevicePoly::fft");
  EXPECT(ronk_ntt_inverse_dev(plan, dy, dy, NULL) == 0, "inverse_in_place");
  dev_to_host(got, dy, n);
  EXPECT(!memcmp(got, x, n * 8), "DevicePoly::ifft round trip");
  /* Plan::forward_host / inverse_host (HeapPoly-sized data through the plan's pinned staging) */
  EXPECT(ronk_ntt_forward(plan, x, got, NULL) == 0 && !memcmp(got, ref, n * 8), "Plan::forward_host");
  EXPECT(ronk_ntt_inverse(plan, ref, got) == 0 && !memcmp(got, x, n * 8), "Plan::inverse_host");
  EXPECT(ronk_plan_destroy(plan) == 0, "ronk_plan_destroy");
  /* evaluate, div_linear (kzg::open: divisor [-z, 1]) */
  uint64_t z = next_field(), *dq = dev_from_host(NULL, n), *dr = dev_from_host(NULL, 1), r = 0, y = 0;
  EXPECT(ronk_poly_eval_dev(P, dx, n, z, dr, NULL) == 0, "ronk_poly_eval_dev");
  dev_to_host(&y, dr, 1);
  EXPECT(y == orc_poly_eval(P, x, n, z), "DevicePoly::evaluate");
  EXPECT(ronk_poly_div_linear_dev(P, dx, n, P - z, 1, dq, dr, NULL) == 0, "ronk_poly_div_linear_dev");
  dev_to_host(&r, dr, 1); dev_to_host(got, dq, n);
  { uint64_t div[2] = {P - z, 1}, *qr = calloc(n, 8), *rr = calloc(n, 8);
    EXPECT(r == y, "remainder == p(z)");
    if (n <= 4096) EXPECT(orc_poly_divrem(P, x, n, div, 2, qr, rr) == 0 && !memcmp(got, qr, n * 8) && rr[0] == r, "div_linear values");
    else { uint64_t t = next_field();   /* p(t) == q(t) (t - z) + r */
           EXPECT(orc_poly_eval(P, x, n, t) == orc_add(P, orc_mul(P, orc_poly_eval(P, got, n - 1, t), orc_sub(P, t, z)), r), "p = q (x - z) + r"); }
    free(qr); free(rr); }
  /* div_rem by a general divisor with the device status word */
  if (n <= 4096) {
    size_t d2 = 5;
    uint64_t *b = fresh(d2), *db = dev_from_host(b, d2), *drem = dev_from_host(NULL, n), *qr = calloc(n, 8), *rr = calloc(n, 8);
    int status = 0; void* dst = NULL;
    EXPECT(ronk_dev_alloc(&dst, 8) == 0 && ronk_memcpy_h2d(dst, &status, 4) == 0, "status word");
    EXPECT(ronk_poly_divrem_dev(P, dx, n, db, d2, dq, drem, (int*)dst, NULL) == 0, "ronk_poly_divrem_dev");
    EXPECT(ronk_dev_sync() == 0 && ronk_memcpy_d2h(&status, dst, 4) == 0 && status == 0, "status == 0");
    dev_to_host(got, dq, n);
    EXPECT(orc_poly_divrem(P, x, n, b, d2, qr, rr) == 0 && !memcmp(got, qr, n * 8), "div_rem quotient");
    dev_to_host(got, drem, n);
    EXPECT(!memcmp(got, rr, n * 8), "div_rem remainder");
    ronk_dev_free(db); ronk_dev_free(drem); ronk_dev_free(dst); free(b); free(qr); free(rr);
  }
  /* mul (d + d2 - 1 coefficients) and add */
  { size_t d2 = n / 2 + 3;
    uint64_t *b = fresh(d2), *db = dev_from_host(b, d2), *dp = dev_from_host(NULL, n + d2 - 1), *prod = calloc(n + d2 - 1, 8);
    EXPECT(ronk_poly_mul_dev(P, G, dx, n, db, d2, dp, NULL) == 0, "ronk_poly_mul_dev");
    dev_to_host(prod, dp, n + d2 - 1);
    uint64_t t = next_field();
    EXPECT(orc_poly_eval(P, prod, n + d2 - 1, t) == orc_mul(P, orc_poly_eval(P, x, n, t), orc_poly_eval(P, b, d2, t)), "mul homomorphism");
    EXPECT(prod[0] == orc_mul(P, x[0], b[0]) && prod[n + d2 - 2] == orc_mul(P, x[n - 1], b[d2 - 1]), "mul end coefficients");
    EXPECT(ronk_vec_add_dev(P, dx, dy, dq, n, NULL) == 0, "ronk_vec_add_dev");
    dev_to_host(got, dq, n);
    EXPECT(got[0] == orc_add(P, x[0], x[0]) && got[n - 1] == orc_add(P, x[n - 1], x[n - 1]), "DevicePoly::add");   /* dy == x after the round trip */
    ronk_dev_free(db); ronk_dev_free(dp); free(b); free(prod); }
  ronk_dev_free(dx); ronk_dev_free(dy); ronk_dev_free(dq); ronk_dev_free(dr);
  free(x); free(ref); free(got);
}
/* Plan::with_two_lanes + forward_many / inverse_many: K device arrays, HOST arrays of device pointers */
static void replay_many(unsigned log2n, size_t K) {
  const size_t n = (size_t)1 << log2n;
  ronk_plan_opts opts = RONK_PLAN_OPTS_DEFAULT;
  opts.in_flight = 2;
  ronk_plan* plan = NULL;
  EXPECT(ronk_plan_create_opts(&plan, P, G, log2n, 1, -1, &opts) == 0, "create with two lanes");
  EXPECT(ronk_plan_in_flight(plan) == (log2n > 12 ? 2 : 1), "in_flight");
  uint64_t** xs = calloc(K, sizeof *xs);
  const uint64_t** din = calloc(K, sizeof *din);
  uint64_t** dout = calloc(K, sizeof *dout);
  for (size_t i = 0; i < K; i++) { xs[i] = fresh(n); din[i] = dev_from_host(xs[i], n); dout[i] = dev_from_host(NULL, n); }
  EXPECT(ronk_ntt_forward_many_dev(plan, din, dout, K, NULL) == 0, "ronk_ntt_forward_many_dev");
  uint64_t *got = calloc(n, 8), *ref = calloc(n, 8);
  for (size_t i = 0; i < K; i++) {
    dev_to_host(got, dout[i], n);
    EXPECT(orc_fft(P, G, xs[i], ref, n) == 0 && !memcmp(got, ref, n * 8), "forward_many values");
  }
  EXPECT(ronk_ntt_inverse_many_dev(plan, (const uint64_t* const*)dout, dout, K, NULL) == 0, "ronk_ntt_inverse_many_dev (in place)");
  for (size_t i = 0; i < K; i++) { dev_to_host(got, dout[i], n); EXPECT(!memcmp(got, xs[i], n * 8), "inverse_many round trip"); }
  for (size_t i = 0; i < K; i++) { free(xs[i]); ronk_dev_free((void*)din[i]); ronk_dev_free(dout[i]); }
  free(xs); free(din); free(dout); free(got); free(ref);
  EXPECT(ronk_plan_destroy(plan) == 0, "destroy");
}
/* ShardedPlan::new(log2n, false, &[0, 0], 0) -> info -> transform_host: two logical ranks on device 0 */
static void replay_sharded(unsigned log2n) {
  const size_t n = (size_t)1 << log2n;
  int devices[2] = {0, 0};
  ronk_sharded_plan* sp = NULL;
  EXPECT(ronk_sharded_plan_create(&sp, log2n, 0, devices, 2, 0) == 0, "ronk_sharded_plan_create");
  uint64_t rows = 0, cols = 0, per = 0; int chunks = 0;
  EXPECT(ronk_sharded_plan_info(sp, &rows, &cols, &per, &chunks) == 0 && rows * cols == n && per == n / 2 && chunks >= 1, "info");
  uint64_t *x = fresh(n), *got = calloc(n, 8), *ref = calloc(n, 8);
  EXPECT(ronk_ntt_sharded(sp, x, got) == 0, "ronk_ntt_sharded");
  EXPECT(orc_fft(P, G, x, ref, n) == 0 && !memcmp(got, ref, n * 8), "ShardedPlan::transform_host");
  EXPECT(ronk_sharded_sync(sp) == 0 && ronk_sharded_plan_destroy(sp) == 0, "sync / destroy");
  free(x); free(got); free(ref);
}

/* codes.rs: Message::<K>::encode::<N> / decode::<M>, encode_batch, lde, decode_dev -- the same calls, buffers and status word */
static void replay_codes(void) {
  /* Message::<3>::new([1, 2, 3]).encode::<8>() and decode::<8> (reed_solomon.rs:42-106) */
  { uint64_t msg[3] = {1, 2, 3}, xs[8], ys[8], rx[8], ry[8], back[3];
    EXPECT(ronk_rs_encode(P, G, msg, 3, 8, xs, ys) == 0, "ronk_rs_encode");
    EXPECT(orc_rs_encode(P, G, msg, 3, 8, rx, ry) == 0 && !memcmp(xs, rx, 64) && !memcmp(ys, ry, 64), "Message::encode values");
    EXPECT(ronk_rs_decode(P, xs, ys, 3, back) == 0 && !memcmp(back, msg, 24), "Message::decode round trip");
    EXPECT(ronk_rs_encode(P, G, msg, 3, 2, xs, ys) != 0, "N < K is refused"); }
  /* encode_batch(plan_n, msgs, k) and lde(plan_k, plan_n, evals, 1): batch 4, k = 2^12, N = 2^13 */
  { const size_t k = 1u << 12, N = 1u << 13, batch = 4;
    ronk_plan_opts opts = RONK_PLAN_OPTS_DEFAULT;
    ronk_plan *pk = NULL, *pn = NULL;
    EXPECT(ronk_plan_create_opts(&pk, P, G, 12, batch, -1, &opts) == 0 && ronk_plan_create_opts(&pn, P, G, 13, batch, -1, &opts) == 0, "plans");
    uint64_t *msgs = fresh(batch * k), *dm = dev_from_host(msgs, batch * k), *dys = dev_from_host(NULL, batch * N);
    EXPECT(ronk_rs_encode_batch_dev(pn, dm, k, dys, NULL) == 0, "ronk_rs_encode_batch_dev");
    uint64_t *ys = calloc(batch * N, 8), *rx = calloc(N, 8), *ry = calloc(N, 8);
    dev_to_host(ys, dys, batch * N);
    for (size_t b = 0; b < batch; b++)
      EXPECT(orc_rs_encode(P, G, msgs + b * k, k, N, rx, ry) == 0 && !memcmp(ys + b * N, ry, N * 8), "encode_batch values");
    uint64_t *dsmall = dev_from_host(NULL, batch * k), *dco = dev_from_host(NULL, batch * k), *dext = dev_from_host(NULL, batch * N);
    EXPECT(ronk_ntt_forward_dev(pk, dm, dsmall, NULL) == 0, "values on the small domain");
    EXPECT(ronk_lde_batch_dev(pk, pn, dsmall, dco, dext, 1, NULL) == 0, "ronk_lde_batch_dev");
    uint64_t *co = calloc(batch * k, 8), *ext = calloc(batch * N, 8);
    dev_to_host(co, dco, batch * k); dev_to_host(ext, dext, batch * N);
    EXPECT(!memcmp(co, msgs, batch * k * 8), "lde coefficients");
    EXPECT(!memcmp(ext, ys, batch * N * 8), "lde values == encode values");
    /* decode_dev on the first k coordinates of codeword 0: xs = omega_N^j (ronk_lagrange_nodes), 8 zeroed status bytes */
    uint64_t *nodes = calloc(N, 8), *dxs, *dout = dev_from_host(NULL, k), zero = 0, st = 0; void* dst = NULL;
    EXPECT(ronk_lagrange_nodes(P, G, nodes, N) == 0, "nodes");
    dxs = dev_from_host(nodes, k);
    EXPECT(ronk_dev_alloc(&dst, 8) == 0 && ronk_memcpy_h2d(dst, &zero, 8) == 0, "status word");
    EXPECT(ronk_rs_decode_dev(P, dxs, dys, k, dout, (int*)dst, NULL) == 0, "ronk_rs_decode_dev");
    dev_to_host(co, dout, k);
    EXPECT(ronk_memcpy_d2h(&st, dst, 8) == 0 && (uint32_t)st == 0, "decode status");
    EXPECT(!memcmp(co, msgs, k * 8), "decode_dev recovers message 0");
    ronk_dev_free(dm); ronk_dev_free(dys); ronk_dev_free(dsmall); ronk_dev_free(dco); ronk_dev_free(dext); ronk_dev_free(dxs);
    ronk_dev_free(dout); ronk_dev_free(dst);
    free(msgs); free(ys); free(rx); free(ry); free(co); free(ext); free(nodes);
    EXPECT(ronk_plan_destroy(pk) == 0 && ronk_plan_destroy(pn) == 0, "destroy"); }
}
/* device.rs: current_device / OnDevice (ronk_get_device, ronk_set_device), ShardedPlan::with_exchange + exchange(),
 * alloc_blocks + transform on device-resident per-rank blocks; bn254::commit_dev; ronk_trim_workspace */
static void replay_placement(int ndev) {
  int cur = -1;
  EXPECT(ronk_get_device(&cur) == 0 && cur >= 0 && cur < ndev, "ronk_get_device");
  EXPECT(ronk_set_device(cur) == 0, "ronk_set_device(current)");
  EXPECT(ronk_set_device(ndev) == RONK_ERR_INVALID && ronk_set_device(-1) == RONK_ERR_INVALID, "ordinal out of range");
  const unsigned log2n = 16; const size_t n = (size_t)1 << log2n;
  int devices[2] = {0, ndev > 1 ? 1 : 0};
  ronk_sharded_plan* sp = NULL;
  EXPECT(ronk_sharded_plan_create_ex(&sp, log2n, 0, devices, 2, 0, RONK_EXCHANGE_MESH) == 0, "ronk_sharded_plan_create_ex(mesh)");
  EXPECT(ronk_sharded_plan_exchange(sp) == RONK_EXCHANGE_MESH, "ronk_sharded_plan_exchange");
  {   /* ShardedPlan::peer_access: the diagonal is SAME_DEVICE; the off-diagonal pair too when both ranks share device 0,
         DIRECT or STAGED (and counted by the return value) when they sit on two GPUs */
    int pm[4] = {9, 9, 9, 9};
    const int staged = ronk_sharded_plan_peer_access(sp, pm, 4);
    EXPECT(staged >= 0 && pm[0] == RONK_PEER_SAME_DEVICE && pm[3] == RONK_PEER_SAME_DEVICE, "ronk_sharded_plan_peer_access");
    if (ndev > 1) EXPECT((pm[1] == RONK_PEER_DIRECT || pm[1] == RONK_PEER_STAGED) && (pm[2] == RONK_PEER_DIRECT || pm[2] == RONK_PEER_STAGED) &&
                         staged == (pm[1] == RONK_PEER_STAGED) + (pm[2] == RONK_PEER_STAGED), "peer_access across two GPUs");
    else EXPECT(staged == 0 && pm[1] == RONK_PEER_SAME_DEVICE && pm[2] == RONK_PEER_SAME_DEVICE, "peer_access on one GPU");
    EXPECT(ronk_sharded_plan_peer_access(sp, pm, 3) == RONK_ERR_INVALID, "peer_access capacity check");
  }
  uint64_t rows = 0, cols = 0, per = 0; int chunks = 0;
  EXPECT(ronk_sharded_plan_info(sp, &rows, &cols, &per, &chunks) == 0 && per == n / 2, "info");
  /* alloc_blocks: block g on devices[g] (OnDevice around every alloc / copy) */
  uint64_t *x = fresh(n), *ref = calloc(n, 8), *got = calloc(n, 8);
  const uint64_t* din[2]; uint64_t* dout[2];
  const size_t Cw = cols / 2;   /* rank g owns columns [g*Cw, (g+1)*Cw): layout [R][C/W] */
  uint64_t* blk = calloc(per, 8);
  for (int g = 0; g < 2; g++) {
    EXPECT(ronk_set_device(devices[g]) == 0, "select the rank's GPU");
    for (size_t r = 0; r < rows; r++) memcpy(blk + r * Cw, x + r * cols + (size_t)g * Cw, Cw * 8);
    din[g] = dev_from_host(blk, per); dout[g] = dev_from_host(NULL, per);
  }
  EXPECT(ronk_set_device(cur) == 0, "restore");
  EXPECT(ronk_ntt_sharded_dev(sp, din, dout) == 0 && ronk_sharded_sync(sp) == 0, "ShardedPlan::transform on placed blocks");
  EXPECT(orc_fft(P, G, x, ref, n) == 0, "oracle");
  { int ok = 1;   /* rank h ends with X[k1 + R*k2] for k1 in [h*R/W, (h+1)*R/W), laid out [C][R/W] (k2-major): include/ronk_ntt.h */
    const size_t Rw = rows / 2;
    for (int h = 0; h < 2; h++) {
      EXPECT(ronk_set_device(devices[h]) == 0, "select");
      dev_to_host(blk, dout[h], per);
      for (size_t k2 = 0; k2 < cols && ok; k2++)
        for (size_t k1 = 0; k1 < Rw; k1++)
          if (blk[k2 * Rw + k1] != ref[(h * Rw + k1) + rows * k2]) { ok = 0; break; }
    }
    EXPECT(ok, "sharded output blocks == fft"); }
  for (int g = 0; g < 2; g++) { EXPECT(ronk_set_device(devices[g]) == 0, "select"); ronk_dev_free((void*)din[g]); ronk_dev_free(dout[g]); }
  EXPECT(ronk_set_device(cur) == 0, "restore");
  EXPECT(ronk_sharded_plan_destroy(sp) == 0, "destroy");
  /* Exchange::Rccl: two ranks on ONE device are refused (UNSUPPORTED), a missing librccl is RONK_ERR_RCCL; on two GPUs it must work */
  { int rc = ronk_sharded_plan_create_ex(&sp, log2n, 0, devices, 2, 0, RONK_EXCHANGE_RCCL);
    if (devices[0] == devices[1]) EXPECT(rc == RONK_ERR_UNSUPPORTED || rc == RONK_ERR_RCCL, "rccl needs distinct devices");
    else if (rc == 0) {
      EXPECT(ronk_sharded_plan_exchange(sp) == RONK_EXCHANGE_RCCL, "exchange == rccl");
      EXPECT(ronk_ntt_sharded(sp, x, got) == 0 && !memcmp(got, ref, n * 8), "rccl exchange values");
      ronk_sharded_plan_destroy(sp);
    } else EXPECT(rc == RONK_ERR_RCCL, "rccl unavailable -> RONK_ERR_RCCL"); }
  /* bn254::commit_dev: 4 copies of the generator times scalars 1, 2, 3, 4 == commit on host pointers */
  { uint64_t pts[4 * 8], sc[4 * 4], o1[8], o2[8];
    memset(pts, 0, sizeof pts); memset(sc, 0, sizeof sc);
    for (int i = 0; i < 4; i++) { pts[8 * i] = 1; pts[8 * i + 4] = 2; sc[4 * i] = (uint64_t)i + 1; }
    uint64_t *dp = dev_from_host(pts, 32), *ds = dev_from_host(sc, 16);
    EXPECT(ronk_msm_bn254(pts, sc, 4, o1) == 0, "ronk_msm_bn254");
    EXPECT(ronk_msm_bn254_dev(dp, ds, 4, o2, NULL) == 0 && !memcmp(o1, o2, 64), "commit_dev == commit");
    /* bn254::open / open_dev / div_linear_dev: p(x) = 1 + 2x + 3x^2 + 4x^3 over F_r at z = 2, SRS = four copies of G:
       quotient [24, 11, 4, 0], p(2) = 49, proof = commit(quotient) = (24 + 11 + 4) G = 39 G = the same MSM with scalar 39 */
    uint64_t z[4] = {2, 0, 0, 0}, op[8], ov[4], od[8], ovd[4], q[16], rem[4], want[8], s39[4] = {39, 0, 0, 0};
    EXPECT(ronk_kzg_open_bn254(sc, 4, z, pts, 4, op, ov) == 0 && ov[0] == 49 && !ov[1] && !ov[2] && !ov[3], "bn254::open value");
    EXPECT(ronk_msm_bn254(pts, s39, 1, want) == 0 && !memcmp(op, want, 64), "bn254::open proof == 39 G");
    uint64_t *dq = dev_from_host(sc, 16), *drem = dev_from_host(z, 4);
    EXPECT(ronk_kzg_open_bn254_dev(ds, 4, z, dp, dq, od, ovd, NULL) == 0 && !memcmp(od, op, 64) && ovd[0] == 49, "bn254::open_dev");
    EXPECT(ronk_poly_div_linear_bn254_dev(ds, 4, z, dq, drem, NULL) == 0, "bn254::div_linear_dev");
    dev_to_host(q, dq, 16); dev_to_host(rem, drem, 4);
    EXPECT(q[0] == 24 && q[4] == 11 && q[8] == 4 && q[12] == 0 && rem[0] == 49, "div_linear_dev quotient / remainder");
    EXPECT(ronk_kzg_open_bn254(sc, 4, z, pts, 3, op, ov) == RONK_ERR_INDEX, "SRS shorter than the polynomial -> panic");
    ronk_dev_free(dq); ronk_dev_free(drem);
    ronk_dev_free(dp); ronk_dev_free(ds); }
  EXPECT(ronk_trim_workspace() == 0, "ronk_trim_workspace");
  free(x); free(ref); free(got); free(blk);
}

/* prime64.rs: Prime64::<P, G>::assert_prime, AcceleratedPrime::{fft_gpu, dft_gpu, evaluate_gpu, mul_gpu,
 * quotient_and_remainder_gpu}, AcceleratedPrimeLagrange::ifft_gpu, PrimePlan::{new, path, forward, inverse} for a generic odd
 * 64-bit prime: the same entry points as for Goldilocks with p = P2, g = G2 (the tile kernels over Montgomery arithmetic) */
static void replay_prime64(uint64_t P2, uint64_t G2, size_t D, size_t D2) {
  EXPECT(ronk_check_prime(P2) == 0, "Prime64::assert_prime");
  uint64_t *c = malloc(D * 8), *b = malloc(D2 * 8), *out = calloc(D, 8), *nodes = calloc(D, 8), *ref = calloc(D + D2, 8), *back = calloc(D, 8);
  for (size_t i = 0; i < D; i++) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; c[i] = rng_state % P2; }
  for (size_t i = 0; i < D2; i++) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; b[i] = rng_state % P2; }
  EXPECT(ronk_fft(P2, G2, c, out, nodes, D) == 0, "prime64 ronk_fft");
  EXPECT(orc_fft(P2, G2, c, ref, D) == 0 && !memcmp(out, ref, D * 8), "prime64 fft values");
  EXPECT(orc_lagrange_nodes(P2, G2, ref, D) == 0 && !memcmp(nodes, ref, D * 8), "prime64 fft nodes");
  EXPECT(ronk_ifft(P2, G2, out, back, D) == 0 && !memcmp(back, c, D * 8), "prime64 ifft round trip");
  memset(out, 0, D * 8);
  EXPECT(ronk_dft(P2, G2, c, out, D) == 0 && ronk_lagrange_nodes(P2, G2, nodes, D) == 0, "prime64 ronk_dft + nodes");
  EXPECT(orc_fft(P2, G2, c, ref, D) == 0 && !memcmp(out, ref, D * 8), "prime64 dft values");
  uint64_t y = 0, x = c[D / 2];
  EXPECT(ronk_poly_eval(P2, c, D, x, &y) == 0 && y == orc_poly_eval(P2, c, D, x), "prime64 evaluate");
  uint64_t* prod = calloc(D + D2 - 1, 8);
  EXPECT(ronk_poly_mul(P2, G2, c, D, b, D2, prod) == 0, "prime64 ronk_poly_mul");
  orc_poly_mul(P2, c, D, b, D2, ref);
  EXPECT(!memcmp(prod, ref, (D + D2 - 1) * 8), "prime64 mul values");
  if (D <= 4096) {
    uint64_t *q = calloc(D, 8), *r = calloc(D, 8), *qr = calloc(D, 8), *rr = calloc(D, 8);
    EXPECT(ronk_poly_divrem(P2, c, D, b, D2, q, r) == 0, "prime64 ronk_poly_divrem");
    EXPECT(orc_poly_divrem(P2, c, D, b, D2, qr, rr) == 0 && !memcmp(q, qr, D * 8) && !memcmp(r, rr, D * 8), "prime64 divrem values");
    free(q); free(r); free(qr); free(rr);
  }
  /* Prime64::sqrt_many / euler_criterion_many (FieldExt over arrays): squares of the first coefficients */
  { size_t m = D < 256 ? D : 256;
    uint64_t *sq = malloc(m * 8), *r0 = malloc(m * 8), *r1 = malloc(m * 8), *o0 = malloc(m * 8), *o1 = malloc(m * 8);
    for (size_t i = 0; i < m; i++) sq[i] = orc_mul(P2, c[i], c[i]);
    EXPECT(ronk_vec_sqrt(P2, sq, r0, r1, m) == 0 && orc_vec_sqrt(P2, sq, o0, o1, m) == 0 && !memcmp(r0, o0, m * 8) && !memcmp(r1, o1, m * 8),
           "prime64 sqrt_many");
    EXPECT(ronk_vec_euler(P2, c, r0, m) == 0, "prime64 euler_criterion_many");
    orc_vec_euler(P2, c, o0, m);
    EXPECT(!memcmp(r0, o0, m * 8), "prime64 euler values");
    free(sq); free(r0); free(r1); free(o0); free(o1); }
  /* PrimePlan::new(log2n, 1): ronk_plan_create(&raw, P, G, log2n, 1, -1); path(); forward; inverse; drop */
  unsigned lg = 0; while (((size_t)1 << lg) < D) lg++;
  ronk_plan* pl = NULL;
  EXPECT(ronk_plan_create(&pl, P2, G2, lg, 1, -1) == 0, "PrimePlan::new");
  EXPECT(ronk_plan_path(pl) == (lg >= 4 ? 2 : 0), "PrimePlan::path: the tile kernels over Montgomery arithmetic");
  memset(out, 0, D * 8);
  EXPECT(ronk_ntt_forward(pl, c, out, NULL) == 0 && orc_fft(P2, G2, c, ref, D) == 0 && !memcmp(out, ref, D * 8), "PrimePlan::forward");
  EXPECT(ronk_ntt_inverse(pl, out, back) == 0 && !memcmp(back, c, D * 8), "PrimePlan::inverse");
  EXPECT(ronk_plan_destroy(pl) == 0, "PrimePlan drop");
  free(c); free(b); free(out); free(nodes); free(ref); free(back); free(prod);
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
  /* device.rs */
  replay_device(10); replay_device(12); replay_device(16); replay_device(20);
  replay_many(10, 3); replay_many(16, 5); replay_many(20, 4);
  replay_sharded(16); replay_sharded(20);
  replay_codes();
  replay_placement(ndev);
  /* prime64.rs: a prime above 2^63 (2-adicity 34), one below 2^62, a 32-bit one */
  replay_prime64(0xFFFFFFFC00000001ull, 10, 16, 3); replay_prime64(0xFFFFFFFC00000001ull, 10, 4096, 100);
  replay_prime64(0xFFFFFFFC00000001ull, 10, 1u << 16, 5); replay_prime64(0x3A00000000000001ull, 3, 1u << 14, 1u << 14);
  replay_prime64(0xC0000001ull, 5, 1024, 1024);
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
