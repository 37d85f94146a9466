/*
 * ronk_ntt.h -- C ABI of libronk_ntt.so, the MI355X-native (gfx950, hand-written HIP)
 * finite-field / NTT / polynomial engine that sits behind ronkathon's
 * `Polynomial<B, F, D>` + `Field` / `FiniteField` trait surface.
 *
 * ronkathon (Rust) has no FFI of its own: the seam is its generic trait surface.  Each
 * entry point below states which reference item it replaces (paths relative to the
 * ronkathon repository).  INTEGRATION.md shows the Rust `extern "C"` block and the trait
 * impls a maintainer adds on the reference side.
 *
 * Conventions
 *  - every field element is a canonical residue in [0, p) stored as uint64_t -- exactly what
 *    `PrimeField<P>{ value: usize }` holds (src/algebra/field/prime/mod.rs:39-42); Montgomery
 *    form never crosses this boundary;
 *  - transforms are natural order in, natural order out, with omega = g^((p-1)/n)
 *    (src/algebra/field/mod.rs:70-75, src/polynomial/mod.rs:240-323);
 *  - buffers are caller-owned, no ownership transfer, no callbacks; functions without a
 *    `_dev` suffix take HOST pointers (and stage through HBM), `_dev` functions take DEVICE
 *    pointers and enqueue on `stream` (a hipStream_t, NULL = the null stream) without
 *    synchronising;
 *  - the reference reports errors by panicking; here every function returns 0 or a negative
 *    RONK_ERR_* code, and the Rust shim turns a non-zero code back into the same panic;
 *  - re-entrant: plans are immutable after creation.  A plan owns ONE scratch buffer: calls on one stream are
 *    ordered by the stream, and a `_dev` call that arrives on another stream than the plan's previous call is made
 *    to wait (event) for that call, so concurrent streams never corrupt each other -- they serialise on the plan.
 *    For transforms that should overlap, hand the library a batch or several arrays per call (ronk_plan_opts::in_flight,
 *    ronk_ntt_forward_many_dev) or use one plan per stream.  (While a stream is being captured into a
 *    hipGraph the guard is skipped: a captured graph must own its plan.)
 *  - the 64-bit hot path is the Goldilocks field p = 2^64 - 2^32 + 1 with generator g = 7;
 *    any other odd prime p < 2^64 (e.g. the reference's F_101, F_17, F_127) runs through a
 *    generic Montgomery path so that the reference's own test vectors pass on the GPU.
 *  - there is NO CPU fallback: without a HIP device every compute entry point returns
 *    RONK_ERR_NO_DEVICE.
 */
#ifndef RONK_NTT_H
#define RONK_NTT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RONK_GOLDILOCKS_P 0xFFFFFFFF00000001ull
#define RONK_GOLDILOCKS_G 7ull

/* error codes; messages via ronk_strerror() repeat the reference's panic texts */
#define RONK_OK 0
#define RONK_ERR_NO_ROOT (-1)       /* assert!(p_minus_one % n == 0, "n must divide p^q - 1"), field/mod.rs:72; polynomial/mod.rs:361 */
#define RONK_ERR_ZERO_INVERSE (-2)  /* inverse().unwrap() on ZERO, prime/arithmetic.rs:54, polynomial/mod.rs:196 */
#define RONK_ERR_NOT_POW2 (-3)      /* fft()/ifft() bound `D.is_power_of_two()`, polynomial/mod.rs:274, :431 */
#define RONK_ERR_NOT_PRIME (-4)     /* is_prime() panic "input is not a prime number", prime/mod.rs:92-100 */
#define RONK_ERR_NO_GENERATOR (-5)  /* find_primitive_element panic, prime/mod.rs:122 */
#define RONK_ERR_INDEX (-6)         /* slice index / unwrap-on-None panics (zero divisor, ragged division) */
#define RONK_ERR_INVALID (-7)       /* NULL pointer, zero length, bad plan */
#define RONK_ERR_HIP (-8)           /* a HIP runtime call failed; ronk_last_hip_error() has the text */
#define RONK_ERR_UNSUPPORTED (-9)   /* size outside what the kernels cover (stated per function) */
#define RONK_ERR_NO_DEVICE (-10)    /* no HIP device: the library never computes on the CPU */
#define RONK_ERR_NOT_ON_CURVE (-11) /* assert!(point.is_on_curve(), "Point is not on curve"), src/curve/mod.rs:79 */
#define RONK_ERR_NOT_RESIDUE (-13)  /* assert!(self.euler_criterion(), "Element is not a quadratic residue"), prime/mod.rs:179 */
#define RONK_ERR_RCCL (-12)         /* librccl.so could not be loaded or an RCCL call failed; ronk_last_hip_error() has the text */

const char* ronk_strerror(int code);
const char* ronk_last_hip_error(void);
int ronk_device_count(int* count);

/* ---- field: src/algebra/field/mod.rs:17-76, src/algebra/field/prime/{mod,arithmetic}.rs ---- */

/* FiniteField::PRIMITIVE_ELEMENT (prime/mod.rs:87-90, :110-123): the reference's heuristic for
 * small primes, the explicit generator 7 for Goldilocks (the heuristic returns a non-generator
 * there).  Host-side integer logic, no device work. */
int ronk_primitive_element(uint64_t p, uint64_t* g);
/* FiniteField::primitive_root_of_unity(n) (field/mod.rs:70-75) */
int ronk_root_of_unity(uint64_t p, uint64_t g, uint64_t n, uint64_t* out);
/* PrimeField::new's primality assertion (prime/mod.rs:48-51, :92-100): 0 or RONK_ERR_NOT_PRIME */
int ronk_check_prime(uint64_t p);

/* Element-wise Field operators over arrays (Add/Sub/Mul/Neg, prime/arithmetic.rs:3-65;
 * Field::inverse prime/mod.rs:62-72 -> RONK_ERR_ZERO_INVERSE if any a[i] == 0;
 * Field::pow prime/mod.rs:74-84).  These are also what Polynomial Add/Sub/Neg reduce to. */
int ronk_vec_add(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int ronk_vec_sub(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int ronk_vec_mul(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int ronk_vec_neg(uint64_t p, const uint64_t* a, uint64_t* out, size_t n);
int ronk_vec_inv(uint64_t p, const uint64_t* a, uint64_t* out, size_t n);
int ronk_vec_pow(uint64_t p, const uint64_t* a, uint64_t e, uint64_t* out, size_t n);
/* FieldExt (src/algebra/field/mod.rs:79-84) over arrays.
 * euler_criterion (prime/mod.rs:142-172): out[i] = 1 when a[i]^((p-1)/2) == 1, else 0 (ZERO is no residue by this test).
 * sqrt (prime/mod.rs:174-226, Tonelli-Shanks): (r0[i], r1[i]) = the two roots of a[i], SMALLER FIRST as the reference returns
 * them; ZERO -> (0, 0); a non-residue is the reference's assert -> RONK_ERR_NOT_RESIDUE (the _dev form raises *d_status, which
 * may be NULL, and stores (0, 0) there).  p must be an odd prime (over F_2 the reference's search for a non-residue does not
 * terminate: RONK_ERR_UNSUPPORTED).  d_r0 may alias d_a. */
int ronk_vec_euler(uint64_t p, const uint64_t* a, uint64_t* out, size_t n);
int ronk_vec_sqrt(uint64_t p, const uint64_t* a, uint64_t* r0, uint64_t* r1, size_t n);
int ronk_vec_euler_dev(uint64_t p, const uint64_t* d_a, uint64_t* d_out, size_t n, void* stream);
int ronk_vec_sqrt_dev(uint64_t p, const uint64_t* d_a, uint64_t* d_r0, uint64_t* d_r1, size_t n, int* d_status, void* stream);
int ronk_vec_add_dev(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
int ronk_vec_sub_dev(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
int ronk_vec_mul_dev(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);

/* ---- transforms: src/polynomial/mod.rs ---- */

typedef struct ronk_plan ronk_plan;

/* A plan fixes (p, g, n = 2^log2n, batch) on one device: twiddle tables in HBM, the pass
 * decomposition, and a scratch buffer of batch*n elements.  `batch` polynomials are stored
 * back to back ([batch][n], row-major).  device < 0 = current device.
 * Errors: RONK_ERR_NO_ROOT if 2^log2n does not divide p-1; RONK_ERR_NOT_PRIME. */
int ronk_plan_create(ronk_plan** out, uint64_t p, uint64_t g, uint32_t log2n, uint64_t batch, int device);
/* Same with planner tuning (-1 = default): tile_log2_columns = log2 of the widest tile (columns per workgroup;
 * default 4 -> one 128 KiB-LDS workgroup per CU at 2^11 rows, best for one transform at a time; 2 -> two
 * workgroups per CU, better when several transforms are in flight on different streams);
 * twiddle_matrix_log2_max = largest full inter-pass twiddle matrix (default: 18 = L2-resident only, and the whole matrix
 * at 2^21 / 2^22 points, where it pays for its extra read). */
int ronk_plan_create_tuned(ronk_plan** out, uint64_t p, uint64_t g, uint32_t log2n, uint64_t batch, int device,
                           int tile_log2_columns, int twiddle_matrix_log2_max);
/* Same through an options block (start from RONK_PLAN_OPTS_DEFAULT, then set what you need; -1 = default everywhere).
 * in_flight: 1 = every call runs on the caller's stream only; 2 = the library keeps TWO transforms in flight behind
 * this one handle: a batched call (batch >= 2, multi-pass sizes) runs the second half of the batch on an internal side
 * stream -- fork / join by events on the caller's stream, so stream order as the caller sees it is unchanged -- and
 * ronk_ntt_forward_many_dev / _inverse_many_dev spread K independent arrays over the two lanes.  -1 = automatic = 1
 * (measured: the lanes pay for independent arrays, not for the halves of one batched launch).
 * The reference has no counterpart (its fft() is a
 * single-threaded recursion, src/polynomial/mod.rs:295-323); this is how a caller with many independent polynomials
 * (kzg / Reed-Solomon batches) gets the concurrent rate without managing streams and plans itself. */
typedef struct ronk_plan_opts {
  int tile_log2_columns;        /* as in ronk_plan_create_tuned */
  int twiddle_matrix_log2_max;  /* as in ronk_plan_create_tuned */
  int in_flight;                /* -1 auto, 1, 2 */
  int split_log2_rows;          /* two-pass plans (2^13 .. 2^24): log2 of the first pass's rows, 0 = the planner's (balanced)
                                   choice; values that leave a pass outside 2^4 .. 2^12 rows are ignored */
  int three_pass_from_log2;     /* 0 = the planner's choice (23; 24 for ONE transform of 2^23); 13 .. 25: the smallest log2n that is
                                   split in THREE passes (the fused multiply asks for two-pass plans of a batch of two at 2^23).
                                   (Round 6: a named field in the place of reserved[0], which carried this knob unnamed -- same
                                   layout, same size.) */
  int reserved[3];              /* zero */
} ronk_plan_opts;
#define RONK_PLAN_OPTS_DEFAULT { -1, -1, -1, 0, 0, { 0, 0, 0 } }
int ronk_plan_create_opts(ronk_plan** out, uint64_t p, uint64_t g, uint32_t log2n, uint64_t batch, int device,
                          const ronk_plan_opts* opts);
/* 1 or 2: the lanes the plan actually uses (see ronk_plan_opts::in_flight) */
int ronk_plan_in_flight(const ronk_plan* plan);
int ronk_plan_destroy(ronk_plan* plan);
/* Which kernel family the plan runs on (2^4 <= n <= 2^30 for the tiled ones):
 *   1 = the tiled Goldilocks path (p = 2^64 - 2^32 + 1 with the explicit generator 7): shift twiddles inside a register round;
 *   2 = the SAME tile kernels over Montgomery arithmetic (R = 2^64): any other odd prime p < 2^64 -- PrimeField<P> is generic
 *       over P, src/algebra/field/prime/mod.rs:39-52 -- and Goldilocks with another generator, whenever g is a quadratic
 *       non-residue (then omega_n = g^((p-1)/n) has order exactly n for every power of two n | p - 1).  Coefficients stay
 *       canonical; twiddle tables hold w * 2^64 mod p.  About twice the arithmetic of path 1 per coefficient;
 *   0 = the radix-2 path (n < 16, n > 2^30, or a g that generates no full 2-power subgroup -- the reference's recursion
 *       src/polynomial/mod.rs:295-323 is then not the DFT and is restated stage by stage): one HBM pass per stage, n <= 2^32. */
int ronk_plan_path(const ronk_plan* plan);

/* Polynomial::<Monomial,F,D>::fft() (polynomial/mod.rs:273-323; same values as dft() :240-258).
 * `nodes`, if non-NULL, receives Lagrange::nodes = [omega^i] (mod.rs:358-365), n elements. */
int ronk_ntt_forward(ronk_plan* plan, const uint64_t* in, uint64_t* out, uint64_t* nodes);
/* Polynomial::<Lagrange<F>,F,D>::ifft() (polynomial/mod.rs:430-484), includes the D^-1 scale */
int ronk_ntt_inverse(ronk_plan* plan, const uint64_t* in, uint64_t* out);
/* device-resident forms; in == out is allowed; asynchronous on `stream` */
int ronk_ntt_forward_dev(ronk_plan* plan, const uint64_t* d_in, uint64_t* d_out, void* stream);
int ronk_ntt_inverse_dev(ronk_plan* plan, const uint64_t* d_in, uint64_t* d_out, void* stream);
/* `count` independent arrays of [batch][n] elements each in one call (Polynomial::fft / ifft of `count` unrelated
 * polynomials, src/polynomial/mod.rs:273-292, :430-453): d_in / d_out are HOST arrays of `count` DEVICE pointers.
 * With in_flight = 2 the arrays alternate between the caller's stream and the plan's side stream (second scratch);
 * asynchronous on `stream`, which sees the results of all of them in stream order. */
int ronk_ntt_forward_many_dev(ronk_plan* plan, const uint64_t* const* d_in, uint64_t* const* d_out, size_t count,
                              void* stream);
int ronk_ntt_inverse_many_dev(ronk_plan* plan, const uint64_t* const* d_in, uint64_t* const* d_out, size_t count,
                              void* stream);
/* Lagrange::<F>::new's node table [omega^i], i < n (polynomial/mod.rs:358-365) */
int ronk_lagrange_nodes(uint64_t p, uint64_t g, uint64_t* nodes, size_t n);

/* One-shot host-pointer forms of the two calls above for a single polynomial, the direct analogues of
 * `poly.fft()` / `lagrange.ifft()`: plans (twiddles, scratch) are kept in an internal LRU cache.
 * n not a power of two -> RONK_ERR_NOT_POW2; n does not divide p-1 -> RONK_ERR_NO_ROOT. */
int ronk_fft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, uint64_t* nodes, size_t n);
int ronk_ifft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n);

/* Polynomial::dft() for ANY n dividing p-1 (polynomial/mod.rs:240-258), e.g. n = 3, 5, 7, 25.
 * Direct O(n^2) kernel; power-of-two n >= 16 over Goldilocks is routed to the NTT. n <= 2^16. */
int ronk_dft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n);

/* plan introspection for benchmarks: number of kernel launches per transform, and the
 * average device time of each launch over `iters` forward (inverse != 0: inverse) transforms
 * measured with hipEvents on `stream`.  ms must hold ronk_plan_num_passes() floats. */
int ronk_plan_num_passes(const ronk_plan* plan);
int ronk_plan_time_passes(ronk_plan* plan, const uint64_t* d_in, uint64_t* d_out, int inverse, int iters,
                          float* ms, void* stream);

/* ---- polynomial arithmetic: src/polynomial/arithmetic.rs ---- */

/* impl Mul (arithmetic.rs:97-119): out has d + d2 - 1 coefficients.  Goldilocks: NTT -> pointwise
 * -> inverse NTT on the padded size; other primes: schoolbook kernel.  d, d2 >= 1. */
int ronk_poly_mul(uint64_t p, uint64_t g, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out);
int ronk_poly_mul_dev(uint64_t p, uint64_t g, const uint64_t* d_a, size_t d, const uint64_t* d_b, size_t d2,
                      uint64_t* d_out, void* stream);
/* impl Add / Sub (arithmetic.rs:16-68): rhs zero-extended or truncated to d = len(lhs) */
int ronk_poly_add(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out);
int ronk_poly_sub(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out);

/* ---- callers either side of the path ---- */

/* Polynomial::<Monomial>::evaluate (polynomial/mod.rs:133-139): sum c_i x^i */
int ronk_poly_eval(uint64_t p, const uint64_t* c, size_t d, uint64_t x, uint64_t* out);
/* same, coefficients resident in HBM, result (ONE element) written to device memory, asynchronous on `stream`
 * (a hipStream_t).  One 8 B/coefficient read of the array (chunked Horner, csrc/scan_kernels.h). */
int ronk_poly_eval_dev(uint64_t p, const uint64_t* d_c, size_t d, uint64_t x, uint64_t* d_out, void* stream);
/* Polynomial::<Lagrange<F>>::evaluate (polynomial/mod.rs:382-415): barycentric evaluation at x from the
 * values c[j] at nodes[j].  As in the reference, x equal to a node yields ZERO (its fold multiplies by
 * l(x) = 0); coincident nodes -> RONK_ERR_ZERO_INVERSE.  n <= 2^16 (O(n^2) weights, like the reference). 
 * More than 2^16 nodes: only node tables of the form Lagrange::new builds (nodes[i] = omega^i, omega of order n; any n | p-1)
 * -- then prod_{m != j}(x_j - x_m) = n / x_j and prod_i (x - x_i) = x^n - 1 give the same value in O(n); other tables of
 * that size are RONK_ERR_UNSUPPORTED (the _dev form sets bit 2 of *d_status, which it then requires). */
int ronk_lagrange_eval(uint64_t p, const uint64_t* c, const uint64_t* nodes, size_t n, uint64_t x, uint64_t* out);
/* quotient_and_remainder (polynomial/mod.rs:170-225) behind impl Div / Rem (arithmetic.rs:121-146);
 * quot and rem both have d coefficients.  Used by kzg::open (src/kzg/setup.rs:63-78). */
int ronk_poly_divrem(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* quot,
                     uint64_t* rem);
/* kzg::open's polynomial step on device (src/kzg/setup.rs:63-78: `poly.div([-z, 1])`): division by the linear
 * divisor b0 + b1*x, b1 != 0, as an affine suffix scan.  d_quot receives d coefficients (the top one ZERO, like the
 * reference's D-long quotient); d_rem (may be NULL) receives ONE element, the remainder's constant coefficient (its
 * other coefficients are ZERO).  d_quot may be d_c itself (the quotient written over the dividend); any other overlap
 * of the two is not allowed.  Up to 2^23 coefficients, out of place, with a 16-byte aligned dividend and b0 != 0, the call is ONE
 * launch that moves the algorithmic 16 bytes per coefficient; otherwise two launches (24 bytes).  Asynchronous on
 * `stream`.  NOT for hipGraph capture (nor is ronk_poly_eval_dev): the
 * workspace comes from an event-guarded pool whose slot would be baked into the graph while later calls reuse it; capture
 * the plan entry points (ronk_ntt_forward_dev / inverse_dev with a plan the graph owns) instead. */
int ronk_poly_div_linear_dev(uint64_t p, const uint64_t* d_c, size_t d, uint64_t b0, uint64_t b1, uint64_t* d_quot,
                             uint64_t* d_rem, void* stream);
/* Reed-Solomon Message::encode::<N> (src/codes/reed_solomon.rs:42-52): xs[i] = omega_N^i,
 * ys[i] = poly(omega_N^i) -- a size-N DFT of the zero-padded K-coefficient message. */
int ronk_rs_encode(uint64_t p, uint64_t g, const uint64_t* msg, size_t k, size_t n, uint64_t* xs, uint64_t* ys);
/* Reed-Solomon Message::decode (src/codes/reed_solomon.rs:54-106): Lagrange interpolation through the first k
 * coordinates (xs[j], ys[j]) of a (possibly erased) codeword -> the k message coefficients.  Coincident nodes are
 * the reference's `numerator / denominator` panic -> RONK_ERR_ZERO_INVERSE.  k <= 2^14 (O(k^2) work) for arbitrary nodes; for
 * the node sequences Message::encode produces (xs[j] = q^j, q of any order > k; Goldilocks) an O(k log k) form -- two
 * convolutions on the NTT path, the same interpolating polynomial -- is chosen on the device from k = 1024 on and covers
 * k <= 2^21; other node sets of that size are RONK_ERR_UNSUPPORTED (the _dev form: bit 2 of *d_status, then required). */
int ronk_rs_decode(uint64_t p, const uint64_t* xs, const uint64_t* ys, size_t k, uint64_t* out);
/* Batched Message::encode::<N> on device (src/codes/reed_solomon.rs:42-52), the production shape of a
 * Reed-Solomon / low-degree extension (1024 x 2^16): d_msgs holds plan.batch compact messages of k coefficients,
 * d_ys receives plan.batch x N y-coordinates (x_i = omega_N^i: ronk_lagrange_nodes).  The zero padding of
 * `Polynomial::from(message)` is implicit (no padded copy) for Goldilocks plans with N >= 2^13. */
int ronk_rs_encode_batch_dev(ronk_plan* plan, const uint64_t* d_msgs, size_t k, uint64_t* d_ys, void* stream);
/* Low-degree extension: a batch of polynomials given by their values on {omega_K^i} (plan_k: n = K) -> their values on
 * coset_shift * {omega_N^i} (plan_n: n = N >= K, same batch and modulus).  = Message::encode::<N> of lagrange_poly.ifft()
 * (src/polynomial/mod.rs:430-453, src/codes/reed_solomon.rs:42-52), the coefficients multiplied by coset_shift^i first when
 * coset_shift != 1 (any field; 0 is RONK_ERR_UNSUPPORTED).  d_coeffs: batch x K scratch that receives the coefficients; d_out: batch x N. */
int ronk_lde_batch_dev(ronk_plan* plan_k, ronk_plan* plan_n, const uint64_t* d_evals, uint64_t* d_coeffs, uint64_t* d_out,
                       uint64_t coset_shift, void* stream);

/* kzg::commit (src/kzg/setup.rs:45-60): sum_i points[i] * scalars[i] with the reference's AffinePoint Add / Mul<ScalarField>
 * (src/curve/mod.rs:152-211) on y^2 = x^3 + a x + b over the quadratic extension F_p[u]/(u^2 - nr) of a small prime
 * field (p < 2^32; PlutoExtendedCurve: p = 101, nr = 99 (X^2 + 2), a = 0, b = 3 -- src/curve/pluto_curve.rs:39-51,
 * src/algebra/field/extension/gf_101_2.rs:12-18).  A point is 5 words: x0 x1 y0 y1 inf (inf != 0: Infinity).
 * n_points < n is the reference's assert (RONK_ERR_INDEX); an off-curve point is AffinePoint::new's panic
 * (RONK_ERR_NOT_ON_CURVE).  kzg::open = ronk_poly_divrem by [-z, 1] over the scalar field, then this. */
typedef struct ronk_curve { uint64_t p, nr, a, b; } ronk_curve;
int ronk_curve_msm(const ronk_curve* curve, const uint64_t* points, size_t n_points, const uint64_t* scalars, size_t n,
                   uint64_t out[5]);

/* kzg::commit on a production-size curve: sum_i scalars[i] * points[i] over BN254 (alt_bn128) G1,
 * y^2 = x^3 + 3 over F_p, p = 21888242871839275222246405745257275088696311157297823662689037894645226208583, by the bucket
 * method on the GPU (csrc/msm_kernels.h).  The reference's commit is the same sum as a fold of AffinePoint Mul / Add over
 * its 17-element toy group (src/kzg/setup.rs:48-60, src/curve/mod.rs:157-211); its field traits are usize-wide
 * (src/algebra/mod.rs:8-13), so a 254-bit curve is a new type on the Rust side.
 * points: n x 8 words -- x then y, each 4 x 64-bit little-endian limbs, standard (non-Montgomery) form, < p; (0, 0) is the
 * point at infinity.  scalars: n x 4 words, any 256-bit integers (e.g. residues mod the group order r).  out: 8 words, same
 * encoding as a point.  RONK_ERR_NOT_ON_CURVE: a coordinate >= p or y^2 != x^3 + 3 (AffinePoint::new's assert,
 * src/curve/mod.rs:79).  The _dev form takes device-resident points / scalars and a HOST result pointer: it enqueues
 * on `stream`, waits for it, and finishes the last ~270 dependent doublings on the host. */
int ronk_msm_bn254(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t out[8]);
int ronk_msm_bn254_dev(const uint64_t* d_points, const uint64_t* d_scalars, size_t n, uint64_t out[8], void* stream);

/* kzg::open on the same curve (src/kzg/setup.rs:63-78): `poly.div([-eval_point, ONE])` over the SCALAR field of BN254,
 * r = 21888242871839275222246405745257275088548364400416034343698204186575808495617, then `commit(quotient, g1_srs)`.
 * coeffs: n x 4 words (4 x 64-bit little-endian limbs, standard form; taken mod r), ascending degree.  z: 4 words.
 * ronk_poly_div_linear_bn254_dev: the division alone -- quotient_and_remainder with a monic linear divisor
 * (src/polynomial/mod.rs:170-225): d_quot receives n entries, the top one ZERO (the reference's D-long quotient), d_rem
 * (device, 4 words, may be NULL) the remainder's constant term poly(z).  A suffix scan over 256-bit elements
 * (csrc/fr_scan_kernels.h, csrc/bn254_fr.h); synchronises `stream` (the multiplier tables are per call).
 * ronk_kzg_open_bn254(_dev): the division followed by ronk_msm_bn254_dev over (srs, quotient) -- the opening proof -- and
 * poly(z) in out_value (may be NULL).  srs: n x 8 words (points as for ronk_msm_bn254); n_srs < n is the reference's
 * assert (RONK_ERR_INDEX); the _dev form takes device-resident coefficients / SRS and an n x 4-word device buffer for
 * the quotient. */
int ronk_poly_div_linear_bn254_dev(const uint64_t* d_coeffs, size_t n, const uint64_t z[4], uint64_t* d_quot, uint64_t* d_rem,
                                   void* stream);
int ronk_kzg_open_bn254_dev(const uint64_t* d_coeffs, size_t n, const uint64_t z[4], const uint64_t* d_srs, uint64_t* d_quot,
                            uint64_t out_point[8], uint64_t out_value[4], void* stream);
int ronk_kzg_open_bn254(const uint64_t* coeffs, size_t n, const uint64_t z[4], const uint64_t* srs, size_t n_srs,
                        uint64_t out_point[8], uint64_t out_value[4]);

/* ---- multi-GPU four-step building blocks (one process per GPU; the exchange between the two
 *      phases is an RCCL all-to-all issued by the host side, see ronkathon_amd/dist.py) ----
 * n = 2^log2n split as R x C with R = 2^(log2n - log2n/2) rows and C = 2^(log2n/2) columns;
 * rank `rank` of `world` owns columns [rank*C/world, (rank+1)*C/world) of the R x C input
 * (layout [R][C/world], row-major) and, after the exchange, rows [rank*R/world, ...) of the
 * twiddled intermediate (layout [R/world][C]); its output block is X[k1 + R*k2] for its k1
 * range, laid out [C][R/world] (k2-major). */
typedef struct ronk_dist_plan ronk_dist_plan;
int ronk_dist_plan_create(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world, int device);
int ronk_dist_plan_destroy(ronk_dist_plan* plan);
/* The same with the exchange split in `chunks` column chunks (a power of two, >= 16 columns per chunk): phase 1 of
 * chunk j writes the contiguous piece d_send[j*R*Cwc ..), Cwc = C/world/chunks, as `world` blocks [R/world][Cwc] (block h
 * for rank h), so the caller can ship chunk j while chunk j+1 is computed; the receiver stores the block of (source rank g,
 * chunk j) at d_recv[(g*chunks + j)*(R/world)*Cwc ..).  chunks = 1 is ronk_dist_plan_create. */
int ronk_dist_plan_create_chunked(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world, int device,
                                  int chunks);
/* The same over ANY field the single-GPU plans cover (round 6): p an odd prime with 2^log2n | p - 1 and g a primitive element
 * (a quadratic non-residue suffices; otherwise RONK_ERR_UNSUPPORTED -- the four-step has no radix-2 fallback); omega =
 * g^((p-1)/n) as in PrimeField<P> (src/algebra/field/mod.rs:70-75, prime/mod.rs:39-52).  The phases run the tile kernels over
 * Montgomery arithmetic.  ronk_dist_plan_create(_chunked) are the shorthands for (RONK_GOLDILOCKS_P, RONK_GOLDILOCKS_G). */
int ronk_dist_plan_create_p(ronk_dist_plan** out, uint64_t p, uint64_t g, uint32_t log2n, int inverse, int rank, int world,
                            int device, int chunks);
int ronk_dist_phase1_chunk_dev(ronk_dist_plan* plan, int chunk, const uint64_t* d_in, uint64_t* d_send, void* stream);
/* phase 1: R-point NTTs down the local columns, times omega_n^{c*k1}; output is written as `world`
 * consecutive send blocks, block h = rows k1 in h's range, layout [R/world][C/world] */
int ronk_dist_phase1_dev(ronk_dist_plan* plan, const uint64_t* d_in, uint64_t* d_send, void* stream);
/* phase 2: d_recv holds `world` blocks [R/world][C/world] (block g from rank g); C-point NTTs along
 * each local row k1; d_out[k2*(R/world) + (k1 - k1_0)] = X[k1 + R*k2] */
int ronk_dist_phase2_dev(ronk_dist_plan* plan, const uint64_t* d_recv, uint64_t* d_out, void* stream);

/* ---- device-resident forms of the callers above: no allocation, copy or synchronisation per call (workspace from an
 *      event-guarded pool), asynchronous on `stream`.  Conditions the reference reports by panicking are reported through a
 *      caller-owned device word `d_status` (the caller zeroes it and reads it when it synchronises; NULL where noted =
 *      "do not care"): non-zero = RONK_ERR_ZERO_INVERSE unless stated otherwise.  Inputs must be canonical residues. ---- */
int ronk_vec_neg_dev(uint64_t p, const uint64_t* d_a, uint64_t* d_out, size_t n, void* stream);
int ronk_vec_pow_dev(uint64_t p, const uint64_t* d_a, uint64_t e, uint64_t* d_out, size_t n, void* stream);
int ronk_vec_inv_dev(uint64_t p, const uint64_t* d_a, uint64_t* d_out, size_t n, int* d_status, void* stream);
/* Polynomial::dft (polynomial/mod.rs:240-258) for any n | p-1; see ronk_dft for the size limits */
int ronk_dft_dev(uint64_t p, uint64_t g, const uint64_t* d_in, uint64_t* d_out, size_t n, void* stream);
/* Polynomial::<Lagrange<F>>::evaluate (polynomial/mod.rs:382-415); d_out = ONE element; d_status may be NULL */
int ronk_lagrange_eval_dev(uint64_t p, const uint64_t* d_c, const uint64_t* d_nodes, size_t n, uint64_t x, uint64_t* d_out,
                           int* d_status, void* stream);
/* quotient_and_remainder (polynomial/mod.rs:170-225), any prime, any divisor; *d_status (required) receives 0 or the
 * RONK_ERR_* code of the reference's panic; d_rem may alias d_a.  The long-division kernel follows the reference's loop
 * (one workgroup, d * d2 steps).  Goldilocks, d2 >= 64 and d - d2 + 1 >= 2048: the operands' degrees are read back first
 * (the ONE exception to "no synchronisation per call": one stream synchronisation, 24 bytes) and, for a full-length divisor,
 * the O(n log n) Newton form on the NTT path runs, as behind ronk_poly_divrem.  A capturing stream keeps the long division. */
int ronk_poly_divrem_dev(uint64_t p, const uint64_t* d_a, size_t d, const uint64_t* d_b, size_t d2, uint64_t* d_quot,
                         uint64_t* d_rem, int* d_status, void* stream);
/* Round 6: the O(n log n) form serves EVERY odd prime whose p - 1 has the 2-adicity of the product sizes (2^(ceil(log2 d) + 1)
 * divides p - 1), not only Goldilocks -- behind ronk_poly_divrem and ronk_poly_divrem_dev alike; the products' transform
 * root is any quadratic non-residue found by the library (a product does not depend on it), so no generator is asked for.
 * Under stream capture ronk_poly_divrem_dev cannot probe the degrees: it captures the long division while that is a matter
 * of milliseconds (d2 * (d - d2 + 1) <= 1e9) and returns RONK_ERR_UNSUPPORTED beyond.
 *
 * quotient_and_remainder for FULL-LENGTH operands (a[d-1] != 0, b[d2-1] != 0, d >= d2): the same O(n log n) form with nothing
 * read back -- the divisor's leading coefficient is inverted on the device and the promise is checked there (*d_status =
 * RONK_ERR_INVALID when a top coefficient is ZERO: the outputs are then meaningless) -- so the call is asynchronous on `stream`
 * and capturable at any size (warm the workspace with one call outside the capture).  Fields as above, else
 * RONK_ERR_UNSUPPORTED.  For full-length operands the reference's loop is plain Euclidean division (mod.rs:170-225). */
int ronk_poly_divrem_full_dev(uint64_t p, const uint64_t* d_a, size_t d, const uint64_t* d_b, size_t d2, uint64_t* d_quot,
                              uint64_t* d_rem, int* d_status, void* stream);
/* Message::decode (src/codes/reed_solomon.rs:54-106); d_status may be NULL */
int ronk_rs_decode_dev(uint64_t p, const uint64_t* d_xs, const uint64_t* d_ys, size_t k, uint64_t* d_out, int* d_status,
                       void* stream);
/* kzg::commit (src/kzg/setup.rs:45-60); *d_status (required): bit 0 = RONK_ERR_NOT_ON_CURVE, bit 1 = RONK_ERR_ZERO_INVERSE.
 * With ronk_poly_div_linear_dev, kzg::open (setup.rs:63-78) never leaves the device. */
int ronk_curve_msm_dev(const ronk_curve* curve, const uint64_t* d_points, size_t n_points, const uint64_t* d_scalars, size_t n,
                       uint64_t* d_out, int* d_status, void* stream);

/* ---- the sharded transform as ONE call for a single-process host (the Rust host of BASELINE config 5): rank g of
 *      ndev = devices[g]; the exchange is a mesh of peer copies over xGMI issued by the library on per-peer copy
 *      streams, in `chunks` column chunks so that a chunk travels while the next one is computed (chunks <= 0: default,
 *      up to 4).  The reference has no counterpart: `Polynomial<B, F, D>` holds its coefficients inline
 *      (src/polynomial/mod.rs:34-44), so a degree this large never exists there.
 *      Errors: RONK_ERR_UNSUPPORTED (fewer than 16 rows / columns per rank and chunk, ndev not a power of two),
 *      RONK_ERR_INVALID (device ordinal out of range), RONK_ERR_HIP. */
typedef struct ronk_sharded_plan ronk_sharded_plan;
int ronk_sharded_plan_create(ronk_sharded_plan** out, uint32_t log2n, int inverse, const int* devices, int ndev, int chunks);
/* The same with the exchange chosen per plan: RONK_EXCHANGE_MESH = hipMemcpyPeerAsync copies, one copy stream per peer
 * (all xGMI links of a device busy at once); RONK_EXCHANGE_RCCL = every chunk's W x W blocks as one ncclGroup of
 * ncclSend / ncclRecv pairs (librccl.so is dlopen()ed on first use: RONK_ERR_RCCL if it is missing or a call fails;
 * ranks must be on distinct devices, else RONK_ERR_UNSUPPORTED).  Same results either way. */
#define RONK_EXCHANGE_MESH 0
#define RONK_EXCHANGE_RCCL 1
int ronk_sharded_plan_create_ex(ronk_sharded_plan** out, uint32_t log2n, int inverse, const int* devices, int ndev, int chunks,
                                int exchange);
/* The same over any field ronk_dist_plan_create_p takes (round 6): (p, g) as there; every rank's phases run the tile kernels
 * over Montgomery arithmetic.  ronk_sharded_plan_create(_ex) are the shorthands for the Goldilocks field. */
int ronk_sharded_plan_create_p(ronk_sharded_plan** out, uint64_t p, uint64_t g, uint32_t log2n, int inverse, const int* devices,
                               int ndev, int chunks, int exchange);
int ronk_sharded_plan_exchange(const ronk_sharded_plan* plan);
/* Diagnostics (bench.py --workload sharded): ONE transform in three SERIALISED stages -- every rank's phase 1, the whole
 * exchange, every rank's phase 2 -- all devices drained between them; ms[0..2] = wall milliseconds per stage.  The achieved
 * rate per directed link is n * 8 / ndev^2 bytes / ms[1].  Same d_out as ronk_ntt_sharded_dev (which overlaps the stages). */
int ronk_sharded_time_stages(ronk_sharded_plan* plan, const uint64_t* const* d_in, uint64_t* const* d_out, float* ms);
/* How a block travels between the ranks of a mesh-exchange plan, decided at plan creation and kept (never silent):
 * matrix[g * ndev + h] = RONK_PEER_SAME_DEVICE (ranks g and h share a GPU), RONK_PEER_DIRECT (hipDeviceCanAccessPeer said yes and
 * peer access is enabled: xGMI / PCIe peer-to-peer) or RONK_PEER_STAGED (refused: hipMemcpyPeerAsync stages through host
 * memory -- correct, roughly an order of magnitude slower).  `matrix` may be NULL; capacity >= ndev * ndev otherwise.
 * Returns the number of STAGED pairs (0 on a healthy xGMI node), or a negative error.  RONK_REQUIRE_PEER=1 in the environment
 * makes ronk_sharded_plan_create(_ex) fail with RONK_ERR_UNSUPPORTED instead of accepting a staged pair.
 * (The reference has no counterpart: it is single-threaded CPU code) */
#define RONK_PEER_SAME_DEVICE 0
#define RONK_PEER_DIRECT 1
#define RONK_PEER_STAGED 2
int ronk_sharded_plan_peer_access(const ronk_sharded_plan* plan, int* matrix, int capacity);
int ronk_sharded_plan_destroy(ronk_sharded_plan* plan);
/* R, C (n = R*C), elements per rank (n / ndev) and the number of column chunks in use; any pointer may be NULL */
int ronk_sharded_plan_info(const ronk_sharded_plan* plan, uint64_t* rows, uint64_t* cols, uint64_t* per_rank, int* chunks);
/* device-resident: d_in[g] = rank g's [R][C/ndev] column block on devices[g], d_out[g] = its [C][R/ndev] block of the
 * natural-order result (layouts as for ronk_dist_*).  Enqueues on the plan's own streams and returns; the buffers may
 * be reused after ronk_sharded_sync().  Successive calls pipeline (events guard the plan's send/receive buffers). */
int ronk_ntt_sharded_dev(ronk_sharded_plan* plan, const uint64_t* const* d_in, uint64_t* const* d_out);
int ronk_sharded_sync(ronk_sharded_plan* plan);
/* host pointers, natural order in and out (n elements each): scatter, transform, gather; synchronous */
int ronk_ntt_sharded(ronk_sharded_plan* plan, const uint64_t* in, uint64_t* out);

/* ---- small device-memory helpers so a non-HIP host (ctypes, cgo, JNI) can stay device-resident ---- */
int ronk_dev_alloc(void** ptr, size_t bytes);
int ronk_dev_free(void* ptr);
int ronk_memcpy_h2d(void* dst, const void* src, size_t bytes);
int ronk_memcpy_d2h(void* dst, const void* src, size_t bytes);
int ronk_dev_sync(void);
/* The device the helpers above act on (the calling thread's current HIP device; plans carry their own ordinal).  A host that
 * places one block per GPU (device::DevicePoly of the Rust shim, the sharded transform's per-rank blocks) selects it around
 * every alloc / copy / free.  RONK_ERR_INVALID for an ordinal outside [0, ronk_device_count). */
int ronk_set_device(int device);
int ronk_get_device(int* device);
/* Releases the library's cached device workspace (the event-guarded buffer pool behind the Newton division, the fast
 * Reed-Solomon decode and the scans) once the work that used it has finished; buffers above 256 MiB are never cached.
 * Safe at any time; the next call that needs workspace allocates again. */
int ronk_trim_workspace(void);

#ifdef __cplusplus
}
#endif
#endif
