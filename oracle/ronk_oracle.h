/*
 * ronk_oracle.h -- CPU restatement of ronkathon's prime-field / polynomial hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ronkathon_amd/ (the product) may include,
 * link or dlopen this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * Every function restates the RESULT semantics of the reference (canonical residues
 * in [0,p), natural-order transforms with omega = g^((p-1)/n), output lengths, panic
 * conditions), citing the reference file:line (paths relative to /root/reference).
 * Products are widened to 128 bit so the restatement is also correct for p >= 2^32,
 * where the reference's `usize` arithmetic overflows.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against every golden
 * vector the reference's own tests hold for the path.  The
 * reference is Rust (nightly-2024-06-10) and no Rust toolchain exists in this image,
 * so there is no oracle/_ref build; see DESIGN.md "Oracle".
 */
#ifndef RONK_ORACLE_H
#define RONK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes: 0 ok; negative = the reference would panic here */
#define ORC_OK 0
#define ORC_PANIC_NO_ROOT (-1)      /* assert!(p_minus_one % n == 0), field/mod.rs:72 */
#define ORC_PANIC_ZERO_INVERSE (-2) /* inverse().unwrap() on zero, prime/arithmetic.rs:54 */
#define ORC_PANIC_NOT_POW2 (-3)     /* fft/ifft compile-time bound, polynomial/mod.rs:274 */
#define ORC_PANIC_NOT_PRIME (-4)    /* is_prime panic, prime/mod.rs:92-100 */
#define ORC_PANIC_NO_GENERATOR (-5) /* find_primitive_element panic, prime/mod.rs:122 */
#define ORC_PANIC_INDEX (-6)        /* slice index out of bounds / unwrap on None */
#define ORC_PANIC_NOT_RESIDUE (-13) /* assert!(self.euler_criterion(), "Element is not a quadratic residue"), prime/mod.rs:179 */
#define ORC_PANIC_NOT_ON_CURVE (-11) /* assert!(point.is_on_curve(), "Point is not on curve"), curve/mod.rs:79 */

/* ---- prime field: src/algebra/field/prime/{mod,arithmetic}.rs ---- */
int orc_is_prime(uint64_t p);                                   /* prime/mod.rs:92-100 */
int orc_find_primitive_element(uint64_t p, uint64_t* g);        /* prime/mod.rs:110-123 */
uint64_t orc_new(uint64_t p, uint64_t v);                       /* prime/mod.rs:48-51 */
uint64_t orc_add(uint64_t p, uint64_t a, uint64_t b);           /* prime/arithmetic.rs:3-7 */
uint64_t orc_sub(uint64_t p, uint64_t a, uint64_t b);           /* prime/arithmetic.rs:19-28 */
uint64_t orc_neg(uint64_t p, uint64_t a);                       /* prime/arithmetic.rs:61-65 */
uint64_t orc_mul(uint64_t p, uint64_t a, uint64_t b);           /* prime/arithmetic.rs:34-38 */
uint64_t orc_pow(uint64_t p, uint64_t a, uint64_t e);           /* prime/mod.rs:74-84 */
int orc_inverse(uint64_t p, uint64_t a, uint64_t* out);         /* prime/mod.rs:62-72 */
int orc_div(uint64_t p, uint64_t a, uint64_t b, uint64_t* out); /* prime/arithmetic.rs:50-55 */
int orc_rem(uint64_t p, uint64_t a, uint64_t b, uint64_t* out); /* prime/arithmetic.rs:67-71 */
int orc_primitive_root_of_unity(uint64_t p, uint64_t g, uint64_t n, uint64_t* out); /* field/mod.rs:70-75 */

/* element-wise vector forms of the above (what Polynomial Add/Sub/Neg reduce to) */
void orc_vec_add(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void orc_vec_sub(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void orc_vec_mul(uint64_t p, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void orc_vec_neg(uint64_t p, const uint64_t* a, uint64_t* out, size_t n);
int orc_vec_inv(uint64_t p, const uint64_t* a, uint64_t* out, size_t n);
void orc_vec_pow(uint64_t p, const uint64_t* a, uint64_t e, uint64_t* out, size_t n);
/* FieldExt (field/mod.rs:79-84) */
int orc_euler_criterion(uint64_t p, uint64_t a);                              /* prime/mod.rs:172 */
int orc_sqrt(uint64_t p, uint64_t a, uint64_t* r0, uint64_t* r1);             /* prime/mod.rs:174-226 */
void orc_vec_euler(uint64_t p, const uint64_t* a, uint64_t* out, size_t n);
int orc_vec_sqrt(uint64_t p, const uint64_t* a, uint64_t* r0, uint64_t* r1, size_t n);

/* ---- polynomial: src/polynomial/{mod,arithmetic}.rs ---- */
int orc_lagrange_nodes(uint64_t p, uint64_t g, uint64_t* nodes, size_t n);              /* mod.rs:358-365 */
int orc_dft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n);       /* mod.rs:240-258 */
int orc_fft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n);       /* mod.rs:273-323 */
int orc_ifft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n);      /* mod.rs:430-484 */
void orc_poly_add(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out); /* arithmetic.rs:16-35 */
void orc_poly_sub(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out); /* arithmetic.rs:49-68 */
void orc_poly_neg(uint64_t p, const uint64_t* a, size_t d, uint64_t* out);                                 /* arithmetic.rs:77-95 */
void orc_poly_mul(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2, uint64_t* out); /* arithmetic.rs:97-119, out has d+d2-1 */
int orc_poly_divrem(uint64_t p, const uint64_t* a, size_t d, const uint64_t* b, size_t d2,
                    uint64_t* quot, uint64_t* rem);                                     /* mod.rs:170-225, both length d */
uint64_t orc_poly_eval(uint64_t p, const uint64_t* c, size_t d, uint64_t x);            /* mod.rs:133-139 */
int orc_lagrange_eval(uint64_t p, const uint64_t* c, const uint64_t* nodes, size_t n, uint64_t x, uint64_t* out); /* mod.rs:382-415 */
void orc_pow_mult(uint64_t p, const uint64_t* c, size_t d, size_t d2, uint64_t coeff, uint64_t* out); /* mod.rs:153-157, out has d+d2 */
size_t orc_degree(const uint64_t* c, size_t d);                                         /* mod.rs:113-115 */
uint64_t orc_leading_coefficient(const uint64_t* c, size_t d);                          /* mod.rs:120-122 */
void orc_poly_from(const uint64_t* c, size_t n, uint64_t* out, size_t d);               /* mod.rs:503-515 */

/* ---- callers either side of the path ("next" rows) ---- */
/* Reed-Solomon encode: src/codes/reed_solomon.rs:42-52; x[i]=w^i, y[i]=poly(w^i), i<n */
int orc_rs_encode(uint64_t p, uint64_t g, const uint64_t* msg, size_t k, size_t n, uint64_t* xs, uint64_t* ys);
/* codes/reed_solomon.rs:54-106: interpolate the first k coordinates back to the k message coefficients */
int orc_rs_decode(uint64_t p, const uint64_t* xs, const uint64_t* ys, size_t k, uint64_t* out);
/* KZG open quotient: src/kzg/setup.rs:63-78; poly / (x - z), length d */
int orc_kzg_open_quotient(uint64_t p, const uint64_t* coeffs, size_t d, uint64_t z, uint64_t* quot);

/* ---- CPU baseline helpers (same algorithm as orc_fft, root excluded from timing) ---- */
/* in-place recursive even/odd FFT exactly as fft_recursive (allocating), omega given */
void orc_fft_recursive(uint64_t p, uint64_t* values, size_t n, uint64_t omega);

#ifdef __cplusplus
}
#endif
/* ---- curve arithmetic behind kzg::commit: src/curve/mod.rs, src/kzg/setup.rs:45-60 ----
 * y^2 = x^3 + a x + b over F_p[u]/(u^2 - nr); a point is 5 words x0 x1 y0 y1 inf */
typedef struct orc_curve { uint64_t p, nr, a, b; } orc_curve;
int orc_curve_is_on_curve(const orc_curve* c, const uint64_t pt[5]);                                  /* curve/mod.rs:129-138 */
int orc_curve_add(const orc_curve* c, const uint64_t p1[5], const uint64_t p2[5], uint64_t out[5]);   /* curve/mod.rs:176-211 */
int orc_curve_mul(const orc_curve* c, const uint64_t pt[5], uint64_t k, uint64_t out[5]);             /* curve/mod.rs:152-166 */
int orc_kzg_commit(const orc_curve* c, const uint64_t* srs, size_t n_srs, const uint64_t* coeffs, size_t n, uint64_t out[5]);

#endif
