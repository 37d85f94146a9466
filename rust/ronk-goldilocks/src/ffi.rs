//! `extern "C"` block for include/ronk_ntt.h (the subset the shim calls) and the code -> panic mapping.
//!
//! Every function returns 0 or a negative `RONK_ERR_*`; the reference reports the same conditions by panicking, so
//! [`check`] panics with `ronk_strerror(code)`, which repeats the reference's panic texts ("n must divide p^q - 1",
//! `called Option::unwrap() on a None value`, ...): `#[should_panic]` tests such as
//! ronkathon `src/polynomial/tests.rs:46-55` and `src/algebra/field/prime/mod.rs:386-391` keep passing.
use core::ffi::{c_char, c_int};

pub const P: u64 = 0xFFFF_FFFF_0000_0001;
pub const G: u64 = 7;

extern "C" {
  pub fn ronk_strerror(code: c_int) -> *const c_char;
  pub fn ronk_last_hip_error() -> *const c_char;
  /// `Polynomial::<Monomial,F,D>::fft` (src/polynomial/mod.rs:273-323); `nodes` may be null
  pub fn ronk_fft(p: u64, g: u64, input: *const u64, output: *mut u64, nodes: *mut u64, n: usize) -> c_int;
  /// `Polynomial::<Lagrange<F>,F,D>::ifft` (src/polynomial/mod.rs:430-484)
  pub fn ronk_ifft(p: u64, g: u64, input: *const u64, output: *mut u64, n: usize) -> c_int;
  /// `Polynomial::dft` for any n | p-1 (src/polynomial/mod.rs:240-258)
  pub fn ronk_dft(p: u64, g: u64, input: *const u64, output: *mut u64, n: usize) -> c_int;
  /// `Lagrange` node table [omega^i] (src/polynomial/mod.rs:358-365)
  pub fn ronk_lagrange_nodes(p: u64, g: u64, nodes: *mut u64, n: usize) -> c_int;
  /// `impl Mul for Polynomial` (src/polynomial/arithmetic.rs:97-119): d + d2 - 1 outputs
  pub fn ronk_poly_mul(p: u64, g: u64, a: *const u64, d: usize, b: *const u64, d2: usize, out: *mut u64) -> c_int;
  /// `quotient_and_remainder` (src/polynomial/mod.rs:170-225): quot and rem have d coefficients each
  pub fn ronk_poly_divrem(p: u64, a: *const u64, d: usize, b: *const u64, d2: usize, quot: *mut u64, rem: *mut u64) -> c_int;
  /// `Polynomial::<Monomial>::evaluate` (src/polynomial/mod.rs:133-139)
  pub fn ronk_poly_eval(p: u64, c: *const u64, d: usize, x: u64, out: *mut u64) -> c_int;
  /// `Polynomial::<Lagrange<F>>::evaluate` (src/polynomial/mod.rs:382-415)
  pub fn ronk_lagrange_eval(p: u64, c: *const u64, nodes: *const u64, n: usize, x: u64, out: *mut u64) -> c_int;
  /// `Message::decode` (src/codes/reed_solomon.rs:54-106)
  pub fn ronk_rs_decode(p: u64, xs: *const u64, ys: *const u64, k: usize, out: *mut u64) -> c_int;
  /// `kzg::commit` (src/kzg/setup.rs:48-60) over BN254 G1: points n x [x: 4 limbs, y: 4 limbs], scalars n x 4 limbs
  pub fn ronk_msm_bn254(points: *const u64, scalars: *const u64, n: usize, out: *mut u64) -> c_int;
}

/// 0 -> (), anything else -> the reference's panic
#[inline]
pub fn check(rc: c_int) {
  if rc != 0 {
    let msg = unsafe { std::ffi::CStr::from_ptr(ronk_strerror(rc)) }.to_string_lossy().into_owned();
    if rc == -8 {
      let hip = unsafe { std::ffi::CStr::from_ptr(ronk_last_hip_error()) }.to_string_lossy().into_owned();
      panic!("{msg}: {hip}");
    }
    panic!("{msg}");
  }
}
