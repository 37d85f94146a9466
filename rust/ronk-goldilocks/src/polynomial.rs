//! GPU-backed polynomial operations for `Polynomial<_, Goldilocks, D>`, each one the counterpart of a reference item
//! (ronkathon src/polynomial/mod.rs, src/polynomial/arithmetic.rs) with the same signature and panics.
//!
//! As a separate crate these can only be offered through the extension trait [`Accelerated`] (Rust's orphan rule forbids
//! `impl Mul for Polynomial<..>` outside ronkathon, and inherent methods cannot be added from outside).  Vendored in-tree
//! (feature `in_tree`, see in_tree/README.md) the same bodies become specialisations, so call sites such as
//! `src/kzg/setup.rs:63-78` and `src/codes/reed_solomon.rs:42-106` keep calling `poly.fft()`, `a * b`, `a / b`, `p.evaluate(x)`.
//!
//! Pointer casts: `Goldilocks` is `repr(transparent)` over `u64`, so `[Goldilocks; D]` IS `[u64; D]`.
//! Plans (twiddle tables, scratch) live in the library's LRU cache keyed by (p, g, n): nothing is created per call.
use ronkathon::{
  algebra::field::Field,
  polynomial::{Lagrange, Monomial, Polynomial},
};

use crate::{
  ffi::{self, check, G, P},
  field::Goldilocks,
};

#[inline]
fn cptr<const D: usize>(a: &[Goldilocks; D]) -> *const u64 { a.as_ptr() as *const u64 }
#[inline]
fn mptr<const D: usize>(a: &mut [Goldilocks; D]) -> *mut u64 { a.as_mut_ptr() as *mut u64 }

/// A zeroed `[Goldilocks; N]` on the HEAP: the output buffers of the methods below never live in this crate's stack
/// frames (`[Goldilocks::ZERO; N]` as a local is N * 8 bytes of stack -- 32 MiB at the headline size).  The array is moved
/// out of the box only into the returned `Polynomial`, i.e. into the caller's return slot: the reference's value type
/// keeps `[F; D]` inline (mod.rs:34-44), so a caller that wants D >= 2^17 needs a large stack for the VALUE itself --
/// or `device::HeapPoly` / `device::DevicePoly`, which have no such limit.
#[inline]
fn heap_zeroed<const N: usize>() -> Box<[Goldilocks; N]> {
  match vec![Goldilocks::ZERO; N].into_boxed_slice().try_into() {
    Ok(b) => b,
    Err(_) => unreachable!("the vector has exactly N elements"),
  }
}

/// Operations on a monomial-basis polynomial over Goldilocks, executed by libronk_ntt.so on the GPU.
pub trait Accelerated<const D: usize> {
  /// `Polynomial::fft` (mod.rs:273-323): natural order in and out, omega = 7^((p-1)/D); also fills `Lagrange::nodes`.
  /// Panics like the reference: D not a power of two (the `where` bound there), D does not divide p - 1.
  fn fft_gpu(&self) -> Polynomial<Lagrange<Goldilocks>, Goldilocks, D>;
  /// `Polynomial::dft` (mod.rs:240-258) for any D | p - 1
  fn dft_gpu(&self) -> Polynomial<Lagrange<Goldilocks>, Goldilocks, D>;
  /// `Polynomial::<Monomial>::evaluate` (mod.rs:133-139)
  fn evaluate_gpu(&self, x: Goldilocks) -> Goldilocks;
  /// `impl Mul` (arithmetic.rs:97-119): D + D2 - 1 coefficients
  fn mul_gpu<const D2: usize>(
    &self,
    rhs: &Polynomial<Monomial, Goldilocks, D2>,
  ) -> Polynomial<Monomial, Goldilocks, { D + D2 - 1 }>
  where [(); D + D2 - 1]:;
  /// `quotient_and_remainder` (mod.rs:170-225), both with D coefficients; behind `impl Div` / `impl Rem` (arithmetic.rs:121-146)
  fn quotient_and_remainder_gpu<const D2: usize>(&self, rhs: &Polynomial<Monomial, Goldilocks, D2>) -> (Self, Self)
  where Self: Sized;
}

impl<const D: usize> Accelerated<D> for Polynomial<Monomial, Goldilocks, D> {
  fn fft_gpu(&self) -> Polynomial<Lagrange<Goldilocks>, Goldilocks, D> {
    let mut out = heap_zeroed::<D>();
    let mut nodes = vec![Goldilocks::ZERO; D];
    check(unsafe { ffi::ronk_fft(P, G, cptr(&self.coefficients), mptr(&mut *out), nodes.as_mut_ptr() as *mut u64, D) });
    Polynomial { coefficients: *out, basis: Lagrange { nodes } }
  }

  fn dft_gpu(&self) -> Polynomial<Lagrange<Goldilocks>, Goldilocks, D> {
    // two calls: the values (any D | p - 1: direct kernel or Bluestein on the NTT path) and the node table omega^i, which
    // only ronk_fft returns alongside (it exists for powers of two only; the table itself is the same call either way)
    let mut out = heap_zeroed::<D>();
    let mut nodes = vec![Goldilocks::ZERO; D];
    check(unsafe { ffi::ronk_dft(P, G, cptr(&self.coefficients), mptr(&mut *out), D) });
    check(unsafe { ffi::ronk_lagrange_nodes(P, G, nodes.as_mut_ptr() as *mut u64, D) });
    Polynomial { coefficients: *out, basis: Lagrange { nodes } }
  }

  fn evaluate_gpu(&self, x: Goldilocks) -> Goldilocks {
    let mut y = 0u64;
    check(unsafe { ffi::ronk_poly_eval(P, cptr(&self.coefficients), D, x.0, &mut y) });
    Goldilocks(y)
  }

  fn mul_gpu<const D2: usize>(
    &self,
    rhs: &Polynomial<Monomial, Goldilocks, D2>,
  ) -> Polynomial<Monomial, Goldilocks, { D + D2 - 1 }>
  where [(); D + D2 - 1]:
  {
    let mut out = heap_zeroed::<{ D + D2 - 1 }>();
    check(unsafe { ffi::ronk_poly_mul(P, G, cptr(&self.coefficients), D, cptr(&rhs.coefficients), D2, mptr(&mut *out)) });
    Polynomial::<Monomial, Goldilocks, { D + D2 - 1 }>::new(*out)
  }

  fn quotient_and_remainder_gpu<const D2: usize>(&self, rhs: &Polynomial<Monomial, Goldilocks, D2>) -> (Self, Self) {
    let (mut q, mut r) = (heap_zeroed::<D>(), heap_zeroed::<D>());
    check(unsafe {
      ffi::ronk_poly_divrem(P, cptr(&self.coefficients), D, cptr(&rhs.coefficients), D2, mptr(&mut *q), mptr(&mut *r))
    });
    (Polynomial::<Monomial, Goldilocks, D>::new(*q), Polynomial::<Monomial, Goldilocks, D>::new(*r))
  }
}

/// Operations on a Lagrange-basis polynomial (values at the D-th roots of unity).
pub trait AcceleratedLagrange<const D: usize> {
  /// `Polynomial::<Lagrange<F>>::ifft` (mod.rs:430-484), including the D^-1 scale
  fn ifft_gpu(&self) -> Polynomial<Monomial, Goldilocks, D>;
  /// `Polynomial::<Lagrange<F>>::evaluate` (mod.rs:382-415), barycentric; evaluating AT a node returns ZERO as in the reference
  fn evaluate_gpu(&self, x: Goldilocks) -> Goldilocks;
}

impl<const D: usize> AcceleratedLagrange<D> for Polynomial<Lagrange<Goldilocks>, Goldilocks, D> {
  fn ifft_gpu(&self) -> Polynomial<Monomial, Goldilocks, D> {
    let mut out = heap_zeroed::<D>();
    check(unsafe { ffi::ronk_ifft(P, G, cptr(&self.coefficients), mptr(&mut *out), D) });
    Polynomial::<Monomial, Goldilocks, D>::new(*out)
  }

  fn evaluate_gpu(&self, x: Goldilocks) -> Goldilocks {
    let mut y = 0u64;
    check(unsafe {
      ffi::ronk_lagrange_eval(P, cptr(&self.coefficients), self.basis.nodes.as_ptr() as *const u64, D, x.0, &mut y)
    });
    Goldilocks(y)
  }
}

/// `Message::decode` (src/codes/reed_solomon.rs:54-106) for Goldilocks coordinates: interpolation through the first K points
pub fn rs_decode<const K: usize>(xs: &[Goldilocks; K], ys: &[Goldilocks; K]) -> [Goldilocks; K] {
  let mut out = heap_zeroed::<K>();
  check(unsafe { ffi::ronk_rs_decode(P, cptr(xs), cptr(ys), K, mptr(&mut *out)) });
  *out
}

#[cfg(test)]
mod tests {
  //! the reference's own polynomial tests (src/polynomial/tests.rs), re-stated for the 64-bit field; need a GPU
  use super::*;

  fn poly() -> Polynomial<Monomial, Goldilocks, 4> {
    Polynomial::<Monomial, Goldilocks, 4>::new([1u64, 2, 3, 4].map(Goldilocks))
  }

  #[test]
  fn fft_matches_dft_and_round_trips() {
    let p = poly();
    assert_eq!(p.fft_gpu(), p.dft());                       // the reference's O(D^2) definition on the host
    assert_eq!(p.fft_gpu().ifft_gpu(), p);
    assert_eq!(p.dft_gpu(), p.dft());
  }

  #[test]
  fn mul_div_eval() {
    let (a, b) = (poly(), Polynomial::<Monomial, Goldilocks, 2>::new([Goldilocks(5), Goldilocks(1)]));
    assert_eq!(a.mul_gpu(&b), a * b);
    let (q, r) = a.quotient_and_remainder_gpu(&b);
    // the reference's `quotient_and_remainder` is private (mod.rs:170): its public faces are `impl Div` / `impl Rem`
    // (arithmetic.rs:121-146); `Polynomial` is `Copy`, so `a` and `b` can be used twice
    assert_eq!((q, r), (a / b, a % b));
    assert_eq!(a.evaluate_gpu(Goldilocks(2)), a.evaluate(Goldilocks(2)));
  }

  #[test]
  #[should_panic]
  fn no_roots_of_unity() {
    // 3 does not divide a power of two, but does divide p - 1 -> dft works; 7 elements: 7 does not divide p - 1 = 2^32 * 3 * 5 * 17 * 257 * 65537
    let p = Polynomial::<Monomial, Goldilocks, 7>::new([Goldilocks(1); 7]);
    let _ = p.dft_gpu();
  }
}
