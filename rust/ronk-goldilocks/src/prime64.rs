//! `Prime64<P, G>`: ANY odd 64-bit prime as a `FiniteField` implementor, with the same GPU-backed polynomial operations.
//!
//! ronkathon's field is generic over its modulus -- `PrimeField<const P: usize>` (src/algebra/field/prime/mod.rs:39-52) --
//! but its arithmetic (`a * b % P` in `usize`, prime/arithmetic.rs:34-38) overflows from P = 2^32 on, and its `Polynomial`
//! methods are CPU recursions.  `Prime64` is that same generic type for the 64-bit range: `P` the modulus, `G` its
//! `PRIMITIVE_ELEMENT` (the reference's heuristic, prime/mod.rs:110-123, is not reliable for large P -- it returns a
//! non-generator for Goldilocks -- so the generator is part of the type).  Scalar arithmetic is host arithmetic on `u128`, as
//! in `field.rs`; arrays go to the GPU through the same C ABI with `p = P, g = G`: the library runs its tile kernels over
//! Montgomery arithmetic for every such prime whose `G` is a quadratic non-residue (`ronk_plan_path() == 2`,
//! include/ronk_ntt.h; csrc/field_policy.h), about 1.2x the Goldilocks time per transform, and the radix-2 path otherwise.
//! `Goldilocks` stays its own type: `Prime64<0xFFFF_FFFF_0000_0001, 7>` computes the same values on the same kernels.
//!
//! Not compiled in the engine's build image (no rustc); the FFI sequence below is replayed by
//! tests/cpp/test_rust_ffi_replay.c (`replay_prime64`) through the real library on a GPU.
use core::{
  fmt,
  hash::Hash,
  iter::{Product, Sum},
  ops::{Add, AddAssign, Div, DivAssign, Mul, MulAssign, Neg, Rem, Sub, SubAssign},
  str::FromStr,
};

use rand::{
  distributions::{Distribution, Standard},
  Rng,
};
use ronkathon::{
  algebra::{
    field::{Field, FieldExt, FiniteField},
    Finite,
  },
  polynomial::{Lagrange, Monomial, Polynomial},
};

use crate::ffi::{self, check};

/// Canonical residue mod `P`.  `repr(transparent)`: `[Prime64<P, G>; D]` is layout-identical to `[u64; D]`.
#[repr(transparent)]
#[derive(Debug, Copy, Clone, PartialEq, Eq, Hash, Default, PartialOrd)]
pub struct Prime64<const P: u64, const G: u64>(pub u64);

impl<const P: u64, const G: u64> Prime64<P, G> {
  /// `PrimeField::new` (prime/mod.rs:48-51): reduces.  Primality of `P` is the library's check (`ronk_check_prime`, a
  /// deterministic Miller-Rabin instead of the reference's trial division): [`Prime64::assert_prime`].
  pub const fn new(value: u64) -> Self { Self(value % P) }

  /// panics with the reference's message ("input is not a prime number", prime/mod.rs:49) unless `P` is prime
  pub fn assert_prime() { check(unsafe { ffi::ronk_check_prime(P) }); }
}

impl<const P: u64, const G: u64> Finite for Prime64<P, G> {
  const ORDER: usize = P as usize;
}

impl<const P: u64, const G: u64> Field for Prime64<P, G> {
  const ONE: Self = Self(1 % P);
  const ZERO: Self = Self(0);

  /// prime/mod.rs:62-72: Fermat, `None` for zero
  fn inverse(&self) -> Option<Self> {
    if self.0 == 0 {
      return None;
    }
    Some(self.pow(Self::ORDER - 2))
  }

  /// prime/mod.rs:74-84: the same value by square-and-multiply
  fn pow(self, mut power: usize) -> Self {
    let (mut acc, mut base) = (Self::ONE, self);
    while power != 0 {
      if power & 1 == 1 {
        acc *= base;
      }
      base *= base;
      power >>= 1;
    }
    acc
  }
}

impl<const P: u64, const G: u64> FiniteField for Prime64<P, G> {
  const PRIMITIVE_ELEMENT: Self = Self(G % P);
}

/// `FieldExt` (src/algebra/field/mod.rs:79-84) as `PrimeField<P>` implements it (prime/mod.rs:142-226): field.rs holds the
/// statement-by-statement Tonelli-Shanks shared with `Goldilocks`
impl<const P: u64, const G: u64> FieldExt for Prime64<P, G> {
  fn sqrt(&self) -> Option<(Self, Self)> { crate::field::sqrt_of(*self) }

  fn euler_criterion(&self) -> bool { crate::field::euler_criterion_of(*self) }
}

impl<const P: u64, const G: u64> Prime64<P, G> {
  /// `sqrt` of every element on the GPU (`ronk_vec_sqrt`, Montgomery arithmetic): (smaller root, larger root)
  pub fn sqrt_many(a: &[Self]) -> Vec<(Self, Self)> {
    let (mut r0, mut r1) = (vec![Self(0); a.len()], vec![Self(0); a.len()]);
    check(unsafe { ffi::ronk_vec_sqrt(P, a.as_ptr() as *const u64, r0.as_mut_ptr() as *mut u64, r1.as_mut_ptr() as *mut u64, a.len()) });
    r0.into_iter().zip(r1).collect()
  }

  /// `euler_criterion` of every element on the GPU (`ronk_vec_euler`)
  pub fn euler_criterion_many(a: &[Self]) -> Vec<bool> {
    let mut out = vec![0u64; a.len()];
    check(unsafe { ffi::ronk_vec_euler(P, a.as_ptr() as *const u64, out.as_mut_ptr(), a.len()) });
    out.into_iter().map(|v| v == 1).collect()
  }
}

impl<const P: u64, const G: u64> Add for Prime64<P, G> {
  type Output = Self;

  fn add(self, rhs: Self) -> Self { Self(((self.0 as u128 + rhs.0 as u128) % P as u128) as u64) }
}
impl<const P: u64, const G: u64> AddAssign for Prime64<P, G> {
  fn add_assign(&mut self, rhs: Self) { *self = *self + rhs; }
}
impl<const P: u64, const G: u64> Sum for Prime64<P, G> {
  fn sum<I: Iterator<Item = Self>>(iter: I) -> Self { iter.reduce(|x, y| x + y).unwrap_or(Self::ZERO) }
}
impl<const P: u64, const G: u64> Sub for Prime64<P, G> {
  type Output = Self;

  fn sub(self, rhs: Self) -> Self {
    let (diff, over) = self.0.overflowing_sub(rhs.0);
    Self(if over { diff.wrapping_add(P) } else { diff })
  }
}
impl<const P: u64, const G: u64> SubAssign for Prime64<P, G> {
  fn sub_assign(&mut self, rhs: Self) { *self = *self - rhs; }
}
impl<const P: u64, const G: u64> Mul for Prime64<P, G> {
  type Output = Self;

  fn mul(self, rhs: Self) -> Self { Self(((self.0 as u128 * rhs.0 as u128) % P as u128) as u64) }
}
impl<const P: u64, const G: u64> MulAssign for Prime64<P, G> {
  fn mul_assign(&mut self, rhs: Self) { *self = *self * rhs; }
}
impl<const P: u64, const G: u64> Product for Prime64<P, G> {
  fn product<I: Iterator<Item = Self>>(iter: I) -> Self { iter.reduce(|x, y| x * y).unwrap_or(Self::ONE) }
}
impl<const P: u64, const G: u64> Div for Prime64<P, G> {
  type Output = Self;

  #[allow(clippy::suspicious_arithmetic_impl)]
  fn div(self, rhs: Self) -> Self { self * rhs.inverse().unwrap() }
}
impl<const P: u64, const G: u64> DivAssign for Prime64<P, G> {
  fn div_assign(&mut self, rhs: Self) { *self = *self / rhs; }
}
impl<const P: u64, const G: u64> Neg for Prime64<P, G> {
  type Output = Self;

  fn neg(self) -> Self::Output { Self::ZERO - self }
}
impl<const P: u64, const G: u64> Rem for Prime64<P, G> {
  type Output = Self;

  fn rem(self, rhs: Self) -> Self { self - (self / rhs) * rhs }
}

// ---- conversions, Display, sampling (prime/mod.rs:125-140, :227-270)
impl<const P: u64, const G: u64> From<usize> for Prime64<P, G> {
  fn from(val: usize) -> Self { Self::new(val as u64) }
}
impl<const P: u64, const G: u64> From<u32> for Prime64<P, G> {
  fn from(val: u32) -> Self { Self::new(val as u64) }
}
impl<const P: u64, const G: u64> From<u64> for Prime64<P, G> {
  fn from(val: u64) -> Self { Self::new(val) }
}
impl<const P: u64, const G: u64> From<Prime64<P, G>> for usize {
  fn from(value: Prime64<P, G>) -> Self { value.0 as usize }
}
impl<const P: u64, const G: u64> From<i32> for Prime64<P, G> {
  fn from(value: i32) -> Self {
    let abs = Self::new(value.unsigned_abs() as u64);
    if value.is_positive() {
      abs
    } else {
      -abs
    }
  }
}
impl<const P: u64, const G: u64> FromStr for Prime64<P, G> {
  type Err = ();

  fn from_str(s: &str) -> Result<Self, Self::Err> {
    let num: u64 = str::parse(s).expect("failed to parse string into usize");
    Ok(Self::new(num))
  }
}
impl<const P: u64, const G: u64> fmt::Display for Prime64<P, G> {
  fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result { write!(f, "{}", self.0) }
}
impl<const P: u64, const G: u64> Distribution<Prime64<P, G>> for Standard {
  /// uniform on [0, P): rejection from the bits the prime has
  #[inline]
  fn sample<R: Rng + ?Sized>(&self, rng: &mut R) -> Prime64<P, G> {
    let mask = if P.leading_zeros() == 0 { u64::MAX } else { (1u64 << (64 - P.leading_zeros())) - 1 };
    loop {
      let next = rng.next_u64() & mask;
      if next < P {
        return Prime64(next);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- polynomial operations
#[inline]
fn cptr<const P: u64, const G: u64, const D: usize>(a: &[Prime64<P, G>; D]) -> *const u64 { a.as_ptr() as *const u64 }
#[inline]
fn mptr<const P: u64, const G: u64, const D: usize>(a: &mut [Prime64<P, G>; D]) -> *mut u64 { a.as_mut_ptr() as *mut u64 }
#[inline]
fn heap_zeroed<const P: u64, const G: u64, const N: usize>() -> Box<[Prime64<P, G>; N]> {
  match vec![Prime64::<P, G>(0); N].into_boxed_slice().try_into() {
    Ok(b) => b,
    Err(_) => unreachable!("the vector has exactly N elements"),
  }
}

/// `Polynomial<Monomial, Prime64<P, G>, D>` on the GPU: the counterparts of `polynomial::Accelerated` for a generic prime
/// (same reference items, same panics: src/polynomial/mod.rs:133-139, :240-323, src/polynomial/arithmetic.rs:97-146).
pub trait AcceleratedPrime<const P: u64, const G: u64, const D: usize> {
  fn fft_gpu(&self) -> Polynomial<Lagrange<Prime64<P, G>>, Prime64<P, G>, D>;
  fn dft_gpu(&self) -> Polynomial<Lagrange<Prime64<P, G>>, Prime64<P, G>, D>;
  fn evaluate_gpu(&self, x: Prime64<P, G>) -> Prime64<P, G>;
  fn mul_gpu<const D2: usize>(
    &self,
    rhs: &Polynomial<Monomial, Prime64<P, G>, D2>,
  ) -> Polynomial<Monomial, Prime64<P, G>, { D + D2 - 1 }>
  where [(); D + D2 - 1]:;
  fn quotient_and_remainder_gpu<const D2: usize>(&self, rhs: &Polynomial<Monomial, Prime64<P, G>, D2>) -> (Self, Self)
  where Self: Sized;
}

impl<const P: u64, const G: u64, const D: usize> AcceleratedPrime<P, G, D> for Polynomial<Monomial, Prime64<P, G>, D> {
  fn fft_gpu(&self) -> Polynomial<Lagrange<Prime64<P, G>>, Prime64<P, G>, D> {
    let mut out = heap_zeroed::<P, G, D>();
    let mut nodes = vec![Prime64::<P, G>(0); D];
    check(unsafe { ffi::ronk_fft(P, G, cptr(&self.coefficients), mptr(&mut *out), nodes.as_mut_ptr() as *mut u64, D) });
    Polynomial { coefficients: *out, basis: Lagrange { nodes } }
  }

  fn dft_gpu(&self) -> Polynomial<Lagrange<Prime64<P, G>>, Prime64<P, G>, D> {
    let mut out = heap_zeroed::<P, G, D>();
    let mut nodes = vec![Prime64::<P, G>(0); D];
    check(unsafe { ffi::ronk_dft(P, G, cptr(&self.coefficients), mptr(&mut *out), D) });
    check(unsafe { ffi::ronk_lagrange_nodes(P, G, nodes.as_mut_ptr() as *mut u64, D) });
    Polynomial { coefficients: *out, basis: Lagrange { nodes } }
  }

  fn evaluate_gpu(&self, x: Prime64<P, G>) -> Prime64<P, G> {
    let mut y = 0u64;
    check(unsafe { ffi::ronk_poly_eval(P, cptr(&self.coefficients), D, x.0, &mut y) });
    Prime64(y)
  }

  fn mul_gpu<const D2: usize>(
    &self,
    rhs: &Polynomial<Monomial, Prime64<P, G>, D2>,
  ) -> Polynomial<Monomial, Prime64<P, G>, { D + D2 - 1 }>
  where [(); D + D2 - 1]:
  {
    let mut out = heap_zeroed::<P, G, { D + D2 - 1 }>();
    check(unsafe { ffi::ronk_poly_mul(P, G, cptr(&self.coefficients), D, cptr(&rhs.coefficients), D2, mptr(&mut *out)) });
    Polynomial::<Monomial, Prime64<P, G>, { D + D2 - 1 }>::new(*out)
  }

  fn quotient_and_remainder_gpu<const D2: usize>(&self, rhs: &Polynomial<Monomial, Prime64<P, G>, D2>) -> (Self, Self) {
    let (mut q, mut r) = (heap_zeroed::<P, G, D>(), heap_zeroed::<P, G, D>());
    check(unsafe {
      ffi::ronk_poly_divrem(P, cptr(&self.coefficients), D, cptr(&rhs.coefficients), D2, mptr(&mut *q), mptr(&mut *r))
    });
    (Polynomial::<Monomial, Prime64<P, G>, D>::new(*q), Polynomial::<Monomial, Prime64<P, G>, D>::new(*r))
  }
}

/// `Polynomial::<Lagrange<F>>::ifft` (mod.rs:430-484) for a generic prime
pub trait AcceleratedPrimeLagrange<const P: u64, const G: u64, const D: usize> {
  fn ifft_gpu(&self) -> Polynomial<Monomial, Prime64<P, G>, D>;
}
impl<const P: u64, const G: u64, const D: usize> AcceleratedPrimeLagrange<P, G, D>
  for Polynomial<Lagrange<Prime64<P, G>>, Prime64<P, G>, D>
{
  fn ifft_gpu(&self) -> Polynomial<Monomial, Prime64<P, G>, D> {
    let mut out = heap_zeroed::<P, G, D>();
    check(unsafe { ffi::ronk_ifft(P, G, cptr(&self.coefficients), mptr(&mut *out), D) });
    Polynomial::<Monomial, Prime64<P, G>, D>::new(*out)
  }
}

/// A plan for batches of 2^log2n-point transforms over `Prime64<P, G>` on host slices (`ronk_plan_create`, `ronk_ntt_forward`,
/// `ronk_ntt_inverse`); `path()` says which kernels it runs: 2 = the tile kernels over Montgomery arithmetic, 0 = radix-2.
pub struct PrimePlan<const P: u64, const G: u64> {
  raw:       *mut ffi::RonkPlan,
  pub log2n: u32,
  pub batch: usize,
}
unsafe impl<const P: u64, const G: u64> Send for PrimePlan<P, G> {}

impl<const P: u64, const G: u64> PrimePlan<P, G> {
  pub fn new(log2n: u32, batch: usize) -> Self {
    let mut raw: *mut ffi::RonkPlan = core::ptr::null_mut();
    check(unsafe { ffi::ronk_plan_create(&mut raw, P, G, log2n, batch as u64, -1) });
    Self { raw, log2n, batch }
  }

  pub fn n(&self) -> usize { 1usize << self.log2n }

  pub fn path(&self) -> i32 { unsafe { ffi::ronk_plan_path(self.raw) } }

  pub fn forward(&self, input: &[Prime64<P, G>], output: &mut [Prime64<P, G>]) {
    assert!(input.len() == self.batch * self.n() && output.len() == input.len());
    check(unsafe {
      ffi::ronk_ntt_forward(self.raw, input.as_ptr() as *const u64, output.as_mut_ptr() as *mut u64, core::ptr::null_mut())
    });
  }

  pub fn inverse(&self, input: &[Prime64<P, G>], output: &mut [Prime64<P, G>]) {
    assert!(input.len() == self.batch * self.n() && output.len() == input.len());
    check(unsafe { ffi::ronk_ntt_inverse(self.raw, input.as_ptr() as *const u64, output.as_mut_ptr() as *mut u64) });
  }
}
impl<const P: u64, const G: u64> Drop for PrimePlan<P, G> {
  fn drop(&mut self) {
    if !self.raw.is_null() {
      unsafe { ffi::ronk_plan_destroy(self.raw) };
    }
  }
}

#[cfg(test)]
mod tests {
  //! need a GPU.  The primes of the engine's GPU suite (tests/test_gpu_mont.py).
  use super::*;

  type F = Prime64<0xFFFF_FFFC_0000_0001, 10>;   // prime above 2^63, 2-adicity 34

  #[test]
  fn generic_prime_runs_the_tile_kernels_and_round_trips() {
    F::assert_prime();
    let plan = PrimePlan::<0xFFFF_FFFC_0000_0001, 10>::new(16, 1);
    assert_eq!(plan.path(), 2);
    let x: Vec<F> = (0..1u64 << 16).map(|i| F::new(i.wrapping_mul(0x9E37_79B9_7F4A_7C15))).collect();
    let (mut y, mut z) = (vec![F::ZERO; x.len()], vec![F::ZERO; x.len()]);
    plan.forward(&x, &mut y);
    plan.inverse(&y, &mut z);
    assert_eq!(x, z);
  }

  #[test]
  fn fft_is_the_reference_dft() {
    let p = Polynomial::<Monomial, F, 16>::new(core::array::from_fn(|i| F::new(i as u64 + 1)));
    assert_eq!(p.fft_gpu(), p.dft());   // Polynomial::dft on the host: the reference's own O(D^2) definition, generic over F
    assert_eq!(p.fft_gpu().ifft_gpu(), p);
    let q = Polynomial::<Monomial, F, 3>::new([F::new(5), F::new(1), F::new(7)]);
    assert_eq!(p.mul_gpu(&q), p * q);
  }
}
