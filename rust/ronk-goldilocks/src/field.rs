//! `Goldilocks`: the 64-bit `FiniteField` implementor (ronkathon's `PrimeField<P>` computes `a * b % P` in `usize` and
//! overflows for P >= 2^32, src/algebra/field/prime/arithmetic.rs:34-38).  Every item of the trait surface of
//! src/algebra/field/mod.rs:17-76 is spelled out below, in the order of the supertrait list; scalar arithmetic stays on
//! the host exactly like `PrimeField` (a single field element is a value type) -- arrays go to the GPU (polynomial.rs).
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
use ronkathon::algebra::{
  field::{Field, FieldExt, FiniteField},
  Finite,
};

use crate::ffi::P;

/// Canonical residue mod p = 2^64 - 2^32 + 1.  `repr(transparent)`: `[Goldilocks; D]` is layout-identical to `[u64; D]`,
/// which is what the C ABI takes (`PrimeField<P>{ value: usize }` has no repr guarantee; src/algebra/field/prime/mod.rs:39-42).
#[repr(transparent)]
#[derive(Debug, Copy, Clone, PartialEq, Eq, Hash, Default, PartialOrd)]
pub struct Goldilocks(pub u64);

impl Goldilocks {
  /// `PrimeField::new` (prime/mod.rs:48-51): reduces, no primality check needed for the fixed prime
  pub const fn new(value: u64) -> Self { Self(value % P) }
}

// ---- Finite (src/algebra/mod.rs:8-13)
impl Finite for Goldilocks {
  const ORDER: usize = P as usize;
}

// ---- Field (src/algebra/field/mod.rs:17-50)
impl Field for Goldilocks {
  const ONE: Self = Self(1);
  const ZERO: Self = Self(0);

  /// prime/mod.rs:62-72: Fermat, `None` for zero
  fn inverse(&self) -> Option<Self> {
    if self.0 == 0 {
      return None;
    }
    Some(self.pow(Self::ORDER - 2))
  }

  /// prime/mod.rs:74-84 computes a^power by a doubly-recursive square-and-multiply; same value, O(log power)
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

// ---- FieldExt (src/algebra/field/mod.rs:79-84; PrimeField's impl: prime/mod.rs:142-226)
/// `PrimeField::euler_criterion` for any 64-bit implementor: `self.pow((P - 1) / 2) == ONE`
pub(crate) fn euler_criterion_of<F: FiniteField>(x: F) -> bool { x.pow((F::ORDER - 1) / 2) == F::ONE }

/// `PrimeField::sqrt` (Tonelli-Shanks, prime/mod.rs:174-226) statement by statement for any 64-bit implementor whose
/// `PartialOrd` compares canonical residues: ZERO -> (ZERO, ZERO); a non-residue panics with the reference's message; the
/// pair comes back smaller root first.
pub(crate) fn sqrt_of<F: FiniteField + PartialOrd>(x: F) -> Option<(F, F)> {
  if x == F::ZERO {
    return Some((F::ZERO, F::ZERO));
  }
  assert!(euler_criterion_of(x), "Element is not a quadratic residue");
  // P - 1 = q * 2^s, q odd
  let (mut q, mut s) = (F::ORDER - 1, 0u32);
  while q % 2 == 0 {
    q /= 2;
    s += 1;
  }
  // the first z >= 2 that is not a quadratic residue
  let mut z = F::ONE + F::ONE;
  while euler_criterion_of(z) {
    z += F::ONE;
  }
  let (mut m, mut c, mut t, mut r) = (s, z.pow(q), x.pow(q), x.pow((q + 1) / 2));
  loop {
    if t == F::ONE {
      return if -r < r { Some((-r, r)) } else { Some((r, -r)) };
    }
    let (mut i, mut t_pow) = (1u32, t.pow(2));
    while t_pow != F::ONE {
      t_pow = t_pow.pow(2);
      i += 1;
    }
    let b = c.pow(2_usize.pow(m - i - 1));
    m = i;
    c = b.pow(2);
    t *= c;
    r *= b;
  }
}

impl FieldExt for Goldilocks {
  fn sqrt(&self) -> Option<(Self, Self)> { sqrt_of(*self) }

  fn euler_criterion(&self) -> bool { euler_criterion_of(*self) }
}

impl Goldilocks {
  /// `euler_criterion` of every element on the GPU (`ronk_vec_euler`): 1 for a non-zero square, else 0
  pub fn euler_criterion_many(a: &[Self]) -> Vec<bool> {
    let mut out = vec![0u64; a.len()];
    crate::ffi::check(unsafe { crate::ffi::ronk_vec_euler(P, a.as_ptr() as *const u64, out.as_mut_ptr(), a.len()) });
    out.into_iter().map(|v| v == 1).collect()
  }

  /// `sqrt` of every element on the GPU (`ronk_vec_sqrt`): (smaller root, larger root); panics like the reference when an
  /// element is no residue
  pub fn sqrt_many(a: &[Self]) -> Vec<(Self, Self)> {
    let (mut r0, mut r1) = (vec![Self(0); a.len()], vec![Self(0); a.len()]);
    crate::ffi::check(unsafe {
      crate::ffi::ronk_vec_sqrt(P, a.as_ptr() as *const u64, r0.as_mut_ptr() as *mut u64, r1.as_mut_ptr() as *mut u64, a.len())
    });
    r0.into_iter().zip(r1).collect()
  }
}

// ---- FiniteField (src/algebra/field/mod.rs:52-76); primitive_root_of_unity is the provided method
impl FiniteField for Goldilocks {
  /// explicit: the reference's `find_primitive_element` heuristic (prime/mod.rs:110-123) returns 3, which does not
  /// generate the 2^32-subgroup; 7 is the generator every Goldilocks implementation uses
  const PRIMITIVE_ELEMENT: Self = Self(7);
}

// ---- Add / AddAssign / Sum (prime/arithmetic.rs:3-17)
impl Add for Goldilocks {
  type Output = Self;

  fn add(self, rhs: Self) -> Self { Self(((self.0 as u128 + rhs.0 as u128) % P as u128) as u64) }
}
impl AddAssign for Goldilocks {
  fn add_assign(&mut self, rhs: Self) { *self = *self + rhs; }
}
impl Sum for Goldilocks {
  fn sum<I: Iterator<Item = Self>>(iter: I) -> Self { iter.reduce(|x, y| x + y).unwrap_or(Self::ZERO) }
}

// ---- Sub / SubAssign (prime/arithmetic.rs:19-32)
impl Sub for Goldilocks {
  type Output = Self;

  fn sub(self, rhs: Self) -> Self {
    let (diff, over) = self.0.overflowing_sub(rhs.0);
    Self(if over { diff.wrapping_add(P) } else { diff })
  }
}
impl SubAssign for Goldilocks {
  fn sub_assign(&mut self, rhs: Self) { *self = *self - rhs; }
}

// ---- Mul / MulAssign / Product (prime/arithmetic.rs:34-48)
impl Mul for Goldilocks {
  type Output = Self;

  fn mul(self, rhs: Self) -> Self { Self(((self.0 as u128 * rhs.0 as u128) % P as u128) as u64) }
}
impl MulAssign for Goldilocks {
  fn mul_assign(&mut self, rhs: Self) { *self = *self * rhs; }
}
impl Product for Goldilocks {
  fn product<I: Iterator<Item = Self>>(iter: I) -> Self { iter.reduce(|x, y| x * y).unwrap_or(Self::ONE) }
}

// ---- Div / DivAssign (prime/arithmetic.rs:50-59): division by zero is `inverse().unwrap()`'s panic
impl Div for Goldilocks {
  type Output = Self;

  #[allow(clippy::suspicious_arithmetic_impl)]
  fn div(self, rhs: Self) -> Self { self * rhs.inverse().unwrap() }
}
impl DivAssign for Goldilocks {
  fn div_assign(&mut self, rhs: Self) { *self = *self / rhs; }
}

// ---- Neg / Rem (prime/arithmetic.rs:61-71)
impl Neg for Goldilocks {
  type Output = Self;

  fn neg(self) -> Self::Output { Self::ZERO - self }
}
impl Rem for Goldilocks {
  type Output = Self;

  fn rem(self, rhs: Self) -> Self { self - (self / rhs) * rhs }
}

// ---- conversions (prime/mod.rs:227-270)
impl From<usize> for Goldilocks {
  fn from(val: usize) -> Self { Self::new(val as u64) }
}
impl From<u32> for Goldilocks {
  fn from(val: u32) -> Self { Self::new(val as u64) }
}
impl From<u64> for Goldilocks {
  fn from(val: u64) -> Self { Self::new(val) }
}
impl From<Goldilocks> for usize {
  fn from(value: Goldilocks) -> Self { value.0 as usize }
}
impl From<i32> for Goldilocks {
  fn from(value: i32) -> Self {
    let abs = Self::new(value.unsigned_abs() as u64);
    if value.is_positive() {
      abs
    } else {
      -abs
    }
  }
}
impl FromStr for Goldilocks {
  type Err = ();

  fn from_str(s: &str) -> Result<Self, Self::Err> {
    let num: u64 = str::parse(s).expect("failed to parse string into usize");
    Ok(Self::new(num))
  }
}

// ---- Display (prime/mod.rs:125-127)
impl fmt::Display for Goldilocks {
  fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result { write!(f, "{}", self.0) }
}

// ---- Distribution<..> for Standard (prime/mod.rs:129-140): rejection sampling of canonical values, here from 64 bits
impl Distribution<Goldilocks> for Standard {
  #[inline]
  fn sample<R: Rng + ?Sized>(&self, rng: &mut R) -> Goldilocks {
    loop {
      let next = rng.next_u64();
      if next < P {
        return Goldilocks(next);
      }
    }
  }
}

#[cfg(test)]
mod tests {
  use super::*;

  #[test]
  fn root_of_unity_convention() {
    // omega_64 = 7^((p-1)/64) = 2^39: what makes every twiddle inside a 64-point sub-transform a shift on the GPU
    assert_eq!(Goldilocks::primitive_root_of_unity(64), Goldilocks(1u64 << 39));
    assert_eq!(Goldilocks(5) / Goldilocks(5), Goldilocks::ONE);
    assert_eq!(-Goldilocks(1), Goldilocks(P - 1));
    assert_eq!(Goldilocks::from(-1i32), Goldilocks(P - 1));
  }

  #[test]
  #[should_panic]
  fn zero_has_no_inverse() { let _ = Goldilocks::ONE / Goldilocks::ZERO; }
}
