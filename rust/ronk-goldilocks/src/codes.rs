//! Reed-Solomon over Goldilocks: the shapes of ronkathon's `src/codes/reed_solomon.rs`, bound to the GPU.
//!
//! The reference's `Message<K, P>`, `Codeword<N, K, P>` and `Coordinate<N, P>` are MONOMORPHIC over `PrimeField<P>`
//! (reed_solomon.rs:15-17, :22-24, :29-35): they cannot hold a `Goldilocks`, and `PrimeField<P>` cannot hold a 64-bit modulus
//! (`a * b % P` in `usize`, prime/arithmetic.rs:34-38).  So the 64-bit code gets the same three types with the modulus fixed
//! -- same field names, same `encode::<N>` / `decode::<M>` signatures, same panics -- and, beside them, the batched device
//! forms a production encoder uses (1024 x 2^16: BASELINE config 4), which have no counterpart in the reference.
//!
//! * `Message::encode::<N>`  = reed_solomon.rs:42-52: `x_i = omega_N^i`, `y_i = poly(omega_N^i)`          -> `ronk_rs_encode`
//! * `Message::decode::<M>`  = reed_solomon.rs:54-106: interpolation through the first K coordinates   -> `ronk_rs_decode`
//! * [`encode_batch`]        = `encode::<N>` of `plan.batch` messages at once, y-coordinates only       -> `ronk_rs_encode_batch_dev`
//! * [`lde`]                 = `lagrange.ifft()` (mod.rs:430-453) then `encode::<N>` on a coset         -> `ronk_lde_batch_dev`
//! * [`decode_dev`]          = `decode` on device-resident coordinates                                -> `ronk_rs_decode_dev`
use core::ffi::{c_int, c_void};
use std::ptr;

use crate::{
  device::{DevicePoly, OnDevice, Plan},
  ffi::{self, check, G, P},
  field::Goldilocks,
};

/// reed_solomon.rs:13-18
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Message<const K: usize> {
  /// The data that is to be encoded.
  pub data: [Goldilocks; K],
}

/// reed_solomon.rs:27-35
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct Coordinate<const N: usize> {
  pub x: Goldilocks,
  pub y: Goldilocks,
}

/// reed_solomon.rs:20-25
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Codeword<const N: usize, const K: usize> {
  pub data: [Coordinate<N>; N],
}

impl<const K: usize> Message<K> {
  /// reed_solomon.rs:39
  pub fn new(data: [Goldilocks; K]) -> Self { Self { data } }

  /// reed_solomon.rs:42-52.  Panics like the reference: `N < K` ("Code size must be greater than or equal to K", :109-111),
  /// `N` does not divide `p - 1` ("n must divide p^q - 1", field/mod.rs:72).
  pub fn encode<const N: usize>(self) -> Codeword<N, K> {
    assert!(N >= K, "Code size must be greater than or equal to K");
    let (mut xs, mut ys) = (vec![0u64; N], vec![0u64; N]);
    check(unsafe { ffi::ronk_rs_encode(P, G, self.data.as_ptr() as *const u64, K, N, xs.as_mut_ptr(), ys.as_mut_ptr()) });
    Codeword { data: core::array::from_fn(|i| Coordinate { x: Goldilocks(xs[i]), y: Goldilocks(ys[i]) }) }
  }

  /// reed_solomon.rs:54-106: the first K coordinates of a (possibly erased) codeword determine the message.  Coincident
  /// x-coordinates are the reference's `numerator / denominator` panic (`unwrap` on `inverse()` of zero).
  pub fn decode<const M: usize>(codeword: Codeword<M, K>) -> Self {
    assert!(M >= K, "Code size must be greater than or equal to K");
    let xs: Vec<u64> = codeword.data.iter().take(K).map(|c| c.x.0).collect();
    let ys: Vec<u64> = codeword.data.iter().take(K).map(|c| c.y.0).collect();
    let mut out = vec![0u64; K];
    check(unsafe { ffi::ronk_rs_decode(P, xs.as_ptr(), ys.as_ptr(), K, out.as_mut_ptr()) });
    Message { data: core::array::from_fn(|i| Goldilocks(out[i])) }
  }
}

/// `encode::<N>` of `plan.batch` messages in one launch pair: `msgs` holds `batch` compact messages of `k` coefficients,
/// the result `batch` x N y-coordinates (N = `plan.n()`; the x-coordinates are the same `omega_N^i` for every codeword:
/// `ronk_lagrange_nodes`).  The zero padding of `Polynomial::from(message)` is implicit (no padded copy is made).
pub fn encode_batch(plan: &Plan, msgs: &DevicePoly, k: usize) -> DevicePoly {
  assert!(k >= 1 && k <= plan.n(), "Code size must be greater than or equal to K");
  assert!(msgs.len() == plan.batch * k);
  plan.same_device(msgs);
  let _g = OnDevice::new(plan.device());
  let out = DevicePoly::alloc_on(plan.device(), plan.batch * plan.n());
  check(unsafe { ffi::ronk_rs_encode_batch_dev(plan.raw(), msgs.as_ptr(), k, out.as_mut_ptr(), ptr::null_mut()) });
  out
}

/// Low-degree extension of a batch: polynomials given by their values on `{omega_K^i}` (`plan_k`) -> their values on
/// `coset_shift * {omega_N^i}` (`plan_n`, N >= K, same batch).  Returns (values on the larger domain, coefficients).
pub fn lde(plan_k: &Plan, plan_n: &Plan, evals: &DevicePoly, coset_shift: Goldilocks) -> (DevicePoly, DevicePoly) {
  assert!(plan_k.batch == plan_n.batch && plan_n.n() >= plan_k.n());
  assert!(evals.len() == plan_k.batch * plan_k.n());
  assert!(plan_k.device() == plan_n.device(), "the two plans of an extension live on one GPU");
  plan_k.same_device(evals);
  let _g = OnDevice::new(plan_k.device());
  let coeffs = DevicePoly::alloc_on(evals.device(), evals.len());
  let out = DevicePoly::alloc_on(evals.device(), plan_n.batch * plan_n.n());
  check(unsafe {
    ffi::ronk_lde_batch_dev(plan_k.raw(), plan_n.raw(), evals.as_ptr(), coeffs.as_mut_ptr(), out.as_mut_ptr(), coset_shift.0, ptr::null_mut())
  });
  (out, coeffs)
}

/// `Message::decode` on device-resident coordinates (k of each); the reference's panics arrive through the status word
pub fn decode_dev(xs: &DevicePoly, ys: &DevicePoly) -> DevicePoly {
  assert!(xs.len() == ys.len());
  xs.same_device(ys);
  let _g = OnDevice::new(xs.device());   // `ronk_rs_decode_dev` (kernels and workspace) runs on the CURRENT device
  let k = xs.len();
  let out = DevicePoly::alloc_on(xs.device(), k);
  let status = DevicePoly::alloc_on(xs.device(), 1);
  let zero = 0u64;
  check(unsafe { ffi::ronk_memcpy_h2d(status.as_mut_ptr() as *mut c_void, &zero as *const u64 as *const c_void, 8) });
  check(unsafe { ffi::ronk_rs_decode_dev(P, xs.as_ptr(), ys.as_ptr(), k, out.as_mut_ptr(), status.as_mut_ptr() as *mut c_int, ptr::null_mut()) });
  let st = status.to_host()[0].0 as u32;   // the library's int status word (low half of the 8 zeroed bytes)
  if st & 4 != 0 {
    check(-9);   // RONK_ERR_UNSUPPORTED: more than 2^14 nodes that are not the q^j sequence `encode` produces
  } else if st != 0 {
    check(-2);   // RONK_ERR_ZERO_INVERSE: coincident nodes (the reference's `numerator / denominator` panic)
  }
  out
}

#[cfg(test)]
mod tests {
  //! the reference's own tests (reed_solomon.rs:121-218), re-stated for the 64-bit field; need a GPU
  use ronkathon::algebra::field::FiniteField;

  use super::*;

  #[test]
  fn encode_is_the_dft_of_the_padded_message() {
    let m = Message::<3>::new([Goldilocks(1), Goldilocks(2), Goldilocks(3)]);
    let cw = m.clone().encode::<8>();
    let w = Goldilocks::primitive_root_of_unity(8);
    for (i, c) in cw.data.iter().enumerate() {
      assert_eq!(c.x, w.pow(i));
      assert_eq!(c.y, Goldilocks(1) + Goldilocks(2) * c.x + Goldilocks(3) * c.x * c.x);
    }
    assert_eq!(Message::<3>::decode::<8>(cw), m);
  }

  #[test]
  #[should_panic]
  fn code_shorter_than_message() { let _ = Message::<4>::new([Goldilocks(1); 4]).encode::<2>(); }

  #[test]
  fn batch_and_lde_on_device() {
    let (k, n, batch) = (1usize << 12, 1usize << 13, 4usize);
    let msgs: Vec<Goldilocks> = (0..(batch * k) as u64).map(|i| Goldilocks::new(i.wrapping_mul(0x9E37_79B9_7F4A_7C15))).collect();
    let (plan_k, plan_n) = (Plan::new(12, batch), Plan::new(13, batch));
    let d = DevicePoly::from_host(&msgs);
    let ys = encode_batch(&plan_n, &d, k).to_host();
    // value form on the small domain, extended to the large one with shift 1 = the same codewords
    let mut small = DevicePoly::alloc(batch * k);
    plan_k.forward(&d, &mut small);
    let (ext, coeffs) = lde(&plan_k, &plan_n, &small, Goldilocks(1));
    assert_eq!(coeffs.to_host(), msgs);
    assert_eq!(ext.to_host(), ys);
  }
}
