//! `kzg::commit` / `kzg::open` on a production-size group: BN254 (alt_bn128) G1 through the GPU's bucket-method MSM.
//!
//! ronkathon's `commit` folds `g1 * coeff` over `AffinePoint<PlutoExtendedCurve>` (src/kzg/setup.rs:48-60); its field traits
//! are `usize`-wide (src/algebra/mod.rs:8-13), so a 254-bit curve is a new pair of plain-data types here rather than an
//! `EllipticCurve` implementor: 4 x 64-bit little-endian limbs, standard (non-Montgomery) form, `(0, 0)` = infinity --
//! byte-identical to what `ronk_msm_bn254` takes, so slices are passed without conversion.
//!
//! `open` is the reference's `kzg::open` (src/kzg/setup.rs:63-78) on the same curve: `poly.div([-eval_point, ONE])` over the
//! SCALAR field F_r (a suffix scan over 256-bit elements on the GPU) followed by `commit(quotient, g1_srs)`.
use crate::{
  device::{DevicePoly, OnDevice},
  ffi,
};

/// element of F_p (coordinates) or an integer scalar, 4 x 64-bit little-endian limbs
pub type Limbs = [u64; 4];

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq, Default)]
pub struct G1Affine {
  pub x: Limbs,
  pub y: Limbs,
}

impl G1Affine {
  pub const INFINITY: Self = Self { x: [0; 4], y: [0; 4] };
  pub const GENERATOR: Self = Self { x: [1, 0, 0, 0], y: [2, 0, 0, 0] };
  pub fn is_infinity(&self) -> bool { *self == Self::INFINITY }
}

/// `kzg::commit(coeffs, g1_srs)` (src/kzg/setup.rs:48-60): sum_i g1_srs[i] * coeffs[i]; panics like the reference when
/// the SRS is shorter than the coefficient vector or a point is not on the curve (`AffinePoint::new`, src/curve/mod.rs:79)
pub fn commit(coeffs: &[Limbs], g1_srs: &[G1Affine]) -> G1Affine {
  assert!(g1_srs.len() >= coeffs.len());
  let mut out = G1Affine::INFINITY;
  ffi::check(unsafe {
    ffi::ronk_msm_bn254(g1_srs.as_ptr() as *const u64, coeffs.as_ptr() as *const u64, coeffs.len(), &mut out as *mut G1Affine as *mut u64)
  });
  out
}

/// the same sum with the SRS and the scalars resident in HBM (`ronk_msm_bn254_dev`): `points` = n x 8 limbs
/// (x then y), `scalars` = n x 4 limbs, both as [`DevicePoly`] buffers of 64-bit words on the same GPU.  With
/// `DevicePoly::div_linear` before it, `kzg::open` (src/kzg/setup.rs:63-78) followed by `commit` never leaves the device.
pub fn commit_dev(scalars: &DevicePoly, g1_srs: &DevicePoly, n: usize) -> G1Affine {
  assert!(scalars.len() >= 4 * n && g1_srs.len() >= 8 * n);
  scalars.same_device(g1_srs);
  let _g = OnDevice::new(scalars.device());   // `ronk_msm_bn254_dev` (kernels and workspace) runs on the CURRENT device
  let mut out = G1Affine::INFINITY;
  ffi::check(unsafe {
    ffi::ronk_msm_bn254_dev(g1_srs.as_ptr(), scalars.as_ptr(), n, &mut out as *mut G1Affine as *mut u64, core::ptr::null_mut())
  });
  out
}


/// `kzg::open(coeffs, eval_point, g1_srs)` (src/kzg/setup.rs:63-78) over BN254: the opening proof `commit(poly / (x - z), srs)` and
/// the evaluation poly(z).  Coefficients and `z` are integers mod r as 4 x 64-bit limbs.  Panics like the reference when the
/// SRS is shorter than the coefficient vector (`commit`'s assert, setup.rs:53) or a point is not on the curve.
pub fn open(coeffs: &[Limbs], eval_point: Limbs, g1_srs: &[G1Affine]) -> (G1Affine, Limbs) {
  let (mut out, mut value) = (G1Affine::INFINITY, [0u64; 4]);
  ffi::check(unsafe {
    ffi::ronk_kzg_open_bn254(
      coeffs.as_ptr() as *const u64, coeffs.len(), eval_point.as_ptr(), g1_srs.as_ptr() as *const u64, g1_srs.len(),
      &mut out as *mut G1Affine as *mut u64, value.as_mut_ptr(),
    )
  });
  (out, value)
}

/// the same with the polynomial and the SRS resident in HBM: `coeffs` = n x 4 words, `g1_srs` = n x 8 words; returns the proof,
/// poly(z) and the quotient (n x 4 words on the same GPU, top entry ZERO like the reference's D-long quotient)
pub fn open_dev(coeffs: &DevicePoly, eval_point: Limbs, g1_srs: &DevicePoly, n: usize) -> (G1Affine, Limbs, DevicePoly) {
  assert!(coeffs.len() >= 4 * n && g1_srs.len() >= 8 * n);
  coeffs.same_device(g1_srs);
  let _g = OnDevice::new(coeffs.device());
  let quot = DevicePoly::alloc_on(coeffs.device(), 4 * n);
  let (mut out, mut value) = (G1Affine::INFINITY, [0u64; 4]);
  ffi::check(unsafe {
    ffi::ronk_kzg_open_bn254_dev(
      coeffs.as_ptr(), n, eval_point.as_ptr(), g1_srs.as_ptr(), quot.as_mut_ptr(), &mut out as *mut G1Affine as *mut u64,
      value.as_mut_ptr(), core::ptr::null_mut(),
    )
  });
  (out, value, quot)
}

/// `poly.div([-z, ONE])` alone (quotient_and_remainder with a monic linear divisor, src/polynomial/mod.rs:170-225) over F_r
pub fn div_linear_dev(coeffs: &DevicePoly, eval_point: Limbs, n: usize) -> (DevicePoly, Limbs) {
  assert!(coeffs.len() >= 4 * n);
  let _g = OnDevice::new(coeffs.device());
  let quot = DevicePoly::alloc_on(coeffs.device(), 4 * n);
  let rem = DevicePoly::alloc_on(coeffs.device(), 4);
  ffi::check(unsafe {
    ffi::ronk_poly_div_linear_bn254_dev(coeffs.as_ptr(), n, eval_point.as_ptr(), quot.as_mut_ptr(), rem.as_mut_ptr(), core::ptr::null_mut())
  });
  let r = rem.to_host();
  (quot, [r[0].0, r[1].0, r[2].0, r[3].0])
}
