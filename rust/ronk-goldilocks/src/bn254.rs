//! `kzg::commit` / `kzg::open` on a production-size group: BN254 (alt_bn128) G1 through the GPU's bucket-method MSM.
//!
//! ronkathon's `commit` folds `g1 * coeff` over `AffinePoint<PlutoExtendedCurve>` (src/kzg/setup.rs:48-60); its field traits
//! are `usize`-wide (src/algebra/mod.rs:8-13), so a 254-bit curve is a new pair of plain-data types here rather than an
//! `EllipticCurve` implementor: 4 x 64-bit little-endian limbs, standard (non-Montgomery) form, `(0, 0)` = infinity --
//! byte-identical to what `ronk_msm_bn254` takes, so slices are passed without conversion.
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
