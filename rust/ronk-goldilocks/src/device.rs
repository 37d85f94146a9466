//! Heap- and device-resident polynomials for sizes the reference's value type cannot hold.
//!
//! `Polynomial<B, F, D>` stores `[F; D]` INLINE (ronkathon src/polynomial/mod.rs:34-44): at the headline size D = 2^22 that
//! is 32 MiB per value, moved through the stack by every `fn(self) -> Self` of the reference -- a default 8 MiB thread
//! stack cannot hold even one.  The types below carry the same data without a const-generic length:
//!
//! * [`HeapPoly`]   -- coefficients in a `Vec<Goldilocks>`; every operation goes through the host-pointer entry points
//!   (pinned staging and chunked copy/compute overlap happen inside the library);
//! * [`DevicePoly`] -- coefficients resident in HBM (`ronk_dev_alloc`), operations through the `_dev` entry points of
//!   include/ronk_ntt.h: a KZG `open` (`poly.div([-z, 1])`, src/kzg/setup.rs:63-78) or an NTT-multiply chain never
//!   crosses PCIe between steps;
//! * [`Plan`]       -- RAII handle of `ronk_plan` (twiddles + scratch for one (n, batch)); `forward_many` hands the library
//!   K unrelated polynomials per call (two transforms in flight inside the library);
//! * [`ShardedPlan`] -- the four-step transform over the GPUs of one node as one call (BASELINE config 5).
//!
//! Semantics (values, panics) are the reference's: `fft` = `Polynomial::fft` (mod.rs:273-323), `ifft` = mod.rs:430-484,
//! `mul` = `impl Mul` (arithmetic.rs:97-119), `evaluate` = mod.rs:133-139, `div_linear` = `impl Div` by a degree-1
//! divisor (arithmetic.rs:121-133 through mod.rs:170-225).
use core::ffi::{c_int, c_void};
use std::ptr;

use crate::{
  ffi::{self, check, RonkPlan, RonkPlanOpts, RonkShardedPlan, G, P},
  field::Goldilocks,
};

#[inline]
fn log2_exact(n: usize) -> u32 {
  // `[(); D.is_power_of_two() as usize - 1]:` of fft()/ifft() (mod.rs:274, :431), as a run-time panic
  assert!(n.is_power_of_two(), "fft/ifft need a power-of-two number of terms");
  n.trailing_zeros()
}

// ------------------------------------------------------------------------------------------------- Plan
/// `ronk_plan`: everything that is fixed per (n = 2^log2n, batch) -- twiddle tables and scratch in HBM.
pub struct Plan {
  raw:       *mut RonkPlan,
  pub log2n: u32,
  pub batch: usize,
  /// the GPU the plan's tables and scratch live on (the current device at creation): every polynomial handed to it must
  /// live there too -- its kernels dereference both
  device:    i32,
}
// the library serialises concurrent calls on one plan internally (stream_mu / staging lock)
unsafe impl Send for Plan {}
unsafe impl Sync for Plan {}

impl Plan {
  /// library defaults (`ronk_plan_create`)
  pub fn new(log2n: u32, batch: usize) -> Self { Self::with_opts(log2n, batch, RonkPlanOpts::default()) }

  /// two transforms in flight behind this handle (`ronk_plan_opts::in_flight = 2`): what `forward_many` needs
  pub fn with_two_lanes(log2n: u32, batch: usize) -> Self {
    Self::with_opts(log2n, batch, RonkPlanOpts { in_flight: 2, ..RonkPlanOpts::default() })
  }

  pub fn with_opts(log2n: u32, batch: usize, opts: RonkPlanOpts) -> Self {
    let mut raw: *mut RonkPlan = ptr::null_mut();
    let device = current_device();
    check(unsafe { ffi::ronk_plan_create_opts(&mut raw, P, G, log2n, batch as u64, device, &opts) });
    Self { raw, log2n, batch, device }
  }

  pub fn n(&self) -> usize { 1usize << self.log2n }

  pub fn device(&self) -> i32 { self.device }

  /// panics unless `p` lives on the plan's GPU (a kernel of GPU a dereferencing memory of GPU b faults when peer access is off)
  pub(crate) fn same_device(&self, p: &DevicePoly) {
    assert!(p.device == self.device, "polynomial on GPU {} handed to a plan on GPU {}", p.device, self.device);
  }

  /// the `ronk_plan*` for the crate's other modules (codes.rs)
  pub(crate) fn raw(&self) -> *mut RonkPlan { self.raw }

  pub fn in_flight(&self) -> i32 { unsafe { ffi::ronk_plan_in_flight(self.raw) } }

  /// host slices of `batch * n` elements (pinned staging inside the library)
  pub fn forward_host(&self, input: &[Goldilocks], output: &mut [Goldilocks]) {
    assert!(input.len() == self.batch * self.n() && output.len() == input.len());
    check(unsafe { ffi::ronk_ntt_forward(self.raw, input.as_ptr() as *const u64, output.as_mut_ptr() as *mut u64, ptr::null_mut()) });
  }

  pub fn inverse_host(&self, input: &[Goldilocks], output: &mut [Goldilocks]) {
    assert!(input.len() == self.batch * self.n() && output.len() == input.len());
    check(unsafe { ffi::ronk_ntt_inverse(self.raw, input.as_ptr() as *const u64, output.as_mut_ptr() as *mut u64) });
  }

  /// device-resident, asynchronous on the null stream; `src` and `dst` may be the same polynomial (see `*_in_place`)
  pub fn forward(&self, src: &DevicePoly, dst: &mut DevicePoly) {
    assert!(src.len == self.batch * self.n() && dst.len == src.len);
    self.same_device(src); self.same_device(dst);
    let _g = OnDevice::new(self.device);
    check(unsafe { ffi::ronk_ntt_forward_dev(self.raw, src.ptr, dst.ptr, ptr::null_mut()) });
  }

  pub fn inverse(&self, src: &DevicePoly, dst: &mut DevicePoly) {
    assert!(src.len == self.batch * self.n() && dst.len == src.len);
    self.same_device(src); self.same_device(dst);
    let _g = OnDevice::new(self.device);
    check(unsafe { ffi::ronk_ntt_inverse_dev(self.raw, src.ptr, dst.ptr, ptr::null_mut()) });
  }

  pub fn forward_in_place(&self, p: &mut DevicePoly) {
    assert!(p.len == self.batch * self.n());
    self.same_device(p);
    let _g = OnDevice::new(self.device);
    check(unsafe { ffi::ronk_ntt_forward_dev(self.raw, p.ptr, p.ptr, ptr::null_mut()) });
  }

  pub fn inverse_in_place(&self, p: &mut DevicePoly) {
    assert!(p.len == self.batch * self.n());
    self.same_device(p);
    let _g = OnDevice::new(self.device);
    check(unsafe { ffi::ronk_ntt_inverse_dev(self.raw, p.ptr, p.ptr, ptr::null_mut()) });
  }

  /// K unrelated polynomials in ONE call (`ronk_ntt_forward_many_dev`): with two lanes the library overlaps the load /
  /// store phases of one transform with the butterflies of another
  pub fn forward_many(&self, src: &[&DevicePoly], dst: &mut [&mut DevicePoly]) {
    assert!(src.len() == dst.len());
    let ins: Vec<*const u64> = src.iter().map(|p| { assert!(p.len == self.batch * self.n()); self.same_device(p); p.ptr as *const u64 }).collect();
    let outs: Vec<*mut u64> = dst.iter().map(|p| { assert!(p.len == self.batch * self.n()); self.same_device(p); p.ptr }).collect();
    let _g = OnDevice::new(self.device);
    check(unsafe { ffi::ronk_ntt_forward_many_dev(self.raw, ins.as_ptr(), outs.as_ptr(), ins.len(), ptr::null_mut()) });
  }

  pub fn inverse_many(&self, src: &[&DevicePoly], dst: &mut [&mut DevicePoly]) {
    assert!(src.len() == dst.len());
    let ins: Vec<*const u64> = src.iter().map(|p| { assert!(p.len == self.batch * self.n()); self.same_device(p); p.ptr as *const u64 }).collect();
    let outs: Vec<*mut u64> = dst.iter().map(|p| { assert!(p.len == self.batch * self.n()); self.same_device(p); p.ptr }).collect();
    let _g = OnDevice::new(self.device);
    check(unsafe { ffi::ronk_ntt_inverse_many_dev(self.raw, ins.as_ptr(), outs.as_ptr(), ins.len(), ptr::null_mut()) });
  }
}

impl Drop for Plan {
  fn drop(&mut self) {
    if !self.raw.is_null() {
      unsafe { ffi::ronk_plan_destroy(self.raw) };
    }
  }
}

// ------------------------------------------------------------------------------------------------- DevicePoly
/// `len` canonical residues resident in HBM (monomial coefficients or values on the roots of unity -- the basis is the
/// caller's knowledge, exactly like the `B` type parameter of the reference)
pub struct DevicePoly {
  ptr:    *mut u64,
  len:    usize,
  /// the GPU the memory lives on: alloc / copy / sync / free select it first (`ronk_set_device`)
  device: i32,
}
unsafe impl Send for DevicePoly {}

/// the calling thread's current device (`ronk_get_device`)
pub fn current_device() -> i32 {
  let mut d = 0 as c_int;
  check(unsafe { ffi::ronk_get_device(&mut d) });
  d
}

/// selects `device` for the scope, restores the previous one on drop (crate-wide: codes.rs and bn254.rs bracket their FFI
/// calls with it -- the `_dev` entry points run on the calling thread's CURRENT device)
pub(crate) struct OnDevice(i32);
impl OnDevice {
  pub(crate) fn new(device: i32) -> Self {
    let prev = current_device();
    if prev != device {
      check(unsafe { ffi::ronk_set_device(device) });
    }
    Self(prev)
  }
}
impl Drop for OnDevice {
  fn drop(&mut self) { unsafe { ffi::ronk_set_device(self.0) }; }
}

impl DevicePoly {
  /// uninitialised device memory for `len` elements on the current device
  pub fn alloc(len: usize) -> Self { Self::alloc_on(current_device(), len) }

  /// ... on GPU `device`: one block per rank of a [`ShardedPlan`] lives on that rank's GPU
  pub fn alloc_on(device: i32, len: usize) -> Self {
    assert!(len > 0);
    let _g = OnDevice::new(device);
    let mut p: *mut c_void = ptr::null_mut();
    check(unsafe { ffi::ronk_dev_alloc(&mut p, len * 8) });
    Self { ptr: p as *mut u64, len, device }
  }

  /// upload (`Polynomial::new(coefficients)` for a device-resident polynomial)
  pub fn from_host(coefficients: &[Goldilocks]) -> Self { Self::from_host_on(current_device(), coefficients) }

  pub fn from_host_on(device: i32, coefficients: &[Goldilocks]) -> Self {
    let d = Self::alloc_on(device, coefficients.len());
    let _g = OnDevice::new(device);
    check(unsafe { ffi::ronk_memcpy_h2d(d.ptr as *mut c_void, coefficients.as_ptr() as *const c_void, d.len * 8) });
    d
  }

  pub fn len(&self) -> usize { self.len }

  pub fn is_empty(&self) -> bool { self.len == 0 }

  pub fn device(&self) -> i32 { self.device }

  /// panics unless `other` lives on the same GPU: the kernels of one GPU dereference both operands
  pub(crate) fn same_device(&self, other: &DevicePoly) {
    assert!(other.device == self.device, "operands on different GPUs ({} and {})", self.device, other.device);
  }

  /// raw device pointers for the crate's other modules (codes.rs, bn254.rs) and for callers that bind further entry points
  pub fn as_ptr(&self) -> *const u64 { self.ptr }

  pub fn as_mut_ptr(&self) -> *mut u64 { self.ptr }

  /// download (synchronises with the polynomial's device first)
  pub fn to_host(&self) -> Vec<Goldilocks> {
    let mut v = vec![Goldilocks(0); self.len];
    let _g = OnDevice::new(self.device);
    check(unsafe { ffi::ronk_dev_sync() });
    check(unsafe { ffi::ronk_memcpy_d2h(v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * 8) });
    v
  }

  /// `Polynomial::fft` with a cached plan of the library (one-shot: no `Plan` to keep)
  pub fn fft(&self) -> DevicePoly {
    let _g = OnDevice::new(self.device);
    let plan = Plan::new(log2_exact(self.len), 1);
    let mut out = DevicePoly::alloc_on(self.device, self.len);
    plan.forward(self, &mut out);
    check(unsafe { ffi::ronk_dev_sync() });   // `plan` (its scratch) is dropped on return
    out
  }

  /// `Polynomial::<Lagrange>::ifft`
  pub fn ifft(&self) -> DevicePoly {
    let _g = OnDevice::new(self.device);
    let plan = Plan::new(log2_exact(self.len), 1);
    let mut out = DevicePoly::alloc_on(self.device, self.len);
    plan.inverse(self, &mut out);
    check(unsafe { ffi::ronk_dev_sync() });
    out
  }

  /// `impl Mul` (arithmetic.rs:97-119): `self.len + rhs.len - 1` coefficients; NTT - pointwise - inverse NTT on the device
  pub fn mul(&self, rhs: &DevicePoly) -> DevicePoly {
    self.same_device(rhs);
    let _g = OnDevice::new(self.device);
    let out = DevicePoly::alloc_on(self.device, self.len + rhs.len - 1);
    check(unsafe { ffi::ronk_poly_mul_dev(P, G, self.ptr, self.len, rhs.ptr, rhs.len, out.ptr, ptr::null_mut()) });
    out
  }

  /// `Polynomial::<Monomial>::evaluate` (mod.rs:133-139); one 8 B/coefficient pass
  pub fn evaluate(&self, x: Goldilocks) -> Goldilocks {
    let _g = OnDevice::new(self.device);
    let y = DevicePoly::alloc_on(self.device, 1);
    check(unsafe { ffi::ronk_poly_eval_dev(P, self.ptr, self.len, x.0, y.ptr, ptr::null_mut()) });
    y.to_host()[0]
  }

  /// `self / [b0, b1]` and the remainder's constant term -- `kzg::open`'s `poly.div(divisor)` with divisor
  /// `[-eval_point, ONE]` (src/kzg/setup.rs:63-78).  The quotient has `len` coefficients, the top one ZERO, like the
  /// reference's D-long quotient.  Panics like the reference for b1 == 0 (`leading_coefficient().inverse().unwrap()`).
  pub fn div_linear(&self, b0: Goldilocks, b1: Goldilocks) -> (DevicePoly, Goldilocks) {
    let _g = OnDevice::new(self.device);
    let quot = DevicePoly::alloc_on(self.device, self.len);
    let rem = DevicePoly::alloc_on(self.device, 1);
    check(unsafe { ffi::ronk_poly_div_linear_dev(P, self.ptr, self.len, b0.0, b1.0, quot.ptr, rem.ptr, ptr::null_mut()) });
    let r = rem.to_host()[0];
    (quot, r)
  }

  /// `self / rhs`, `self % rhs` for any divisor (quotient_and_remainder, mod.rs:170-225): both with `len` coefficients.
  /// Large operands (divisor >= 64 coefficients, quotient >= 2048) run the O(n log n) Newton form inside the library.
  pub fn div_rem(&self, rhs: &DevicePoly) -> (DevicePoly, DevicePoly) {
    self.same_device(rhs);
    let _g = OnDevice::new(self.device);
    let (quot, rem) = (DevicePoly::alloc_on(self.device, self.len), DevicePoly::alloc_on(self.device, self.len));
    let mut status = 0 as c_int;
    let d_status = DevicePoly::alloc_on(self.device, 1);
    check(unsafe { ffi::ronk_memcpy_h2d(d_status.ptr as *mut c_void, &status as *const c_int as *const c_void, 4) });
    check(unsafe {
      ffi::ronk_poly_divrem_dev(P, self.ptr, self.len, rhs.ptr, rhs.len, quot.ptr, rem.ptr, d_status.ptr as *mut c_int, ptr::null_mut())
    });
    check(unsafe { ffi::ronk_dev_sync() });
    check(unsafe { ffi::ronk_memcpy_d2h(&mut status as *mut c_int as *mut c_void, d_status.ptr as *const c_void, 4) });
    check(status);   // the reference's panic (zero divisor, ragged division), reported through the device status word
    (quot, rem)
  }

  /// `div_rem` for operands the caller KNOWS to be full-length (`self[len - 1] != 0`, `rhs[len - 1] != 0`, `self.len >= rhs.len`):
  /// `ronk_poly_divrem_full_dev` -- the O(n log n) form with nothing read back before the work is queued (one synchronisation at
  /// the end, for the status word).  Panics with "invalid argument" when the promise does not hold.
  pub fn div_rem_full(&self, rhs: &DevicePoly) -> (DevicePoly, DevicePoly) {
    self.same_device(rhs);
    let _g = OnDevice::new(self.device);
    let (quot, rem) = (DevicePoly::alloc_on(self.device, self.len), DevicePoly::alloc_on(self.device, self.len));
    let mut status = 0 as c_int;
    let d_status = DevicePoly::alloc_on(self.device, 1);
    check(unsafe {
      ffi::ronk_poly_divrem_full_dev(P, self.ptr, self.len, rhs.ptr, rhs.len, quot.ptr, rem.ptr, d_status.ptr as *mut c_int, ptr::null_mut())
    });
    check(unsafe { ffi::ronk_dev_sync() });
    check(unsafe { ffi::ronk_memcpy_d2h(&mut status as *mut c_int as *mut c_void, d_status.ptr as *const c_void, 4) });
    check(status);
    (quot, rem)
  }

  /// element-wise `impl Add` / `impl Sub` of two equally long polynomials (arithmetic.rs:16-68), `impl Mul` of two
  /// Lagrange-basis polynomials on the same nodes
  pub fn add(&self, rhs: &DevicePoly) -> DevicePoly { self.zip(rhs, ffi::ronk_vec_add_dev) }

  pub fn sub(&self, rhs: &DevicePoly) -> DevicePoly { self.zip(rhs, ffi::ronk_vec_sub_dev) }

  pub fn pointwise_mul(&self, rhs: &DevicePoly) -> DevicePoly { self.zip(rhs, ffi::ronk_vec_mul_dev) }

  fn zip(
    &self,
    rhs: &DevicePoly,
    f: unsafe extern "C" fn(u64, *const u64, *const u64, *mut u64, usize, *mut c_void) -> c_int,
  ) -> DevicePoly {
    self.same_device(rhs);
    let _g = OnDevice::new(self.device);
    assert!(self.len == rhs.len);
    let out = DevicePoly::alloc_on(self.device, self.len);
    check(unsafe { f(P, self.ptr, rhs.ptr, out.ptr, self.len, ptr::null_mut()) });
    out
  }
}

impl Drop for DevicePoly {
  fn drop(&mut self) {
    if !self.ptr.is_null() {
      let _g = OnDevice::new(self.device);
      unsafe {
        ffi::ronk_dev_sync();   // nothing enqueued on ITS device's null stream may still read or write this buffer
        ffi::ronk_dev_free(self.ptr as *mut c_void);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- HeapPoly
/// Coefficients on the heap; operations through the host-pointer entry points (each call stages through HBM).
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct HeapPoly {
  pub coefficients: Vec<Goldilocks>,
}

impl HeapPoly {
  pub fn new(coefficients: Vec<Goldilocks>) -> Self {
    assert!(!coefficients.is_empty());
    Self { coefficients }
  }

  pub fn len(&self) -> usize { self.coefficients.len() }

  pub fn is_empty(&self) -> bool { self.coefficients.is_empty() }

  fn cptr(&self) -> *const u64 { self.coefficients.as_ptr() as *const u64 }

  /// `Polynomial::fft`: values at omega^i, and the node table `Lagrange::new` would build (mod.rs:358-365)
  pub fn fft(&self) -> (HeapPoly, Vec<Goldilocks>) {
    let n = self.len();
    let (mut out, mut nodes) = (vec![Goldilocks(0); n], vec![Goldilocks(0); n]);
    check(unsafe { ffi::ronk_fft(P, G, self.cptr(), out.as_mut_ptr() as *mut u64, nodes.as_mut_ptr() as *mut u64, n) });
    (HeapPoly { coefficients: out }, nodes)
  }

  pub fn ifft(&self) -> HeapPoly {
    let n = self.len();
    let mut out = vec![Goldilocks(0); n];
    check(unsafe { ffi::ronk_ifft(P, G, self.cptr(), out.as_mut_ptr() as *mut u64, n) });
    HeapPoly { coefficients: out }
  }

  pub fn mul(&self, rhs: &HeapPoly) -> HeapPoly {
    let mut out = vec![Goldilocks(0); self.len() + rhs.len() - 1];
    check(unsafe { ffi::ronk_poly_mul(P, G, self.cptr(), self.len(), rhs.cptr(), rhs.len(), out.as_mut_ptr() as *mut u64) });
    HeapPoly { coefficients: out }
  }

  pub fn evaluate(&self, x: Goldilocks) -> Goldilocks {
    let mut y = 0u64;
    check(unsafe { ffi::ronk_poly_eval(P, self.cptr(), self.len(), x.0, &mut y) });
    Goldilocks(y)
  }

  /// `(self / rhs, self % rhs)`, both with `self.len()` coefficients (mod.rs:170-225)
  pub fn div_rem(&self, rhs: &HeapPoly) -> (HeapPoly, HeapPoly) {
    let n = self.len();
    let (mut q, mut r) = (vec![Goldilocks(0); n], vec![Goldilocks(0); n]);
    check(unsafe {
      ffi::ronk_poly_divrem(P, self.cptr(), n, rhs.cptr(), rhs.len(), q.as_mut_ptr() as *mut u64, r.as_mut_ptr() as *mut u64)
    });
    (HeapPoly { coefficients: q }, HeapPoly { coefficients: r })
  }

  /// division by `[b0, b1]` on the device (upload once, quotient back once)
  pub fn div_linear(&self, b0: Goldilocks, b1: Goldilocks) -> (HeapPoly, Goldilocks) {
    let (q, r) = DevicePoly::from_host(&self.coefficients).div_linear(b0, b1);
    (HeapPoly { coefficients: q.to_host() }, r)
  }
}

// ------------------------------------------------------------------------------------------------- ShardedPlan
/// `ronk_sharded_plan`: one 2^log2n-point transform over `devices` (a power of two of them) as a four-step NTT whose
/// exchange the library issues itself.  The reference has no counterpart (a degree this large never exists there).
pub struct ShardedPlan {
  raw:         *mut RonkShardedPlan,
  pub n:       usize,
  pub ndev:    usize,
  pub devices: Vec<i32>,
}
unsafe impl Send for ShardedPlan {}

/// how the transpose of the four-step transform travels between the GPUs
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Exchange {
  /// `hipMemcpyPeerAsync` copies, one copy stream per destination device (`RONK_EXCHANGE_MESH`)
  Mesh,
  /// one `ncclGroup` of `ncclSend` / `ncclRecv` pairs per column chunk through RCCL over xGMI (`RONK_EXCHANGE_RCCL`);
  /// the ranks must sit on distinct GPUs
  Rccl,
}

impl ShardedPlan {
  /// the library's default exchange (the peer-copy mesh)
  pub fn new(log2n: u32, inverse: bool, devices: &[i32], chunks: i32) -> Self {
    Self::with_exchange(log2n, inverse, devices, chunks, Exchange::Mesh)
  }

  /// `ronk_sharded_plan_create_ex`: panics (`RONK_ERR_RCCL`) when RCCL is asked for and `librccl.so` cannot be loaded
  pub fn with_exchange(log2n: u32, inverse: bool, devices: &[i32], chunks: i32, exchange: Exchange) -> Self {
    let mut raw: *mut RonkShardedPlan = ptr::null_mut();
    let ex = match exchange { Exchange::Mesh => ffi::EXCHANGE_MESH, Exchange::Rccl => ffi::EXCHANGE_RCCL };
    check(unsafe {
      ffi::ronk_sharded_plan_create_ex(&mut raw, log2n, inverse as c_int, devices.as_ptr(), devices.len() as c_int, chunks, ex)
    });
    Self { raw, n: 1usize << log2n, ndev: devices.len(), devices: devices.to_vec() }
  }

  /// `ronk_sharded_plan_create_p`: the same transform over any odd 64-bit prime `p` with 2^log2n | p - 1 and primitive element
  /// `g` -- what `Prime64<P, G>` hands over (`ShardedPlan::for_prime(P, G, ..)`); the phases run the tile kernels over Montgomery
  /// arithmetic.  The reference's field is generic over its modulus (src/algebra/field/prime/mod.rs:39-52).
  pub fn for_prime(p: u64, g: u64, log2n: u32, inverse: bool, devices: &[i32], chunks: i32, exchange: Exchange) -> Self {
    let mut raw: *mut RonkShardedPlan = ptr::null_mut();
    let ex = match exchange { Exchange::Mesh => ffi::EXCHANGE_MESH, Exchange::Rccl => ffi::EXCHANGE_RCCL };
    check(unsafe {
      ffi::ronk_sharded_plan_create_p(&mut raw, p, g, log2n, inverse as c_int, devices.as_ptr(), devices.len() as c_int, chunks, ex)
    });
    Self { raw, n: 1usize << log2n, ndev: devices.len(), devices: devices.to_vec() }
  }

  /// the exchange this plan runs (`ronk_sharded_plan_exchange`)
  pub fn exchange(&self) -> Exchange {
    match unsafe { ffi::ronk_sharded_plan_exchange(self.raw) } {
      ffi::EXCHANGE_RCCL => Exchange::Rccl,
      rc if rc < 0 => { check(rc); unreachable!() },
      _ => Exchange::Mesh,
    }
  }

  /// How blocks travel between the ranks of the mesh exchange (`ronk_sharded_plan_peer_access`): the row-major
  /// `ndev x ndev` matrix of `ffi::PEER_SAME_DEVICE` / `PEER_DIRECT` / `PEER_STAGED`, and the number of STAGED pairs -- pairs
  /// whose copies the runtime routes through host memory because peer access was refused.  0 on a healthy xGMI node; a caller
  /// that measures should assert it.
  pub fn peer_access(&self) -> (Vec<i32>, usize) {
    let mut m = vec![0 as c_int; self.ndev * self.ndev];
    let staged = unsafe { ffi::ronk_sharded_plan_peer_access(self.raw, m.as_mut_ptr(), m.len() as c_int) };
    check(if staged < 0 { staged } else { 0 });
    (m, staged as usize)
  }

  /// uninitialised per-rank blocks, block g on `devices[g]` (what `transform` takes on both sides)
  pub fn alloc_blocks(&self) -> Vec<DevicePoly> {
    let per = self.info().2 as usize;
    self.devices.iter().map(|&d| DevicePoly::alloc_on(d, per)).collect()
  }

  /// (rows R, columns C, elements per rank, column chunks in use)
  pub fn info(&self) -> (u64, u64, u64, i32) {
    let (mut r, mut c, mut per, mut ch) = (0u64, 0u64, 0u64, 0 as c_int);
    check(unsafe { ffi::ronk_sharded_plan_info(self.raw, &mut r, &mut c, &mut per, &mut ch) });
    (r, c, per, ch)
  }

  /// host vectors of n elements, natural order in and out (scatter, transform, gather; synchronous)
  pub fn transform_host(&self, input: &[Goldilocks], output: &mut [Goldilocks]) {
    assert!(input.len() == self.n && output.len() == self.n);
    check(unsafe { ffi::ronk_ntt_sharded(self.raw, input.as_ptr() as *const u64, output.as_mut_ptr() as *mut u64) });
  }

  /// device-resident blocks, one per rank on that rank's GPU (layouts: include/ronk_ntt.h); asynchronous, see `sync`
  pub fn transform(&self, blocks_in: &[&DevicePoly], blocks_out: &mut [&mut DevicePoly]) {
    assert!(blocks_in.len() == self.ndev && blocks_out.len() == self.ndev);
    let per = self.info().2 as usize;
    for (g, &d) in self.devices.iter().enumerate() {
      assert!(blocks_in[g].device == d && blocks_out[g].device == d, "block {g} must live on GPU {d}");
      assert!(blocks_in[g].len == per && blocks_out[g].len == per);
    }
    let ins: Vec<*const u64> = blocks_in.iter().map(|p| p.ptr as *const u64).collect();
    let outs: Vec<*mut u64> = blocks_out.iter().map(|p| p.ptr).collect();
    check(unsafe { ffi::ronk_ntt_sharded_dev(self.raw, ins.as_ptr(), outs.as_ptr()) });
  }

  pub fn sync(&self) { check(unsafe { ffi::ronk_sharded_sync(self.raw) }); }
}

impl Drop for ShardedPlan {
  fn drop(&mut self) {
    if !self.raw.is_null() {
      unsafe { ffi::ronk_sharded_plan_destroy(self.raw) };
    }
  }
}

#[cfg(test)]
mod tests {
  //! need a GPU; sizes the inline `[F; D]` type cannot reach without a 32 MiB stack frame
  use ronkathon::algebra::field::{Field, FiniteField};

  use super::*;

  fn ramp(n: usize) -> Vec<Goldilocks> { (0..n as u64).map(|i| Goldilocks::new(i.wrapping_mul(0x9E37_79B9_7F4A_7C15))).collect() }

  #[test]
  fn device_round_trip_2_22() {
    let x = ramp(1 << 22);
    let d = DevicePoly::from_host(&x);
    assert_eq!(d.fft().ifft().to_host(), x);
  }

  #[test]
  fn heap_and_device_agree() {
    let x = HeapPoly::new(ramp(1 << 16));
    let (y, nodes) = x.fft();
    assert_eq!(nodes[1], Goldilocks::primitive_root_of_unity(1 << 16));
    assert_eq!(DevicePoly::from_host(&x.coefficients).fft().to_host(), y.coefficients);
    assert_eq!(y.ifft(), x);
  }

  #[test]
  fn open_quotient_on_device() {
    // kzg::open: q = (p - p(z)) / (x - z)  <=>  p = q (x - z) + r with r = p(z)
    let p = DevicePoly::from_host(&ramp(1 << 20));
    let z = Goldilocks(0x1234_5678_9ABC_DEF1);
    let (q, r) = p.div_linear(-z, Goldilocks::ONE);
    assert_eq!(r, p.evaluate(z));
    let t = Goldilocks(0xFEED_FACE_1234_5);
    assert_eq!(p.evaluate(t), q.evaluate(t) * (t - z) + r);
  }

  #[test]
  fn many_in_one_call() {
    let plan = Plan::with_two_lanes(19, 1);
    assert_eq!(plan.in_flight(), 2);
    let xs: Vec<Vec<Goldilocks>> = (0..4).map(|k| ramp(1 << 19).into_iter().map(|v| v + Goldilocks(k)).collect()).collect();
    let src: Vec<DevicePoly> = xs.iter().map(|x| DevicePoly::from_host(x)).collect();
    let mut dst: Vec<DevicePoly> = (0..4).map(|_| DevicePoly::alloc(1 << 19)).collect();
    plan.forward_many(&src.iter().collect::<Vec<_>>(), &mut dst.iter_mut().collect::<Vec<_>>());
    for (x, y) in xs.iter().zip(&dst) {
      assert_eq!(y.to_host(), HeapPoly::new(x.clone()).fft().0.coefficients);
    }
  }
}
