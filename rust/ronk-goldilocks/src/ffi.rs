//! `extern "C"` block for include/ronk_ntt.h (every entry point the shim calls) and the code -> panic mapping.
//!
//! Every function returns 0 or a negative `RONK_ERR_*`; the reference reports the same conditions by panicking, so
//! [`check`] panics with `ronk_strerror(code)`, which repeats the reference's panic texts ("n must divide p^q - 1",
//! `called Option::unwrap() on a None value`, ...): `#[should_panic]` tests such as
//! ronkathon `src/polynomial/tests.rs:46-55` and `src/algebra/field/prime/mod.rs:386-391` keep passing.
//!
//! tests/test_cpp_host_mirror.py (engine repository) checks every declaration below against the header: same symbol,
//! same number of parameters, same parameter TYPES (u64 = uint64_t, usize = size_t, c_int = int, ...).
use core::ffi::{c_char, c_int, c_void};

pub const P: u64 = 0xFFFF_FFFF_0000_0001;
pub const G: u64 = 7;
/// `RONK_EXCHANGE_MESH` / `RONK_EXCHANGE_RCCL` (include/ronk_ntt.h)
pub const EXCHANGE_MESH: c_int = 0;
pub const EXCHANGE_RCCL: c_int = 1;
pub const PEER_SAME_DEVICE: c_int = 0;
pub const PEER_DIRECT: c_int = 1;
pub const PEER_STAGED: c_int = 2;

/// `ronk_plan` (opaque)
#[repr(C)]
pub struct RonkPlan {
  _private: [u8; 0],
}
/// `ronk_sharded_plan` (opaque)
#[repr(C)]
pub struct RonkShardedPlan {
  _private: [u8; 0],
}
/// `ronk_plan_opts`: -1 = library default for every field
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RonkPlanOpts {
  pub tile_log2_columns:       c_int,
  pub twiddle_matrix_log2_max: c_int,
  /// 1 = caller's stream only, 2 = two transforms in flight behind one handle, -1 = automatic
  pub in_flight:               c_int,
  /// two-pass plans: log2 of the first pass's rows, 0 = the planner's (balanced) choice
  pub split_log2_rows:         c_int,
  /// 0 = the planner's choice; 13 .. 25: the smallest log2n that is split in three passes
  pub three_pass_from_log2:    c_int,
  pub reserved:                [c_int; 3],
}
impl Default for RonkPlanOpts {
  fn default() -> Self {
    Self { tile_log2_columns: -1, twiddle_matrix_log2_max: -1, in_flight: -1, split_log2_rows: 0, three_pass_from_log2: 0, reserved: [0; 3] }
  }
}

extern "C" {
  pub fn ronk_strerror(code: c_int) -> *const c_char;
  pub fn ronk_last_hip_error() -> *const c_char;
  pub fn ronk_device_count(count: *mut c_int) -> c_int;

  // ---- host-pointer one-shot forms (what `Polynomial<_, Goldilocks, D>` with its inline `[F; D]` calls)
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
  /// `Message::encode::<N>` (src/codes/reed_solomon.rs:42-52): xs[i] = omega_N^i, ys[i] = poly(omega_N^i)
  pub fn ronk_rs_encode(p: u64, g: u64, msg: *const u64, k: usize, n: usize, xs: *mut u64, ys: *mut u64) -> c_int;
  /// `kzg::commit` (src/kzg/setup.rs:48-60) over BN254 G1: points n x [x: 4 limbs, y: 4 limbs], scalars n x 4 limbs
  pub fn ronk_msm_bn254(points: *const u64, scalars: *const u64, n: usize, out: *mut u64) -> c_int;
  /// the same with device-resident points and scalars (the result comes back to the host: 8 limbs)
  pub fn ronk_msm_bn254_dev(d_points: *const u64, d_scalars: *const u64, n: usize, out: *mut u64, stream: *mut c_void) -> c_int;
  /// `poly.div([-z, ONE])` over BN254's scalar field on device-resident coefficients (n x 4 words): quotient (n entries, top
  /// ZERO) and the remainder's constant term poly(z) (4 device words, may be null) -- the first half of `kzg::open`,
  /// src/kzg/setup.rs:63-78
  pub fn ronk_poly_div_linear_bn254_dev(
    d_coeffs: *const u64, n: usize, z: *const u64, d_quot: *mut u64, d_rem: *mut u64, stream: *mut c_void,
  ) -> c_int;
  /// `kzg::open` over BN254: the division, then `commit(quotient, srs)`; `out_value` (may be null) receives poly(z)
  pub fn ronk_kzg_open_bn254_dev(
    d_coeffs: *const u64, n: usize, z: *const u64, d_srs: *const u64, d_quot: *mut u64, out_point: *mut u64, out_value: *mut u64,
    stream: *mut c_void,
  ) -> c_int;
  pub fn ronk_kzg_open_bn254(
    coeffs: *const u64, n: usize, z: *const u64, srs: *const u64, n_srs: usize, out_point: *mut u64, out_value: *mut u64,
  ) -> c_int;

  // ---- plans and device-resident forms (device.rs: `Plan`, `DevicePoly`): coefficients stay in HBM between calls
  pub fn ronk_plan_create(out: *mut *mut RonkPlan, p: u64, g: u64, log2n: u32, batch: u64, device: c_int) -> c_int;
  pub fn ronk_plan_create_tuned(
    out: *mut *mut RonkPlan, p: u64, g: u64, log2n: u32, batch: u64, device: c_int, tile_log2_columns: c_int,
    twiddle_matrix_log2_max: c_int,
  ) -> c_int;
  pub fn ronk_plan_create_opts(
    out: *mut *mut RonkPlan, p: u64, g: u64, log2n: u32, batch: u64, device: c_int, opts: *const RonkPlanOpts,
  ) -> c_int;
  pub fn ronk_plan_in_flight(plan: *const RonkPlan) -> c_int;
  /// which kernels the plan runs: 1 = tiled Goldilocks, 2 = the tile kernels over Montgomery arithmetic (any other odd prime whose
  /// g is a quadratic non-residue), 0 = the radix-2 path
  pub fn ronk_plan_path(plan: *const RonkPlan) -> c_int;
  /// `PrimeField::new`'s `is_prime` assertion (src/algebra/field/prime/mod.rs:48-51) as a deterministic Miller-Rabin
  pub fn ronk_check_prime(p: u64) -> c_int;
  /// `FieldExt::euler_criterion` over an array (src/algebra/field/prime/mod.rs:142-172): out[i] = 1 for a non-zero square
  pub fn ronk_vec_euler(p: u64, a: *const u64, out: *mut u64, n: usize) -> c_int;
  /// `FieldExt::sqrt` over an array (src/algebra/field/prime/mod.rs:174-226): (smaller root, larger root) per element; a
  /// non-residue is the reference's assert ("Element is not a quadratic residue")
  pub fn ronk_vec_sqrt(p: u64, a: *const u64, r0: *mut u64, r1: *mut u64, n: usize) -> c_int;
  pub fn ronk_plan_destroy(plan: *mut RonkPlan) -> c_int;
  /// host pointers through the plan's pinned staging ring (`nodes` may be null)
  pub fn ronk_ntt_forward(plan: *mut RonkPlan, input: *const u64, output: *mut u64, nodes: *mut u64) -> c_int;
  pub fn ronk_ntt_inverse(plan: *mut RonkPlan, input: *const u64, output: *mut u64) -> c_int;
  /// device pointers, asynchronous on `stream` (a hipStream_t; null = the null stream)
  pub fn ronk_ntt_forward_dev(plan: *mut RonkPlan, d_in: *const u64, d_out: *mut u64, stream: *mut c_void) -> c_int;
  pub fn ronk_ntt_inverse_dev(plan: *mut RonkPlan, d_in: *const u64, d_out: *mut u64, stream: *mut c_void) -> c_int;
  /// `count` unrelated device arrays in one call (two lanes inside the library)
  pub fn ronk_ntt_forward_many_dev(
    plan: *mut RonkPlan, d_in: *const *const u64, d_out: *const *mut u64, count: usize, stream: *mut c_void,
  ) -> c_int;
  pub fn ronk_ntt_inverse_many_dev(
    plan: *mut RonkPlan, d_in: *const *const u64, d_out: *const *mut u64, count: usize, stream: *mut c_void,
  ) -> c_int;
  pub fn ronk_poly_mul_dev(
    p: u64, g: u64, d_a: *const u64, d: usize, d_b: *const u64, d2: usize, d_out: *mut u64, stream: *mut c_void,
  ) -> c_int;
  pub fn ronk_poly_eval_dev(p: u64, d_c: *const u64, d: usize, x: u64, d_out: *mut u64, stream: *mut c_void) -> c_int;
  /// `kzg::open`'s `poly.div([-z, 1])` (src/kzg/setup.rs:63-78): quotient by b0 + b1 x, remainder's constant term
  pub fn ronk_poly_div_linear_dev(
    p: u64, d_c: *const u64, d: usize, b0: u64, b1: u64, d_quot: *mut u64, d_rem: *mut u64, stream: *mut c_void,
  ) -> c_int;
  pub fn ronk_poly_divrem_dev(
    p: u64, d_a: *const u64, d: usize, d_b: *const u64, d2: usize, d_quot: *mut u64, d_rem: *mut u64,
    d_status: *mut c_int, stream: *mut c_void,
  ) -> c_int;
  /// the same for FULL-LENGTH operands (top coefficients non-zero, d >= d2): nothing is read back, so the call is asynchronous
  /// and capturable at any size; `*d_status = RONK_ERR_INVALID` when a top coefficient turns out to be ZERO
  pub fn ronk_poly_divrem_full_dev(
    p: u64, d_a: *const u64, d: usize, d_b: *const u64, d2: usize, d_quot: *mut u64, d_rem: *mut u64,
    d_status: *mut c_int, stream: *mut c_void,
  ) -> c_int;
  /// batched `Message::encode::<N>` (codes/reed_solomon.rs:42-52): `plan.batch` messages of k coefficients -> batch x N y-coordinates
  pub fn ronk_rs_encode_batch_dev(plan: *mut RonkPlan, d_msgs: *const u64, k: usize, d_ys: *mut u64, stream: *mut c_void) -> c_int;
  /// low-degree extension of a batch: values on {omega_K^i} -> values on coset_shift * {omega_N^i} (ifft, then encode::<N>)
  pub fn ronk_lde_batch_dev(
    plan_k: *mut RonkPlan, plan_n: *mut RonkPlan, d_evals: *const u64, d_coeffs: *mut u64, d_out: *mut u64, coset_shift: u64,
    stream: *mut c_void,
  ) -> c_int;
  /// `Message::decode` (codes/reed_solomon.rs:54-106) on device-resident coordinates; `d_status` may be null
  pub fn ronk_rs_decode_dev(
    p: u64, d_xs: *const u64, d_ys: *const u64, k: usize, d_out: *mut u64, d_status: *mut c_int, stream: *mut c_void,
  ) -> c_int;
  pub fn ronk_vec_add_dev(p: u64, a: *const u64, b: *const u64, out: *mut u64, n: usize, stream: *mut c_void) -> c_int;
  pub fn ronk_vec_sub_dev(p: u64, a: *const u64, b: *const u64, out: *mut u64, n: usize, stream: *mut c_void) -> c_int;
  pub fn ronk_vec_mul_dev(p: u64, a: *const u64, b: *const u64, out: *mut u64, n: usize, stream: *mut c_void) -> c_int;
  pub fn ronk_dev_alloc(ptr: *mut *mut c_void, bytes: usize) -> c_int;
  pub fn ronk_dev_free(ptr: *mut c_void) -> c_int;
  pub fn ronk_memcpy_h2d(dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
  pub fn ronk_memcpy_d2h(dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
  pub fn ronk_dev_sync() -> c_int;
  /// the calling thread's current device: what `ronk_dev_alloc` / `ronk_memcpy_*` / `ronk_dev_sync` act on
  pub fn ronk_set_device(device: c_int) -> c_int;
  pub fn ronk_get_device(device: *mut c_int) -> c_int;
  /// releases the library's cached device workspace (buffers above 256 MiB are never cached)
  pub fn ronk_trim_workspace() -> c_int;

  // ---- the sharded four-step transform (BASELINE config 5: 2^26 over the GPUs of one node) as one call
  pub fn ronk_sharded_plan_create(
    out: *mut *mut RonkShardedPlan, log2n: u32, inverse: c_int, devices: *const c_int, ndev: c_int, chunks: c_int,
  ) -> c_int;
  /// the same with the exchange chosen per plan: [`EXCHANGE_MESH`] (hipMemcpyPeerAsync, one copy stream per peer) or
  /// [`EXCHANGE_RCCL`] (ncclGroup of Send / Recv pairs through a dlopen'ed librccl: the "RCCL all-to-all over xGMI")
  pub fn ronk_sharded_plan_create_ex(
    out: *mut *mut RonkShardedPlan, log2n: u32, inverse: c_int, devices: *const c_int, ndev: c_int, chunks: c_int,
    exchange: c_int,
  ) -> c_int;
  /// the same over any odd 64-bit prime `p` with 2^log2n | p - 1 and primitive element `g` (the phases run the tile kernels over
  /// Montgomery arithmetic); `Prime64<P, G>`'s sharded transform
  pub fn ronk_sharded_plan_create_p(
    out: *mut *mut RonkShardedPlan, p: u64, g: u64, log2n: u32, inverse: c_int, devices: *const c_int, ndev: c_int,
    chunks: c_int, exchange: c_int,
  ) -> c_int;
  pub fn ronk_sharded_plan_exchange(plan: *const RonkShardedPlan) -> c_int;
  pub fn ronk_sharded_plan_peer_access(plan: *const RonkShardedPlan, matrix: *mut c_int, capacity: c_int) -> c_int;
  pub fn ronk_sharded_plan_destroy(plan: *mut RonkShardedPlan) -> c_int;
  pub fn ronk_sharded_plan_info(
    plan: *const RonkShardedPlan, rows: *mut u64, cols: *mut u64, per_rank: *mut u64, chunks: *mut c_int,
  ) -> c_int;
  pub fn ronk_ntt_sharded_dev(plan: *mut RonkShardedPlan, d_in: *const *const u64, d_out: *const *mut u64) -> c_int;
  pub fn ronk_sharded_sync(plan: *mut RonkShardedPlan) -> c_int;
  pub fn ronk_ntt_sharded(plan: *mut RonkShardedPlan, input: *const u64, output: *mut u64) -> c_int;
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
