//! ronk-goldilocks: the reference-side binding of libronk_ntt.so (MI355X-native NTT / polynomial engine).
//!
//! * [`field::Goldilocks`] -- a 64-bit implementor of ronkathon's `Finite` / `Field` / `FiniteField`
//!   (src/algebra/mod.rs:8-13, src/algebra/field/mod.rs:17-76) with every operator, conversion, `Display`, `FromStr` and
//!   `Distribution` impl `PrimeField<P>` has (src/algebra/field/prime/{mod,arithmetic}.rs).  Generic code --
//!   `Polynomial<B, F, D>`, `src/kzg`, `src/codes` -- compiles against it unchanged.
//! * [`polynomial::Accelerated`] / [`polynomial::AcceleratedLagrange`] -- `fft` / `ifft` / `dft` / `Mul` / `Div` / `Rem` /
//!   `evaluate` on the GPU through the C ABI (include/ronk_ntt.h), bit-exact with the reference's CPU results.
//! * [`device::HeapPoly`] / [`device::DevicePoly`] / [`device::Plan`] / [`device::ShardedPlan`] -- heap- and HBM-resident
//!   polynomials for the sizes the inline `[F; D]` of the reference cannot hold (2^22 coefficients = 32 MiB per value),
//!   plans, K transforms per call, and the transform sharded over the GPUs of a node.
//! * [`bn254::commit`] -- `kzg::commit` (src/kzg/setup.rs:48-60) over BN254 G1 through the GPU's bucket-method MSM.
//! * `in_tree/` -- how the same bodies become specialisations when vendored inside ronkathon, so call sites do not change.
//!
//! The nightly features mirror ronkathon's own (src/lib.rs:15-24); `generic_const_exprs` is needed for `D + D2 - 1`.
#![allow(incomplete_features)]
#![feature(generic_const_exprs)]
#![feature(const_trait_impl)]
#![feature(effects)]

pub mod bn254;
pub mod device;
pub mod ffi;
pub mod field;
pub mod polynomial;

pub use device::{DevicePoly, HeapPoly, Plan, ShardedPlan};
pub use field::Goldilocks;
pub use polynomial::{rs_decode, Accelerated, AcceleratedLagrange};
