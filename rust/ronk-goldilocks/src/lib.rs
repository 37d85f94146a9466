//! ronk-goldilocks: the reference-side binding of libronk_ntt.so (MI355X-native NTT / polynomial engine).
//!
//! * [`field::Goldilocks`] -- a 64-bit implementor of ronkathon's `Finite` / `Field` / `FiniteField`
//!   (src/algebra/mod.rs:8-13, src/algebra/field/mod.rs:17-76) with every operator, conversion, `Display`, `FromStr` and
//!   `Distribution` impl `PrimeField<P>` has (src/algebra/field/prime/{mod,arithmetic}.rs).  Code that is generic over
//!   `F: FiniteField` -- `Polynomial<B, F, D>` and everything built on it -- compiles against it unchanged.  `src/codes` and
//!   `src/kzg` are NOT generic: `Message<K, P>` is fixed to `PrimeField<P>` (src/codes/reed_solomon.rs:15-17, :37-52) and
//!   `kzg::{commit, open}` to `PlutoScalarField` / `PlutoExtendedCurve` (src/kzg/setup.rs:48-78); they keep working on the
//!   small fields they were written for, and their 64-bit counterparts are [`codes`] (same `Message` / `Codeword` /
//!   `Coordinate` shapes over `Goldilocks`) and [`bn254`] + `DevicePoly::div_linear` (commit / open on BN254).
//! * [`prime64::Prime64`] -- the same for ANY odd 64-bit prime, modulus and primitive element as const generics: the 64-bit
//!   counterpart of the reference's generic `PrimeField<const P: usize>`; its arrays run the same tile kernels over Montgomery
//!   arithmetic ([`prime64::AcceleratedPrime`], [`prime64::PrimePlan`]).
//! * [`polynomial::Accelerated`] / [`polynomial::AcceleratedLagrange`] -- `fft` / `ifft` / `dft` / `Mul` / `Div` / `Rem` /
//!   `evaluate` on the GPU through the C ABI (include/ronk_ntt.h), bit-exact with the reference's CPU results.
//! * [`device::HeapPoly`] / [`device::DevicePoly`] / [`device::Plan`] / [`device::ShardedPlan`] -- heap- and HBM-resident
//!   polynomials for the sizes the inline `[F; D]` of the reference cannot hold (2^22 coefficients = 32 MiB per value),
//!   plans, K transforms per call, and the transform sharded over the GPUs of a node.
//! * [`bn254::commit`] -- `kzg::commit` (src/kzg/setup.rs:48-60) over BN254 G1 through the GPU's bucket-method MSM.
//! * [`codes`] -- `Message::encode::<N>` / `decode::<M>` (src/codes/reed_solomon.rs:37-107) over `Goldilocks`, plus the batched
//!   device forms (`encode_batch`, `lde`, `decode_dev`).
//! * `in_tree/ronkathon.patch` -- the unified diff that vendors these sources INSIDE ronkathon as specialisations
//!   (`default fn` + `Goldilocks` impls), so that `poly.fft()`, `a * b`, `a / b`, `p.evaluate(x)` call sites do not change.
//!
//! The nightly features mirror ronkathon's own (src/lib.rs:15-24); `generic_const_exprs` is needed for `D + D2 - 1`.
#![allow(incomplete_features)]
#![feature(generic_const_exprs)]
#![feature(const_trait_impl)]
#![feature(effects)]

pub mod bn254;
pub mod codes;
pub mod device;
pub mod ffi;
pub mod field;
pub mod polynomial;
pub mod prime64;

pub use device::{DevicePoly, Exchange, HeapPoly, Plan, ShardedPlan};
pub use field::Goldilocks;
pub use polynomial::{rs_decode, Accelerated, AcceleratedLagrange};
pub use prime64::{AcceleratedPrime, AcceleratedPrimeLagrange, Prime64, PrimePlan};
