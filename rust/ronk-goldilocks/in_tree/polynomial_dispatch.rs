// in_tree/polynomial_dispatch.rs -- the edit to ronkathon's src/polynomial/{mod,arithmetic}.rs that routes
// `Polynomial<_, Goldilocks, D>` to the GPU while every other field keeps the reference's bodies.
// (A sketch in the reference's own style; it is applied by hand when vendoring -- see README.md.)

// ---- src/polynomial/mod.rs ------------------------------------------------------------------------------------------
/// private dispatch: one method per specialisable inherent method
pub(crate) trait MonomialOps<F: FiniteField, const D: usize> {
  fn fft_impl(&self) -> Polynomial<Lagrange<F>, F, D>;
  fn dft_impl(&self) -> Polynomial<Lagrange<F>, F, D>;
  fn evaluate_impl(&self, x: F) -> F;
}
impl<F: FiniteField, const D: usize> MonomialOps<F, D> for Polynomial<Monomial, F, D> {
  default fn fft_impl(&self) -> Polynomial<Lagrange<F>, F, D> { /* the body of mod.rs:273-294 (fft_recursive driver) */ unimplemented!() }
  default fn dft_impl(&self) -> Polynomial<Lagrange<F>, F, D> { /* the body of mod.rs:240-258 */ unimplemented!() }
  default fn evaluate_impl(&self, x: F) -> F { /* the body of mod.rs:133-139 */ unimplemented!() }
}
impl<const D: usize> MonomialOps<Goldilocks, D> for Polynomial<Monomial, Goldilocks, D> {
  fn fft_impl(&self) -> Polynomial<Lagrange<Goldilocks>, Goldilocks, D> { Accelerated::fft_gpu(self) }
  fn dft_impl(&self) -> Polynomial<Lagrange<Goldilocks>, Goldilocks, D> { Accelerated::dft_gpu(self) }
  fn evaluate_impl(&self, x: Goldilocks) -> Goldilocks { Accelerated::evaluate_gpu(self, x) }
}
impl<F: FiniteField, const D: usize> Polynomial<Monomial, F, D> {
  pub fn fft(&self) -> Polynomial<Lagrange<F>, F, D>
  where [(); D.is_power_of_two() as usize - 1]: {
    self.fft_impl()
  }
  pub fn dft(&self) -> Polynomial<Lagrange<F>, F, D> { self.dft_impl() }
  pub fn evaluate(&self, x: F) -> F { self.evaluate_impl(x) }
}
// (the same pattern, `LagrangeOps`, for `ifft` mod.rs:430-484 and the barycentric `evaluate` mod.rs:382-415)

// ---- src/polynomial/arithmetic.rs -----------------------------------------------------------------------------------
impl<F: FiniteField, const D: usize, const D2: usize> Mul<Polynomial<Monomial, F, D2>> for Polynomial<Monomial, F, D>
where [(); D + D2 - 1]:
{
  type Output = Polynomial<Monomial, F, { D + D2 - 1 }>;
  default fn mul(self, rhs: Polynomial<Monomial, F, D2>) -> Self::Output { /* arithmetic.rs:110-118, unchanged */ unimplemented!() }
}
impl<const D: usize, const D2: usize> Mul<Polynomial<Monomial, Goldilocks, D2>> for Polynomial<Monomial, Goldilocks, D>
where [(); D + D2 - 1]:
{
  fn mul(self, rhs: Polynomial<Monomial, Goldilocks, D2>) -> Self::Output { self.mul_gpu(&rhs) }
}
// Div / Rem (arithmetic.rs:121-146) call quotient_and_remainder, which dispatches the same way:
//   default fn quotient_and_remainder_impl(..) = mod.rs:170-225;   Goldilocks: self.quotient_and_remainder_gpu(&rhs)
