"""ronkathon_amd -- MI355X-native finite-field / NTT / polynomial engine behind ronkathon's
`Polynomial<B, F, D>` + `Field` / `FiniteField` surface.

Layout: csrc/ (hand-written HIP for gfx950 + the C ABI of include/ronk_ntt.h), and a thin host
mirror of the reference interface for this path: field.py (PrimeField / FiniteField,
src/algebra/field), polynomial.py (Polynomial, Monomial, Lagrange, src/polynomial),
callers.py (Reed-Solomon encode, KZG open quotient), dist.py (multi-GPU four-step).
Importing the package requires the built shared library; there is no CPU fallback.
"""
from ._lib import GOLDILOCKS_G, GOLDILOCKS_P, Plan, RonkPanic, device_count  # noqa: F401
from .field import (AESField, GoldilocksField, PlutoBaseField, PlutoScalarField, PrimeField)  # noqa: F401
from .polynomial import Lagrange, Monomial, Polynomial  # noqa: F401
