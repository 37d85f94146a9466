"""Host mirror of ronkathon's `Polynomial<B: Basis, F: FiniteField, const D: usize>`
(reference src/polynomial/mod.rs, src/polynomial/arithmetic.rs) over the C ABI.

Same names, argument meaning and error behaviour as the reference; the const generic D is
`len(coefficients)`.  Coefficients live in a numpy uint64 array of canonical residues (the
`[F; D]` of the reference, 8 bytes per element); every method that computes on them calls
into libronk_ntt.so (HIP kernels) -- none of the arithmetic below happens in Python.
"""
import numpy as np

from . import _lib as L


class Monomial:
    """`Monomial` basis marker (polynomial/mod.rs:55-59)."""
    def __eq__(self, o): return isinstance(o, Monomial)
    def __hash__(self): return 0
    def __repr__(self): return "Monomial"


class Lagrange:
    """`Lagrange<F>{ nodes }` (polynomial/mod.rs:65-72)."""
    def __init__(self, nodes):
        self.nodes = L.arr(nodes)
    def __eq__(self, o): return isinstance(o, Lagrange) and np.array_equal(self.nodes, o.nodes)
    def __repr__(self): return "Lagrange(%d nodes)" % self.nodes.size


def _coeffs(field, c):
    if isinstance(c, np.ndarray) and c.dtype == np.uint64:
        c = np.ascontiguousarray(c)
        # the kernels assume canonical residues (include/ronk_ntt.h); PrimeField::new reduces with `% P`
        # (prime/mod.rs:48-51), so do the same for raw 64-bit values instead of returning wrong sums
        if c.size and int(c.max()) >= field.ORDER:
            c = c % np.uint64(field.ORDER)
        return c
    return L.arr([int(x) % field.ORDER for x in c])


class Polynomial:
    def __init__(self, field, coefficients, basis=None):
        self.field = field
        self.coefficients = _coeffs(field, coefficients)
        self.basis = basis if basis is not None else Monomial()

    # ---- constructors
    @classmethod
    def new(cls, field, coefficients):
        """Polynomial::<Monomial, F, D>::new (mod.rs:98)."""
        return cls(field, coefficients, Monomial())

    @classmethod
    def new_lagrange(cls, field, coefficients):
        """Polynomial::<Lagrange<F>, F, D>::new (mod.rs:358-365): asserts (ORDER-1) % n == 0, builds nodes."""
        c = _coeffs(field, coefficients)
        nodes = np.empty(c.size, dtype=np.uint64)
        L.check(L.lib.ronk_lagrange_nodes(field.ORDER, field._G, L.ptr(nodes), c.size))
        return cls(field, c, Lagrange(nodes))

    @classmethod
    def from_coeffs(cls, field, coeffs, D):
        """From<[F; N]> (mod.rs:503-515): zero-pad or truncate to D."""
        c = _coeffs(field, coeffs)
        out = np.zeros(D, dtype=np.uint64)
        k = min(D, c.size)
        out[:k] = c[:k]
        return cls(field, out, Monomial())

    # ---- shape helpers (pure bookkeeping on the host array)
    def num_terms(self): return int(self.coefficients.size)           # mod.rs:87
    @property
    def D(self): return int(self.coefficients.size)

    def degree(self):                                                  # mod.rs:113-115
        nz = np.flatnonzero(self.coefficients)
        return int(nz[-1]) if nz.size else 0

    def leading_coefficient(self):                                     # mod.rs:120-122
        nz = np.flatnonzero(self.coefficients)
        return self.field(int(self.coefficients[nz[-1]])) if nz.size else self.field.ZERO

    def __eq__(self, o):                                               # derived PartialEq (mod.rs:34)
        return (isinstance(o, Polynomial) and o.field is self.field and self.basis == o.basis
                and np.array_equal(self.coefficients, o.coefficients))

    def __repr__(self):
        return "Polynomial<%r, %s, %d>%s" % (self.basis, self.field.__name__, self.D,
                                              self.coefficients[:8].tolist())

    def __str__(self):
        """`Display` (Monomial: mod.rs:326-342, Lagrange: mod.rs:487-501): `c0 + c1x^1 + ...` / `c0*l_0(x) + ...`"""
        c = self.coefficients.tolist()
        if isinstance(self.basis, Monomial):
            return " + ".join(("%d" % v) if i == 0 else ("%dx^%d" % (v, i)) for i, v in enumerate(c))
        nodes = self.basis.nodes.tolist()         # the subscript is the NODE, as in the reference
        return " + ".join("%d*l_%d(x)" % (v, nd) for v, nd in zip(c, nodes))

    def _mono(self):
        if not isinstance(self.basis, Monomial):
            raise TypeError("operation is only implemented for the Monomial basis")

    # ---- evaluation
    def evaluate(self, x):
        """Monomial: sum c_i x^i (mod.rs:133-139).  Lagrange: barycentric formula (mod.rs:382-415)."""
        x = int(x) % self.field.ORDER
        if isinstance(self.basis, Monomial):
            return self.field(L.out_scalar(L.lib.ronk_poly_eval, self.field.ORDER, L.ptr(self.coefficients), self.D, x))
        return self.field(L.out_scalar(L.lib.ronk_lagrange_eval, self.field.ORDER, L.ptr(self.coefficients),
                                       L.ptr(self.basis.nodes), self.D, x))

    # ---- arithmetic (arithmetic.rs)
    def _binary(self, fn, rhs):
        self._mono(); rhs._mono()
        out = np.empty(self.D, dtype=np.uint64)
        L.check(fn(self.field.ORDER, L.ptr(self.coefficients), self.D, L.ptr(rhs.coefficients), rhs.D, L.ptr(out)))
        return Polynomial(self.field, out, Monomial())

    def __add__(self, rhs): return self._binary(L.lib.ronk_poly_add, rhs)      # arithmetic.rs:16-35
    def __sub__(self, rhs): return self._binary(L.lib.ronk_poly_sub, rhs)      # arithmetic.rs:49-68

    def __neg__(self):                                                          # arithmetic.rs:77-95
        self._mono()
        return Polynomial(self.field, self.field.vec_neg(self.coefficients), Monomial())

    def __mul__(self, rhs):                                                     # arithmetic.rs:97-119: D + D2 - 1
        self._mono(); rhs._mono()
        out = np.empty(self.D + rhs.D - 1, dtype=np.uint64)
        L.check(L.lib.ronk_poly_mul(self.field.ORDER, self.field._G, L.ptr(self.coefficients), self.D,
                                    L.ptr(rhs.coefficients), rhs.D, L.ptr(out)))
        return Polynomial(self.field, out, Monomial())

    def pow_mult(self, D2, coeff):
        """self * coeff * x^D2 (mod.rs:153-157), D + D2 coefficients."""
        self._mono()
        c = np.full(self.D, int(coeff) % self.field.ORDER, dtype=np.uint64)
        out = np.zeros(self.D + D2, dtype=np.uint64)
        out[D2:] = self.field.vec_mul(self.coefficients, c)
        return Polynomial(self.field, out, Monomial())

    def quotient_and_remainder(self, rhs):
        """Euclidean division (mod.rs:170-225); both results have D coefficients."""
        self._mono(); rhs._mono()
        q = np.empty(self.D, dtype=np.uint64); r = np.empty(self.D, dtype=np.uint64)
        L.check(L.lib.ronk_poly_divrem(self.field.ORDER, L.ptr(self.coefficients), self.D, L.ptr(rhs.coefficients),
                                       rhs.D, L.ptr(q), L.ptr(r)))
        return Polynomial(self.field, q, Monomial()), Polynomial(self.field, r, Monomial())

    def __truediv__(self, rhs): return self.quotient_and_remainder(rhs)[0]      # arithmetic.rs:121-133
    div = __truediv__
    def __mod__(self, rhs): return self.quotient_and_remainder(rhs)[1]          # arithmetic.rs:135-146

    # ---- transforms
    def dft(self):
        """Polynomial::dft (mod.rs:240-258): any D dividing ORDER - 1; natural order; Lagrange result."""
        self._mono()
        out = np.empty(self.D, dtype=np.uint64)
        L.check(L.lib.ronk_dft(self.field.ORDER, self.field._G, L.ptr(self.coefficients), L.ptr(out), self.D))
        return Polynomial.new_lagrange(self.field, out)

    def fft(self):
        """Polynomial::fft (mod.rs:273-292): D a power of two dividing ORDER - 1 (else the reference's panics)."""
        self._mono()
        out = np.empty(self.D, dtype=np.uint64)
        nodes = np.empty(self.D, dtype=np.uint64)
        L.check(L.lib.ronk_fft(self.field.ORDER, self.field._G, L.ptr(self.coefficients), L.ptr(out), L.ptr(nodes), self.D))
        return Polynomial(self.field, out, Lagrange(nodes))

    def ifft(self):
        """Polynomial::<Lagrange>::ifft (mod.rs:430-453), including the D^-1 scale."""
        if not isinstance(self.basis, Lagrange):
            raise TypeError("ifft is defined on the Lagrange basis")
        out = np.empty(self.D, dtype=np.uint64)
        L.check(L.lib.ronk_ifft(self.field.ORDER, self.field._G, L.ptr(self.coefficients), L.ptr(out), self.D))
        return Polynomial(self.field, out, Monomial())
