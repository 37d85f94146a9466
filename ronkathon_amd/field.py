"""Host mirror of ronkathon's `Field` / `FiniteField` traits and `PrimeField<P>`
(reference src/algebra/field/mod.rs:17-84, src/algebra/field/prime/mod.rs:39-140,
src/algebra/field/prime/arithmetic.rs:3-71).

`PrimeField(P)` returns the class of residues mod P (the const-generic `PrimeField<P>`);
`GoldilocksField` is the new 64-bit implementor (explicit PRIMITIVE_ELEMENT = 7) that the
reference's `usize` arithmetic cannot represent.

A single element is host data, exactly like the Rust value type: scalar operators are plain
integer arithmetic on the canonical residue (this is the trait surface, not the data path).
Everything that touches ARRAYS of elements -- `Polynomial` and the `vec_*` class methods
below -- goes through the C ABI into the HIP kernels.
"""
import numpy as np

from . import _lib as L

_classes = {}


def PrimeField(P):
    """`PrimeField<const P: usize>` (prime/mod.rs:39-42).  Non-prime P panics (prime/mod.rs:92-100)."""
    P = int(P)
    if P in _classes:
        return _classes[P]
    L.check(L.lib.ronk_check_prime(P))          # const fn is_prime(P)
    g = L.out_scalar(L.lib.ronk_primitive_element, P)

    class _F(_Element):
        ORDER = P
        _G = g

    _F.__name__ = _F.__qualname__ = "PrimeField<%d>" % P
    _F.ZERO, _F.ONE = _F(0), _F(1)
    _F.PRIMITIVE_ELEMENT = _F(g)                 # prime/mod.rs:87-90
    _classes[P] = _F
    return _F


class _Element:
    __slots__ = ("value",)
    ORDER = None

    def __init__(self, value=0):                 # PrimeField::new: value % P (prime/mod.rs:48-51)
        self.value = int(value) % self.ORDER

    @classmethod
    def from_str(cls, s):                        # FromStr (prime/mod.rs:262-270): parse, then new
        return cls(int(str(s).strip()))

    @classmethod
    def sample(cls, rng):                        # Distribution<PrimeField<P>> for Standard (prime/mod.rs:129-140):
        while True:                              # 28-bit draws (next_u32 >> 4) until one is below ORDER
            draws = rng.integers(0, 1 << 32, size=4096, dtype=np.uint64) >> np.uint64(4)   # (batched: for a small
            ok = np.nonzero(draws < np.uint64(cls.ORDER))[0]                               #  P almost all are rejected)
            if ok.size:
                return cls(int(draws[ok[0]]))

    # --- Field trait (field/mod.rs:17-51)
    def inverse(self):                           # prime/mod.rs:62-72: None for zero
        if self.value == 0:
            return None
        return self.pow(self.ORDER - 2)

    def pow(self, power):                        # prime/mod.rs:74-84 (value of the recursion)
        return type(self)(pow(self.value, int(power), self.ORDER) if power else 1)

    # --- FiniteField trait (field/mod.rs:54-76)
    @classmethod
    def primitive_root_of_unity(cls, n):
        return cls(L.out_scalar(L.lib.ronk_root_of_unity, cls.ORDER, cls._G, int(n)))  # panics: n must divide p^q - 1

    # --- FieldExt trait (field/mod.rs:79-84; prime/mod.rs:142-226)
    def euler_criterion(self):                   # self.pow((P - 1) / 2).value == 1
        return self.pow((self.ORDER - 1) // 2).value == 1

    def sqrt(self):
        """Tonelli-Shanks as the reference writes it: (smaller root, larger root); ZERO -> (0, 0); a non-residue is the
        reference's assert.  (Over F_2 the reference's search for a non-residue never ends: reported as unsupported.)"""
        cls, P = type(self), self.ORDER
        if self.value == 0:
            return (cls(0), cls(0))
        if P == 2:
            raise L.RonkPanic(L.ERR_UNSUPPORTED)
        if not self.euler_criterion():
            raise L.RonkPanic(L.ERR_NOT_RESIDUE)
        q, s = P - 1, 0
        while q % 2 == 0:
            q //= 2; s += 1
        z = cls(2)
        while z.euler_criterion():
            z = z + cls(1)
        m, c, t, r = s, z.pow(q), self.pow(q), self.pow((q + 1) // 2)
        while True:
            if t.value == 1:
                nr = -r
                return (nr, r) if nr.value < r.value else (r, nr)
            i, t_pow = 1, t.pow(2)
            while t_pow.value != 1:
                t_pow = t_pow.pow(2); i += 1
            b = c.pow(2 ** (m - i - 1))
            m, c = i, b.pow(2)
            t, r = t * c, r * b

    # --- operator impls (prime/arithmetic.rs:3-71)
    def _coerce(self, o):
        return o if isinstance(o, type(self)) else type(self)(o)

    def __add__(self, o): return type(self)(self.value + self._coerce(o).value)
    def __sub__(self, o): return type(self)(self.value - self._coerce(o).value)
    def __mul__(self, o): return type(self)(self.value * self._coerce(o).value)
    def __neg__(self): return type(self)(-self.value)

    def __truediv__(self, o):                    # self * rhs.inverse().unwrap()
        inv = self._coerce(o).inverse()
        if inv is None:
            raise L.RonkPanic(L.ERR_ZERO_INVERSE)
        return self * inv

    def __mod__(self, o):                        # Rem: self - (self / rhs) * rhs
        o = self._coerce(o)
        return self - (self / o) * o

    __radd__ = __add__
    __rmul__ = __mul__
    def __eq__(self, o): return isinstance(o, _Element) and o.ORDER == self.ORDER and o.value == self.value
    def __hash__(self): return hash((self.ORDER, self.value))
    def __int__(self): return self.value
    def __index__(self): return self.value
    def __repr__(self): return "%d" % self.value  # Display (prime/mod.rs:125-127)

    # --- array forms of the operators: these run on the GPU through the C ABI
    @classmethod
    def _v(cls, fn, a, b):
        a, b = L.arr(a), L.arr(b)
        out = np.empty_like(a)
        L.check(fn(cls.ORDER, L.ptr(a), L.ptr(b), L.ptr(out), a.size))
        return out

    @classmethod
    def vec_add(cls, a, b): return cls._v(L.lib.ronk_vec_add, a, b)
    @classmethod
    def vec_sub(cls, a, b): return cls._v(L.lib.ronk_vec_sub, a, b)
    @classmethod
    def vec_mul(cls, a, b): return cls._v(L.lib.ronk_vec_mul, a, b)

    @classmethod
    def vec_neg(cls, a):
        a = L.arr(a); out = np.empty_like(a)
        L.check(L.lib.ronk_vec_neg(cls.ORDER, L.ptr(a), L.ptr(out), a.size))
        return out

    @classmethod
    def vec_inv(cls, a):
        a = L.arr(a); out = np.empty_like(a)
        L.check(L.lib.ronk_vec_inv(cls.ORDER, L.ptr(a), L.ptr(out), a.size))
        return out

    @classmethod
    def vec_pow(cls, a, e):
        a = L.arr(a); out = np.empty_like(a)
        L.check(L.lib.ronk_vec_pow(cls.ORDER, L.ptr(a), int(e), L.ptr(out), a.size))
        return out

    @classmethod
    def vec_euler(cls, a):
        """FieldExt::euler_criterion over an array: 1 where a[i] is a non-zero square, else 0"""
        a = L.arr(a); out = np.empty_like(a)
        L.check(L.lib.ronk_vec_euler(cls.ORDER, L.ptr(a), L.ptr(out), a.size))
        return out

    @classmethod
    def vec_sqrt(cls, a):
        """FieldExt::sqrt over an array: (smaller roots, larger roots); a non-residue anywhere is the reference's assert"""
        a = L.arr(a); r0 = np.empty_like(a); r1 = np.empty_like(a)
        L.check(L.lib.ronk_vec_sqrt(cls.ORDER, L.ptr(a), L.ptr(r0), L.ptr(r1), a.size))
        return r0, r1


PlutoBaseField = PrimeField(101)     # prime/mod.rs:26
PlutoScalarField = PrimeField(17)    # prime/mod.rs:30
AESField = PrimeField(2)             # prime/mod.rs:34
GoldilocksField = PrimeField(L.GOLDILOCKS_P)
