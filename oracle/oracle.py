"""ctypes binding of oracle/libronk_oracle.so (TEST INFRASTRUCTURE ONLY).

Mirrors ronk_oracle.h one to one; arrays are numpy uint64.  A negative return code is
what the reference would have panicked on and is raised as OraclePanic.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libronk_oracle.so")

GOLDILOCKS_P = 0xFFFFFFFF00000001
GOLDILOCKS_G = 7

PANICS = {
    -1: "n must divide p^q - 1",
    -2: "called `Option::unwrap()` on a `None` value (inverse of zero)",
    -3: "number of coefficients is not a power of two",
    -4: "input is not a prime number",
    -5: "generator not found",
    -6: "index out of bounds / unwrap on None",
    -11: "Point is not on curve",
    -13: "Element is not a quadratic residue",
}


class OraclePanic(Exception):
    def __init__(self, code):
        super().__init__(PANICS.get(code, "panic %d" % code))
        self.code = code


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "ronk_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def _load():
    build()
    lib = C.CDLL(_SO)
    u64, sz, pu = C.c_uint64, C.c_size_t, C.POINTER(C.c_uint64)
    sig = {
        "orc_is_prime": (C.c_int, [u64]),
        "orc_find_primitive_element": (C.c_int, [u64, pu]),
        "orc_new": (u64, [u64, u64]),
        "orc_add": (u64, [u64, u64, u64]),
        "orc_sub": (u64, [u64, u64, u64]),
        "orc_neg": (u64, [u64, u64]),
        "orc_mul": (u64, [u64, u64, u64]),
        "orc_pow": (u64, [u64, u64, u64]),
        "orc_inverse": (C.c_int, [u64, u64, pu]),
        "orc_div": (C.c_int, [u64, u64, u64, pu]),
        "orc_rem": (C.c_int, [u64, u64, u64, pu]),
        "orc_primitive_root_of_unity": (C.c_int, [u64, u64, u64, pu]),
        "orc_vec_add": (None, [u64, pu, pu, pu, sz]),
        "orc_vec_sub": (None, [u64, pu, pu, pu, sz]),
        "orc_vec_mul": (None, [u64, pu, pu, pu, sz]),
        "orc_vec_neg": (None, [u64, pu, pu, sz]),
        "orc_vec_inv": (C.c_int, [u64, pu, pu, sz]),
        "orc_vec_pow": (None, [u64, pu, u64, pu, sz]),
        "orc_euler_criterion": (C.c_int, [u64, u64]),
        "orc_sqrt": (C.c_int, [u64, u64, pu, pu]),
        "orc_vec_euler": (None, [u64, pu, pu, sz]),
        "orc_vec_sqrt": (C.c_int, [u64, pu, pu, pu, sz]),
        "orc_lagrange_nodes": (C.c_int, [u64, u64, pu, sz]),
        "orc_dft": (C.c_int, [u64, u64, pu, pu, sz]),
        "orc_fft": (C.c_int, [u64, u64, pu, pu, sz]),
        "orc_ifft": (C.c_int, [u64, u64, pu, pu, sz]),
        "orc_poly_add": (None, [u64, pu, sz, pu, sz, pu]),
        "orc_poly_sub": (None, [u64, pu, sz, pu, sz, pu]),
        "orc_poly_neg": (None, [u64, pu, sz, pu]),
        "orc_poly_mul": (None, [u64, pu, sz, pu, sz, pu]),
        "orc_poly_divrem": (C.c_int, [u64, pu, sz, pu, sz, pu, pu]),
        "orc_poly_eval": (u64, [u64, pu, sz, u64]),
        "orc_lagrange_eval": (C.c_int, [u64, pu, pu, sz, u64, pu]),
        "orc_pow_mult": (None, [u64, pu, sz, sz, u64, pu]),
        "orc_degree": (sz, [pu, sz]),
        "orc_leading_coefficient": (u64, [pu, sz]),
        "orc_poly_from": (None, [pu, sz, pu, sz]),
        "orc_rs_encode": (C.c_int, [u64, u64, pu, sz, sz, pu, pu]),
        "orc_rs_decode": (C.c_int, [u64, pu, pu, sz, pu]),
        "orc_curve_is_on_curve": (C.c_int, [C.c_void_p, pu]),
        "orc_curve_add": (C.c_int, [C.c_void_p, pu, pu, pu]),
        "orc_curve_mul": (C.c_int, [C.c_void_p, pu, u64, pu]),
        "orc_kzg_commit": (C.c_int, [C.c_void_p, pu, sz, pu, sz, pu]),
        "orc_kzg_open_quotient": (C.c_int, [u64, pu, sz, u64, pu]),
        "orc_fft_recursive": (None, [u64, pu, sz, u64]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


_lib = _load()
_PU = C.POINTER(C.c_uint64)


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def _p(a):
    return a.ctypes.data_as(_PU)


def _chk(rc):
    if rc != 0:
        raise OraclePanic(rc)


def _scalar_out(fn, *args):
    out = C.c_uint64(0)
    _chk(fn(*args, C.byref(out)))
    return out.value


# ---- field ----
def is_prime(p):
    return _lib.orc_is_prime(p) == 0


def find_primitive_element(p):
    return _scalar_out(_lib.orc_find_primitive_element, p)


def new(p, v): return _lib.orc_new(p, v)
def add(p, a, b): return _lib.orc_add(p, a, b)
def sub(p, a, b): return _lib.orc_sub(p, a, b)
def neg(p, a): return _lib.orc_neg(p, a)
def mul(p, a, b): return _lib.orc_mul(p, a, b)
def pow_(p, a, e): return _lib.orc_pow(p, a, e)
def inverse(p, a): return _scalar_out(_lib.orc_inverse, p, a)
def div(p, a, b): return _scalar_out(_lib.orc_div, p, a, b)
def rem(p, a, b): return _scalar_out(_lib.orc_rem, p, a, b)
def primitive_root_of_unity(p, g, n): return _scalar_out(_lib.orc_primitive_root_of_unity, p, g, n)


def _vec2(fn, p, a, b):
    a, b = _arr(a), _arr(b)
    assert a.shape == b.shape
    out = np.empty_like(a)
    fn(p, _p(a), _p(b), _p(out), a.size)
    return out


def vec_add(p, a, b): return _vec2(_lib.orc_vec_add, p, a, b)
def vec_sub(p, a, b): return _vec2(_lib.orc_vec_sub, p, a, b)
def vec_mul(p, a, b): return _vec2(_lib.orc_vec_mul, p, a, b)


def vec_neg(p, a):
    a = _arr(a); out = np.empty_like(a)
    _lib.orc_vec_neg(p, _p(a), _p(out), a.size)
    return out


def vec_inv(p, a):
    a = _arr(a); out = np.empty_like(a)
    _chk(_lib.orc_vec_inv(p, _p(a), _p(out), a.size))
    return out


def vec_pow(p, a, e):
    a = _arr(a); out = np.empty_like(a)
    _lib.orc_vec_pow(p, _p(a), e, _p(out), a.size)
    return out


# ---- FieldExt (field/mod.rs:79-84, prime/mod.rs:142-226) ----
def euler_criterion(p, a):
    return bool(_lib.orc_euler_criterion(p, int(a)))


def sqrt(p, a):
    """(smaller root, larger root); OraclePanic(-13) for a non-residue"""
    r0, r1 = C.c_uint64(0), C.c_uint64(0)
    _chk(_lib.orc_sqrt(p, int(a), C.byref(r0), C.byref(r1)))
    return int(r0.value), int(r1.value)


def vec_euler(p, a):
    a = _arr(a); out = np.empty_like(a)
    _lib.orc_vec_euler(p, _p(a), _p(out), a.size)
    return out


def vec_sqrt(p, a):
    a = _arr(a); r0 = np.empty_like(a); r1 = np.empty_like(a)
    _chk(_lib.orc_vec_sqrt(p, _p(a), _p(r0), _p(r1), a.size))
    return r0, r1


# ---- polynomial ----
def lagrange_nodes(p, g, n):
    out = np.empty(n, dtype=np.uint64)
    _chk(_lib.orc_lagrange_nodes(p, g, _p(out), n))
    return out


def _xform(fn, p, g, x):
    x = _arr(x); out = np.empty_like(x)
    _chk(fn(p, g, _p(x), _p(out), x.size))
    return out


def dft(p, g, x): return _xform(_lib.orc_dft, p, g, x)
def fft(p, g, x): return _xform(_lib.orc_fft, p, g, x)
def ifft(p, g, x): return _xform(_lib.orc_ifft, p, g, x)


def fft_recursive_inplace(p, values, omega):
    """Timed leg of the CPU baseline: the reference's recursion, root supplied."""
    assert values.dtype == np.uint64 and values.flags.c_contiguous
    _lib.orc_fft_recursive(p, _p(values), values.size, omega)


def poly_add(p, a, b):
    a, b = _arr(a), _arr(b); out = np.empty_like(a)
    _lib.orc_poly_add(p, _p(a), a.size, _p(b), b.size, _p(out))
    return out


def poly_sub(p, a, b):
    a, b = _arr(a), _arr(b); out = np.empty_like(a)
    _lib.orc_poly_sub(p, _p(a), a.size, _p(b), b.size, _p(out))
    return out


def poly_neg(p, a):
    a = _arr(a); out = np.empty_like(a)
    _lib.orc_poly_neg(p, _p(a), a.size, _p(out))
    return out


def poly_mul(p, a, b):
    a, b = _arr(a), _arr(b); out = np.empty(a.size + b.size - 1, dtype=np.uint64)
    _lib.orc_poly_mul(p, _p(a), a.size, _p(b), b.size, _p(out))
    return out


def poly_divrem(p, a, b):
    a, b = _arr(a), _arr(b)
    q = np.empty_like(a); r = np.empty_like(a)
    _chk(_lib.orc_poly_divrem(p, _p(a), a.size, _p(b), b.size, _p(q), _p(r)))
    return q, r


def poly_eval(p, c, x):
    c = _arr(c)
    return _lib.orc_poly_eval(p, _p(c), c.size, x)


def lagrange_eval(p, c, nodes, x):
    c, nodes = _arr(c), _arr(nodes)
    return _scalar_out(_lib.orc_lagrange_eval, p, _p(c), _p(nodes), c.size, x)


def pow_mult(p, c, d2, coeff):
    c = _arr(c); out = np.empty(c.size + d2, dtype=np.uint64)
    _lib.orc_pow_mult(p, _p(c), c.size, d2, coeff, _p(out))
    return out


def degree(c):
    c = _arr(c)
    return _lib.orc_degree(_p(c), c.size)


def leading_coefficient(c):
    c = _arr(c)
    return _lib.orc_leading_coefficient(_p(c), c.size)


def poly_from(c, d):
    c = _arr(c); out = np.empty(d, dtype=np.uint64)
    _lib.orc_poly_from(_p(c), c.size, _p(out), d)
    return out


def rs_encode(p, g, msg, n):
    msg = _arr(msg)
    xs = np.empty(n, dtype=np.uint64); ys = np.empty(n, dtype=np.uint64)
    _chk(_lib.orc_rs_encode(p, g, _p(msg), msg.size, n, _p(xs), _p(ys)))
    return xs, ys


def rs_decode(p, xs, ys, k):
    """Message::decode (codes/reed_solomon.rs:54-106): the first k coordinates -> k message coefficients"""
    xs = _arr(xs); ys = _arr(ys)
    assert xs.size >= k and ys.size >= k
    out = np.empty(k, dtype=np.uint64)
    _chk(_lib.orc_rs_decode(p, _p(xs), _p(ys), k, _p(out)))
    return out


# ---- curve arithmetic behind kzg::commit.  A point is 5 words: x0 x1 y0 y1 inf.
class Curve(C.Structure):
    """y^2 = x^3 + a x + b over F_p[u]/(u^2 - nr)  (src/curve/pluto_curve.rs:27-51, extension/gf_101_2.rs:12-18)"""
    _fields_ = [("p", C.c_uint64), ("nr", C.c_uint64), ("a", C.c_uint64), ("b", C.c_uint64)]


PLUTO_CURVE = Curve(101, 99, 0, 3)          # X^2 + 2 irreducible -> u^2 = -2 = 99; y^2 = x^3 + 3
INFINITY = [0, 0, 0, 0, 1]


def point(x0, y0, x1=0, y1=0):
    return [x0, x1, y0, y1, 0]


def curve_is_on_curve(c, pt):
    return bool(_lib.orc_curve_is_on_curve(C.byref(c), _p(_arr(pt))))


def curve_add(c, p1, p2):
    out = np.empty(5, dtype=np.uint64)
    _chk(_lib.orc_curve_add(C.byref(c), _p(_arr(p1)), _p(_arr(p2)), _p(out)))
    return out.tolist()


def curve_mul(c, pt, k):
    out = np.empty(5, dtype=np.uint64)
    _chk(_lib.orc_curve_mul(C.byref(c), _p(_arr(pt)), k, _p(out)))
    return out.tolist()


def kzg_commit(c, coeffs, srs):
    """kzg::commit (kzg/setup.rs:45-60); srs: list of points"""
    flat = _arr([w for pt in srs for w in pt]); co = _arr(coeffs)
    out = np.empty(5, dtype=np.uint64)
    _chk(_lib.orc_kzg_commit(C.byref(c), _p(flat), len(srs), _p(co), co.size, _p(out)))
    return out.tolist()


def kzg_open_quotient(p, coeffs, z):
    c = _arr(coeffs); q = np.empty_like(c)
    _chk(_lib.orc_kzg_open_quotient(p, _p(c), c.size, z, _p(q)))
    return q
