"""The callers either side of the polynomial path, over the C ABI.

* Reed-Solomon `Message::encode::<N>` (reference src/codes/reed_solomon.rs:42-52): evaluating the
  K-coefficient message at omega_N^i, i < N, is a size-N DFT of the zero-padded message.
* KZG `open` quotient (reference src/kzg/setup.rs:63-78): poly / (x - z).
* KZG `commit` / `open` (reference src/kzg/setup.rs:45-78): the multi-scalar multiplication over the SRS.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .polynomial import Polynomial


class Message:
    """`Message<K, P>` (codes/reed_solomon.rs:14-18)."""

    def __init__(self, field, data):
        self.field = field
        self.data = L.arr([int(x) % field.ORDER for x in data])

    def encode(self, N):
        """-> list of (x, y) coordinates, `Codeword<N, K, P>` (codes/reed_solomon.rs:42-52)."""
        k = self.data.size
        xs = np.empty(N, dtype=np.uint64); ys = np.empty(N, dtype=np.uint64)
        L.check(L.lib.ronk_rs_encode(self.field.ORDER, self.field._G, L.ptr(self.data), k, N, L.ptr(xs), L.ptr(ys)))
        return xs, ys


    @classmethod
    def decode(cls, field, xs, ys, K):
        """`Message::decode::<M>` (codes/reed_solomon.rs:54-106): interpolate the first K coordinates of a
        codeword (after erasures: any K surviving coordinates, in any order) back to the K message symbols."""
        xs = L.arr([int(v) % field.ORDER for v in xs]); ys = L.arr([int(v) % field.ORDER for v in ys])
        if xs.size < K or ys.size < K:
            raise L.RonkPanic(L.ERR_INDEX, "Code size must be greater than or equal to K")  # assert_ge::<M, K>()
        out = np.empty(K, dtype=np.uint64)
        L.check(L.lib.ronk_rs_decode(field.ORDER, L.ptr(xs), L.ptr(ys), K, L.ptr(out)))
        msg = cls.__new__(cls)
        msg.field, msg.data = field, out
        return msg


def kzg_open_quotient(field, coeffs, eval_point):
    """`kzg::open`'s polynomial step (kzg/setup.rs:63-78): Polynomial::new(coeffs).div([-z, 1])."""
    poly = Polynomial.new(field, coeffs)
    divisor = Polynomial.new(field, [int(-field(eval_point)), 1])
    return (poly / divisor).coefficients


class Curve(C.Structure):
    """y^2 = x^3 + a x + b over F_p[u]/(u^2 - nr) (src/curve/pluto_curve.rs:27-51, extension/gf_101_2.rs:12-18).
    Points are 5-word lists [x0, x1, y0, y1, inf] (`AffinePoint::Point(x, y)` / `AffinePoint::Infinity`)."""
    _fields_ = [("p", C.c_uint64), ("nr", C.c_uint64), ("a", C.c_uint64), ("b", C.c_uint64)]


PlutoExtendedCurve = Curve(101, 99, 0, 3)      # X^2 + 2 -> u^2 = -2; EQUATION_A = 0, EQUATION_B = 3
INFINITY = [0, 0, 0, 0, 1]


def kzg_commit(curve, coeffs, g1_srs):
    """`kzg::commit` (kzg/setup.rs:45-60): sum_i g1_srs[i] * coeffs[i]; panics if the SRS is shorter"""
    pts = L.arr([int(w) for pt in g1_srs for w in pt])
    sc = L.arr([int(c) for c in coeffs])
    out = np.empty(5, dtype=np.uint64)
    L.check(L.lib.ronk_curve_msm(C.byref(curve), L.ptr(pts), len(g1_srs), L.ptr(sc), sc.size, L.ptr(out)))
    return out.tolist()


def kzg_open(curve, scalar_field, coeffs, eval_point, g1_srs):
    """`kzg::open::<D>` (kzg/setup.rs:63-78): commit(poly.div([-z, 1]).coefficients, g1_srs)"""
    return kzg_commit(curve, kzg_open_quotient(scalar_field, coeffs, eval_point), g1_srs)


# ---- kzg::commit on BN254 G1 (bucket-method MSM, csrc/msm_kernels.h)
BN254_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN254_G1 = (1, 2)


def _limbs4(v):
    return [(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def msm_bn254(points, scalars):
    """sum_i scalars[i] * points[i]: `kzg::commit` (kzg/setup.rs:48-60) with a 254-bit group.  points: (x, y) integer
    pairs or None for the point at infinity; scalars: non-negative integers < 2^256.  Returns (x, y) or None."""
    assert len(points) == len(scalars)      # the reference zips and asserts srs.len() >= coeffs.len()
    n = len(points)
    pw = np.zeros((n, 8), dtype=np.uint64)
    sw = np.zeros((n, 4), dtype=np.uint64)
    for i, (pt, k) in enumerate(zip(points, scalars)):
        if pt is not None:
            pw[i, :4] = _limbs4(pt[0]); pw[i, 4:] = _limbs4(pt[1])
        sw[i] = _limbs4(k)
    out = np.zeros(8, dtype=np.uint64)
    L.check(L.lib.ronk_msm_bn254(L.ptr(pw), L.ptr(sw), n, L.ptr(out)))
    x = sum(int(out[i]) << (64 * i) for i in range(4))
    y = sum(int(out[4 + i]) << (64 * i) for i in range(4))
    return None if x == 0 and y == 0 else (x, y)


def kzg_open_bn254(coeffs, z, srs):
    """`kzg::open` (kzg/setup.rs:63-78) over BN254: the polynomial (integer coefficients, taken mod the group order r) divided by
    (x - z), the quotient committed against `srs` ((x, y) pairs or None).  Returns (proof point or None, poly(z))."""
    n = len(coeffs)
    cw = np.zeros((n, 4), dtype=np.uint64)
    for i, c in enumerate(coeffs):
        cw[i] = _limbs4(int(c))
    pw = np.zeros((len(srs), 8), dtype=np.uint64)
    for i, pt in enumerate(srs):
        if pt is not None:
            pw[i, :4] = _limbs4(pt[0]); pw[i, 4:] = _limbs4(pt[1])
    zw = np.array(_limbs4(int(z)), dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    val = np.zeros(4, dtype=np.uint64)
    L.check(L.lib.ronk_kzg_open_bn254(L.ptr(cw), n, L.ptr(zw), L.ptr(pw), len(srs), L.ptr(out), L.ptr(val)))
    x = sum(int(out[i]) << (64 * i) for i in range(4))
    y = sum(int(out[4 + i]) << (64 * i) for i in range(4))
    return (None if x == 0 and y == 0 else (x, y)), sum(int(val[i]) << (64 * i) for i in range(4))
