"""The callers either side of the polynomial path, over the C ABI.

* Reed-Solomon `Message::encode::<N>` (reference src/codes/reed_solomon.rs:42-52): evaluating the
  K-coefficient message at omega_N^i, i < N, is a size-N DFT of the zero-padded message.
* KZG `open` quotient (reference src/kzg/setup.rs:63-78): poly / (x - z).
"""
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
